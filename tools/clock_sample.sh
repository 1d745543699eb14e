# Samples the GPU's clocks / power with rocm-smi while bench.py runs (developer tool; run through tools/gpu.sh): bash tools/clock_sample.sh TAG
TAG=${1:-clk}; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out/${TAG}_clock_samples.txt
echo "# idle:" > $O; rocm-smi --showclocks --showpower 2>&1 | grep -i "sclk\|mclk\|power\|fclk" >> $O
( for i in $(seq 1 400); do echo "t=$(date +%s.%N | cut -c1-14)"; rocm-smi --showclocks --showpower 2>&1 | grep -i "sclk\|mclk\|Power (W)\|Average Graphics\|Current Socket"; sleep 0.4; done >> $O ) &
SAMPLER=$!
timeout 400 python bench.py --steps 60 --warmup 5 --no-roofline --no-cpu-baseline --no-eager-baseline --no-dense-text-line --no-extra-lines 2>&1 | grep '^{' > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_bench.json
echo "# bench finished at t=$(date +%s.%N | cut -c1-14)" >> $O
kill $SAMPLER 2>/dev/null; wait $SAMPLER 2>/dev/null
exit 0
