# One GPU-box round (run through gpurun): parity tests, smoke, bench (+ its baselines), the --h2d / --accum-freq 8 lines, rocprofv3 kernel
# trace, PMC passes.   usage: gpurun --timeout 2700 -- 'bash tools/gpu_round.sh TAG [pmc]'   -> gpurun_out/TAG_*
TAG=${1:-round}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/parity_report.txt
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/${TAG}_tests.log
cp gpurun_out/parity_report.txt gpurun_out/${TAG}_parity_report.txt 2>/dev/null
timeout 300 python __graft_entry__.py smoke > gpurun_out/${TAG}_smoke.log 2>&1
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.log 2>&1
timeout 300 python bench.py --steps 8 --warmup 2 --h2d --no-cpu-baseline --no-eager-baseline --no-dense-text-line > gpurun_out/${TAG}_bench_h2d.log 2>&1
timeout 400 python bench.py --steps 2 --warmup 2 --accum-freq 8 --no-cpu-baseline --no-eager-baseline --no-dense-text-line > gpurun_out/${TAG}_bench_accum8.log 2>&1
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-eager-baseline --no-dense-text-line --no-roofline --serial-towers --no-wgrad-pair > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof.log 2>&1
DB=$(find /tmp/prof -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_kernel_stats.txt 2>&1
# the same trace of the step AS SHIPPED (towers overlapped on two streams: a kernel's duration includes what it shares the chip with)
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof2 -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-eager-baseline --no-dense-text-line --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof_overlap.log 2>&1
DB=$(find /tmp/prof2 -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_kernel_stats_overlap.txt 2>&1
if [ "$2" = "pmc" ]; then
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CU_CYCLES"; do
T=$(echo $C | cut -d' ' -f1)
timeout 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_$T -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-eager-baseline --no-dense-text-line --no-roofline --serial-towers --no-wgrad-pair > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc_$T.log 2>&1
find /tmp/pmc_$T -name "*counter_collection.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc_$T.csv \;
done
fi
