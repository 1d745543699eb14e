# where the accumulation path's time goes: kernel trace of --accum-freq 2 next to the plain step (every kernel alone on the chip)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out; T=${1:-a1}
Q="--no-cpu-baseline --no-eager-baseline --no-dense-text-line --no-roofline"
timeout 300 python bench.py --steps 4 --warmup 2 --accum-freq 2 $Q > $O/${T}_accum2.log 2>&1
timeout 300 python bench.py --steps 4 --warmup 2 --accum-freq 2 --serial-towers --no-wgrad-pair $Q > $O/${T}_accum2_serial.log 2>&1
timeout 300 python bench.py --steps 6 --warmup 2 --serial-towers --no-wgrad-pair $Q > $O/${T}_plain_serial.log 2>&1
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pa -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --accum-freq 2 $Q --serial-towers --no-wgrad-pair > $O/${T}_prof.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find /tmp/pa -name "*.db" | head -1) > $O/${T}_kernel_stats.txt 2>&1
for f in accum2 accum2_serial plain_serial; do grep '^{' $O/${T}_$f.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$f', d['value'], d['ms_per_step'])
"; done; head -30 $O/${T}_kernel_stats.txt | cut -c1-150
