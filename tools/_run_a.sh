cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 800 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 > gpurun_out/a_tests.log
timeout 300 python bench.py --steps 6 --warmup 2 > gpurun_out/a_bench.log 2>&1
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o a -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/a_prof.log 2>&1
DB=$(find /tmp/prof -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB > $GRAFT_REPO_ROOT/gpurun_out/a_kernel_stats.txt 2>&1
nproc > $GRAFT_REPO_ROOT/gpurun_out/a_nproc.txt
