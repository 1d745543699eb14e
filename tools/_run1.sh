cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 700 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r2_tests.log
timeout 240 python tools/gemm_bench.py - > gpurun_out/r2_tn.log 2>&1
timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r2_bench.log 2>&1
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o r2 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/r2_prof.log 2>&1
DB=$(find /tmp/prof -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB > $GRAFT_REPO_ROOT/gpurun_out/r2_kernel_stats.txt 2>&1
ls -la /tmp/prof/* >> $GRAFT_REPO_ROOT/gpurun_out/r2_prof.log
