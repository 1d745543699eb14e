cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r_tests.log
timeout 300 python tools/sweep.py nt > gpurun_out/r_nt.log 2>&1
timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/r_bench.log 2>&1
