cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python tools/sweep.py stagger > gpurun_out/f_stagger.log 2>&1
timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/f_bench_auto.log 2>&1
timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --gemm-variant $((63<<21)) > gpurun_out/f_bench_off.log 2>&1
timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/f_bench_auto2.log 2>&1
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 > gpurun_out/f_tests.log
