cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OCN_WGRAD_STREAM=0 timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/k_bench_off.log 2>&1
OCN_WGRAD_STREAM=1 timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/k_bench_on.log 2>&1
OCN_WGRAD_STREAM=0 timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/k_bench_off2.log 2>&1
timeout 600 python -m pytest tests/test_model_gpu.py -m gpu -x -q 2>&1 | tail -3 > gpurun_out/k_tests.log
