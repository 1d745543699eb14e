cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_gemm_bench_shapes_gpu.py -x -q 2>&1 | tail -4 > gpurun_out/r2p_tests.log
timeout 600 python -m pytest tests/test_model_gpu.py -x -q -k "not vith14 and not vitl14" 2>&1 | tail -4 >> gpurun_out/r2p_tests.log
for i in 1 2; do timeout 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-eager-baseline 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['roofline']['achieved'])"; done >> gpurun_out/r2p_tests.log
timeout 120 python tools/sweep.py epiabl 2>&1 | grep "epi 1" >> gpurun_out/r2p_tests.log
