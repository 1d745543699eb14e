cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python tools/sweep.py attn > gpurun_out/c_attn.log 2>&1
timeout 600 python tools/sweep.py nt > gpurun_out/c_nt.log 2>&1
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "gemm_nt or attention" 2>&1 | tail -5 > gpurun_out/c_tests.log
timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/c_bench.log 2>&1
