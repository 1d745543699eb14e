cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k attention 2>&1 | tail -5 > gpurun_out/j_attn.log
timeout 300 python tools/sweep.py attnw >> gpurun_out/j_attn.log 2>&1
timeout 300 python tools/sweep.py attn >> gpurun_out/j_attn.log 2>&1
timeout 300 python -m pytest tests/test_model_gpu.py -m gpu -x -q 2>&1 | tail -3 >> gpurun_out/j_attn.log
timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/j_bench.log 2>&1
