cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "attention or gemm_nt" 2>&1 | tail -3 > gpurun_out/o_attn.log
timeout 300 python tools/sweep.py attnw >> gpurun_out/o_attn.log 2>&1
timeout 300 python tools/sweep.py ntstore > gpurun_out/o_ntstore.log 2>&1
timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/o_bench.log 2>&1
timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --gemm-variant $((2<<8)) > gpurun_out/o_bench_nt.log 2>&1
