cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for E in 0 1 2 3; do timeout 120 python tools/gemm_trace.py 204800 3072 768 $E 2>&1 | grep -v amdgpu | head -12 >> gpurun_out/g_trace.log; done
timeout 300 python tools/sweep.py tn > gpurun_out/g_tn.log 2>&1
timeout 300 python tools/sweep.py attnp > gpurun_out/g_attnp.log 2>&1
timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/g_bench.log 2>&1
