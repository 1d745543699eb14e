cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 > gpurun_out/e_tests.log
for E in 0 1 2 3; do timeout 120 python tools/gemm_trace.py 204800 3072 768 $E >> gpurun_out/e_trace.log 2>&1; done
timeout 120 python tools/gemm_trace.py 315392 2048 512 0 >> gpurun_out/e_trace.log 2>&1
timeout 120 python tools/gemm_trace.py 204800 768 3072 0 >> gpurun_out/e_trace.log 2>&1
rocprofv3 -L > gpurun_out/e_counters.txt 2>&1
