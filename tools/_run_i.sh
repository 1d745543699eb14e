cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python tools/sweep.py attnw > gpurun_out/i_attnw.log 2>&1
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k attention 2>&1 | tail -3 >> gpurun_out/i_attnw.log
