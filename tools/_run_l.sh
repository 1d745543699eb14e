cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python tools/overlap_probe.py > gpurun_out/l_overlap.log 2>&1
timeout 600 python -m pytest tests/test_model_gpu.py tests/test_ddp_gpu.py -m gpu -x -q 2>&1 | tail -3 > gpurun_out/l_tests.log
