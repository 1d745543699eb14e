cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k attention 2>&1 | tail -3 > gpurun_out/n_attn.log
timeout 300 python tools/sweep.py attn >> gpurun_out/n_attn.log 2>&1
timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/n_bench.log 2>&1
OCN_BENCH_ONE_DEVICE=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 2 --steps 3 --warmup 1 --local-batch 1024 --dist-backend gloo --no-cpu-baseline > gpurun_out/n_bench2.log 2>&1
timeout 600 python -m pytest tests/test_model_gpu.py tests/test_ddp_gpu.py -m gpu -x -q 2>&1 | tail -3 >> gpurun_out/n_attn.log
