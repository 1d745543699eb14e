# configs 4 / 5 after the streamed attention kernels (kernel traces), and bench.py's two-rank path at FULL model size on one GPU (gloo transport)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out; T=${1:-m2}
Q="--no-cpu-baseline --no-eager-baseline --no-dense-text-line --no-roofline"
R="--nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29751"
OCN_BENCH_ONE_DEVICE=1 timeout 600 python -m torch.distributed.run $R bench.py --gpus 2 --local-batch 2048 --steps 4 --warmup 2 $Q --dist-backend gloo > $O/${T}_w2_ddp.log 2>&1
OCN_BENCH_ONE_DEVICE=1 timeout 600 python -m torch.distributed.run $R bench.py --gpus 2 --local-batch 2048 --steps 4 --warmup 2 $Q --dist-backend gloo --native-allreduce > $O/${T}_w2_native.log 2>&1
timeout 300 python bench.py --local-batch 2048 --data-ranks 2 --steps 4 --warmup 2 $Q > $O/${T}_w1_dataranks2.log 2>&1
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/p5 -o t -- python $GRAFT_REPO_ROOT/bench.py --model ViT-H-14 --siglip --local-batch 1024 --grad-checkpointing --steps 2 --warmup 1 $Q --serial-towers --no-wgrad-pair > $O/${T}_h14_prof.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find /tmp/p5 -name "*.db" | head -1) > $O/${T}_h14_kernel_stats.txt 2>&1
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/p4 -o t -- python $GRAFT_REPO_ROOT/bench.py --model ViT-L-14 --local-batch 2048 --grad-checkpointing --steps 2 --warmup 1 $Q --serial-towers --no-wgrad-pair > $O/${T}_l14_prof.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find /tmp/p4 -name "*.db" | head -1) > $O/${T}_l14_kernel_stats.txt 2>&1
