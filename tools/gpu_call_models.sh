cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
Q="--no-cpu-baseline --no-eager-baseline --no-dense-text-line --no-roofline"
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/p5 -o t -- python $GRAFT_REPO_ROOT/bench.py --model ViT-H-14 --siglip --local-batch 1024 --grad-checkpointing --steps 2 --warmup 1 $Q --serial-towers --no-wgrad-pair > $O/c3_h14_prof.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find /tmp/p5 -name "*.db" | head -1) > $O/c3_h14_kernel_stats.txt 2>&1
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/p4 -o t -- python $GRAFT_REPO_ROOT/bench.py --model ViT-L-14 --local-batch 2048 --grad-checkpointing --steps 2 --warmup 1 $Q --serial-towers --no-wgrad-pair > $O/c3_l14_prof.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find /tmp/p4 -name "*.db" | head -1) > $O/c3_l14_kernel_stats.txt 2>&1
cd $GRAFT_REPO_ROOT
timeout 300 python bench.py --model ViT-H-14 --siglip --local-batch 1024 --grad-checkpointing --steps 3 --warmup 1 $Q > $O/c3_h14_bench.log 2>&1
timeout 300 python bench.py --model ViT-L-14 --local-batch 2048 --grad-checkpointing --steps 3 --warmup 1 $Q > $O/c3_l14_bench.log 2>&1
