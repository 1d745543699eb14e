# configs 4 / 5 with partial block recompute (--keep-blocks auto) next to full recompute (--keep-blocks 0)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out; T=${1:-k1}
Q="--no-cpu-baseline --no-eager-baseline --no-dense-text-line --no-roofline"
timeout 300 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "oracle" > $O/${T}_tests.log 2>&1
timeout 400 python bench.py --model ViT-H-14 --siglip --local-batch 1024 --grad-checkpointing --keep-blocks auto --steps 3 --warmup 1 $Q > $O/${T}_h14_auto.log 2>&1
timeout 400 python bench.py --model ViT-H-14 --siglip --local-batch 1024 --grad-checkpointing --keep-blocks 0 --steps 3 --warmup 1 $Q > $O/${T}_h14_all.log 2>&1
timeout 400 python bench.py --model ViT-L-14 --local-batch 2048 --grad-checkpointing --keep-blocks auto --steps 3 --warmup 1 $Q > $O/${T}_l14_auto.log 2>&1
timeout 400 python bench.py --model ViT-L-14 --local-batch 2048 --grad-checkpointing --keep-blocks 0 --steps 3 --warmup 1 $Q > $O/${T}_l14_all.log 2>&1
tail -2 $O/${T}_tests.log; for f in h14_auto h14_all l14_auto l14_all; do tail -c 600 $O/${T}_$f.log | tr '\n' ' ' | tail -c 400; echo; done
