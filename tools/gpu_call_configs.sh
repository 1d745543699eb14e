# full-depth sanity lines of further registered configs (3 steps each; not the headline metric)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out; T=${1:-g1}
Q="--no-cpu-baseline --no-eager-baseline --no-dense-text-line --no-roofline --steps 3 --warmup 1"
timeout 300 python bench.py --model ViT-B-16 --local-batch 1024 $Q > $O/${T}_b16.log 2>&1
timeout 300 python bench.py --model ViT-S-32 --local-batch 4096 $Q > $O/${T}_s32.log 2>&1
timeout 300 python bench.py --model ViT-L-14-336 --local-batch 512 --grad-checkpointing $Q > $O/${T}_l14_336.log 2>&1
timeout 400 python bench.py --model ViT-g-14 --local-batch 512 --grad-checkpointing $Q > $O/${T}_g14.log 2>&1
timeout 400 python bench.py --model ViT-bigG-14 --local-batch 512 --grad-checkpointing $Q > $O/${T}_bigg14.log 2>&1
for f in b16 s32 l14_336 g14 bigg14; do grep '^{' $O/${T}_$f.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$f', d['value'], d['ms_per_step'], d['peak_hbm_gb_rank0'], [v for k, v in d.items() if 'tflops' in k], d['config'].get('grad_checkpointing', '')[:90])
" || tail -3 $O/${T}_$f.log; done
