cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for m in all ln attn none; do
OCN_WGRAD_PAIR=$m timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-eager-baseline > gpurun_out/r2v_pair_$m.log 2>&1
done
