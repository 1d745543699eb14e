cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
B="--steps 10 --warmup 4 --no-cpu-baseline --no-eager-baseline --no-dense-text-line --no-roofline"
OCN_IMAGE_SPLIT=1 timeout 300 python bench.py $B > gpurun_out/r2s_1.log 2>&1
OCN_IMAGE_SPLIT=2 timeout 300 python bench.py $B > gpurun_out/r2s_2.log 2>&1
OCN_IMAGE_SPLIT=3 timeout 300 python bench.py $B > gpurun_out/r2s_3.log 2>&1
