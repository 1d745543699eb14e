cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out; T=${1:-c4}
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "attention" --maxfail=10 2>&1 | tail -15 > $O/${T}_attn_tests.log
timeout 600 python -m pytest tests/test_model_gpu.py -q -k "head_dim_80 or vith14 or vitl14" --maxfail=10 2>&1 | tail -15 >> $O/${T}_attn_tests.log
timeout 300 python tools/ab_attn_long.py > $O/${T}_ab_attn_long.txt 2>&1
timeout 300 python tools/ab_attn_bwd.py > $O/${T}_ab_attn_bwd.txt 2>&1
