cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for m in 0 1 2 3 0; do
  if [ $m = 0 ]; then unset OCN_LIB_PATH; else export OCN_LIB_PATH=$GRAFT_REPO_ROOT/tools/probes/libopenclip_hip_prio$m.so; fi
  echo "=== OCN_PRIO_MODE=$m"
  timeout 200 python tools/gemm_bench.py 5 0,1,2,3 2>&1 | grep -v amdgpu | grep -v "rel_l2\|variant 5:\|TN variant"
  timeout 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-eager-baseline 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('bench ms/step', d['ms_per_step'], 'nt', d['roofline']['achieved'], 'tn', d['roofline']['gemm_tn_kernel']['achieved'])"
done > gpurun_out/r2t_prio.log 2>&1
