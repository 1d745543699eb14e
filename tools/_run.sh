cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
B="timeout 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-eager-baseline --no-roofline"
for cfg in "--force-ddp --tuning 10=3" "--force-ddp --tuning 10=3 --tuning 11=2" "--force-ddp --tuning 10=2" "--tuning 10=3" "--force-ddp" ""; do
  $B $cfg 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('[$cfg]', d['ms_per_step'])"
done > gpurun_out/r2o_ddp_knobs.log 2>&1
