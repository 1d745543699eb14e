# persistent NT GEMM grid oversubscription (developer knob 10) and wgrad M-splits per CU (knob 11) in the SHIPPED two-stream step
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out; T=${1:-w1}
Q="--no-cpu-baseline --no-eager-baseline --no-dense-text-line --no-roofline --steps 10 --warmup 3"
for rep in 1 2; do
for v in 0 2 3 4; do
  timeout 200 python bench.py $Q --tuning 10=$v 2>&1 | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('knob10=$v rep $rep', d['value'], d['ms_per_step'])
" >> $O/${T}_wgs.txt
done; done
cat $O/${T}_wgs.txt
