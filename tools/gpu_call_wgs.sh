# persistent NT GEMM grid oversubscription (developer knob 10; KNOB=11: wgrad M-splits per CU) in the SHIPPED two-stream step
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out; T=${1:-w1}
Q="--no-cpu-baseline --no-eager-baseline --no-dense-text-line --no-roofline --steps 10 --warmup 3"
for rep in 1 2; do
for v in ${VALUES:-0 2 3 4}; do
  timeout 200 python bench.py $Q --tuning ${KNOB:-10}=$v 2>&1 | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('knob${KNOB:-10}=$v rep $rep', d['value'], d['ms_per_step'])
" >> $O/${T}_wgs.txt
done; done
cat $O/${T}_wgs.txt
