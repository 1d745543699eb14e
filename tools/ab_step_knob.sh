# whole-step A/B of one developer knob on one box, alternating processes: bash tools/ab_step_knob.sh KEY=VALUE [rounds]
KV=${1:?KEY=VALUE}; N=${2:-3}
cd $GRAFT_REPO_ROOT
Q="--no-cpu-baseline --no-eager-baseline --no-dense-text-line --no-extra-lines --no-config-lines --no-roofline --no-clock-sample"
for i in $(seq $N); do
  python bench.py --steps 12 --warmup 3 $Q 2>/dev/null | grep "^{" | python -c "import json,sys; print('shipped      ', json.loads(sys.stdin.read())['ms_per_step'])"
  python bench.py --steps 12 --warmup 3 $Q --tuning $KV 2>/dev/null | grep "^{" | python -c "import json,sys; print('--tuning $KV', json.loads(sys.stdin.read())['ms_per_step'])"
done
