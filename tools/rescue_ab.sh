# GPU-box call: the training step with the GEMMs' rescue form off / on, alternating processes (what the form costs when nothing holds CUs)
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
QUIET="--no-cpu-baseline --no-eager-baseline --no-dense-text-line --no-extra-lines --no-config-lines --no-roofline"
for i in 1 2 3; do for m in off on; do
  timeout 300 python bench.py --steps 20 --warmup 5 $QUIET --tile-rescue $m 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$m', d['ms_per_step'], d['value'])" >> $O/${1}_rescue_ab.txt
done; done
