# One budgeted GPU-box call of round 6 (run through tools/gpu.sh): bash tools/gpu_call_r6.sh TAG "STEPS..."
TAG=${1:-call}; STEPS=${2:-"tests bench"}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
QUIET="--no-cpu-baseline --no-eager-baseline --no-dense-text-line --no-extra-lines --no-config-lines"
DEVLIB=$GRAFT_REPO_ROOT/open_clip_amd/libopenclip_hip_dev.so
has() { case " $STEPS " in *" $1 "*) return 0;; esac; return 1; }
t0=$(date +%s); stamp() { echo "$1 done at +$(( $(date +%s) - t0 )) s" >> $O/${TAG}_timeline.txt; }
if has newtests; then
  rm -f $O/parity_report.txt
  timeout ${NEW_TESTS_TIMEOUT:-1500} python -m pytest ${NEW_TESTS:-tests/test_bf16_stream_gpu.py} -q --maxfail=12 --durations=12 ${NEW_TESTS_K:+-k "$NEW_TESTS_K"} 2>&1 | tail -70 > $O/${TAG}_newtests.log
  cp $O/parity_report.txt $O/${TAG}_newtests_parity_report.txt 2>/dev/null; stamp newtests
fi
if has tests; then
  rm -f $O/parity_report.txt
  timeout 3000 python -m pytest tests -m gpu -q --maxfail=12 --durations=20 2>&1 | tail -80 > $O/${TAG}_tests.log
  cp $O/parity_report.txt $O/${TAG}_parity_report.txt 2>/dev/null
  timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3 > $O/${TAG}_smoke.log; stamp tests
fi
if has bench; then timeout 1500 python bench.py --steps 20 --warmup 5 > $O/${TAG}_bench.log 2> $O/${TAG}_bench.err; stamp bench; fi
if has benchquick; then timeout 600 python bench.py --steps 20 --warmup 5 $QUIET ${BENCH_ARGS} > $O/${TAG}_benchquick.log 2>&1; stamp benchquick; fi
if has streams; then  # the three residual-stream modes of the image tower, alternating processes
  for i in 1 2; do for m in fp32 bf16 bf16-fp32grad; do
    timeout 300 python bench.py --steps 12 --warmup 3 --no-roofline $QUIET --image-stream $m 2>&1 | grep '^{' >> $O/${TAG}_streams_$m.json
  done; done; stamp streams
fi
if has rescue; then  # the GEMMs' rescue form: single launches with CUs held (static 1-3 workgroups per CU against rescue), then the step with the form off / on
  timeout 600 python tools/occupancy_hazard_probe.py 2>&1 | grep -v "^/opt" > $O/${TAG}_occupancy_probe.txt
  for i in 1 2 3; do for m in off on; do
    timeout 300 python bench.py --steps 20 --warmup 5 $QUIET --no-roofline --tile-rescue $m 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$m', d['ms_per_step'], d['value'])" >> $O/${TAG}_rescue_ab.txt
  done; done; stamp rescue
fi
if has cmd; then bash -c "$GPU_CMD" > $O/${TAG}_cmd.log 2>&1; stamp cmd; fi
cd /tmp; export TMPDIR=/tmp
if has prof; then  # every kernel alone on the chip (one stream, no wgrad side stream)
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 $QUIET --no-roofline --serial-towers --no-wgrad-pair ${BENCH_ARGS} > $O/${TAG}_prof.log 2>&1
  python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find /tmp/prof -name "*.db" | head -1) > $O/${TAG}_kernel_stats.txt 2>&1; stamp prof
fi
if has profov; then  # the step AS SHIPPED (towers overlapped: a kernel's duration includes what it shares the chip with)
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof2 -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 $QUIET --no-roofline ${BENCH_ARGS} > $O/${TAG}_prof_overlap.log 2>&1
  python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find /tmp/prof2 -name "*.db" | head -1) > $O/${TAG}_kernel_stats_overlap.txt 2>&1; stamp profov
fi
pmc_pass() {  # $1 = file tag, $2 = counters
  timeout 400 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d /tmp/pmc_$1 -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 $QUIET --no-roofline --serial-towers --no-wgrad-pair ${BENCH_ARGS} > $O/${TAG}_pmc_$1.log 2>&1
  find /tmp/pmc_$1 -name "*counter_collection.csv" -exec cp {} $O/${TAG}_pmc_$1.csv \;
}
if has pmc; then
  pmc_pass FETCH_SIZE FETCH_SIZE; pmc_pass WRITE_SIZE WRITE_SIZE
  python $GRAFT_REPO_ROOT/tools/pmc_stats.py $O/${TAG}_pmc_FETCH_SIZE.csv $O/${TAG}_pmc_WRITE_SIZE.csv $O/${TAG}_pmc_traffic.json $(cat $GRAFT_REPO_ROOT/.head_sha 2>/dev/null) > $O/${TAG}_pmc_hbm_traffic.txt 2>&1
  rm -f $O/${TAG}_pmc_FETCH_SIZE.csv $O/${TAG}_pmc_WRITE_SIZE.csv; stamp pmc
fi
if has mfma; then
  pmc_pass SQ_VALU_MFMA_BUSY_CYCLES "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CU_CYCLES"
  python $GRAFT_REPO_ROOT/tools/pmc_mfma.py $O/${TAG}_pmc_SQ_VALU_MFMA_BUSY_CYCLES.csv > $O/${TAG}_pmc_mfma_util.txt 2>&1
  rm -f $O/${TAG}_pmc_SQ_VALU_MFMA_BUSY_CYCLES.csv; stamp mfma
fi
if has logits; then  # the loss at config 3's sizes: wall times un-profiled first, then per kernel and form under rocprofv3
  timeout 300 python $GRAFT_REPO_ROOT/tools/logits_probe.py sharded naive local 2>&1 | grep -v "^/opt" > $O/${TAG}_logits_wall.txt
  for f in sharded naive; do
    timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_logits_$f -o t -- python $GRAFT_REPO_ROOT/tools/logits_probe.py $f > /dev/null 2>&1
    python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find /tmp/prof_logits_$f -name "*.db" | head -1) 2>&1 | head -24 > $O/${TAG}_logits_kernel_stats_$f.txt
  done
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_logits -o t -- python $GRAFT_REPO_ROOT/tools/logits_probe.py sharded naive local > $O/${TAG}_logits_probe.txt 2>&1
  python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find /tmp/prof_logits -name "*.db" | head -1) > $O/${TAG}_logits_kernel_stats.txt 2>&1; stamp logits
fi
echo "end +$(( $(date +%s) - t0 )) s" >> $O/${TAG}_timeline.txt
