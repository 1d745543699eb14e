# One budgeted GPU-box call of round 3 (run through tools/gpu.sh): bash tools/gpu_call_r3.sh TAG "STEPS..."
#   steps: tests | newtests | abgelu | abstep | bench | yard | calib | prof | profov | pmc | mfma | lines | native
TAG=${1:-call}; STEPS=${2:-"tests bench"}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
QUIET="--no-cpu-baseline --no-eager-baseline --no-dense-text-line"
has() { case " $STEPS " in *" $1 "*) return 0;; esac; return 1; }
t0=$(date +%s); stamp() { echo "$1 done at +$(( $(date +%s) - t0 )) s" >> $O/${TAG}_timeline.txt; }
nproc > $O/${TAG}_host.txt; free -g | head -2 >> $O/${TAG}_host.txt
if has tests; then
  rm -f $O/parity_report.txt
  timeout 1500 python -m pytest tests -m gpu -q --maxfail=12 --durations=15 2>&1 | tail -60 > $O/${TAG}_tests.log
  cp $O/parity_report.txt $O/${TAG}_parity_report.txt 2>/dev/null
  timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3 > $O/${TAG}_smoke.log; stamp tests
fi
if has newtests; then
  rm -f $O/parity_report.txt
  timeout 900 python -m pytest ${NEW_TESTS:-tests/test_bench_size_gpu.py} -q --maxfail=12 --durations=10 2>&1 | tail -40 > $O/${TAG}_newtests.log
  cp $O/parity_report.txt $O/${TAG}_newtests_parity_report.txt 2>/dev/null; stamp newtests
fi
if [ ! -d $GRAFT_REPO_ROOT/_ab_r02 ]; then STEPS=$(echo " $STEPS " | sed 's/ abgelu / /; s/ abstep / /'); fi  # the round-2 tree (untracked) is needed for the A/B steps
if has abgelu; then  # GELU / dGELU epilogue GEMMs: this tree (8-bit gelu') against the round-2 tree (bf16 gelu'), alternating
  for i in 1 2; do
    timeout 200 python tools/ab_gelu_epilogues.py >> $O/${TAG}_abgelu_new.txt 2>&1
    timeout 200 python tools/ab_gelu_epilogues.py --root $GRAFT_REPO_ROOT/_ab_r02 >> $O/${TAG}_abgelu_r02.txt 2>&1
  done; stamp abgelu
fi
if has abstep; then  # whole step: this tree against the round-2 tree on the same box, alternating
  for i in 1 2; do
    timeout 300 python bench.py --steps 12 --warmup 3 --no-roofline $QUIET 2>&1 | grep '^{' >> $O/${TAG}_abstep_new.json
    (cd _ab_r02 && timeout 300 python bench.py --steps 12 --warmup 3 --no-roofline $QUIET 2>&1 | grep '^{' >> $O/${TAG}_abstep_r02.json)
  done; stamp abstep
fi
if has bench; then timeout 700 python bench.py --steps 20 --warmup 5 > $O/${TAG}_bench.log 2>&1; stamp bench; fi
if has lines; then
  timeout 300 python bench.py --steps 8 --warmup 2 --h2d $QUIET > $O/${TAG}_bench_h2d.log 2>&1
  timeout 300 python bench.py --steps 8 --warmup 2 --deterministic $QUIET --no-roofline > $O/${TAG}_bench_det.log 2>&1
  timeout 400 python bench.py --steps 2 --warmup 2 --accum-freq 8 $QUIET > $O/${TAG}_bench_accum8.log 2>&1; stamp lines
fi
if has native; then
  timeout 300 python bench.py --steps 10 --warmup 3 --no-roofline $QUIET --force-ddp > $O/${TAG}_bench_ddp.log 2>&1
  timeout 300 python bench.py --steps 10 --warmup 3 --no-roofline $QUIET --native-allreduce > $O/${TAG}_bench_native_allreduce.log 2>&1
  timeout 300 python bench.py --steps 10 --warmup 3 --no-roofline $QUIET > $O/${TAG}_bench_noddp.log 2>&1; stamp native
fi
if has yard; then timeout 400 python tools/gemm_vendor_yardstick.py > $O/${TAG}_gemm_vs_vendor.txt 2>&1; stamp yard; fi
cd /tmp; export TMPDIR=/tmp
if has calib; then
  timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/cal_f -o p -- $GRAFT_REPO_ROOT/tools/probes/fetch_calib_bin > $O/${TAG}_calib_run.log 2>&1
  timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/cal_w -o p -- $GRAFT_REPO_ROOT/tools/probes/fetch_calib_bin >> $O/${TAG}_calib_run.log 2>&1
  find /tmp/cal_f -name "*counter_collection.csv" -exec cp {} $O/${TAG}_calib_fetch.csv \;
  find /tmp/cal_w -name "*counter_collection.csv" -exec cp {} $O/${TAG}_calib_write.csv \;
  python $GRAFT_REPO_ROOT/tools/pmc_calib.py $O/${TAG}_calib_fetch.csv $O/${TAG}_calib_write.csv > $O/${TAG}_fetch_size_calibration.txt 2>&1; stamp calib
fi
if has prof; then  # every kernel alone on the chip (one stream, no wgrad side stream)
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 $QUIET --no-roofline --serial-towers --no-wgrad-pair > $O/${TAG}_prof.log 2>&1
  python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find /tmp/prof -name "*.db" | head -1) > $O/${TAG}_kernel_stats.txt 2>&1; stamp prof
fi
if has profov; then  # the step AS SHIPPED (towers overlapped: a kernel's duration includes what it shares the chip with)
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof2 -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 $QUIET --no-roofline > $O/${TAG}_prof_overlap.log 2>&1
  python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find /tmp/prof2 -name "*.db" | head -1) > $O/${TAG}_kernel_stats_overlap.txt 2>&1; stamp profov
fi
pmc_pass() {  # $1 = file tag, $2 = counters
  timeout 400 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d /tmp/pmc_$1 -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 $QUIET --no-roofline --serial-towers --no-wgrad-pair > $O/${TAG}_pmc_$1.log 2>&1
  find /tmp/pmc_$1 -name "*counter_collection.csv" -exec cp {} $O/${TAG}_pmc_$1.csv \;
}
if has pmc; then pmc_pass FETCH_SIZE FETCH_SIZE; pmc_pass WRITE_SIZE WRITE_SIZE; stamp pmc; fi
if has mfma; then pmc_pass SQ_VALU_MFMA_BUSY_CYCLES "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CU_CYCLES"; stamp mfma; fi
echo "end +$(( $(date +%s) - t0 )) s" >> $O/${TAG}_timeline.txt
