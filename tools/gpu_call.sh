# One budgeted GPU-box call (run through gpurun): usage  gpurun --timeout 1500 -- 'bash tools/gpu_call.sh TAG "STEPS..."'
#   steps: quick (changed tests + smoke), tests (full pytest -m gpu + smoke), ab (deep-ring A/B), bench, lines (--h2d / --accum-freq 8), prof, pmc, mfma
TAG=${1:-call}; STEPS=${2:-"quick bench prof"}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
QUIET="--no-cpu-baseline --no-eager-baseline --no-dense-text-line"
has() { case " $STEPS " in *" $1 "*) return 0;; esac; return 1; }
t0=$(date +%s); stamp() { echo "$1 done at +$(( $(date +%s) - t0 )) s" >> $O/${TAG}_timeline.txt; }
if has quick; then
  rm -f $O/parity_report.txt
  timeout 600 python -m pytest ${QUICK_TESTS:-tests/test_bench_gpu.py tests/test_model_gpu.py::test_overlapped_towers_equal_one_stream} -x -q --durations=8 2>&1 | tail -25 > $O/${TAG}_tests.log
  cp $O/parity_report.txt $O/${TAG}_parity_report.txt 2>/dev/null
  timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3 > $O/${TAG}_smoke.log; stamp quick
fi
if has tests; then
  rm -f $O/parity_report.txt
  timeout 1100 python -m pytest tests -m gpu -x -q --durations=12 2>&1 | tail -30 > $O/${TAG}_tests.log
  cp $O/parity_report.txt $O/${TAG}_parity_report.txt 2>/dev/null
  timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3 > $O/${TAG}_smoke.log; stamp tests
fi
if has ab; then
  timeout 300 python tools/ab_deep_ring.py > $O/${TAG}_deep_ring.txt 2>&1
  for V in 0 268435456; do
    timeout 300 python bench.py --steps 12 --warmup 3 --no-roofline $QUIET --gemm-variant $V 2>&1 | grep '^{' > $O/${TAG}_bench_variant_$V.json
  done; stamp ab
fi
if has buckets; then  # packed text attention launched in buckets of equal block count vs every workgroup sized for context_length
  for BK in 1 0 1 0; do
    OCN_ATTN_BUCKETS=$BK timeout 300 python bench.py --steps 12 --warmup 3 --no-roofline $QUIET 2>&1 | grep '^{' >> $O/${TAG}_buckets_$BK.json
  done; stamp buckets
fi
if has tnpair; then  # out-proj + QKV wgrads in one launch vs two (developer knob 13)
  timeout 300 python tools/ab_tn_pair.py > $O/${TAG}_tn_pair.txt 2>&1
  for KN in 0 1 0 1; do
    timeout 300 python bench.py --steps 12 --warmup 3 --no-roofline $QUIET --tuning 13=$KN 2>&1 | grep '^{' >> $O/${TAG}_tnpair_$KN.json
  done; stamp tnpair
fi
if has lngrid; then  # LayerNorm backward: 16-wave workgroups, one per CU (default) against other grid sizes
  timeout 300 python tools/ab_ln_bwd.py 0,128,256,512,1024 > $O/${TAG}_ln_grid.txt 2>&1; stamp lngrid
fi
if has pooled; then  # the last block of each tower only on the pooled rows (model.py::_PooledBlockFn) vs the full block
  for PK in 1 0 1 0; do
    OCN_POOLED_LAST_BLOCK=$PK timeout 300 python bench.py --steps 12 --warmup 3 --no-roofline $QUIET 2>&1 | grep '^{' >> $O/${TAG}_pooled_$PK.json
  done; stamp pooled
fi
if has bench; then timeout 500 python bench.py --steps 20 --warmup 5 > $O/${TAG}_bench.log 2>&1; stamp bench; fi
if has lines; then
  timeout 300 python bench.py --steps 8 --warmup 2 --h2d $QUIET > $O/${TAG}_bench_h2d.log 2>&1
  timeout 400 python bench.py --steps 2 --warmup 2 --accum-freq 8 $QUIET > $O/${TAG}_bench_accum8.log 2>&1; stamp lines
fi
cd /tmp; export TMPDIR=/tmp
if has prof; then
  # every kernel alone on the chip (one stream, no wgrad side stream), then the step AS SHIPPED (towers overlapped: a kernel's duration includes what it shares the chip with)
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 $QUIET --no-roofline --serial-towers --no-wgrad-pair > $O/${TAG}_prof.log 2>&1
  python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find /tmp/prof -name "*.db" | head -1) > $O/${TAG}_kernel_stats.txt 2>&1
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof2 -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 $QUIET --no-roofline > $O/${TAG}_prof_overlap.log 2>&1
  python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find /tmp/prof2 -name "*.db" | head -1) > $O/${TAG}_kernel_stats_overlap.txt 2>&1; stamp prof
fi
pmc_pass() {  # $1 = file tag, $2 = counters
  timeout 400 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d /tmp/pmc_$1 -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 $QUIET --no-roofline --serial-towers --no-wgrad-pair > $O/${TAG}_pmc_$1.log 2>&1
  find /tmp/pmc_$1 -name "*counter_collection.csv" -exec cp {} $O/${TAG}_pmc_$1.csv \;
}
if has pmc; then pmc_pass FETCH_SIZE FETCH_SIZE; pmc_pass WRITE_SIZE WRITE_SIZE; stamp pmc; fi
if has mfma; then pmc_pass SQ_VALU_MFMA_BUSY_CYCLES "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CU_CYCLES"; stamp mfma; fi
echo "end +$(( $(date +%s) - t0 )) s" >> $O/${TAG}_timeline.txt
