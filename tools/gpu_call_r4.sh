# One budgeted GPU-box call of round 4 (run through tools/gpu.sh): bash tools/gpu_call_r4.sh TAG "STEPS..."
#   steps: gemmtests | tests | abnt | abnt_knobs | abstep | bench | prof | profov | pmc | mfma | lines
TAG=${1:-call}; STEPS=${2:-"tests bench"}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
QUIET="--no-cpu-baseline --no-eager-baseline --no-dense-text-line --no-extra-lines"
OLD=$GRAFT_REPO_ROOT/_ab/libopenclip_hip_r03.so; DEVLIB=$GRAFT_REPO_ROOT/open_clip_amd/libopenclip_hip_dev.so
has() { case " $STEPS " in *" $1 "*) return 0;; esac; return 1; }
t0=$(date +%s); stamp() { echo "$1 done at +$(( $(date +%s) - t0 )) s" >> $O/${TAG}_timeline.txt; }
nproc > $O/${TAG}_host.txt; free -g | head -2 >> $O/${TAG}_host.txt
if has gemmtests; then
  timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_gemm_bench_shapes_gpu.py -q -k "gemm" --maxfail=12 2>&1 | tail -40 > $O/${TAG}_gemmtests.log; stamp gemmtests
fi
if has newtests; then
  rm -f $O/parity_report.txt
  timeout 1500 python -m pytest ${NEW_TESTS:-tests/test_bench_size_gpu.py} -q --maxfail=12 --durations=10 ${NEW_TESTS_K:+-k "$NEW_TESTS_K"} 2>&1 | tail -60 > $O/${TAG}_newtests.log
  cp $O/parity_report.txt $O/${TAG}_newtests_parity_report.txt 2>/dev/null; stamp newtests
fi
if has tests; then
  rm -f $O/parity_report.txt
  timeout 1800 python -m pytest tests -m gpu -q --maxfail=12 --durations=15 2>&1 | tail -60 > $O/${TAG}_tests.log
  cp $O/parity_report.txt $O/${TAG}_parity_report.txt 2>/dev/null
  timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3 > $O/${TAG}_smoke.log; stamp tests
fi
if has abnt; then  # per-shape NT GEMMs: this tree against the round-3 library, alternating
  for i in 1 2; do
    OCN_LIB_PATH=$OLD timeout 200 python tools/ab_nt.py --json $O/${TAG}_abnt.jsonl >> $O/${TAG}_abnt_r03.txt 2>&1
    timeout 200 python tools/ab_nt.py --json $O/${TAG}_abnt.jsonl >> $O/${TAG}_abnt_new.txt 2>&1
  done; stamp abnt
fi
if has abnt_knobs; then  # cache-policy flips of the epilogues (developer build): 2 = stores, 8 = operand loads, 10 = both
  for k in 0 2 8 10; do OCN_LIB_PATH=$DEVLIB timeout 200 python tools/ab_nt.py --knob $k --only dgelu --json $O/${TAG}_abnt.jsonl >> $O/${TAG}_abnt_knobs.txt 2>&1; done
  for k in 0 2 8; do OCN_LIB_PATH=$DEVLIB timeout 200 python tools/ab_nt.py --knob $k --only res --json $O/${TAG}_abnt.jsonl >> $O/${TAG}_abnt_knobs.txt 2>&1; done
  stamp abnt_knobs
fi
if has abtail; then  # half-tile tail round of the persistent NT GEMM on / off (developer build, knob 0x200000 = whole tail tiles), alternating
  for i in 1 2; do
    OCN_LIB_PATH=$DEVLIB timeout 200 python tools/ab_nt.py --knob 0 --json $O/${TAG}_abtail.jsonl >> $O/${TAG}_abtail_on.txt 2>&1
    OCN_LIB_PATH=$DEVLIB timeout 200 python tools/ab_nt.py --knob 2097152 --json $O/${TAG}_abtail.jsonl >> $O/${TAG}_abtail_off.txt 2>&1
  done; stamp abtail
fi
if has yard; then timeout 400 python tools/gemm_vendor_yardstick.py > $O/${TAG}_gemm_vs_vendor.txt 2>&1; stamp yard; fi
if has abstep; then  # whole step: this tree against the round-3 library on the same box, alternating
  for i in 1 2; do
    timeout 300 python bench.py --steps 12 --warmup 3 --no-roofline $QUIET 2>&1 | grep '^{' >> $O/${TAG}_abstep_new.json
    (cd $GRAFT_REPO_ROOT/_ab/r03tree && timeout 300 python bench.py --steps 12 --warmup 3 --no-roofline --no-cpu-baseline --no-eager-baseline --no-dense-text-line 2>&1 | grep '^{' >> $O/${TAG}_abstep_r03.json)
  done; stamp abstep
fi
if has bench; then timeout 900 python bench.py --steps 20 --warmup 5 > $O/${TAG}_bench.log 2>&1; stamp bench; fi
if has models; then  # BASELINE configs 4 / 5 through the same bench on one GPU (sanity lines, 3 steps)
  Q2="--no-cpu-baseline --no-eager-baseline --no-dense-text-line --no-roofline --no-extra-lines"
  timeout 300 python bench.py --model ViT-H-14 --siglip --local-batch 1024 --grad-checkpointing --steps 3 --warmup 1 $Q2 2>&1 | grep '^{' > $O/${TAG}_h14_bench.json
  timeout 300 python bench.py --model ViT-L-14 --local-batch 2048 --grad-checkpointing --steps 3 --warmup 1 $Q2 2>&1 | grep '^{' > $O/${TAG}_l14_bench.json; stamp models
fi
if has lines; then
  timeout 300 python bench.py --steps 8 --warmup 2 --h2d $QUIET > $O/${TAG}_bench_h2d.log 2>&1
  timeout 300 python bench.py --steps 8 --warmup 2 --deterministic $QUIET --no-roofline > $O/${TAG}_bench_det.log 2>&1; stamp lines
fi
cd /tmp; export TMPDIR=/tmp
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
if has pmc; then
  pmc_pass FETCH_SIZE FETCH_SIZE; pmc_pass WRITE_SIZE WRITE_SIZE
  python $GRAFT_REPO_ROOT/tools/pmc_stats.py $O/${TAG}_pmc_FETCH_SIZE.csv $O/${TAG}_pmc_WRITE_SIZE.csv $O/${TAG}_pmc_traffic.json $(cat $GRAFT_REPO_ROOT/.head_sha 2>/dev/null) > $O/${TAG}_pmc_hbm_traffic.txt 2>&1; stamp pmc
fi
if has mfma; then
  pmc_pass SQ_VALU_MFMA_BUSY_CYCLES "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CU_CYCLES"
  python $GRAFT_REPO_ROOT/tools/pmc_mfma.py $O/${TAG}_pmc_SQ_VALU_MFMA_BUSY_CYCLES.csv > $O/${TAG}_pmc_mfma_util.txt 2>&1; stamp mfma
fi
echo "end +$(( $(date +%s) - t0 )) s" >> $O/${TAG}_timeline.txt
