# One budgeted GPU-box call of round 5 (run through tools/gpu.sh): bash tools/gpu_call_r5.sh TAG "STEPS..."
#   steps: tests | newtests | bench | benchquick | abstep | abgelu | abstagger | stagger | band | shapes | geluform | tnepi | models | lines | attn | profmodels | prof | profov | pmc | mfma
TAG=${1:-call}; STEPS=${2:-"tests bench"}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
QUIET="--no-cpu-baseline --no-eager-baseline --no-dense-text-line --no-extra-lines --no-config-lines"
OLD=$GRAFT_REPO_ROOT/tools/probes/libopenclip_hip_r04.so; DEVLIB=$GRAFT_REPO_ROOT/open_clip_amd/libopenclip_hip_dev.so
has() { case " $STEPS " in *" $1 "*) return 0;; esac; return 1; }
t0=$(date +%s); stamp() { echo "$1 done at +$(( $(date +%s) - t0 )) s" >> $O/${TAG}_timeline.txt; }
nproc > $O/${TAG}_host.txt; free -g | head -2 >> $O/${TAG}_host.txt; rocm-smi --showproductname 2>/dev/null | head -12 >> $O/${TAG}_host.txt
if has newtests; then
  rm -f $O/parity_report.txt
  timeout 1500 python -m pytest ${NEW_TESTS:-tests/test_parity_at_size_gpu.py tests/test_reference_dropin_gpu.py} -q --maxfail=12 --durations=12 ${NEW_TESTS_K:+-k "$NEW_TESTS_K"} 2>&1 | tail -70 > $O/${TAG}_newtests.log
  cp $O/parity_report.txt $O/${TAG}_newtests_parity_report.txt 2>/dev/null; stamp newtests
fi
if has tests; then
  rm -f $O/parity_report.txt
  timeout 2400 python -m pytest tests -m gpu -q --maxfail=12 --durations=20 2>&1 | tail -80 > $O/${TAG}_tests.log
  cp $O/parity_report.txt $O/${TAG}_parity_report.txt 2>/dev/null
  timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3 > $O/${TAG}_smoke.log; stamp tests
fi
if has bench; then timeout 1200 python bench.py --steps 20 --warmup 5 > $O/${TAG}_bench.log 2> $O/${TAG}_bench.err; stamp bench; fi
if has benchquick; then timeout 600 python bench.py --steps 20 --warmup 5 $QUIET > $O/${TAG}_benchquick.log 2>&1; stamp benchquick; fi
if has abstep; then  # whole step: this tree's library against the round-4 library (same Python tree), alternating processes
  for i in 1 2; do
    timeout 300 python bench.py --steps 12 --warmup 3 --no-roofline $QUIET 2>&1 | grep '^{' >> $O/${TAG}_abstep_new.json
    OCN_LIB_PATH=$OLD OCN_ALLOW_ABI=101 timeout 300 python bench.py --steps 12 --warmup 3 --no-roofline $QUIET 2>&1 | grep '^{' >> $O/${TAG}_abstep_r04.json
  done; stamp abstep
fi
if has abgelu; then  # c_fc + GELU GEMMs: polynomial (product) against the Abramowitz-Stegun form (developer knob), same developer library, alternating
  for i in 1 2; do
    OCN_LIB_PATH=$DEVLIB timeout 200 python tools/ab_nt.py --knob 0 --only +gelu --json $O/${TAG}_abgelu.jsonl >> $O/${TAG}_abgelu_poly.txt 2>&1
    OCN_LIB_PATH=$DEVLIB timeout 200 python tools/ab_nt.py --knob 4194304 --only +gelu --json $O/${TAG}_abgelu.jsonl >> $O/${TAG}_abgelu_as.txt 2>&1
  done; stamp abgelu
fi
if has models; then  # BASELINE configs 4 / 5 through the same bench on one GPU, with the roofline
  Q2="--no-cpu-baseline --no-eager-baseline --no-dense-text-line --no-extra-lines --no-config-lines"
  timeout 400 python bench.py --model ViT-L-14 --local-batch 2048 --grad-checkpointing --steps 4 --warmup 1 $Q2 2>&1 | grep '^{' > $O/${TAG}_l14_bench.json
  timeout 400 python bench.py --model ViT-H-14 --siglip --local-batch 1024 --grad-checkpointing --steps 4 --warmup 1 $Q2 2>&1 | grep '^{' > $O/${TAG}_h14_bench.json; stamp models
fi
if has lines; then
  timeout 300 python bench.py --steps 8 --warmup 2 --deterministic $QUIET --no-roofline > $O/${TAG}_bench_det.log 2>&1
  timeout 300 python bench.py --steps 8 --warmup 2 --native-comm --native-allreduce $QUIET --no-roofline > $O/${TAG}_bench_native_comm.log 2>&1
  timeout 300 python bench.py --steps 8 --warmup 2 --h2d $QUIET --no-roofline > $O/${TAG}_bench_h2d.log 2>&1
  timeout 300 python bench.py --steps 8 --warmup 2 --force-ddp $QUIET --no-roofline > $O/${TAG}_bench_ddp1.log 2>&1; stamp lines
fi
if has stagger; then  # start-stagger classes of the persistent NT GEMM (developer build, ocn_set_tuning key 3) on the four GELU / dGELU shapes, alternating
  for i in 1 2; do
    for m in 0 1 2; do OCN_LIB_PATH=$DEVLIB timeout 200 python tools/ab_nt.py --tuning 3=$m --only gelu --json $O/${TAG}_stagger.jsonl >> $O/${TAG}_stagger_mode$m.txt 2>&1; done
    OCN_LIB_PATH=$DEVLIB timeout 200 python tools/ab_nt.py --knob $((63 << 13)) --only gelu --json $O/${TAG}_stagger.jsonl >> $O/${TAG}_stagger_off.txt 2>&1
  done; stamp stagger
fi
if has abstagger; then  # whole step, developer library: start stagger as shipped vs off (knob 63 << 13 of the ablation mask = gemm variant 63 << 21), alternating
  for i in 1 2 3; do
    OCN_LIB_PATH=$DEVLIB timeout 300 python bench.py --steps 12 --warmup 3 --no-roofline $QUIET 2>&1 | grep '^{' >> $O/${TAG}_abstagger_on.json
    OCN_LIB_PATH=$DEVLIB timeout 300 python bench.py --steps 12 --warmup 3 --no-roofline $QUIET --gemm-variant $((63 << 21)) 2>&1 | grep '^{' >> $O/${TAG}_abstagger_off.json
  done; stamp abstagger
fi
if has shapes; then timeout 300 python tools/ab_nt.py --json $O/${TAG}_shapes.jsonl > $O/${TAG}_shapes.txt 2>&1; timeout 300 python tools/gemm_vendor_yardstick.py > $O/${TAG}_gemm_vs_vendor.txt 2>&1; stamp shapes; fi
if has geluform; then  # polynomial vs Abramowitz-Stegun GELU arithmetic under the whole-step parity tests at size (developer library)
  rm -f $O/parity_report.txt
  for f in poly as; do for c in h14 b32; do OCN_LIB_PATH=$DEVLIB timeout 400 python tools/parity_gelu_form_probe.py $f $c >> $O/${TAG}_geluform.log 2>&1; done; done
  cp $O/parity_report.txt $O/${TAG}_geluform_parity_report.txt; stamp geluform
fi
if has tnepi; then timeout 300 python tools/ab_tn_epilogue.py > $O/${TAG}_tn_epilogue.txt 2>&1; stamp tnepi; fi
if has accumloss; then timeout 300 python tools/accum_loss_cost.py > $O/${TAG}_accum_loss_cost.txt 2>&1; stamp accumloss; fi
if has abclock; then  # does the rocm-smi sampler thread cost step time?  alternating
  for i in 1 2 3; do
    timeout 300 python bench.py --steps 20 --warmup 5 $QUIET --no-roofline 2>&1 | grep '^{' >> $O/${TAG}_abclock_on.json
    timeout 300 python bench.py --steps 20 --warmup 5 $QUIET --no-roofline --no-clock-sample 2>&1 | grep '^{' >> $O/${TAG}_abclock_off.json
  done; stamp abclock
fi
if has attnocc; then OCN_LIB_PATH=$DEVLIB timeout 300 python tools/attn_bwd_occupancy_probe.py > $O/${TAG}_attn_occupancy.txt 2>&1; stamp attnocc; fi
if has band; then  # tile-walk band width (knob bits 8..12 of the ablation mask) on the four GELU / dGELU shapes
  for i in 1 2; do
    for b in 3 4 12; do OCN_LIB_PATH=$DEVLIB timeout 200 python tools/ab_nt.py --knob $((b << 8)) --only gelu --json $O/${TAG}_band.jsonl >> $O/${TAG}_band$b.txt 2>&1; done
  done; stamp band
fi
if has attn; then timeout 300 python tools/ab_attn_bwd.py > $O/${TAG}_ab_attn_bwd.txt 2>&1; stamp attn; fi
cd /tmp; export TMPDIR=/tmp
if has profmodels; then  # BASELINE configs 4 / 5: kernel traces of the same bench lines (as shipped: towers overlapped)
  Q2="--no-cpu-baseline --no-eager-baseline --no-dense-text-line --no-extra-lines --no-config-lines --no-roofline --serial-towers --no-wgrad-pair"
  timeout 500 rocprofv3 --kernel-trace --stats -d /tmp/prof_l14 -o t -- python $GRAFT_REPO_ROOT/bench.py --model ViT-L-14 --local-batch 2048 --grad-checkpointing --steps 2 --warmup 1 $Q2 > $O/${TAG}_l14_prof.log 2>&1
  python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find /tmp/prof_l14 -name "*.db" | head -1) > $O/${TAG}_l14_kernel_stats.txt 2>&1
  timeout 500 rocprofv3 --kernel-trace --stats -d /tmp/prof_h14 -o t -- python $GRAFT_REPO_ROOT/bench.py --model ViT-H-14 --siglip --local-batch 1024 --grad-checkpointing --steps 2 --warmup 1 $Q2 > $O/${TAG}_h14_prof.log 2>&1
  python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find /tmp/prof_h14 -name "*.db" | head -1) > $O/${TAG}_h14_kernel_stats.txt 2>&1; stamp profmodels
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
echo "end +$(( $(date +%s) - t0 )) s" >> $O/${TAG}_timeline.txt
