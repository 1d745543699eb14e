# evidence at HEAD: full tests, the bench line, kernel traces (alone / as shipped), PMC traffic, MFMA busy, configs 4 / 5, step A/B against the round-2 tree
T=${1:-f1}
cd $GRAFT_REPO_ROOT
bash tools/gpu_call_r3.sh $T "tests abstep bench prof profov pmc mfma"
O=$GRAFT_REPO_ROOT/gpurun_out
python tools/pmc_stats.py $O/${T}_pmc_FETCH_SIZE.csv $O/${T}_pmc_WRITE_SIZE.csv $O/${T}_pmc_traffic.json $(cat .head_sha 2>/dev/null) > $O/${T}_pmc_hbm_traffic.txt 2>&1
python tools/pmc_mfma.py $O/${T}_pmc_SQ_VALU_MFMA_BUSY_CYCLES.csv > $O/${T}_pmc_mfma_util.txt 2>&1
Q="--no-cpu-baseline --no-eager-baseline --no-dense-text-line --no-roofline"
timeout 300 python bench.py --model ViT-H-14 --siglip --local-batch 1024 --grad-checkpointing --steps 3 --warmup 1 $Q > $O/${T}_h14_bench.log 2>&1
timeout 300 python bench.py --model ViT-L-14 --local-batch 2048 --grad-checkpointing --steps 3 --warmup 1 $Q > $O/${T}_l14_bench.log 2>&1
