# GPU-box call: FETCH_SIZE / WRITE_SIZE of single GEMM shapes (separate passes): bash tools/pmc_one_gemm.sh TAG
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
run() {  # name epi M N K
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/p1g; timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/p1g -o p -- python $R/tools/gemm_one.py nt 0 $2 $3 $4 $5 4 > /dev/null 2>&1
    find /tmp/p1g -name "*counter_collection.csv" -exec cp {} /tmp/p1g_$c.csv \;
  done
  echo "== $1: epilogue $2, [$3 x $4 x $5]" >> $O/${TAG}_pmc_one_gemm.txt
  python $R/tools/pmc_stats.py /tmp/p1g_FETCH_SIZE.csv /tmp/p1g_WRITE_SIZE.csv 2>&1 | grep "gemm_\|calls" >> $O/${TAG}_pmc_one_gemm.txt
}
TAG=$1
run "image c_fc + GELU" 1 204800 3072 768
run "text c_fc + GELU (packed rows)" 1 176128 2048 512
run "image dGELU (dh2)" 3 204800 3072 768
run "image plain, N = 3072" 0 204800 3072 768
