cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 400 python tools/gemm_bench.py 5 0,1,2,3 > gpurun_out/b_gemm.log 2>&1
cd /tmp; export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
timeout 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_$C -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/b_pmc_$C.log 2>&1
find /tmp/pmc_$C -name "*counter_collection.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/b_pmc_$C.csv \;
ls -laR /tmp/pmc_$C >> $GRAFT_REPO_ROOT/gpurun_out/b_pmc_$C.log
done
