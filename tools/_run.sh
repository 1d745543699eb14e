cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
rm -f gpurun_out/parity_report.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r2z_tests.log
cp gpurun_out/parity_report.txt gpurun_out/r2z_parity_report.txt 2>/dev/null
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-eager-baseline > gpurun_out/r2z_bench.log 2>&1
