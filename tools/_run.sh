cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
rm -f gpurun_out/parity_report.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r2u_tests.log
