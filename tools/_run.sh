cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
rm -f gpurun_out/parity_report.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > gpurun_out/r2q_tests.log
timeout 300 python __graft_entry__.py smoke >> gpurun_out/r2q_tests.log 2>&1
cp gpurun_out/parity_report.txt gpurun_out/r2q_parity_report.txt 2>/dev/null
