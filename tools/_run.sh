cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
rm -f gpurun_out/parity_report.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > gpurun_out/r2s_tests.log
cp gpurun_out/parity_report.txt gpurun_out/r2s_parity_report.txt 2>/dev/null
for i in 1 2; do timeout 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-eager-baseline 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('bf16 stream', d['ms_per_step'], d['config']['final_loss'])"; done >> gpurun_out/r2s_tests.log
OCN_GRAD_STREAM=fp32 timeout 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-eager-baseline 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('fp32 stream', d['ms_per_step'], d['config']['final_loss'])" >> gpurun_out/r2s_tests.log
