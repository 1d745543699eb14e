cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -k "autocast or vith14_siglip_full_size_against" 2>&1 | tail -8 > gpurun_out/r2h_t_model.log
timeout 600 python -m pytest tests/test_ddp_gpu.py tests/test_bench_gpu.py -x -q 2>&1 | tail -12 > gpurun_out/r2h_t_ddp.log
timeout 400 python bench.py --steps 10 --warmup 2 > gpurun_out/r2h_bench.log 2>&1
timeout 300 python bench.py --steps 6 --warmup 2 --h2d --no-cpu-baseline --no-eager-baseline > gpurun_out/r2h_bench_h2d.log 2>&1
timeout 400 python bench.py --steps 2 --warmup 1 --accum-freq 8 --no-cpu-baseline --no-eager-baseline > gpurun_out/r2h_bench_accum8.log 2>&1
cp gpurun_out/parity_report.txt gpurun_out/r2h_parity_report.txt 2>/dev/null
