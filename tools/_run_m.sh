cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/m_bench_noroof.log 2>&1
timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/m_bench_roof.log 2>&1
timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/m_bench_noroof2.log 2>&1
timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/m_bench_roof2.log 2>&1
