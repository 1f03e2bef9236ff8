set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q 2>&1 | tail -15 > gpurun_out/r2_t_kernels.log; cat gpurun_out/r2_t_kernels.log
timeout 900 python -m pytest tests/test_gpu_baseline_configs.py -x -q -s -k "shards or distribution" 2>&1 | tail -60 > gpurun_out/r2_t_baseline.log; cat gpurun_out/r2_t_baseline.log
timeout 600 python scripts/kernel_bench.py > gpurun_out/kernel_bench_r2b.json 2> gpurun_out/kernel_bench_r2b.err; tail -3 gpurun_out/kernel_bench_r2b.err
timeout 600 python bench.py > gpurun_out/bench_r2b.json 2> gpurun_out/bench_r2b.err; cat gpurun_out/bench_r2b.json; tail -3 gpurun_out/bench_r2b.err
