set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q 2>&1 | tail -15 > gpurun_out/r2_t_kernels.log; cat gpurun_out/r2_t_kernels.log
timeout 1500 python -m pytest tests/test_gpu_attack.py -x -q 2>&1 | tail -25 > gpurun_out/r2_t_attack.log; cat gpurun_out/r2_t_attack.log
timeout 600 python scripts/kernel_bench.py > gpurun_out/kernel_bench_r2a.json 2> gpurun_out/kernel_bench_r2a.err; tail -3 gpurun_out/kernel_bench_r2a.err
timeout 600 python bench.py > gpurun_out/bench_r2a.json 2> gpurun_out/bench_r2a.err; cat gpurun_out/bench_r2a.json; tail -3 gpurun_out/bench_r2a.err
