set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q -s 2>&1 | tail -120 > gpurun_out/r2_t_all.log; tail -100 gpurun_out/r2_t_all.log
timeout 300 python bench.py --no-span-timing > gpurun_out/bench_r2c.json 2> gpurun_out/bench_r2c.err; cut -c1-1500 gpurun_out/bench_r2c.json; tail -3 gpurun_out/bench_r2c.err
