set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -s 2>&1 | tail -400 > gpurun_out/r2_t_all.log; tail -40 gpurun_out/r2_t_all.log | cut -c1-300
