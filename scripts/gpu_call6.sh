set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 120 python scripts/diag_pearl.py > gpurun_out/diag_pearl.log 2>&1; cat gpurun_out/diag_pearl.log
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -400 > gpurun_out/r2_t_all.log; tail -30 gpurun_out/r2_t_all.log
