set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 280 python scripts/config_runs.py --full --only 4 2>&1 | grep "^configs" | cut -c1-400 > gpurun_out/r2_config4_full.log; cat gpurun_out/r2_config4_full.log
