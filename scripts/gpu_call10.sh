set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 150 python scripts/diag_group.py 24000 > gpurun_out/diag_group.log 2>&1; grep "Trial 0 \|its" gpurun_out/diag_group.log | tail -40 | cut -c1-200
timeout 100 python scripts/config_runs.py --only 5 2>&1 | grep "^configs" | cut -c1-400 > gpurun_out/r2_config5_run.log; cat gpurun_out/r2_config5_run.log
