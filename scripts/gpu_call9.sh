set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 240 python scripts/diag_group.py > gpurun_out/diag_group.log 2>&1; grep -v "amdgpu.ids" gpurun_out/diag_group.log | tail -40 | cut -c1-250
