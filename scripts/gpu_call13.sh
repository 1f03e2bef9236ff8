set -x
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_baseline_configs.py -m gpu -q -s -k "teacher or seethrough" 2>&1 | grep -v "amdgpu\|Warning\|run_backward" | tail -40 | cut -c1-250
