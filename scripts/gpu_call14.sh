set -x
cd $GRAFT_REPO_ROOT
timeout 200 python -m pytest tests/test_gpu_baseline_configs.py tests/test_gpu_attack.py -m gpu -q -k "shards or bert_base or seethrough or first_iterations" 2>&1 | grep -v "amdgpu\|Warning\|run_backward" | tail -30 | cut -c1-250
