# Round 4, GPU call 14: kernel A cache policy (plain / non-temporal) at three list sizes, stand-alone and inside the attack loop.
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
timeout 400 python scripts/nt_loads_probe.py > $OUT/r4_cache_policy_probe.jsonl 2> $OUT/r4_cache_policy_probe.err; cut -c1-420 $OUT/r4_cache_policy_probe.jsonl; tail -2 $OUT/r4_cache_policy_probe.err
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "gm_" 2>&1 | tail -2
timeout 200 python scripts/config_runs.py --only 5 --its 400 > /dev/null 2>&1
for pol in 1 2 3 1 2; do
  timeout 200 python - <<PY 2>&1 | tail -1
import sys, time, torch
sys.path.insert(0, ".")
import breaching_amd
from breaching_amd.cases import build_text_case
dev = torch.device("cuda:0")
case = build_text_case(device=dev, full_size=True, seq_len=32)
cfg = breaching_amd.get_attack_config("tag", ["optim.max_iterations=600", "optim.callback=300", "impl.gm_cache_policy=$pol"])
att = breaching_amd.prepare_attack(case.model, case.loss_fn, cfg, dict(device=dev, dtype=torch.float))
torch.manual_seed(0); torch.cuda.synchronize(); t0 = time.perf_counter()
rec, stats = att.reconstruct(case.server_payload, case.shared_data, {})
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("BERT-base TAG, cache policy $pol: 600 iterations in %.2f s = %.1f it/s incl. start-up; last loss %.4f" % (dt, 600 / dt, stats["Trial_0_Val"][-1]))
PY
done
