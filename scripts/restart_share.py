"""One GPU's share of BASELINE configs[3] through the public API with per-1000-iteration log lines: ResNet-18, four restarts in
flight, N iterations each (default 3000).  `python scripts/restart_share.py 24000` is the full-length share; its
"Trial 0" lines are profiles/r2_config4_share_steady_state.log (flat 1000 iterations / 8.45 s x 4 trials)."""
import sys, os, time, logging, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import breaching_amd
from breaching_amd.cases import build_case
logging.basicConfig(level=logging.INFO, format="%(asctime)s %(name)s %(message)s")
dev = torch.device("cuda:0")
case = build_case("resnet18", "ImageNet", 1, device=dev, gradient_device=dev)
for its in (int(sys.argv[1]) if len(sys.argv) > 1 else 3000,):
    cfg = breaching_amd.get_attack_config("invertinggradients", [f"optim.max_iterations={its}", "restarts.num_trials=4", "optim.callback=1000"])
    att = breaching_amd.prepare_attack(case.model, case.loss_fn, cfg, dict(device=dev, dtype=torch.float))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rec, stats = att.reconstruct(case.server_payload, case.shared_data, {})
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(json.dumps(dict(its=its, wall_s=round(dt, 2), it_per_s=round(4 * its / dt, 1), opt=stats["opt_value"])), flush=True)
