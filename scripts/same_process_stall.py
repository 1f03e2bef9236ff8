"""Reproduction harness for DESIGN.md section 9, first open item: in ONE process, the 4-in-flight restarts of configs[3]
started right after the ResNet-50 / DeepInversion run of configs[2] did not finish within 12 minutes (round 2), while either
configuration alone is fine.  Runs configs[2] (200 iterations) and then configs[3] (4 x 2000 iterations) with INFO logging,
prints how every trial was executed (hipGraph replay / eager launches) and dumps all Python stacks if the second run is still
going after `--dump-after` seconds, so one bounded GPU call shows where it sits.

    timeout 400 python scripts/same_process_stall.py [--dump-after 120] [--skip-first]
"""
import argparse, faulthandler, json, logging, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import breaching_amd
from breaching_amd.cases import build_case, initial_candidate

parser = argparse.ArgumentParser()
parser.add_argument("--dump-after", type=float, default=120.0)
parser.add_argument("--skip-first", action="store_true", help="control: run only the 4-in-flight configuration")
args = parser.parse_args()
logging.basicConfig(level=logging.INFO, format="%(asctime)s %(name)s %(message)s")
dev = torch.device("cuda:0")
setup = dict(device=dev, dtype=torch.float)


def run(name, case, cfg, x0=None):
    attacker = breaching_amd.prepare_attack(case.model, case.loss_fn, cfg, setup)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rec, stats = attacker.reconstruct(case.server_payload, case.shared_data, {}, initial_data=x0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    its = sum(len(v) for k, v in stats.items() if k.startswith("Trial_"))
    print(json.dumps(dict(name=name, iterations=its, wall_s=round(dt, 2), it_per_s=round(its / dt, 1),
                          execution=attacker.last_trial_execution, allocated_GB=round(torch.cuda.memory_allocated(dev) / 2 ** 30, 2),
                          reserved_GB=round(torch.cuda.memory_reserved(dev) / 2 ** 30, 2))), flush=True)


if not args.skip_first:
    case = build_case("resnet50", "ImageNet", 8, device=dev, gradient_device=dev, provide_buffers=True)
    run("configs[2] ResNet-50 B=8 see-through + DeepInversion, 200 its", case,
        breaching_amd.get_attack_config("seethroughgradients", ["optim.max_iterations=200", "optim.callback=100"]),
        initial_candidate(case.data_cfg, 8))
    del case
faulthandler.dump_traceback_later(args.dump_after, repeat=True, file=sys.stderr)
case = build_case("resnet18", "ImageNet", 1, device=dev, gradient_device=dev)
run("configs[3] share: ResNet-18, 4 restarts in flight x 2000 its", case,
    breaching_amd.get_attack_config("invertinggradients", ["optim.max_iterations=2000", "restarts.num_trials=4", "optim.callback=500"]))
faulthandler.cancel_dump_traceback_later()
