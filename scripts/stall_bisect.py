"""Bisect of the same-process slow-down: configs[3]'s share (ResNet-18, 4 restarts in flight)
runs at ~150 instead of ~470 iterations/s when an earlier attack ran in the same process.  One variant per process:

    python scripts/stall_bisect.py --first {none,convnet,resnet18,resnet50,resnet50-nodi} [--empty-cache] [--its 600]
                                   [--first-its 50] [--fresh-streams] [--repeat 1]

prints one JSON line: the variant, iterations/s of every pass of the 4-in-flight run, memory statistics.
"""
import argparse, gc, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import breaching_amd
from breaching_amd.cases import build_case, initial_candidate

parser = argparse.ArgumentParser()
parser.add_argument("--first", default="none")
parser.add_argument("--first-its", type=int, default=50)
parser.add_argument("--its", type=int, default=600)
parser.add_argument("--repeat", type=int, default=1)
parser.add_argument("--width", type=int, default=4)
parser.add_argument("--empty-cache", action="store_true")
parser.add_argument("--tag", default="")
parser.add_argument("--reset-capture-stream", action="store_true", help="forget torch.cuda.graph's shared capture stream before the 4-in-flight run")
parser.add_argument("--burn-streams", type=int, default=0, help="take this many streams from torch's pool before the 4-in-flight run")
parser.add_argument("--dot", default=None, help="dump the first trial's captured graph as DOT to this path prefix")
args = parser.parse_args()
dev = torch.device("cuda:0")
setup = dict(device=dev, dtype=torch.float)
out = dict(variant=vars(args), env={k: os.environ[k] for k in ("GPU_MAX_HW_QUEUES", "BREACH_HIP_GRAPH") if k in os.environ})


def run(case, cfg, x0=None):
    attacker = breaching_amd.prepare_attack(case.model, case.loss_fn, cfg, setup)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rec, stats = attacker.reconstruct(case.server_payload, case.shared_data, {}, initial_data=x0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    its = sum(len(v) for k, v in stats.items() if k.startswith("Trial_"))
    return dict(iterations=its, wall_s=round(dt, 2), it_per_s=round(its / dt, 1), execution=attacker.last_trial_execution,
                reserved_GB=round(torch.cuda.memory_reserved(dev) / 2 ** 30, 2))


if args.first == "convnet":
    case = build_case("convnet", "CIFAR10", 1, device=dev)
    out["first"] = run(case, breaching_amd.get_attack_config("invertinggradients", [f"optim.max_iterations={args.first_its}"]),
                       initial_candidate(case.data_cfg, 1, seed=6))
elif args.first == "resnet18":
    case = build_case("resnet18", "ImageNet", 1, device=dev, gradient_device=dev)
    out["first"] = run(case, breaching_amd.get_attack_config("invertinggradients", [f"optim.max_iterations={args.first_its}"]),
                       initial_candidate(case.data_cfg, 1))
elif args.first.startswith("resnet50"):
    case = build_case("resnet50", "ImageNet", 8, device=dev, gradient_device=dev, provide_buffers=True)
    over = [f"optim.max_iterations={args.first_its}", "optim.callback=100"]
    if args.first == "resnet50-nodi":
        over.append("regularization.deep_inversion.scale=0.0")
    if args.first == "resnet50-nograph":
        over.append("impl.hip_graph=False")
    out["first"] = run(case, breaching_amd.get_attack_config("seethroughgradients", over), initial_candidate(case.data_cfg, 8))
case = None
gc.collect()
if args.empty_cache:
    torch.cuda.empty_cache()
if args.reset_capture_stream:
    torch.cuda.graph.default_capture_stream = None
burned = [torch.cuda.Stream(dev) for _ in range(args.burn_streams)]
out["capture_stream"] = str(torch.cuda.graph.default_capture_stream)
case = build_case("resnet18", "ImageNet", 1, device=dev, gradient_device=dev)
cfg = breaching_amd.get_attack_config("invertinggradients", [f"optim.max_iterations={args.its}", f"restarts.num_trials={args.width}",
                                                             f"impl.trials_in_flight={args.width}", "optim.callback=1000"])
out["passes"] = [run(case, cfg) for _ in range(args.repeat)]
print(json.dumps(out), flush=True)
