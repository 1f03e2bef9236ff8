"""Build-container anchor for bench.py's `cpu_baseline` (kind "port").

bench.py times oracle/restate.py on the GPU box's host cores, because the unmodified reference does not exist there.  This
script times BOTH -- the unmodified reference (through oracle/ref_shim.py) and the restatement -- on the same cores of the
build container, same workload (ResNet-18 / 224 x 224, invertinggradients), so the ratio ties the port's iterations/s to the
reference's.  Output: profiles/r6_cpu_baseline_anchor.json (round 2: r2_...).   python scripts/cpu_baseline_anchor.py [--iters 20]
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

parser = argparse.ArgumentParser()
parser.add_argument("--iters", type=int, default=20)
parser.add_argument("--threads", type=int, default=os.cpu_count())
args = parser.parse_args()
torch.set_num_threads(args.threads)

import breaching_amd
from breaching_amd.cases import build_case, initial_candidate
from oracle import make_golden as mg
from oracle import restate

case = build_case("resnet18", "ImageNet", 1)
x0 = initial_candidate(case.data_cfg, 1)


def timed_reference(its):
    cfg = mg._cfg("invertinggradients", [f"optim.max_iterations={its}", "optim.callback=100000"])
    t0 = time.perf_counter()
    mg._run_reference_attack(cfg, case, x0)
    return time.perf_counter() - t0


def timed_port(its):
    cfg = breaching_amd.get_attack_config("invertinggradients")
    timing = []
    restate.run_attack(case.model, case.loss_fn, cfg, case.server_payload, case.shared_data, initial_data=x0, max_iterations=its,
                       timing=timing)
    return timing[0]


out = dict(workload="ResNet-18 (1000 classes) 1x3x224x224, attack=invertinggradients", threads=args.threads, torch=torch.__version__,
           iterations=args.iters)
timed_reference(2), timed_port(2)  # warm-up: allocator, oneDNN primitives, TorchScript
# the reference call includes attacker construction and the final rescoring; subtract a 2-iteration call to isolate the loop.
# Five alternating repetitions (round 6: single measurements of the ratio scattered between 0.94 and 1.17 on the 8 shared cores of the
# build container): the medians and every repetition are reported.
ref_rates, port_rates = [], []
for _ in range(5):
    t_ref_short, t_ref_long = timed_reference(2), timed_reference(2 + args.iters)
    ref_rates.append(round(args.iters / (t_ref_long - t_ref_short), 3))
    port_rates.append(round(args.iters / timed_port(args.iters), 3))
median = lambda v: sorted(v)[len(v) // 2]  # noqa: E731
out["reference_iterations_per_s"], out["port_iterations_per_s"] = median(ref_rates), median(port_rates)
out["port_over_reference"] = round(out["port_iterations_per_s"] / out["reference_iterations_per_s"], 3)
out["repetitions"] = dict(reference=ref_rates, port=port_rates, ratio=[round(p / r, 3) for p, r in zip(port_rates, ref_rates)])
print(json.dumps(out, indent=1))
with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r6_cpu_baseline_anchor.json"), "w") as f:
    json.dump(out, f, indent=1)
