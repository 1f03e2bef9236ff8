"""Run-to-run spread of the fused loop on one GPU box: N eager-launch runs and N hipGraph-replay runs of the same attack from
the same start.  Reports, per iteration, the largest relative difference of the loss history,
of the candidate after the step and of the best-so-far copy WITHIN a mode (eager vs eager, graph vs graph) and BETWEEN the
modes, plus opt_value -- with MIOpen's default algorithms and again with torch.backends.cudnn.deterministic=True.

    python scripts/mode_spread.py [--runs 3] [--iterations 15] > gpurun_out/mode_spread.json
"""
import argparse, itertools, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import breaching_amd
from breaching_amd import attacker as attacker_module
from breaching_amd.cases import build_case, initial_candidate

parser = argparse.ArgumentParser()
parser.add_argument("--runs", type=int, default=3)
parser.add_argument("--iterations", type=int, default=15)
parser.add_argument("--model", default="convnet")
args = parser.parse_args()
dev = torch.device("cuda:0")
setup = dict(device=dev, dtype=torch.float)
dataset, n = ("CIFAR10", 1) if args.model in ("convnet", "smoothnet") else ("ImageNet", 1)
case = build_case(args.model, dataset, n, device=dev)
x0 = initial_candidate(case.data_cfg, n, seed=6)
over = ["objective.type=euclidean", "objective.scale=0.01", "optim.signed=soft", f"optim.max_iterations={args.iterations}",
        "restarts.scoring=euclidean", "optim.callback=5"]

trace = []
plain_step = attacker_module.FusedTrial.step


def recording_step(self):
    plain_step(self)
    trace.append((self.candidates[0].detach().clone(), self.slots[0]["best"].detach().clone()))


attacker_module.FusedTrial.step = recording_step


def one_run(graph):
    cfg = breaching_amd.get_attack_config("invertinggradients", over + [f"impl.hip_graph={graph}"])
    att = breaching_amd.prepare_attack(case.model, case.loss_fn, cfg, setup)
    torch.manual_seed(7)
    shared = [dict(gradients=list(d["gradients"]), buffers=d["buffers"], metadata=dict(d["metadata"])) for d in case.shared_data]
    trace.clear()
    rec, stats = att.reconstruct(case.server_payload, shared, {}, initial_data=x0)
    xs = torch.stack([t[0] for t in trace]).cpu().double().numpy()
    bests = torch.stack([t[1] for t in trace]).cpu().double().numpy()
    return dict(mode=att.last_trial_execution, hist=np.asarray(stats["Trial_0_Val"], dtype=np.float64), opt=float(stats["opt_value"]),
                xs=xs, bests=bests, rec=rec["data"].cpu().double().numpy())


def rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def spread(pairs):
    pairs = list(pairs)
    if not pairs:
        return None
    its = len(pairs[0][0]["hist"])
    return dict(
        hist_per_it=[max(abs(a["hist"][k] - b["hist"][k]) / abs(b["hist"][k]) for a, b in pairs) for k in range(its)],
        x_per_it=[max(rel(a["xs"][k], b["xs"][k]) for a, b in pairs) for k in range(its)],
        best_per_it=[max(rel(a["bests"][k], b["bests"][k]) for a, b in pairs) for k in range(its)],
        opt_value=max(abs(a["opt"] - b["opt"]) / abs(b["opt"]) for a, b in pairs),
        rec=max(rel(a["rec"], b["rec"]) for a, b in pairs),
    )


report = {}
for deterministic in (False, True):
    torch.backends.cudnn.deterministic = deterministic
    eager = [one_run(False) for _ in range(args.runs)]
    graph = [one_run(True) for _ in range(args.runs)]
    report["deterministic" if deterministic else "default"] = dict(
        modes=[eager[0]["mode"], graph[0]["mode"]],
        opt_values=dict(eager=[r["opt"] for r in eager], graph=[r["opt"] for r in graph]),
        last_hist=dict(eager=[r["hist"][-1] for r in eager], graph=[r["hist"][-1] for r in graph]),
        eager_vs_eager=spread(itertools.combinations(eager, 2)),
        graph_vs_graph=spread(itertools.combinations(graph, 2)),
        graph_vs_eager=spread(itertools.product(graph, eager)),
    )
print(json.dumps(dict(model=args.model, iterations=args.iterations, runs=args.runs, report=report), indent=1))
