"""Which autograd node / ATen op launches which kernel of the attack iteration?

rocprofv3's kernel trace names the ~700 dispatches of a ResNet-18 iteration but not who asked for them.  This runs a few
EAGER iterations of BASELINE configs[1] (same FusedTrial.step body the hipGraph replays) under torch.profiler and walks every
kernel launch up its CPU-side parents: nearest `aten::` op and nearest autograd node (`autograd::engine::evaluate_function:
XBackward0`), or "forward" when there is none.  Output: launches per iteration grouped by (phase, autograd node, aten op,
kernel family) -- where the fills, the gradient-accumulation adds, the copies and MIOpen's helper kernels come from.

    python scripts/op_attribution.py [--iters 4] [--model resnet18]  ->  JSON on stdout, table on stderr
"""
import argparse
import collections
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import torch  # noqa: E402
from gap_census import family  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=4)
    ap.add_argument("--model", default="resnet18")
    args = ap.parse_args()
    import breaching_amd
    from breaching_amd.attacker import FusedTrial
    from breaching_amd.cases import build_case, initial_candidate

    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    case = build_case(args.model, "ImageNet", 1, device=device, gradient_device=device)
    cfg = breaching_amd.get_attack_config("invertinggradients", ["impl.hip_graph=False"])
    attacker = breaching_amd.prepare_attack(case.model, case.loss_fn, cfg, dict(device=device, dtype=torch.float))
    rec_models, labels, _ = attacker.prepare_attack(case.server_payload, case.shared_data)
    attacker.objective.initialize(attacker.loss_fn, cfg.impl, None)
    for reg in attacker.regularizers:
        reg.initialize(rec_models, case.shared_data, labels)
    attacker.objective.prepare(rec_models, case.shared_data)
    x0 = initial_candidate(case.data_cfg, 1).to(device).requires_grad_(True)
    run = FusedTrial(attacker, [x0], labels, rec_models, case.shared_data)
    for _ in range(4):
        run.step()
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile

    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for _ in range(args.iters):
            run.step()
        torch.cuda.synchronize()
    counts = collections.Counter()
    for ev in prof.events():
        kernels = getattr(ev, "kernels", None)
        if not kernels:
            continue
        # only the innermost CPU event that owns the launch (runtime call or op): skip events whose children own kernels too
        if any(getattr(child, "kernels", None) for child in (ev.cpu_children or [])):
            continue
        aten, node = None, None
        cur = ev
        while cur is not None:
            name = cur.name
            if aten is None and name.startswith("aten::"):
                aten = name
            if node is None and "evaluate_function:" in name:
                node = name.split("evaluate_function:")[1].strip()
            cur = cur.cpu_parent
        # custom autograd.Function forwards show up under their class name
        if aten is None:
            cur = ev
            while cur is not None and aten is None:
                if not cur.name.startswith(("hip", "cuda")) and cur.name != ev.name:
                    aten = cur.name
                cur = cur.cpu_parent
        for k in kernels:
            counts[(node or "forward / non-autograd", aten or ev.name, family(k.name))] += 1
    n = args.iters
    rows = [dict(autograd_node=a, op=o, kernel_family=f, launches_per_iteration=round(c / n, 2)) for (a, o, f), c in counts.most_common()]
    by_family = collections.Counter()
    by_node = collections.Counter()
    for r in rows:
        by_family[r["kernel_family"]] += r["launches_per_iteration"]
        by_node[r["autograd_node"]] += r["launches_per_iteration"]
    out = dict(model=args.model, iterations=n, launches_per_iteration=round(sum(counts.values()) / n, 1),
               by_kernel_family={k: round(v, 1) for k, v in by_family.most_common()},
               by_autograd_node={k: round(v, 1) for k, v in by_node.most_common()}, rows=rows)
    print(json.dumps(out, indent=1))
    sys.stderr.write(f"{out['launches_per_iteration']} launches per iteration\n")
    for r in rows[:60]:
        sys.stderr.write(f"  {r['launches_per_iteration']:7.2f}  {r['autograd_node'][:44]:44s} {r['op'][:38]:38s} {r['kernel_family']}\n")


if __name__ == "__main__":
    main()
