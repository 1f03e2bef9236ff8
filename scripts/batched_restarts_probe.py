"""Probe for trial-batched restarts: per-trial parameter gradients of T restarts from ONE batched
forward / double backward (torch.func.vmap(grad) over the trial axis, eval-mode BN) against T separate chains, both replayed
as hipGraphs.  The gradient-matching objective here is a plain torch cosine (not kernel A) so that only the victim-model part
is compared.  Prints one JSON line.

    python scripts/batched_restarts_probe.py [--model resnet18] [--trials 4] [--steps 30] [--device cuda:0]
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.func import functional_call, grad, vmap
from breaching_amd.attacker import use_affine_eval_batchnorm
from breaching_amd.cases import build_case, initial_candidate

p = argparse.ArgumentParser()
p.add_argument("--model", default="resnet18")
p.add_argument("--data", default="ImageNet")
p.add_argument("--trials", type=int, default=4)
p.add_argument("--steps", type=int, default=30)
p.add_argument("--device", default="cuda:0")
p.add_argument("--no-graph", action="store_true")
p.add_argument("--chunk", type=int, default=None, help="vmap chunk_size")
p.add_argument("--only", default=None, choices=["separate", "batched"], help="run one variant only (for a rocprofv3 trace of it)")
p.add_argument("--groups", type=int, default=1, help="batched variant: this many independent trial batches in flight on separate streams")
args = p.parse_args()
dev = torch.device(args.device)
on_gpu = dev.type == "cuda"
case = build_case(args.model, args.data, 1, device=dev, gradient_device=dev if on_gpu else None) if on_gpu else build_case(args.model, args.data, 1, device=dev)
model = use_affine_eval_batchnorm(case.model.to(dev).eval(), "addcmul")  # torch ops only: vmap has no rule for the HIP function
params = {k: v.detach() for k, v in model.named_parameters()}
buffers = {k: v.detach() for k, v in model.named_buffers()}
names = list(params)
data = [g.to(dev) for g in case.shared_data[0]["gradients"]]
labels = case.shared_data[0]["metadata"]["labels"].to(dev)
loss_fn = case.loss_fn
T = args.trials
X = torch.stack([initial_candidate(case.data_cfg, 1, trial=t).to(dev) for t in range(T)]).requires_grad_(True)  # [T,1,3,H,W]
dn = torch.sqrt(sum((d * d).sum() for d in data))


def cosine(grads, lead):  # grads: list of [lead..., *shape]; returns [lead] objective values
    dot = sum((g * d).flatten(lead).sum(-1) for g, d in zip(grads, data))
    rn = torch.sqrt(sum((g * g).flatten(lead).sum(-1) for g in grads))
    return 1 - dot / (rn * dn)


def separate(X):
    outs = []
    for t in range(T):
        x = X[t]
        loss = loss_fn(functional_call(model, (params_req, buffers), (x,)), labels)
        g = torch.autograd.grad(loss, list(params_req.values()), create_graph=True)
        outs.append(cosine(list(g), 0))
    total = torch.stack(outs).sum()
    return torch.autograd.grad(total, X)[0], torch.stack(outs).detach()


def batched(X):
    def loss_one(ps, x):
        return loss_fn(functional_call(model, (ps, buffers), (x,)), labels)

    G = vmap(grad(loss_one), in_dims=(None, 0), chunk_size=args.chunk)(params, X)  # dict of [T, *shape]
    values = cosine([G[k] for k in names], 1)
    return torch.autograd.grad(values.sum(), X)[0], values.detach()


params_req = {k: v.clone().requires_grad_(True) for k, v in params.items()}


def timed(fn):
    for _ in range(3):
        out = fn(X)
    graph = None
    if on_gpu and not args.no_graph:
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = fn(X)
        graph.replay()
    if on_gpu:
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if graph is not None:
            graph.replay()
        else:
            out = fn(X)
    if on_gpu:
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / args.steps * 1e3, out


def timed_groups(fn, groups):
    """`groups` independent batches of T trials, each replayed as its own hipGraph on its own side stream."""
    streams = [torch.cuda.Stream(dev) for _ in range(groups)]
    Xs = [torch.stack([initial_candidate(case.data_cfg, 1, trial=g * T + t).to(dev) for t in range(T)]).requires_grad_(True) for g in range(groups)]
    graphs = []
    for st, Xg in zip(streams, Xs):
        st.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(st):
            for _ in range(3):
                fn(Xg)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                fn(Xg)
            graphs.append(graph)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        for st, graph in zip(streams, graphs):
            with torch.cuda.stream(st):
                graph.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / args.steps * 1e3


if args.groups > 1:
    ms = timed_groups(batched, args.groups)
    print(json.dumps(dict(model=args.model, trials_per_batch=T, groups=args.groups, ms_per_round=round(ms, 3),
                          trial_iterations_per_s=round(args.groups * T / ms * 1e3, 1))))
    raise SystemExit(0)
if args.only is not None:
    ms, (gx, v) = timed(separate if args.only == "separate" else batched)
    print(json.dumps(dict(model=args.model, trials=T, variant=args.only, ms_per_round=round(ms, 3),
                          trial_iterations_per_s=round(T / ms * 1e3, 1))))
    raise SystemExit(0)
ms_sep, (gx_sep, v_sep) = timed(separate)
ms_bat, (gx_bat, v_bat) = timed(batched)
rel = float((gx_sep - gx_bat).abs().max() / gx_sep.abs().max())
print(json.dumps(dict(model=args.model, trials=T, graph=bool(on_gpu and not args.no_graph), separate_ms_per_round=round(ms_sep, 3),
                      batched_ms_per_round=round(ms_bat, 3), speedup=round(ms_sep / ms_bat, 3),
                      trial_iterations_per_s=dict(separate=round(T / ms_sep * 1e3, 1), batched=round(T / ms_bat * 1e3, 1)),
                      values_separate=v_sep.flatten().tolist(), values_batched=v_bat.flatten().tolist(), grad_rel_diff=rel)))
