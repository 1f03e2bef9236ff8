"""Small, bounded target for rocprofv3 --pmc passes: kernel A forward / finalize / backward on a
synthetic gradient list of one BASELINE size, and kernel D on the 53 BatchNorm inputs of ResNet-50 at B = 8.  A handful of
launches each, nothing else on the GPU -- the full bench under --pmc WRITE_SIZE died inside rocprofv3 twice in round 2.

    rocprofv3 --pmc WRITE_SIZE -- python scripts/pmc_target.py --size resnet18|resnet50|bert|bn [--reps 6]
    ... --size mt_resnet50 | mt_bert : the multi-tensor kernels (axpy: read a, b, write out = 3 N 4 B; scale: 2 N 4 B) on that list (round 5)
    ... --size bneval_plain | bneval_tap : kernel E's backward launch over the same 53 activations without / with the DeepInversion
        term riding in it (round 4): same FETCH_SIZE / WRITE_SIZE = the prior's backward costs no traffic of its own
"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from breaching_amd import _lib
from breaching_amd.cases import ResNet, build_text_case
from breaching_amd.gm import GradientMatchPlan

p = argparse.ArgumentParser()
p.add_argument("--size", default="resnet18")
p.add_argument("--reps", type=int, default=6)
args = p.parse_args()
dev = torch.device("cuda:0")
lib = _lib.load()
gen = torch.Generator().manual_seed(0)
if args.size.startswith("bneval"):
    acts = []
    model = ResNet(50, 1000).to(dev).eval()
    bns = [m for m in model.modules() if isinstance(m, torch.nn.BatchNorm2d)]
    hooks = [m.register_forward_hook(lambda m, i, o: acts.append(i[0].detach().contiguous())) for m in bns]
    with torch.no_grad():
        model(torch.randn(8, 3, 224, 224, device=dev))
    st = _lib.current_stream_handle(dev)
    tap = args.size.endswith("tap")
    gout = torch.ones(1, device=dev)
    work = []
    for a, m in zip(acts, bns):
        B, C, hw = a.shape[0], a.shape[1], a.shape[2] * a.shape[3]
        S = lib.bh_bn_eval_slabs(B, C, hw)
        work.append((a, torch.randn_like(a), torch.empty_like(a), torch.empty(C, device=dev), torch.empty(C, device=dev),
                     torch.empty(2 * C * S, dtype=torch.float64, device=dev), torch.randn(2 * C, device=dev),
                     torch.rsqrt(m.running_var + m.eps).contiguous(), (m.running_mean * torch.rsqrt(m.running_var + m.eps)).contiguous(), m, B, C, hw))
    for _ in range(args.reps):
        for a, gy, gx, gw, gb, ws, coef, inv, mi, m, B, C, hw in work:
            _lib.check(lib.bh_bn_eval_bwd(_lib.ptr(gy), _lib.ptr(a), _lib.ptr(m.weight), _lib.ptr(inv), _lib.ptr(mi), _lib.ptr(gx), _lib.ptr(gw),
                                          _lib.ptr(gb), _lib.ptr(ws), _lib.ptr(coef if tap else None), _lib.ptr(gout if tap else None), None, None, None,
                                          B, C, hw, st), "bn_eval_bwd")
    torch.cuda.synchronize()
    print(args.size, sum(a.numel() for a in acts), "elements per pass over the 53 BatchNorm inputs of ResNet-50 at B = 8")
elif args.size == "bn":
    from breaching_amd.priors import BnStatPlan

    acts = []
    model = ResNet(50, 1000).to(dev).eval()
    hooks = [m.register_forward_hook(lambda m, i, o: acts.append(i[0].detach().contiguous())) for m in model.modules()
             if isinstance(m, torch.nn.BatchNorm2d)]
    with torch.no_grad():
        model(torch.randn(8, 3, 224, 224, device=dev))
    bns = [m for m in model.modules() if isinstance(m, torch.nn.BatchNorm2d)]
    plan = BnStatPlan([a.shape for a in acts], [m.running_mean for m in bns], [m.running_var for m in bns], [1.0] * len(acts), dev)
    sums = torch.empty(2 * plan.n_pairs, dtype=torch.float64, device=dev)
    layer_values = torch.empty(plan.n_layers, dtype=torch.float64, device=dev)
    coef = torch.empty(2 * plan.n_channels, device=dev)
    total, ticket = torch.empty(1, device=dev), torch.zeros(1, dtype=torch.int32, device=dev)
    grad_flat = torch.empty(plan.flat_elems, device=dev)
    ptrs = plan.pointers(acts)
    st = _lib.current_stream_handle(dev)
    for _ in range(args.reps):
        _lib.check(lib.bh_bn_sums(plan.n_layers, ptrs, plan.hw_host, _lib.ptr(plan.layers_dev), _lib.ptr(plan.fwd_dev), plan.n_fwd,
                                  _lib.ptr(sums), 0, 0, st), "sums")
        _lib.check(lib.bh_bn_finalize(plan.n_layers, _lib.ptr(plan.layers_dev), _lib.ptr(sums), _lib.ptr(plan.running_mean),
                                      _lib.ptr(plan.running_var), _lib.ptr(coef), _lib.ptr(layer_values), _lib.ptr(total),
                                      _lib.ptr(ticket), 0, st), "finalize")
        _lib.check(lib.bh_bn_bwd(plan.n_layers, ptrs, plan.hw_host, _lib.ptr(plan.layers_dev), _lib.ptr(plan.bwd_dev), plan.n_bwd,
                                 _lib.ptr(coef), None, _lib.ptr(grad_flat), st), "bwd")
    torch.cuda.synchronize()
    print("bn", sum(a.numel() for a in acts), "elements", float(total))
elif args.size.startswith("mt_"):
    from breaching_amd.gm import ListLayout

    if args.size == "mt_bert":
        case = build_text_case(device=dev, full_size=True, seq_len=32)
        shapes = [tuple(g.shape) for g in case.shared_data[0]["gradients"]][1:]
        del case
    else:
        shapes = [tuple(q.shape) for q in ResNet(50, 1000).parameters()]
    layout = ListLayout(shapes, dev)
    a = [torch.randn(s, generator=gen).to(dev) for s in shapes]
    b = [torch.randn(s, generator=gen).to(dev) for s in shapes]
    out = layout.empty_flat()
    st = _lib.current_stream_handle(dev)
    for _ in range(args.reps):
        _lib.check(lib.bh_mt_axpy(layout.n_tensors, layout.pointers(a), layout.pointers(b), None, -0.0123, _lib.ptr(layout.chunks_dev),
                                  layout.n_chunks, layout.mt_bounds, _lib.ptr(out), st), "axpy")
        _lib.check(lib.bh_mt_scale(layout.n_tensors, layout.pointers(a), -0.0123, _lib.ptr(layout.chunks_dev), layout.n_chunks,
                                   layout.mt_bounds, _lib.ptr(out), st), "scale")
    torch.cuda.synchronize()
    print(args.size, sum(layout.numels), "elements: axpy 3 N 4 B, scale 2 N 4 B per launch")
else:
    if args.size == "bert":
        case = build_text_case(device=dev, full_size=True, seq_len=32)
        shapes = [tuple(g.shape) for g in case.shared_data[0]["gradients"]][1:]  # word-embedding gradient popped
        kind_name = "tag-euclidean"
    else:
        shapes = [tuple(q.shape) for q in ResNet(18 if args.size == "resnet18" else 50, 1000).parameters()]
        kind_name = "cosine-similarity" if args.size == "resnet18" else "euclidean"
    data = [torch.randn(s, generator=gen).to(dev) for s in shapes]
    rec = [torch.randn(s, generator=gen).to(dev) for s in shapes]
    plan = GradientMatchPlan(data)
    kind = _lib.GM_KINDS[kind_name]
    weights = torch.linspace(1, 0.1, len(shapes), device=dev) if kind_name == "tag-euclidean" else None
    for _ in range(args.reps):
        stats = plan.forward(kind, rec, 1.0, 0.1, 1e-7, weights)
        plan.backward(kind, rec, stats, None, weights)
    torch.cuda.synchronize()
    print(args.size, plan.total_elements, "elements", plan.n_rows, "rows", float(stats[0]))
