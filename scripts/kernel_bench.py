"""Per-kernel micro-benchmark on one MI355X: achieved GB/s of every HIP kernel at the BASELINE sizes.

    python scripts/kernel_bench.py > gpurun_out/kernel_bench.json

Kernel A is timed with hipExtLaunchKernelGGL start/stop events (same as bench.py); the other kernels with an event pair
around a burst of back-to-back launches on the stream (burst average, includes the ~1.5 us kernel boundary).
"""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from breaching_amd import _lib, schedules
from breaching_amd.cases import ResNet, build_text_case
from breaching_amd.gm import GradientMatchPlan
from breaching_amd.priors import launch_tv_norm, ctypes_offset

dev = torch.device("cuda:0")
lib = _lib.load()
out = {}


def burst(fn, reps=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3  # us


def gm_case(name, shapes):
    gen = torch.Generator().manual_seed(0)
    data = [torch.randn(s, generator=gen).to(dev) for s in shapes]
    rec = [torch.randn(s, generator=gen).to(dev) for s in shapes]
    plan = GradientMatchPlan(data)
    n = plan.total_elements
    res = {}
    for kind_name in ("cosine-similarity", "euclidean", "tag-euclidean"):
        kind = _lib.GM_KINDS[kind_name]
        weights = torch.linspace(1, 0.1, len(shapes), device=dev) if kind_name == "tag-euclidean" else None
        plan.enable_timing()
        for _ in range(30):
            stats = plan.forward(kind, rec, 1.0, 0.1, 1e-7, weights)
            plan.backward(kind, rec, stats, None, weights)
        t = plan.drain_timers()
        fwd = sorted(t["fwd"])[5:]
        bwd = sorted(t["bwd"])[5:]
        f, b = sum(fwd) / len(fwd), sum(bwd) / len(bwd)
        res[kind_name] = dict(fwd_us=round(f, 2), fwd_GBs=round(2 * n * 4 / f / 1e3, 1), bwd_us=round(b, 2), bwd_GBs=round(3 * n * 4 / b / 1e3, 1))
    res.update(tensors=len(shapes), elements=n, chunks=plan.n_chunks)
    out[f"kernelA_{name}"] = res


torch.manual_seed(0)
gm_case("resnet18", [tuple(p.shape) for p in ResNet(18, 1000).parameters()])
gm_case("resnet50", [tuple(p.shape) for p in ResNet(50, 1000).parameters()])
bert = build_text_case(full_size=True, seq_len=8)
gm_case("bert_base", [tuple(g.shape) for g in bert.shared_data[0]["gradients"]][1:])
del bert

# kernel C: TV + norm, value and gradient
for B in (1, 8):
    x = torch.randn(B, 3, 224, 224, device=dev)
    g = torch.empty_like(x)
    parts = torch.empty(_lib.BH_PRIOR_MAX_GRID * 2, dtype=torch.float64, device=dev)
    for opp, p, q, tag in ((False, 1, 1, "p1q1"), (True, 2, 0.5, "p2q0.5_opp")):
        us = burst(lambda: launch_tv_norm(x, 0.2, p, q, 1e-8, opp, 1e-6, 2.0, grad_out=g, partials=parts))
        out[f"kernelC_B{B}_{tag}"] = dict(us=round(us, 2), algorithmic_bytes=2 * x.numel() * 4, GBs=round(2 * x.numel() * 4 / us / 1e3, 1))

# kernel B: candidate step
for B in (1, 8):
    n = B * 3 * 224 * 224
    x, g, gr, m, v, best = (torch.randn(n, device=dev) for _ in range(6))
    v.abs_()
    state = torch.zeros(_lib.BH_STATE_WORDS, dtype=torch.int32, device=dev)
    hist = torch.zeros(64, dtype=torch.float32, device=dev)
    loss = torch.ones(1, device=dev)
    sched = torch.from_numpy(schedules.adam_schedule_table([0.1] * 64, 0.9, 0.999)).to(dev)
    P = _lib.StepParams()
    P.n, P.plane, P.channels, P.boxed, P.sign_mode, P.max_iterations = n, 224 * 224, 3, 1, 1, 64
    for c in range(3):
        P.lo[c], P.hi[c] = -2.0, 2.0
    P.beta1, P.beta2, P.eps = 0.9, 0.999, 1e-8
    st = _lib.current_stream_handle(dev)
    lib.bh_state_reset(_lib.ptr(state), st)
    lib.bh_loss_commit(_lib.ptr(state), _lib.ptr(hist), 64, _lib.ptr(loss), None, 0, None, None, st)
    us = burst(lambda: lib.bh_candidate_step(_lib.ptr(state), _lib.ptr(sched), P, _lib.ptr(x), _lib.ptr(g), _lib.ptr(gr), None, _lib.ptr(m), _lib.ptr(v), _lib.ptr(best), st))
    out[f"kernelB_B{B}"] = dict(us=round(us, 2), algorithmic_bytes=7 * n * 4, GBs=round(7 * n * 4 / us / 1e3, 1))
    us = burst(lambda: lib.bh_loss_commit(_lib.ptr(state), _lib.ptr(hist), 64, _lib.ptr(loss), None, 0, None, None, st), reps=20)
    out["loss_commit_us"] = round(us, 2)

# kernel D: DeepInversion BN statistics over the 53 BN inputs of ResNet-50 at B=8 (355.6 MB)
acts = []
def hook(m, i, o):
    acts.append(i[0].detach())
model = ResNet(50, 1000).to(dev).eval()
hs = [m.register_forward_hook(hook) for m in model.modules() if isinstance(m, torch.nn.BatchNorm2d)]
with torch.no_grad():
    model(torch.randn(8, 3, 224, 224, device=dev))
for h in hs:
    h.remove()
total = sum(a.numel() for a in acts)
bufs = []
for a in acts:
    B_, C = a.shape[0], a.shape[1]
    HW = a.numel() // (B_ * C)
    S = lib.bh_bnstat_slabs(B_, C, HW)
    bufs.append(dict(x=a.contiguous(), B=B_, C=C, HW=HW, sums=torch.empty(C * S * 2, dtype=torch.float64, device=dev),
                     scratch=torch.empty(2 * C, dtype=torch.float64, device=dev), out=torch.empty(1 + 2 * C, device=dev),
                     rm=torch.zeros(C, device=dev), rv=torch.ones(C, device=dev), grad=torch.empty_like(a)))
def d_fwd():
    st = _lib.current_stream_handle(dev)  # the capture stream while a graph is being recorded
    for b in bufs:
        lib.bh_bnstat_sums(_lib.ptr(b["x"]), b["B"], b["C"], b["HW"], _lib.ptr(b["sums"]), st)
        lib.bh_bnstat_finalize(_lib.ptr(b["sums"]), b["B"], b["C"], b["HW"], _lib.ptr(b["rm"]), _lib.ptr(b["rv"]), _lib.ptr(b["out"]),
                               ctypes_offset(b["out"], 1), _lib.ptr(b["scratch"]), st)
def d_bwd():
    st = _lib.current_stream_handle(dev)
    for b in bufs:
        lib.bh_bnstat_bwd(_lib.ptr(b["x"]), b["B"], b["C"], b["HW"], ctypes_offset(b["out"], 1), None, _lib.ptr(b["grad"]), st)
def graphed(fn):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    return g.replay
us_f, us_b = burst(graphed(d_fwd), reps=10, warm=2), burst(graphed(d_bwd), reps=10, warm=2)
out["kernelD_resnet50_B8"] = dict(layers=len(acts), elements=total, fwd_us=round(us_f, 1), fwd_GBs=round(total * 4 / us_f / 1e3, 1),
                                  bwd_us=round(us_b, 1), bwd_GBs=round(2 * total * 4 / us_b / 1e3, 1),
                                  note="53 x (sums + finalize) launches forward, 53 launches backward, replayed from a hipGraph")
print(json.dumps(out, indent=1))
