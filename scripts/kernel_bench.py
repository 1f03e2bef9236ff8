"""Per-kernel micro-benchmark on one MI355X: achieved GB/s of every HIP kernel at the BASELINE sizes.

    python scripts/kernel_bench.py > gpurun_out/kernel_bench.json

Kernel A is timed with hipExtLaunchKernelGGL start/stop events (same as bench.py); the other kernels with an event pair
around a burst of back-to-back launches on the stream (burst average, includes the ~1.5 us kernel boundary).
"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from breaching_amd import _lib, schedules
from breaching_amd.cases import ResNet, build_text_case
from breaching_amd.gm import GradientMatchPlan
from breaching_amd.priors import launch_tv_norm

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
        fin = sorted(t["fin"])[5:]
        bwd = sorted(t["bwd"])[5:]
        f, e, b = sum(fwd) / len(fwd), sum(fin) / len(fin), sum(bwd) / len(bwd)
        res[kind_name] = dict(fwd_us=round(f, 2), fwd_GBs=round(2 * n * 4 / f / 1e3, 1), finalize_us=round(e, 2),
                              stage_GBs=round(2 * n * 4 / (f + e) / 1e3, 1), bwd_us=round(b, 2), bwd_GBs=round(3 * n * 4 / b / 1e3, 1))
    # Forward launch right after a kernel REWROTE the whole reconstructed list (what autograd does in the loop): the producer's
    # dirty lines drain to HBM while kernel A streams -- the in-loop condition the warm figures above do not have.
    kind = _lib.GM_KINDS["cosine-similarity"]
    plan.enable_timing()
    for _ in range(25):
        torch._foreach_mul_(rec, 1.0)
        stats = plan.forward(kind, rec, 1.0, 0.1, 1e-7, None)
        torch._foreach_mul_(rec, 1.0)
        plan.backward(kind, rec, stats, None, None)
    t = plan.drain_timers()
    fwd, bwd = sorted(t["fwd"])[5:], sorted(t["bwd"])[5:]
    res["after_producer_rewrite"] = dict(fwd_us=round(sum(fwd) / len(fwd), 2), bwd_us=round(sum(bwd) / len(bwd), 2),
                                         note="each launch preceded by one multi-tensor launch that rewrites all of rec")
    # ... and right after 1 GiB of unrelated writes (the activations / gradients the victim's double backward leaves dirty in
    # L2 and the Infinity Cache in the loop): their write-back competes with kernel A's reads, whatever the list size.
    big = torch.empty(1 << 28, dtype=torch.float32, device=dev)
    plan.enable_timing()
    for _ in range(25):
        big.fill_(1.0)
        stats = plan.forward(kind, rec, 1.0, 0.1, 1e-7, None)
        big.fill_(2.0)
        plan.backward(kind, rec, stats, None, None)
    t = plan.drain_timers()
    fwd, bwd = sorted(t["fwd"])[5:], sorted(t["bwd"])[5:]
    res["after_1GiB_of_writes"] = dict(fwd_us=round(sum(fwd) / len(fwd), 2), bwd_us=round(sum(bwd) / len(bwd), 2))
    del big
    res.update(tensors=len(shapes), elements=n, chunks=plan.n_chunks, rows=plan.n_rows)
    # forward stage (reduction + finalize) against the persistent-grid size.  events: dispatch begin -> end of the forward
    # launch alone / event pair around the finalize launch; burst: back-to-back stage time incl. both launch boundaries
    sweep = {}
    for cap in (128, 256, 384, 512, 768, 1024, 2048):
        p2 = GradientMatchPlan(data, rows_cap=cap)
        p2.enable_timing()
        for _ in range(30):
            p2.forward(0, rec, 1.0, 0.0, 1e-7, None)
        t = p2.drain_timers()
        ev, fin = sorted(t["fwd"])[5:], sorted(t["fin"])[5:]
        us = burst(lambda: p2.forward(0, rec, 1.0, 0.0, 1e-7, None), reps=40)
        sweep[f"cap{cap}"] = dict(rows=p2.n_rows, fwd_event_us=round(sum(ev) / len(ev), 2), finalize_us=round(sum(fin) / len(fin), 2),
                                  stage_burst_us=round(us, 2))
    res["forward_stage_sweep_cosine"] = sweep
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
from breaching_amd.priors import BnStatPlan
bns = [m for m in model.modules() if isinstance(m, torch.nn.BatchNorm2d)]
acts = [a.contiguous() for a in acts]
plan = BnStatPlan([a.shape for a in acts], [m.running_mean for m in bns], [m.running_var for m in bns], [1.0] * len(acts), dev)
sums = torch.empty(2 * plan.n_pairs, dtype=torch.float64, device=dev)
layer_values = torch.empty(plan.n_layers, dtype=torch.float64, device=dev)
coef = torch.empty(2 * plan.n_channels, device=dev)
total_out = torch.empty(1, device=dev)
ticket = torch.zeros(1, dtype=torch.int32, device=dev)
grad_flat = torch.empty(plan.flat_elems, device=dev)
ptrs = plan.pointers(acts)
knobs = dict(grid_cap=0, load_depth=0, finalize_block=0)  # launch arguments (0 = library default)
def d_sums():
    _lib.check(lib.bh_bn_sums(plan.n_layers, ptrs, plan.hw_host, _lib.ptr(plan.layers_dev), _lib.ptr(plan.fwd_dev), plan.n_fwd,
                              _lib.ptr(sums), knobs["grid_cap"], knobs["load_depth"], _lib.current_stream_handle(dev)), "sums")
def d_fin():
    _lib.check(lib.bh_bn_finalize(plan.n_layers, _lib.ptr(plan.layers_dev), _lib.ptr(sums), _lib.ptr(plan.running_mean),
                                  _lib.ptr(plan.running_var), _lib.ptr(coef), _lib.ptr(layer_values), _lib.ptr(total_out),
                                  _lib.ptr(ticket), knobs["finalize_block"], _lib.current_stream_handle(dev)), "finalize")
def d_fwd():
    d_sums(); d_fin()
def d_bwd():
    _lib.check(lib.bh_bn_bwd(plan.n_layers, ptrs, plan.hw_host, _lib.ptr(plan.layers_dev), _lib.ptr(plan.bwd_dev), plan.n_bwd,
                             _lib.ptr(coef), None, _lib.ptr(grad_flat), _lib.current_stream_handle(dev)), "bwd")
us_s, us_fin = burst(d_sums, reps=20, warm=3), burst(d_fin, reps=20, warm=3)
us_f, us_b = burst(d_fwd, reps=20, warm=3), burst(d_bwd, reps=20, warm=3)
out["kernelD_resnet50_B8"] = dict(layers=len(acts), elements=total, fwd_items=plan.n_fwd, bwd_items=plan.n_bwd,
                                  sums_us=round(us_s, 1), finalize_us=round(us_fin, 1),
                                  fwd_us=round(us_f, 1), fwd_GBs=round(total * 4 / us_f / 1e3, 1),
                                  bwd_us=round(us_b, 1), bwd_GBs=round(2 * total * 4 / us_b / 1e3, 1),
                                  note="one sums launch + one finalize launch forward, one launch backward for all 53 layers "
                                       "(burst of back-to-back launches; 355.6 MB read forward, read + written backward)")
sweep = {}
for cap in (512, 1024, 2048, 4096, 1 << 20):
    knobs["grid_cap"] = cap
    sweep[str(cap)] = round(burst(d_sums, reps=20, warm=3), 1)
knobs["grid_cap"] = 0
out["kernelD_resnet50_B8"]["sums_us_by_grid_cap"] = sweep
fin = {}
for threads in (256, 512, 1024):
    knobs["finalize_block"] = threads
    fin[str(threads)] = dict(finalize_us=round(burst(d_fin, reps=20, warm=3), 1), stage_us=round(burst(d_fwd, reps=20, warm=3), 1))
knobs["finalize_block"] = 0
out["kernelD_resnet50_B8"]["finalize_by_block_threads"] = fin
print(json.dumps(out, indent=1))
