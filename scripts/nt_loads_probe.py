"""Cache policy of kernel A's streaming accesses: plain vs non-temporal loads (`global_load_dwordx4 ... nt`) vs non-temporal loads + stores.

    python scripts/nt_loads_probe.py   ->  JSON lines (profiles/r4_cache_policy_probe.jsonl)

For the ResNet-18 (93.5 MB per forward, Infinity-Cache resident), ResNet-50 (204.5 MB) and BERT-base (688.6 MB, HBM) lists and each policy
(_lib.GM_CACHE_KEEP / STREAM / STREAM_ALL): the forward alone back to back; forward and backward alternating (each forward behind a backward
that has just written the gradient list); and the forward behind 1 GiB of unrelated writes (what autograd leaves in the loop).  Timed with
hipExtLaunchKernelGGL events.  The first version of this probe (build-time variants, profiles/r4_nt_loads_probe.jsonl) found the effect.
"""
import importlib.util, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
from breaching_amd import _lib
from breaching_amd.cases import ResNet
from breaching_amd.gm import GradientMatchPlan
dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(0)
big = torch.empty(1 << 28, dtype=torch.float32, device=dev)


def avg(v):
    v = sorted(v)[:-5]
    return sum(v) / len(v)


for name, shapes in (("resnet18", [tuple(p.shape) for p in ResNet(18, 1000).parameters()]), ("resnet50", [tuple(p.shape) for p in ResNet(50, 1000).parameters()]),
                     ("bert_base", bench.bert_base_gradient_shapes())):
    data = [torch.randn(s, generator=gen).to(dev) for s in shapes]
    rec = [torch.randn(s, generator=gen).to(dev) for s in shapes]
    for policy, label in ((_lib.GM_CACHE_KEEP, "keep"), (_lib.GM_CACHE_STREAM, "stream"), (_lib.GM_CACHE_STREAM_ALL, "stream_all"), (_lib.GM_CACHE_KEEP, "keep (again)")):
        plan = GradientMatchPlan(data, cache_policy=policy)
        n = plan.total_elements
        for _ in range(5):
            plan.backward(0, rec, plan.forward(0, rec, 1.0, 0.0, 1e-7, None), None, None)
        plan.enable_timing()
        for _ in range(40):
            plan.forward(0, rec, 1.0, 0.0, 1e-7, None)
        torch.cuda.synchronize()
        alone = avg(plan.drain_timers()["fwd"])
        plan.enable_timing()
        for _ in range(40):
            plan.backward(0, rec, plan.forward(0, rec, 1.0, 0.0, 1e-7, None), None, None)
        torch.cuda.synchronize()
        t = plan.drain_timers()
        behind, bwd = avg(t["fwd"]), avg(t["bwd"])
        plan.enable_timing()
        for _ in range(25):
            big.fill_(1.0)
            stats = plan.forward(0, rec, 1.0, 0.0, 1e-7, None)
            big.fill_(2.0)
            plan.backward(0, rec, stats, None, None)
        torch.cuda.synchronize()
        t = plan.drain_timers()
        gib_f, gib_b = avg(t["fwd"]), avg(t["bwd"])
        print(json.dumps(dict(list=name, elements=n, policy=label, fwd_alone_us=round(alone, 2), fwd_behind_bwd_us=round(behind, 2), bwd_us=round(bwd, 2),
                              fwd_after_1GiB_us=round(gib_f, 2), bwd_after_1GiB_us=round(gib_b, 2), pair_us=round(behind + bwd, 2), pair_after_1GiB_us=round(gib_f + gib_b, 2),
                              fwd_behind_bwd_frac=round(2 * n * 4 / behind / 1e3 / 8000, 4), fwd_after_1GiB_frac=round(2 * n * 4 / gib_f / 1e3 / 8000, 4),
                              bwd_frac=round(3 * n * 4 / bwd / 1e3 / 8000, 4))), flush=True)
