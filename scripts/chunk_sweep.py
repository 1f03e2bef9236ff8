"""Kernel A time vs chunk size (library variants built with -DBH_GM_CHUNK=...)."""
import os, subprocess, sys, json
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, os, json
sys.path.insert(0, %r)
import torch
from breaching_amd import _lib
from breaching_amd.cases import ResNet
from breaching_amd.gm import GradientMatchPlan
dev = torch.device("cuda:0")
out = {}
for depth in (18, 50):
    shapes = [tuple(p.shape) for p in ResNet(depth, 1000).parameters()]
    gen = torch.Generator().manual_seed(0)
    data = [torch.randn(s, generator=gen).to(dev) for s in shapes]
    rec = [torch.randn(s, generator=gen).to(dev) for s in shapes]
    plan = GradientMatchPlan(data); n = plan.total_elements
    plan.enable_timing()
    for _ in range(40):
        stats = plan.forward(0, rec, 1.0); plan.backward(0, rec, stats, None)
    t = plan.drain_timers(); f = sorted(t["fwd"])[5:-5]; b = sorted(t["bwd"])[5:-5]
    out[f"resnet{depth}"] = dict(chunks=plan.n_chunks, fwd_us=round(sum(f)/len(f),2), bwd_us=round(sum(b)/len(b),2))
print(json.dumps(out))
''' % root
for chunk in (2048, 4096, 8192, 16384):
    env = dict(os.environ)
    if chunk != 4096:
        env["BREACH_HIP_LIB"] = os.path.join(root, "breaching_amd", "lib", "variants", f"libbreach_hip_chunk{chunk}.so")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print(chunk, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:], flush=True)
