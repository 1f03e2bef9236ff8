"""Diagnostic (not shipped): HIP objective vs torch ops on the SAME device and SAME x along a trajectory."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from breaching_amd.cases import build_case, initial_candidate
from breaching_amd.gm import HipCosineSimilarity
from oracle import restate

gpu = build_case("convnet", "CIFAR10", 1, device="cuda:0")
x0 = initial_candidate(gpu.data_cfg, 1)
dm = torch.as_tensor(gpu.data_cfg.mean)[None, :, None, None].cuda(); ds = torch.as_tensor(gpu.data_cfg.std)[None, :, None, None].cuda()
xg = x0.clone().cuda().requires_grad_(True)
og = torch.optim.Adam([xg], lr=0.1)
hip = HipCosineSimilarity()
labels = gpu.shared_data[0]["metadata"]["labels"]
gd = gpu.shared_data[0]["gradients"]
for it in range(6):
    res = {}
    for tag in ("torch", "hip", "torch2"):
        loss = gpu.loss_fn(gpu.model(xg), labels)
        g = torch.autograd.grad(loss, tuple(gpu.model.parameters()), create_graph=True)
        obj = hip.gradient_based_loss(list(g), gd) if tag == "hip" else restate.cosine_distance(g, gd)
        (dx,) = torch.autograd.grad(obj, xg)
        res[tag] = (obj.item(), dx.detach().clone(), [t.detach().clone() for t in g])
    a, b, c = res["torch"], res["hip"], res["torch2"]
    rel = lambda u, v: ((u.double() - v.double()).norm() / v.double().norm()).item()
    print(f"it {it}: obj torch {a[0]:.8f} hip {b[0]:.8f} torch2 {c[0]:.8f} | dx rel hip-vs-torch {rel(b[1], a[1]):.2e} torch2-vs-torch {rel(c[1], a[1]):.2e} "
          f"| first-order rel hip-run vs torch-run {max(rel(p, q) for p, q in zip(b[2], a[2])):.2e} | sign mism hip {(b[1].sign()!=a[1].sign()).sum().item()} torch2 {(c[1].sign()!=a[1].sign()).sum().item()}")
    tv = restate.total_variation(xg, 0.2, 1, 1)
    (gtv,) = torch.autograd.grad(tv, xg)
    xg.grad = (a[1] + gtv).sign()
    og.step()
    with torch.no_grad():
        xg.data = torch.max(torch.min(xg, (1 - dm) / ds), -dm / ds)
