"""Diagnostic (not shipped): step both implementations side by side and report where candidates part ways."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from breaching_amd.cases import build_case, initial_candidate
from breaching_amd.gm import HipCosineSimilarity
from breaching_amd.priors import launch_tv_norm
from oracle import restate

cpu = build_case("convnet", "CIFAR10", 1, device="cpu")
gpu = build_case("convnet", "CIFAR10", 1, device="cuda:0")
x0 = initial_candidate(cpu.data_cfg, 1)
dm = torch.as_tensor(cpu.data_cfg.mean)[None, :, None, None]; ds = torch.as_tensor(cpu.data_cfg.std)[None, :, None, None]

xc = x0.clone().requires_grad_(True); xg = x0.clone().cuda().requires_grad_(True)
oc = torch.optim.Adam([xc], lr=0.1); og = torch.optim.Adam([xg], lr=0.1)
hip = HipCosineSimilarity()
for it in range(8):
    outs = {}
    for tag, case, x, dev in (("cpu", cpu, xc, "cpu"), ("gpu", gpu, xg, "cuda:0")):
        m = case.model
        loss = case.loss_fn(m(x), case.shared_data[0]["metadata"]["labels"])
        g = torch.autograd.grad(loss, tuple(m.parameters()), create_graph=True)
        if tag == "cpu":
            obj = restate.cosine_distance(g, case.shared_data[0]["gradients"])
            tv = restate.total_variation(x, 0.2, 1, 1)
            (gm,) = torch.autograd.grad(obj, x, retain_graph=True)
            (gtv,) = torch.autograd.grad(tv, x)
            tvv = tv.detach()
        else:
            obj = hip.gradient_based_loss(list(g), case.shared_data[0]["gradients"])
            (gm,) = torch.autograd.grad(obj, x)
            gtv, partials, grid = launch_tv_norm(x.detach(), 0.2, 1, 1, 1e-8, False)
            tvv = partials[: grid * 2].view(grid, 2).sum(0)[0].float()
        outs[tag] = dict(obj=obj.item(), tv=float(tvv), gm=gm.detach().cpu(), gtv=gtv.detach().cpu())
        x.grad = (gm + gtv).detach().sign()
    a, b = outs["cpu"], outs["gpu"]
    tot_c, tot_g = a["gm"] + a["gtv"], b["gm"] + b["gtv"]
    mism = (tot_c.sign() != tot_g.sign())
    print(f"it {it}: obj {a['obj']:.8f} {b['obj']:.8f} tv {a['tv']:.8f} {b['tv']:.8f} x-diff-before {(xc.detach()-xg.detach().cpu()).abs().max().item():.3e} "
          f"n(x differs) {((xc.detach()-xg.detach().cpu()).abs()>1e-5).sum().item()} sign mismatches {mism.sum().item()} gtv mismatch {(a['gtv']!=b['gtv']).sum().item()} "
          f"gtv maxdiff {(a['gtv']-b['gtv']).abs().max().item():.3e}")
    if mism.any():
        idx = mism.nonzero()[:5]
        for i in idx:
            i = tuple(i.tolist())
            print("    at", i, "cpu gm/gtv", a["gm"][i].item(), a["gtv"][i].item(), "gpu gm/gtv", b["gm"][i].item(), b["gtv"][i].item(), "x", xc.detach()[i].item(), xg.detach().cpu()[i].item())
    oc.step(); og.step()
    with torch.no_grad():
        xc.data = torch.max(torch.min(xc, (1 - dm) / ds), -dm / ds)
        xg.data = torch.max(torch.min(xg, ((1 - dm) / ds).cuda()), (-dm / ds).cuda())
