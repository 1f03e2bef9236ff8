"""Diagnostic (not shipped): is the objective discontinuous in x at ulp scale?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from breaching_amd.cases import build_case, initial_candidate
from oracle import restate

cpu = build_case("convnet", "CIFAR10", 1, device="cpu")
gpu = build_case("convnet", "CIFAR10", 1, device="cuda:0")
x0 = initial_candidate(cpu.data_cfg, 1)
dm = torch.as_tensor(cpu.data_cfg.mean)[None, :, None, None]; ds = torch.as_tensor(cpu.data_cfg.std)[None, :, None, None]

def obj_dx(case, x, dev):
    x = x.detach().clone().to(dev).requires_grad_(True)
    loss = case.loss_fn(case.model(x), case.shared_data[0]["metadata"]["labels"])
    g = torch.autograd.grad(loss, tuple(case.model.parameters()), create_graph=True)
    obj = restate.cosine_distance(g, case.shared_data[0]["gradients"])
    (dx,) = torch.autograd.grad(obj, x)
    return obj.item(), dx.detach().cpu(), loss.item()

xc = x0.clone().requires_grad_(True); xg = x0.clone().cuda().requires_grad_(True)
oc = torch.optim.Adam([xc], lr=0.1); og = torch.optim.Adam([xg], lr=0.1)
for it in range(4):
    occ, dxc, lc = obj_dx(cpu, xc, "cpu")
    ogg, dxg, lg = obj_dx(gpu, xg, "cuda:0")
    ocg, _, lcg = obj_dx(gpu, xc, "cuda:0")      # GPU model at CPU's x
    ogc, _, lgc = obj_dx(cpu, xg.cpu(), "cpu")  # CPU model at GPU's x
    d = (xc.detach() - xg.detach().cpu())
    nz = d.nonzero()
    print(f"it {it}: obj cpu@xc {occ:.8f} gpu@xg {ogg:.8f} gpu@xc {ocg:.8f} cpu@xg {ogc:.8f} | task loss {lc:.6f} {lg:.6f} | #x differ {len(nz)} maxdiff {d.abs().max().item():.3e}")
    for i in nz[:6]:
        i = tuple(i.tolist()); print("     ", i, xc.detach()[i].item(), xg.detach().cpu()[i].item())
    for x, o, dx in ((xc, oc, dxc), (xg, og, dxg)):
        tv = restate.total_variation(x, 0.2, 1, 1)
        (gtv,) = torch.autograd.grad(tv, x)
        x.grad = (dx.to(x.device) + gtv).sign()
        o.step()
        with torch.no_grad():
            x.data = torch.max(torch.min(x, ((1 - dm) / ds).to(x.device)), (-dm / ds).to(x.device))
