"""Diagnostic (not shipped): which side is noisy? fp64 truth vs fp32 CPU vs fp32 GPU candidate gradient at x_k."""
import sys, os, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from breaching_amd.cases import build_case, initial_candidate
from oracle import restate

cpu = build_case("convnet", "CIFAR10", 1, device="cpu")
x0 = initial_candidate(cpu.data_cfg, 1)
dm = torch.as_tensor(cpu.data_cfg.mean)[None, :, None, None]; ds = torch.as_tensor(cpu.data_cfg.std)[None, :, None, None]
labels = cpu.shared_data[0]["metadata"]["labels"]
gd = cpu.shared_data[0]["gradients"]

def dx_of(model, x, gdata, dev, dtype):
    model = copy.deepcopy(model).to(dev, dtype)
    x = x.detach().clone().to(dev, dtype).requires_grad_(True)
    loss = torch.nn.functional.cross_entropy(model(x), labels.to(dev))
    g = torch.autograd.grad(loss, tuple(model.parameters()), create_graph=True)
    obj = restate.cosine_distance(g, [t.to(dev, dtype) for t in gdata])
    (dx,) = torch.autograd.grad(obj, x)
    return dx.detach().cpu().double(), [t.detach().cpu().double() for t in g]

xc = x0.clone().requires_grad_(True)
oc = torch.optim.Adam([xc], lr=0.1)
for it in range(4):
    t64, g64 = dx_of(cpu.model, xc, gd, "cpu", torch.float64)
    c32, gc = dx_of(cpu.model, xc, gd, "cpu", torch.float32)
    g32, gg = dx_of(cpu.model, xc, gd, "cuda:0", torch.float32)
    torch.set_num_threads(1); c32_1, _ = dx_of(cpu.model, xc, gd, "cpu", torch.float32); torch.set_num_threads(os.cpu_count())
    rel = lambda a, b: ((a - b).norm() / b.norm()).item()
    print(f"it {it}: |dx| rms {t64.pow(2).mean().sqrt().item():.3e}  relerr cpu32 {rel(c32, t64):.2e} gpu32 {rel(g32, t64):.2e} cpu32(1thr) {rel(c32_1, t64):.2e} | maxabs cpu {(c32-t64).abs().max().item():.2e} gpu {(g32-t64).abs().max().item():.2e} "
          f"| sign flips vs truth: cpu {(c32.sign()!=t64.sign()).sum().item()} gpu {(g32.sign()!=t64.sign()).sum().item()} | first-order grads relerr cpu {max(rel(a,b) for a,b in zip(gc,g64)):.2e} gpu {max(rel(a,b) for a,b in zip(gg,g64)):.2e}")
    tv = restate.total_variation(xc, 0.2, 1, 1)
    (gtv,) = torch.autograd.grad(tv, xc)
    xc.grad = (c32.float() + gtv).sign()
    oc.step()
    with torch.no_grad():
        xc.data = torch.max(torch.min(xc, (1 - dm) / ds), -dm / ds)
