"""Diagnostic (not shipped): bisect which op's backward breaks hipGraph capture."""
import sys, subprocess, os
if len(sys.argv) == 1:
    for name in ["linear", "conv", "conv_nobias", "bn_eval", "relu", "maxpool", "ce", "conv_bn_relu", "adaptivepool", "flatten_linear", "conv_ce", "sum_only"]:
        r = subprocess.run([sys.executable, __file__, name], capture_output=True, text=True)
        out = (r.stdout + r.stderr)
        tag = "OK" if "replayed" in out else ("SEGV" if "Segmentation" in out or r.returncode < 0 else "ERR")
        print(f"{name:16s} {tag} rc={r.returncode}", [l for l in out.splitlines() if "Error" in l][:2], flush=True)
    sys.exit(0)
import torch, faulthandler
faulthandler.enable()
name = sys.argv[1]
nn = torch.nn
x = torch.randn(2, 3, 16, 16, device="cuda", requires_grad=True)
mods = {
    "linear": (nn.Sequential(nn.Flatten(), nn.Linear(768, 10)), None),
    "conv": (nn.Conv2d(3, 8, 3, padding=1), None),
    "conv_nobias": (nn.Conv2d(3, 8, 3, padding=1, bias=False), None),
    "bn_eval": (nn.BatchNorm2d(3).eval(), None),
    "relu": (nn.ReLU(), None),
    "maxpool": (nn.MaxPool2d(3), None),
    "adaptivepool": (nn.AdaptiveAvgPool2d(1), None),
    "ce": (nn.Sequential(nn.Flatten(), nn.Linear(768, 10)), "ce"),
    "conv_bn_relu": (nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.BatchNorm2d(8).eval(), nn.ReLU()), None),
    "flatten_linear": (nn.Sequential(nn.Flatten(), nn.Linear(768, 10)), None),
    "conv_ce": (nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.Flatten(), nn.Linear(2048, 10)), "ce"),
    "sum_only": (nn.Identity(), None),
}
m, lossk = mods[name]
m = m.cuda()
labels = torch.tensor([1, 2], device="cuda")
static = {}
def body():
    y = m(x)
    loss = torch.nn.functional.cross_entropy(y, labels) if lossk == "ce" else y.sum()
    params = tuple(m.parameters())
    g = torch.autograd.grad(loss, params + (x,))
    static["g"] = g
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): body()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    body()
graph.replay(); torch.cuda.synchronize()
print("replayed")
