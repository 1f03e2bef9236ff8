"""Diagnostic (not shipped): bisect ConvNet prefix whose backward breaks hipGraph capture."""
import sys, subprocess, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) == 1:
    for k in [1, 2, 3, 4, 7, 10, 19, 25, 28]:
        for extra in ([], ["nocudnn"]):
            r = subprocess.run([sys.executable, __file__, str(k)] + extra, capture_output=True, text=True)
            out = (r.stdout + r.stderr)
            tag = "OK" if "replayed" in out else ("SEGV" if "Segmentation" in out or r.returncode < 0 else "ERR")
            print(f"prefix {k:2d} {' '.join(extra):8s} {tag} rc={r.returncode}", [l for l in out.splitlines() if "Error" in l][:2], flush=True)
    sys.exit(0)
import torch, faulthandler
faulthandler.enable()
from breaching_amd.cases import ConvNet
if "nocudnn" in sys.argv: torch.backends.cudnn.enabled = False
k = int(sys.argv[1])
torch.manual_seed(0)
full = ConvNet(64, 10).eval().cuda()
m = torch.nn.Sequential(*list(full.model.children())[:k])
x = torch.randn(1, 3, 32, 32, device="cuda", requires_grad=True)
static = {}
def body():
    y = m(x)
    static["g"] = torch.autograd.grad(y.sum(), tuple(m.parameters()) + (x,))
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): body()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    body()
graph.replay(); torch.cuda.synchronize()
print("replayed")
