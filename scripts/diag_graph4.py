"""Diagnostic (not shipped): which difference to the passing case triggers the capture crash?"""
import sys, subprocess, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) == 1:
    base = ["ce", "paramsonly", "casemodel", "prework"]
    for flags in (base + ["imports"], base + ["imports", "loadlib"], base + ["caselabels"], base + ["x0init"], base + ["faultmode"], base + ["imports", "loadlib", "caselabels", "x0init", "faultmode", "hipobj"]):
        r = subprocess.run([sys.executable, __file__] + flags, capture_output=True, text=True)
        out = (r.stdout + r.stderr)
        tag = "OK" if "replayed" in out else ("SEGV" if "Segmentation" in out or r.returncode < 0 else "ERR")
        print(f"{' '.join(flags):45s} {tag} rc={r.returncode}", [l for l in out.splitlines() if "Error" in l][:2], flush=True)
    sys.exit(0)
import torch, faulthandler
faulthandler.enable()
from breaching_amd.cases import ConvNet, build_case
flags = sys.argv[1:]
if "imports" in flags:
    import breaching_amd.gm, breaching_amd.priors
    from oracle import restate
if "loadlib" in flags:
    from breaching_amd import _lib
    _lib.load()
if "hipobj" in flags:
    from breaching_amd.gm import HipCosineSimilarity
    hip = HipCosineSimilarity()
torch.manual_seed(0)
if "casemodel" in flags:
    case = build_case("convnet", "CIFAR10", 1, device="cuda:0", gradient_device="cuda:0")
    m = case.model
else:
    m = ConvNet(64, 10).eval().cuda()
x = torch.randn(1, 3, 32, 32, device="cuda", requires_grad=True)
labels = torch.tensor([3], device="cuda")
if "caselabels" in flags:
    labels = case.shared_data[0]["metadata"]["labels"]
if "x0init" in flags:
    from breaching_amd.cases import initial_candidate
    x = initial_candidate(case.data_cfg, 1).cuda().requires_grad_(True)
mode = {}
if "faultmode" in flags:
    mode = dict(capture_error_mode="global")
static = {}
def body():
    y = m(x)
    loss = torch.nn.functional.cross_entropy(y, labels) if "ce" in flags else y.sum()
    inputs = tuple(m.parameters()) + (() if "paramsonly" in flags else (x,))
    static["g"] = torch.autograd.grad(loss, inputs)
if "prework" in flags:
    body()
if "noside" in flags:
    for _ in range(3): body()
else:
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): body()
    torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph, **mode):
    body()
graph.replay(); torch.cuda.synchronize()
print("replayed")
