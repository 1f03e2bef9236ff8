"""Diagnostic (not shipped): which part of the iteration breaks under hipGraph capture?"""
import sys, os, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from breaching_amd.cases import build_case, initial_candidate
from breaching_amd.gm import HipCosineSimilarity
from breaching_amd.priors import launch_tv_norm
from oracle import restate

which = sys.argv[1]
opts = sys.argv[3:] if len(sys.argv) > 3 else []
if "nomt" in opts: torch.autograd.set_multithreading_enabled(False)
if "nocudnn" in opts: torch.backends.cudnn.enabled = False
mode = "relaxed" if "relaxed" in opts else ("thread_local" if "tl" in opts else "global")
model_name = sys.argv[2] if len(sys.argv) > 2 else "convnet"
data = "CIFAR10" if model_name == "convnet" else "ImageNet"
case = build_case(model_name, data, 1, device="cuda:0", gradient_device="cuda:0")
x = initial_candidate(case.data_cfg, 1).cuda().requires_grad_(True)
labels = case.shared_data[0]["metadata"]["labels"]; gd = case.shared_data[0]["gradients"]
hip = HipCosineSimilarity()
static = {}

def body():
    if which == "kernels":   # only my kernels on static inputs
        g = static["g"]
        obj = hip.gradient_based_loss([t.detach().requires_grad_(True) for t in g], gd)
        static["o"] = obj
        launch_tv_norm(x.detach(), 0.2, 1, 1, 1e-8, False)
    elif which == "torch":   # model double backward with torch-only objective
        loss = case.loss_fn(case.model(x), labels)
        g = torch.autograd.grad(loss, tuple(case.model.parameters()), create_graph=True)
        obj = restate.cosine_distance(g, gd)
        (dx,) = torch.autograd.grad(obj, x)
        static["dx"] = dx
    elif which == "fwd":
        with torch.no_grad():
            static["y"] = case.model(x)
    elif which == "first":   # first-order only
        loss = case.loss_fn(case.model(x), labels)
        g = torch.autograd.grad(loss, tuple(case.model.parameters()), create_graph=False)
        static["g1"] = g
    else:                    # full: model + HIP objective
        loss = case.loss_fn(case.model(x), labels)
        g = torch.autograd.grad(loss, tuple(case.model.parameters()), create_graph=True)
        obj = hip.gradient_based_loss(list(g), gd)
        (dx,) = torch.autograd.grad(obj, x)
        static["dx"] = dx

loss = case.loss_fn(case.model(x), labels)
static["g"] = [t.detach() for t in torch.autograd.grad(loss, tuple(case.model.parameters()))]
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        body()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
print("warm ok", which, flush=True)
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph, capture_error_mode=mode):
    body()
print("captured", which, flush=True)
for _ in range(3):
    graph.replay()
torch.cuda.synchronize()
print("replayed", which, {k: (v.flatten()[:2].tolist() if torch.is_tensor(v) else None) for k, v in static.items() if k != "g"}, flush=True)
