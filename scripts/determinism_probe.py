"""Which part of an attack iteration is not bit-reproducible from run to run on this box?

The in-flight / pool / graph-vs-eager tests compare runs of our own attack bit for bit when a control run shows that the box
reproduces itself.  On one MI355X box the `legacy` family on a 2-image ConvNet did not (pixels with near-zero gradient follow
rounding through Adam's normalisation, so any ulp-level nondeterminism shows there, while the losses agreed to 1e-4).  This
probe separates the suspects: the victim's parameter gradient alone (MIOpen's backward-weight kernels may use atomics), and
the attack with each prior switched on by itself, for 1 and 2 images, default and deterministic MIOpen algorithms.

    python scripts/determinism_probe.py  ->  JSON lines
"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import breaching_amd
from breaching_amd.cases import build_case

dev = torch.device("cuda:0")
setup = dict(device=dev, dtype=torch.float)


def attack(case, over, seed=3):
    cfg = breaching_amd.get_attack_config("legacy", over)
    att = breaching_amd.prepare_attack(case.model, case.loss_fn, cfg, setup)
    torch.manual_seed(seed)
    shared = [dict(gradients=list(d["gradients"]), buffers=d["buffers"], metadata=dict(d["metadata"])) for d in case.shared_data]
    rec, stats = att.reconstruct(case.server_payload, shared, {})
    return rec["data"].detach().clone(), list(stats["Trial_0_Val"])


for deterministic in (False, True):
    torch.backends.cudnn.deterministic = deterministic
    for images in (1, 2):
        case = build_case("convnet", "CIFAR10", images, device=dev, provide_buffers=True)
        x = torch.randn(images, 3, 32, 32, device=dev, requires_grad=True)
        labels = case.shared_data[0]["metadata"]["labels"]
        grads = []
        for _ in range(3):
            loss = case.loss_fn(case.model(x), labels)
            g = torch.autograd.grad(loss, list(case.model.parameters()), create_graph=True)
            (gx,) = torch.autograd.grad(sum((t * t).sum() for t in g), x)
            grads.append((torch.cat([t.detach().flatten() for t in g]), gx.detach()))
        print(json.dumps(dict(what="victim gradients alone (stock torch modules)", images=images, miopen_deterministic=deterministic,
                              parameter_gradient_bit_identical=all(torch.equal(grads[0][0], o[0]) for o in grads[1:]),
                              second_order_input_gradient_bit_identical=all(torch.equal(grads[0][1], o[1]) for o in grads[1:]))), flush=True)
        base = ["optim.max_iterations=14", "optim.callback=7", "init=randn"]
        variants = {
            "objective only": base + ["regularization.total_variation.scale=0", "regularization.features.scale=0", "regularization.deep_inversion.scale=0"],
            "+ total variation": base + ["regularization.features.scale=0", "regularization.deep_inversion.scale=0"],
            "+ features": base + ["regularization.total_variation.scale=0", "regularization.deep_inversion.scale=0"],
            "+ deep inversion": base + ["regularization.total_variation.scale=0", "regularization.features.scale=0", "regularization.deep_inversion.scale=0.001"],
            "all three": base + ["regularization.deep_inversion.scale=0.001"],
        }
        for name, over in variants.items():
            a, b = attack(case, over), attack(case, over)
            diff = (a[0] - b[0]).abs()
            print(json.dumps(dict(what=f"legacy attack, {name}", images=images, miopen_deterministic=deterministic,
                                  history_bit_identical=a[1] == b[1], candidate_bit_identical=bool(torch.equal(a[0], b[0])),
                                  pixels_differing=int((diff > 0).sum()), max_pixel_difference=float(diff.max()))), flush=True)
