"""BASELINE configs[4] (BERT-base TAG, tag.yaml untouched, 1 000 iterations): how far do two HIP runs part when the embedding start
moves by <= 16 ulp -- the HIP path's OWN reproducibility envelope, next to the reference's (tests/golden/attack_tag_bert_base_1000.npz:
nominal run and its 16-ulp twin).  AdamW with eps = 1e-6 turns gradient components at rounding level into full-size steps, so the
trajectory is sensitive to the last bit once the warm-up is over.

    python scripts/tag_twin_probe.py  ->  JSON lines (relative deviation of the loss history at selected iterations)
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import breaching_amd  # noqa: E402
from breaching_amd.cases import build_text_case, ulp_perturb  # noqa: E402

MARKS = [0, 10, 30, 50, 75, 100, 125, 150, 200, 300, 400, 500, 600, 700, 800, 900, 999]
dev = torch.device("cuda:0")


def run(perturb):
    case = build_text_case(device=dev, full_size=True, seq_len=32)
    cfg = breaching_amd.get_attack_config("tag", ["optim.callback=1000"])
    attacker = breaching_amd.prepare_attack(case.model, case.loss_fn, cfg, dict(device=dev, dtype=torch.float))
    original, calls = attacker._initialize_data, [0]
    gen = torch.Generator().manual_seed(124)

    def cpu_init(shape):  # draw on the CPU like the reference (tests/test_gpu_attack.py::_draw_on_cpu); optionally move the 2nd draw
        device = attacker.setup["device"]
        attacker.setup["device"] = torch.device("cpu")
        try:
            t = original(shape)
        finally:
            attacker.setup["device"] = device
        calls[0] += 1
        t = t.detach()
        if perturb and calls[0] == 2:
            t = ulp_perturb(t, 16, gen)
        t = t.to(device).requires_grad_(True)
        t.grad = torch.zeros_like(t)
        return t

    attacker._initialize_data = cpu_init
    torch.manual_seed(3)
    rec, stats = attacker.reconstruct(case.server_payload, case.shared_data, {})
    return np.asarray(stats["Trial_0_Val"], dtype=np.float64), float(stats["opt_value"]), rec["data"].cpu().numpy()


a, opt_a, tok_a = run(False)
b, opt_b, tok_b = run(False)
c, opt_c, tok_c = run(True)
rel = lambda x, y: (np.abs(x - y) / np.abs(y))  # noqa: E731
print(json.dumps(dict(what="HIP nominal run, repeated", bit_identical=bool(np.array_equal(a, b)), max_rel_dev=float(rel(b, a).max()))))
print(json.dumps(dict(what="HIP run from a start <= 16 ulp away vs HIP nominal", iterations=MARKS,
                      rel_dev=[float(f"{rel(c, a)[m]:.2e}") for m in MARKS], final_loss=[float(a[-1]), float(c[-1])], opt_value=[opt_a, opt_c],
                      tokens_equal=float((tok_a == tok_c).mean()))))
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "attack_tag_bert_base_1000.npz")
if os.path.exists(path):
    gold = np.load(path)
    ref, twin = gold["history"].astype(np.float64), gold["twin_history"].astype(np.float64)
    print(json.dumps(dict(what="reference: run from a start <= 16 ulp away vs its nominal run (CPU, fixture)", iterations=MARKS,
                          rel_dev=[float(f"{rel(twin, ref)[m]:.2e}") for m in MARKS], final_loss=[float(ref[-1]), float(twin[-1])],
                          opt_value=[float(gold["opt_value"]), float(gold["twin_opt_value"])])))
    print(json.dumps(dict(what="HIP nominal vs reference nominal", iterations=MARKS, rel_dev=[float(f"{rel(a, ref)[m]:.2e}") for m in MARKS],
                          final_loss=[float(ref[-1]), float(a[-1])], opt_value=[float(gold["opt_value"]), opt_a])))
