"""Host-side restatement of the optimiser / step-size schedule selection of the reference.

reference: breaching/attacks/auxiliaries/common.py:5-40 (``optimizer_lookup``) and :74-140 (``GradualWarmupScheduler``).

The fused candidate step (kernel B) reads one row of four doubles per iteration, precomputed here on the host exactly
the way ``torch.optim.Adam`` computes its Python scalars (``bias_correction1 = 1 - beta1 ** step`` etc.), so the device
never needs the host during the loop and a captured hipGraph can replay any number of iterations.
"""

import math
from collections import Counter

import numpy as np

# common.py:6-12 -- the Adam-family optimisers served by the fused HIP step.
FUSED_OPTIMIZERS = {
    "adam": dict(betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, decoupled=False),
    "adam-safe": dict(betas=(0.5, 0.99), eps=1e-4, weight_decay=0.0, decoupled=False),
    "bert-adam": dict(betas=(0.9, 0.999), eps=1e-6, weight_decay=0.01, decoupled=True),
}
# common.py:13-18 -- served by torch.optim in the generic loop.
TORCH_OPTIMIZERS = ("momgd", "gd", "l-bfgs")


def optimizer_hparams(optim_name):
    """Hyper-parameters of a fused optimiser, ``None`` for the torch.optim-only ones, ValueError if unknown."""
    key = str(optim_name).lower()
    if key in FUSED_OPTIMIZERS:
        return dict(FUSED_OPTIMIZERS[key])
    if key in TORCH_OPTIMIZERS:
        return None
    raise ValueError(f"Invalid optimizer {optim_name} given.")  # common.py:20


def _base_schedule(step_size, scheduler, max_iterations):
    """Generator of the learning rate after 0, 1, 2, ... ``scheduler.step()`` calls (common.py:22-35)."""
    lr = float(step_size)
    if scheduler == "step-lr":
        # MultiStepLR with float milestones max_it // {2.667, 1.6, 1.142}, gamma 0.1; torch applies
        # lr *= gamma ** multiplicity whenever last_epoch hits a milestone -- including epoch 0 at construction.
        milestones = Counter([max_iterations // 2.667, max_iterations // 1.6, max_iterations // 1.142])
        epoch = 0
        while True:
            if epoch in milestones:
                lr = lr * 0.1 ** milestones[epoch]
            yield lr
            epoch += 1
    elif scheduler == "cosine-decay":
        # CosineAnnealingLR(T_max=max_iterations, eta_min=0), torch's recursive form
        t_max, base = max_iterations, float(step_size)
        yield lr
        epoch = 1
        while True:
            if (epoch - 1 - t_max) % (2 * t_max) == 0:
                lr = lr + base * (1 - math.cos(math.pi / t_max)) / 2
            else:
                lr = (1 + math.cos(math.pi * epoch / t_max)) / (1 + math.cos(math.pi * (epoch - 1) / t_max)) * lr
            yield lr
            epoch += 1
    elif scheduler == "linear":
        base = float(step_size)
        epoch = 0
        while True:
            yield base * max(0.0, float(max_iterations - epoch) / float(max(1, max_iterations)))
            epoch += 1
    else:  # MultiStepLR(milestones=[], gamma=1)
        while True:
            yield lr


def lr_sequence(step_size, scheduler=None, warmup=0, max_iterations=10_000, length=None):
    """Learning rate in effect at iteration k = 0 .. length-1 (``optimizer.step`` happens before ``scheduler.step``).

    With ``warmup > 0`` the reference wraps the schedule in GradualWarmupScheduler(multiplier=1.0): the rate is
    ``base * k / warmup`` for k <= warmup (so iteration 0 runs with lr = 0), the un-stepped inner schedule's value at
    k = warmup + 1, and the inner schedule stepped k - warmup - 1 times afterwards (common.py:93-108, :129-140).
    """
    length = max_iterations if length is None else length
    inner = _base_schedule(step_size, scheduler, max_iterations)
    out = []
    if warmup and warmup > 0:
        base = float(step_size)
        inner_first = next(inner)  # constructed (epoch 0) before the warm-up wrapper, never stepped during warm-up
        for k in range(length):
            if k <= warmup:
                out.append(base * (float(k) / warmup))
            elif k == warmup + 1:
                out.append(inner_first)
            else:
                out.append(next(inner))
    else:
        for _ in range(length):
            out.append(next(inner))
    return out


def adam_schedule_table(lrs, beta1, beta2, weight_decay=0.0):
    """Rows {lr / bias_correction1, sqrt(bias_correction2), 1 - lr * weight_decay, lr} as float64 [len(lrs), 4]."""
    table = np.empty((len(lrs), 4), dtype=np.float64)
    for k, lr in enumerate(lrs):
        step = float(k + 1)
        bias_correction1 = 1 - beta1**step
        bias_correction2 = 1 - beta2**step
        table[k, 0] = lr / bias_correction1
        table[k, 1] = bias_correction2**0.5
        table[k, 2] = 1 - lr * weight_decay
        table[k, 3] = lr
    return table
