"""TEST INFRASTRUCTURE ONLY -- CPU restatement (plain PyTorch ops, fp32) of the reference's optimisation attack.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module; the product
package ``breaching_amd`` never does and has no CPU path of its own.

Why it exists: ``/root/reference`` is not available on the GPU box, so the HIP path is checked there against this
restatement, which in turn is pinned (tests/test_oracle_pinning.py, ``-m "not gpu"``) against golden vectors produced by
running the *unmodified* reference in the build container (oracle/make_golden.py -> tests/golden/*.npz).

Every function names the reference lines it follows (paths relative to the reference checkout).  The victim model's
forward/backward and ``torch.optim`` are PyTorch on both sides (third-party arithmetic).
"""

import math
import time
from collections import defaultdict

import torch


# ---------------------------------------------------------------------------------------------------------------
# objectives  (breaching/attacks/auxiliaries/objectives.py)
# ---------------------------------------------------------------------------------------------------------------
def cosine_distance(rec, data):
    """objectives.py:183-196: 1 - <r,d> / (|r| |d|) accumulated tensor by tensor."""
    dot = rec[0].new_zeros(1)
    rec_sq = rec[0].new_zeros(1)
    data_sq = rec[0].new_zeros(1)
    for r, d in zip(rec, data):
        dot = dot + (r * d).sum()
        rec_sq = rec_sq + r.pow(2).sum()
        data_sq = data_sq + d.pow(2).sum()
    return 1 - dot / (rec_sq.sqrt() * data_sq.sqrt())


def squared_distance(rec, data):
    """objectives.py:89-95."""
    total = rec[0].new_zeros(1)
    for r, d in zip(rec, data):
        total = total + (r - d).pow(2).sum()
    return 0.5 * total


def l1_distance(rec, data):
    """objectives.py:158-166."""
    total = rec[0].new_zeros(1)
    for r, d in zip(rec, data):
        total = total + (r - d).abs().sum()
    return 0.5 * total


def tag_weights(n, scheme, like):
    """objectives.py:115-125."""
    if scheme == "linear":
        return torch.arange(n, 0, -1, dtype=like.dtype, device=like.device) / n
    if scheme == "exp":
        w = torch.arange(n, 0, -1, dtype=like.dtype, device=like.device).softmax(dim=0)
        return w / w[0]
    return like.new_ones(n)


def tag_distance(rec, data, tag_scale, scheme):
    """objectives.py:133-141."""
    weights = tag_weights(len(rec), scheme, rec[0])
    total = rec[0].new_zeros(1)
    for r, d, w in zip(rec, data, weights):
        total = total + (r - d).pow(2).sum() + tag_scale * w * (r - d).abs().sum()
    return 0.5 * total


def masked_cosine_distance(rec, data, mask_value=1e-6):
    """objectives.py:233-244."""
    dot, rec_sq, data_sq = 0.0, 0.0, 0.0
    for r, d in zip(rec, data):
        mask = d.abs() > mask_value
        dot = dot + (r * d * mask).sum()
        rec_sq = rec_sq + (r * mask).pow(2).sum()
        data_sq = data_sq + (d * mask).pow(2).sum()
    return 1 - dot / rec_sq.sqrt() / data_sq.sqrt()


def fast_cosine_distance(rec, data):
    """objectives.py:259-273 (norms detached)."""
    dot = rec[0].new_zeros(1)
    rec_sq = rec[0].new_zeros(1)
    data_sq = rec[0].new_zeros(1)
    for r, d in zip(rec, data):
        dot = dot + (r * d).sum()
        rec_sq = rec_sq + r.detach().pow(2).sum()
        data_sq = data_sq + d.detach().pow(2).sum()
    return 1 - dot / rec_sq.sqrt() / data_sq.sqrt()


def angular_distance(rec, data, fudge=1e-7):
    """objectives.py:210-214."""
    cosine = 1 - cosine_distance(rec, data)
    return torch.acos(cosine.clamp(min=-1 + fudge, max=1 - fudge)) / math.pi


def gradient_objective(kind, rec, data, cfg_objective=None):
    scale = 1.0 if cfg_objective is None else cfg_objective.get("scale", 1.0)
    if kind == "cosine-similarity":
        return cosine_distance(rec, data) * scale
    if kind == "euclidean":
        return squared_distance(rec, data) * scale
    if kind == "l1":
        return l1_distance(rec, data) * scale
    if kind == "tag-euclidean":
        return tag_distance(rec, data, cfg_objective.get("tag_scale", 0.1), cfg_objective.get("scale_scheme", "linear")) * scale
    if kind == "masked-cosine-similarity":
        return masked_cosine_distance(rec, data) * scale
    if kind == "fast-cosine-similarity":
        return fast_cosine_distance(rec, data) * scale
    if kind == "angular":
        return angular_distance(rec, data) * scale
    raise ValueError(f"Unknown objective type {kind} given.")


def multi_step_update(model, loss_fn, candidate, hyperparams):
    """objectives.py:48-72 (FedAvg unroll): `steps` local SGD steps on consecutive slices of the candidate batch, the shared
    quantity is p_local - p_server.  Plain list comprehensions, like the reference."""
    from torch.func import functional_call

    names = [n for n, _ in model.named_parameters()]
    server = [p for _, p in model.named_parameters()]
    buffers = dict(model.named_buffers())
    params = server
    seen, task_loss = 0, None
    for i in range(hyperparams["steps"]):
        data = candidate[seen : seen + hyperparams["data_per_step"]]
        seen = (seen + hyperparams["data_per_step"]) % candidate.shape[0]
        task_loss = loss_fn(functional_call(model, ({**dict(zip(names, params)), **buffers},), (data,)), hyperparams["labels"][i])
        grads = torch.autograd.grad(task_loss, params, create_graph=True)
        params = [p - hyperparams["lr"] * g for p, g in zip(params, grads)]
    return [p_local - p_server for p_local, p_server in zip(params, server)], task_loss


def pearlmutter_estimate(model, loss_fn, gradient_data, candidate, labels, kind="pearlmutter-loss", scale=1.0, eps=1e-3,
                         task_regularization=0.0, implementation="forward"):
    """objectives.py:279-493: objective value, task loss and the finite-difference estimate the reference ADDS to
    candidate.grad.  Out of place (the reference patches the live parameters and restores them, :326-328): with g = dL/dtheta,
    v = d objective / d g (residual :455, or the first-order cosine direction :468-477) and eps_n = eps / |g| (:346),
        forward  (:347-354)  scale * (dL/dx(theta + eps_n v) - dL/dx(theta)) / eps_n
        backward (:375-382)  scale * (dL/dx(theta) - dL/dx(theta - eps_n v)) / eps_n
        central  (:401-413)  scale * (dL/dx(theta + eps_n v / 2) - dL/dx(theta - eps_n v / 2)) / eps_n
        upwind   (:436-444)  scale * (max_0(dL/dx) * D_minus + min_0(dL/dx) * D_plus), max / min taken ALONG DIM 0
    plus task_regularization * dL/dx (:356)."""
    from torch.func import functional_call

    names = [n for n, _ in model.named_parameters()]
    params = [p for _, p in model.named_parameters()]
    buffers = dict(model.named_buffers())
    x = candidate.detach().clone().requires_grad_(True)
    task_loss = loss_fn(model(x), labels)
    *grads, dLdx = torch.autograd.grad(task_loss, (*params, x))
    if kind == "pearlmutter-loss":
        direction = [g - d for g, d in zip(grads, gradient_data)]
        value = 0.5 * scale * torch.stack([r.pow(2).sum() for r in direction]).sum()
    elif kind == "pearlmutter-cosine":
        dot = sum((g * d).sum() for g, d in zip(grads, gradient_data))
        gn = torch.stack([g.pow(2).sum() for g in grads]).sum().sqrt()
        dn = torch.stack([d.pow(2).sum() for d in gradient_data]).sum().sqrt()
        direction = [d / (-gn * dn) + g * (dot / (gn.pow(3) * dn)) for g, d in zip(grads, gradient_data)]
        value = scale * (1 - dot / (gn * dn))
    else:
        raise ValueError(kind)
    eps_n = eps / torch.stack([g.pow(2).sum() for g in grads]).sum().sqrt()

    def gradient_at(moved):
        xs = candidate.detach().clone().requires_grad_(True)
        loss = loss_fn(functional_call(model, ({**dict(zip(names, moved)), **buffers},), (xs,)), labels)
        return torch.autograd.grad(loss, xs)[0]

    def offset(base, mult):  # torch._foreach_add_(params, direction, alpha=mult * eps_n), out of place
        return [torch.add(p.detach(), v, alpha=float(mult * eps_n)) for p, v in zip(base, direction)]

    def central_pair():
        # the reference moves the live parameters by +eps_n/2 and then by -eps_n from THERE (:401-408), so the second point
        # carries the rounding of the first
        at_plus = offset(params, 0.5)
        return gradient_at(at_plus), gradient_at(offset(at_plus, -1.0))

    if implementation == "forward":
        estimate = (gradient_at(offset(params, 1.0)) - dLdx) / eps_n * scale
    elif implementation == "backward":
        estimate = (dLdx - gradient_at(offset(params, -1.0))) / eps_n * scale
    elif implementation == "central":
        plus, minus = central_pair()
        estimate = (plus - minus) / eps_n * scale
    elif implementation == "upwind":
        plus, minus = central_pair()
        d_plus, d_minus = (plus - dLdx) / eps_n, (dLdx - minus) / eps_n
        estimate = (torch.max(dLdx, 0)[0] * d_minus + torch.min(dLdx, 0)[0] * d_plus) * scale
    else:
        raise ValueError(implementation)
    return value.detach(), task_loss.detach(), estimate + task_regularization * dLdx


def orthogonality_penalty(x):
    """regularizers.py:170-178 (the value is not multiplied by `scale` there)."""
    if x.shape[0] == 1:
        return x.new_zeros(())
    B = x.shape[0]
    products = (x.unsqueeze(0) * x.unsqueeze(1)).pow(2).view(B, B, -1).mean(dim=2)
    return products.sum() - products.diagonal().sum()


# ---------------------------------------------------------------------------------------------------------------
# priors  (breaching/attacks/auxiliaries/regularizers.py)
# ---------------------------------------------------------------------------------------------------------------
def total_variation(x, scale=0.1, inner_exp=1, outer_exp=1, double_opponents=False, eps=1e-8):
    """regularizers.py:130-147 written with explicit forward differences instead of the grouped 3x3 convolution:
    dv[i,j] = x[i+1,j] - x[i,j], dh[i,j] = x[i,j+1] - x[i,j], zero beyond the border (padding=1)."""
    if double_opponents:
        x = torch.cat([x, x[:, 0:1] - x[:, 1:2], x[:, 0:1] - x[:, 2:3], x[:, 1:2] - x[:, 2:3]], dim=1)
    padded = torch.nn.functional.pad(x, (0, 1, 0, 1))
    dv = padded[:, :, 1:, :-1] - x
    dh = padded[:, :, :-1, 1:] - x
    terms = ((dv.abs() + eps).pow(inner_exp) + (dh.abs() + eps).pow(inner_exp)).pow(outer_exp)
    return terms.mean() * scale


def norm_penalty(x, scale=0.1, pnorm=2.0):
    """regularizers.py:196-197."""
    return 1 / pnorm * x.pow(pnorm).mean() * scale


class BnStatistic:
    """Forward hook on a BatchNorm2d input: mean / biased variance vs running statistics (math of
    deepinversion.py:93-101, restated from the definition)."""

    def __init__(self, module):
        self.value = None
        self.handle = module.register_forward_hook(self._hook)

    def _hook(self, module, inputs, output):
        x = inputs[0]
        mean = x.mean(dim=(0, 2, 3))
        var = ((x - mean[None, :, None, None]) ** 2).mean(dim=(0, 2, 3))
        self.value = (module.running_var.data - var).norm(2) + (module.running_mean.data - mean).norm(2)

    def close(self):
        self.handle.remove()


class DeepInversionPrior:
    """regularizers.py:203-230: scale * (10 * r_0 + sum_{l>0} r_l) over the BatchNorm2d layers."""

    def __init__(self, scale=0.1, first_bn_multiplier=10):
        self.scale, self.first_bn_multiplier, self.hooks = scale, first_bn_multiplier, []

    def initialize(self, models):
        self.close()
        self.hooks = [[BnStatistic(m) for m in model.modules() if isinstance(m, torch.nn.BatchNorm2d)] for model in models]

    def __call__(self, x):
        total = 0
        for hooks in self.hooks:
            total = total + sum(h.value * (self.first_bn_multiplier if i == 0 else 1.0) for i, h in enumerate(hooks))
        return self.scale * total

    def close(self):
        for hooks in self.hooks:
            for h in hooks:
                h.close()
        self.hooks = []


# ---------------------------------------------------------------------------------------------------------------
# optimiser / schedule  (breaching/attacks/auxiliaries/common.py)
# ---------------------------------------------------------------------------------------------------------------
class Warmup:
    """common.py:74-140 behaviour for multiplier 1.0: rate base*k/total for k <= total, then the wrapped scheduler
    (first its un-stepped value, then stepped once per call)."""

    def __init__(self, optimizer, total, after):
        self.optimizer, self.total, self.after = optimizer, total, after
        self.base = [g["initial_lr"] for g in optimizer.param_groups]
        self.k, self.finished = 0, False
        self._set([b * 0.0 for b in self.base])

    def _set(self, lrs):
        for g, lr in zip(self.optimizer.param_groups, lrs):
            g["lr"] = lr

    def step(self):
        if self.finished:
            self.after.step()
            return
        self.k += 1
        if self.k > self.total:
            self.finished = True
            self._set(self.after.get_last_lr())
        else:
            self._set([b * (float(self.k) / self.total) for b in self.base])


def make_optimizer(params, name, step_size, scheduler=None, warmup=0, max_iterations=10_000):
    """common.py:5-40."""
    key = name.lower()
    if key == "adam":
        opt = torch.optim.Adam(params, lr=step_size)
    elif key == "adam-safe":
        opt = torch.optim.Adam(params, lr=step_size, betas=(0.5, 0.99), eps=1e-4)
    elif key == "bert-adam":
        opt = torch.optim.AdamW(params, lr=step_size, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.01)
    elif key == "momgd":
        opt = torch.optim.SGD(params, lr=step_size, momentum=0.9, nesterov=True)
    elif key == "gd":
        opt = torch.optim.SGD(params, lr=step_size, momentum=0.0)
    elif key == "l-bfgs":
        opt = torch.optim.LBFGS(params, lr=step_size)
    else:
        raise ValueError(f"Invalid optimizer {name} given.")
    S = torch.optim.lr_scheduler
    if scheduler == "step-lr":
        sch = S.MultiStepLR(opt, milestones=[max_iterations // 2.667, max_iterations // 1.6, max_iterations // 1.142], gamma=0.1)
    elif scheduler == "cosine-decay":
        sch = S.CosineAnnealingLR(opt, max_iterations, eta_min=0.0)
    elif scheduler == "linear":
        sch = S.LambdaLR(opt, lambda s: max(0.0, float(max_iterations - s) / float(max(1, max_iterations))))
    else:
        sch = S.MultiStepLR(opt, milestones=[], gamma=1)
    if warmup > 0:
        sch = Warmup(opt, warmup, sch)
    return opt, sch


# ---------------------------------------------------------------------------------------------------------------
# the attack loop  (breaching/attacks/optimization_based_attack.py, base_attack.py)
# ---------------------------------------------------------------------------------------------------------------
def run_attack(model, loss_fn, cfg, server_payload, shared_data, initial_data=None, dryrun=False, max_iterations=None,
               timing=None, device="cpu"):
    """Restatement of ``OptimizationBasedAttacker.reconstruct`` for honest-server vision cases with labels provided
    (optimization_based_attack.py:63-218; base_attack.py:43-74, :169-212).  Returns (dict(data, labels), stats).

    ``device``: "cpu" is the oracle.  "cuda:0" runs the SAME statements with PyTorch-ROCm ops -- the same-GPU control of the
    HIP path (same victim-model arithmetic, torch's own elementwise / reduction kernels for the attack side); used by
    tests/control_same_gpu_torch.py and bench.py's `gpu_torch_baseline` leg, never by the product."""
    device = torch.device(device)
    import copy

    stats = defaultdict(list)
    meta = server_payload[0]["metadata"]
    shape = list(meta.shape)
    dm = torch.as_tensor(meta.mean, dtype=torch.float32, device=device)[None, :, None, None]
    ds = torch.as_tensor(meta.std, dtype=torch.float32, device=device)[None, :, None, None]

    models = []
    for payload, user in zip(server_payload, shared_data):
        m = copy.deepcopy(model).to(dtype=torch.float32, device=device)
        buffers = user["buffers"] if user["buffers"] is not None else payload["buffers"]
        if buffers is None:
            raise NotImplementedError("restatement covers the public/user-buffer (eval mode) cases only")
        m.eval()
        with torch.no_grad():
            for p, s in zip(m.parameters(), payload["parameters"]):
                p.copy_(s.to(device, torch.float32))
            for b, s in zip(m.buffers(), buffers):
                b.copy_(s.to(device, torch.float32))
        models.append(m)
    data_grads = [[g.to(device, torch.float32) for g in user["gradients"]] for user in shared_data]
    labels = shared_data[0]["metadata"]["labels"]
    if labels is None:
        raise NotImplementedError("restatement expects provided labels")
    labels = labels.clone().to(device)

    optim = cfg.optim
    n_iter = optim.max_iterations if max_iterations is None else max_iterations
    regs = cfg.regularization if cfg.regularization is not None else {}
    di = None
    if "deep_inversion" in regs and regs["deep_inversion"]["scale"] > 0:
        di = DeepInversionPrior(**regs["deep_inversion"])

    local_hyperparams = shared_data[0]["metadata"].get("local_hyperparams")

    def objective_and_task(candidate, kind, cfg_obj):
        total, task_total = 0, 0
        for m, dg in zip(models, data_grads):
            m.zero_grad()
            if kind.startswith("pearlmutter"):
                # objectives.py:322-329: the objective itself ADDS its finite-difference estimate to candidate.grad and returns
                # a value without graph; only used inside the closure (scoring uses the plain objectives)
                if local_hyperparams is not None:
                    raise ValueError("This loss is only implemented for local gradients so far.")  # :304-305
                value, task, estimate = pearlmutter_estimate(
                    m, loss_fn, dg, candidate, labels, kind=kind, scale=cfg_obj.get("scale", 1.0), eps=cfg_obj.get("eps", 1e-3),
                    task_regularization=cfg_obj.get("task_regularization", 0.0), implementation=cfg_obj.get("implementation", "forward"))
                with torch.no_grad():
                    candidate.grad += estimate
                total = total + value
                task_total = task_total + task
                continue
            if local_hyperparams is not None:  # FedAvg user: objectives.py:48-72
                grads, task = multi_step_update(m, loss_fn, candidate, local_hyperparams)
            else:
                task = loss_fn(m(candidate), labels)
                grads = torch.autograd.grad(task, tuple(m.parameters()), create_graph=True)
            obj = gradient_objective(kind, grads, dg, cfg_obj)
            if cfg_obj is not None and cfg_obj.get("task_regularization", 0.0) != 0:
                obj = obj + cfg_obj["task_regularization"] * task
            total = total + obj
            task_total = task_total + task.detach()
        return total, task_total

    scores, solutions = [], []
    for trial in range(cfg.restarts.num_trials):
        if di is not None:
            di.initialize(models)
        num_points = shared_data[0]["metadata"]["num_data_points"]
        if cfg.init == "randn":
            candidate = torch.randn([num_points, *shape]).to(device)
        elif cfg.init == "zeros":
            candidate = torch.zeros([num_points, *shape], device=device)
        else:
            raise NotImplementedError(cfg.init)
        if initial_data is not None:
            candidate = initial_data.detach().clone().to(device, torch.float32)
        candidate.requires_grad_(True)
        candidate.grad = torch.zeros_like(candidate)
        best = candidate.detach().clone()
        minimal = torch.as_tensor(float("inf"), device=device)
        optimizer, scheduler = make_optimizer([candidate], optim.optimizer, optim.step_size, optim.step_size_decay,
                                              optim.warmup, optim.max_iterations)
        t0 = time.time()
        for iteration in range(n_iter):
            def closure():
                optimizer.zero_grad(set_to_none=False)  # the reference's torch 1.10 zeroed the gradient (needed by :354)
                total, task = objective_and_task(candidate, cfg.objective.type, cfg.objective)
                for key in regs.keys():
                    r = regs[key]
                    if r["scale"] <= 0:
                        continue
                    if key == "total_variation":
                        total = total + total_variation(candidate, **r)
                    elif key == "norm":
                        total = total + norm_penalty(candidate, **r)
                    elif key == "deep_inversion":
                        total = total + di(candidate)
                    else:
                        raise NotImplementedError(key)
                if torch.is_tensor(total) and total.requires_grad:  # optimization_based_attack.py:164-165
                    total.backward(inputs=candidate)
                with torch.no_grad():
                    if optim.langevin_noise > 0:
                        candidate.grad += optim.langevin_noise * optimizer.param_groups[0]["lr"] * torch.randn_like(candidate.grad)
                    if optim.grad_clip is not None:
                        gn = candidate.grad.norm()
                        if gn > optim.grad_clip:
                            candidate.grad.mul_(optim.grad_clip / (gn + 1e-6))
                    if optim.signed is not None:
                        if optim.signed == "soft":
                            f = 1 - iteration / optim.max_iterations
                            candidate.grad.mul_(f).tanh_().div_(f)
                        elif optim.signed == "hard":
                            candidate.grad.sign_()
                return total

            value = optimizer.step(closure)
            scheduler.step()
            with torch.no_grad():
                if optim.boxed:
                    candidate.data = torch.max(torch.min(candidate, (1 - dm) / ds), -dm / ds)
                if value < minimal:
                    minimal = value.detach()
                    best = candidate.detach().clone()
            if not torch.isfinite(value):
                break
            stats[f"Trial_{trial}_Val"].append(value.item())
            if dryrun:
                break
        if timing is not None:
            timing.append(time.time() - t0)
        solutions.append(best)
        scoring = cfg.restarts.scoring
        if scoring not in ("euclidean", "cosine-similarity"):
            raise NotImplementedError(scoring)
        score, _ = objective_and_task(best, scoring, None)
        score = score.detach()
        scores.append(score if score.isfinite() else torch.as_tensor(float("inf"), device=device))
    if di is not None:
        di.close()
    scores = torch.stack([s.reshape(()) for s in scores])
    opt_val, opt_idx = torch.min(scores, dim=0)
    stats["opt_value"] = opt_val.item()
    solution = solutions[opt_idx] if opt_val.isfinite() else torch.zeros_like(solutions[opt_idx])
    return dict(data=solution, labels=labels), stats
