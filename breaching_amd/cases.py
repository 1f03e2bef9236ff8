"""Synthetic attack cases: victim models and the two dictionaries the attacker consumes.

The reference builds these through its case simulation (``cases.construct_model`` model_preparation.py:17-39,
``HonestServer.distribute_payload`` servers.py:138-147, ``UserSingleStep.compute_local_updates`` users.py:107-186), all
of which needs datasets on disk.  For the hot path only the *shape* of their output matters:

  server_payload = [dict(parameters=[...], buffers=[...] | None, metadata=<data cfg>)]
  shared_data    = [dict(gradients=[...], buffers=None | [...], metadata=dict(num_data_points, labels, local_hyperparams))]

Victim models stay plain ``torch.nn`` (PyTorch-ROCm runs their forward / backward / double backward).  The
architectures are the ones the reference attacks: ``ConvNet`` (model_preparation.py:437-479) and torchvision-shaped
ResNets (reference class resnets.py:45-237 with stem="standard").  Layer construction and initialisation order follow
the reference so that the same seed yields bitwise the same parameters (checked against golden checksums in tests).
"""

import torch

from .config import AttrDict, get_data_config


# -----------------------------------------------------------------------------------------------------------------
# models
# -----------------------------------------------------------------------------------------------------------------
class ConvNet(torch.nn.Module):
    """Eight conv-BN-ReLU stages (widths w,2w,2w,4w,4w,4w | pool | 4w,4w | pool) and a linear head."""

    def __init__(self, width=32, num_classes=10, num_channels=3):
        super().__init__()
        plan = [(num_channels, width), (width, 2 * width), (2 * width, 2 * width), (2 * width, 4 * width),
                (4 * width, 4 * width), (4 * width, 4 * width), "pool", (4 * width, 4 * width), (4 * width, 4 * width),
                "pool"]
        layers = []
        for item in plan:
            if item == "pool":
                layers.append(torch.nn.MaxPool2d(3))
            else:
                layers += [torch.nn.Conv2d(item[0], item[1], kernel_size=3, padding=1), torch.nn.BatchNorm2d(item[1]),
                           torch.nn.ReLU()]
        layers += [torch.nn.Flatten(), torch.nn.Linear(36 * width, num_classes)]
        self.model = torch.nn.Sequential(*layers)

    def forward(self, x):
        return self.model(x)


class _Residual(torch.nn.Module):
    """Basic (two 3x3) or bottleneck (1x1, 3x3, 1x1) residual unit, torchvision v1.5 stride placement."""

    def __init__(self, inplanes, planes, stride, bottleneck, project):
        super().__init__()
        out_planes = planes * (4 if bottleneck else 1)
        # the projection shortcut is created first (RNG order of the reference's _make_layer), registered last
        shortcut = None
        if project:
            shortcut = torch.nn.Sequential(
                torch.nn.Conv2d(inplanes, out_planes, kernel_size=1, stride=stride, bias=False),
                torch.nn.BatchNorm2d(out_planes),
            )
        if bottleneck:
            self.conv1 = torch.nn.Conv2d(inplanes, planes, kernel_size=1, bias=False)
            self.bn1 = torch.nn.BatchNorm2d(planes)
            self.conv2 = torch.nn.Conv2d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
            self.bn2 = torch.nn.BatchNorm2d(planes)
            self.conv3 = torch.nn.Conv2d(planes, out_planes, kernel_size=1, bias=False)
            self.bn3 = torch.nn.BatchNorm2d(out_planes)
        else:
            self.conv1 = torch.nn.Conv2d(inplanes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
            self.bn1 = torch.nn.BatchNorm2d(planes)
            self.conv2 = torch.nn.Conv2d(planes, planes, kernel_size=3, padding=1, bias=False)
            self.bn2 = torch.nn.BatchNorm2d(planes)
        self.bottleneck = bottleneck
        self.shortcut = shortcut

    def forward(self, x):
        out = torch.relu(self.bn1(self.conv1(x)))
        if self.bottleneck:
            out = torch.relu(self.bn2(self.conv2(out)))
            out = self.bn3(self.conv3(out))
        else:
            out = self.bn2(self.conv2(out))
        identity = x if self.shortcut is None else self.shortcut(x)
        return torch.relu(out + identity)


class ResNet(torch.nn.Module):
    """ImageNet ResNet-{18,34,50,101,152}: 7x7/2 stem, 3x3/2 max-pool, four stages, global average pool, linear."""

    _DEPTHS = {18: (False, (2, 2, 2, 2)), 34: (False, (3, 4, 6, 3)), 50: (True, (3, 4, 6, 3)),
               101: (True, (3, 4, 23, 3)), 152: (True, (3, 8, 36, 3))}

    def __init__(self, depth=18, num_classes=1000, num_channels=3):
        super().__init__()
        if depth not in self._DEPTHS:
            raise ValueError(f"Invalid depth {depth} given.")
        bottleneck, counts = self._DEPTHS[depth]
        expansion = 4 if bottleneck else 1
        self.stem = torch.nn.Sequential(
            torch.nn.Conv2d(num_channels, 64, kernel_size=7, stride=2, padding=3, bias=False),
            torch.nn.BatchNorm2d(64), torch.nn.ReLU(), torch.nn.MaxPool2d(kernel_size=3, stride=2, padding=1),
        )
        stages, inplanes, planes = [], 64, 64
        for stage, count in enumerate(counts):
            stride = 1 if stage == 0 else 2
            units = []
            for unit in range(count):
                s = stride if unit == 0 else 1
                project = unit == 0 and (s != 1 or inplanes != planes * expansion)
                units.append(_Residual(inplanes, planes, s, bottleneck, project))
                inplanes = planes * expansion
            stages.append(torch.nn.Sequential(*units))
            planes *= 2
        self.layers = torch.nn.Sequential(*stages)
        self.avgpool = torch.nn.AdaptiveAvgPool2d((1, 1))
        self.linear = torch.nn.Linear(inplanes, num_classes)
        for m in self.modules():  # kaiming-normal(fan_out) convs, unit BN scale (resnets.py:129-134)
            if isinstance(m, torch.nn.Conv2d):
                torch.nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, torch.nn.BatchNorm2d):
                torch.nn.init.constant_(m.weight, 1)
                torch.nn.init.constant_(m.bias, 0)

    def forward(self, x):
        x = self.layers(self.stem(x))
        return self.linear(torch.flatten(self.avgpool(x), 1))


class SmoothNet(torch.nn.Module):
    """Small kink-free classifier (softplus, average pooling): finite differences of its gradients are meaningful, which the
    parity tests of the Pearlmutter objectives need (on ReLU / max-pool nets the finite difference is dominated by kinks)."""

    def __init__(self, width=16, num_classes=10, num_channels=3):
        super().__init__()
        self.features = torch.nn.Sequential(
            torch.nn.Conv2d(num_channels, width, 3, padding=1), torch.nn.Softplus(), torch.nn.AvgPool2d(2),
            torch.nn.Conv2d(width, 2 * width, 3, padding=1), torch.nn.Softplus(), torch.nn.AvgPool2d(2),
            torch.nn.Conv2d(2 * width, 2 * width, 3, padding=1), torch.nn.Softplus(), torch.nn.AdaptiveAvgPool2d(2))
        self.head = torch.nn.Linear(8 * width, num_classes)

    def forward(self, x):
        return self.head(torch.flatten(self.features(x), 1))


def build_model(name, num_classes, seed=0):
    """Seeded random-init victim model in eval mode (public BN buffers: mean 0, var 1)."""
    torch.manual_seed(seed)
    key = name.lower()
    if key == "convnet":
        model = ConvNet(width=64, num_classes=num_classes)
    elif key == "smoothnet":
        model = SmoothNet(num_classes=num_classes)
    elif key.startswith("resnet"):
        model = ResNet(depth=int(key.replace("resnet", "")), num_classes=num_classes)
    else:
        raise ValueError(f"Unknown synthetic victim model {name}.")
    model.eval()
    return model


def parameter_checksum(model):
    """Order-sensitive fp64 checksum of all parameters and buffers (pins bitwise equality with the reference's init)."""
    total = 0.0
    for i, t in enumerate(list(model.parameters()) + list(model.buffers())):
        t = t.detach().double().flatten().cpu()
        if t.numel():
            total += float((t * torch.linspace(1.0, 2.0, t.numel(), dtype=torch.float64)).sum()) * (1 + 0.001 * i)
    return total


# -----------------------------------------------------------------------------------------------------------------
# payload / shared data
# -----------------------------------------------------------------------------------------------------------------
def synthetic_user_data(data_cfg, num_data_points, seed=1, kind="rand"):
    """Normalised synthetic images and fixed labels, drawn with a CPU generator so every device sees the same data."""
    gen = torch.Generator().manual_seed(seed)
    shape = (num_data_points, *data_cfg.shape)
    mean = torch.as_tensor(data_cfg.mean)[None, :, None, None]
    std = torch.as_tensor(data_cfg.std)[None, :, None, None]
    if kind == "rand":
        x = (torch.rand(shape, generator=gen) - mean) / std
    else:
        x = torch.randn(shape, generator=gen)
    labels = torch.randint(0, data_cfg.classes, (num_data_points,), generator=gen).sort()[0]
    return x, labels


def initial_candidate(data_cfg, num_data_points, seed=2, trial=0):
    gen = torch.Generator().manual_seed(seed + trial)
    return torch.randn((num_data_points, *data_cfg.shape), generator=gen)


def ulp_perturb(x, ulps, gen):
    """x moved by a random integer in [-ulps, ulps] units in the last place, element-wise (twin starting points: the
    parity fixtures measure how far the reference's own trajectories spread under such a perturbation)."""
    step = torch.nextafter(x.abs(), torch.full_like(x, float("inf"))) - x.abs()
    return x + step * torch.randint(-ulps, ulps + 1, x.shape, generator=gen).to(x.dtype)


def honest_payload(model, data_cfg, public_buffers=True):
    """servers.py:138-147 -- references to the live parameters, public buffers for an honest-but-curious server."""
    return [dict(parameters=[p for p in model.parameters()],
                 buffers=[b for b in model.buffers()] if public_buffers else None, metadata=data_cfg)]


def single_step_update(model, loss_fn, x, labels, provide_labels=True, provide_buffers=False):
    """users.py:107-186 for one local step: the plain gradient of the loss on the user's batch."""
    was_training = model.training
    if provide_buffers:
        model.train()
    loss = loss_fn(model(x), labels)
    grads = torch.autograd.grad(loss, tuple(model.parameters()))
    buffers = [b.clone().detach() for b in model.buffers()] if provide_buffers else None
    model.train(was_training)
    return [dict(gradients=[g.detach() for g in grads], buffers=buffers,
                 metadata=dict(num_data_points=x.shape[0], labels=labels if provide_labels else None,
                               local_hyperparams=None))]


def build_case(model_name="convnet", data_name="CIFAR10", num_data_points=1, device="cpu", seed_model=0, seed_data=1,
               provide_labels=True, provide_buffers=False, classes=None, gradient_device=None):
    """Everything a parity run needs.  ``gradient_device`` is where the user's gradient is computed (defaults to CPU so
    that HIP runs and CPU oracle runs observe bitwise the same gradient)."""
    data_cfg = get_data_config(data_name)
    if classes is not None:
        data_cfg.classes = classes
    model = build_model(model_name, data_cfg.classes, seed_model)
    loss_fn = torch.nn.CrossEntropyLoss()
    x_true, labels = synthetic_user_data(data_cfg, num_data_points, seed_data)
    gdev = torch.device(gradient_device or "cpu")
    user_model = model.to(gdev)
    shared = single_step_update(user_model, loss_fn, x_true.to(gdev), labels.to(gdev), provide_labels, provide_buffers)
    device = torch.device(device)
    model = model.to(device)
    for entry in shared:
        entry["gradients"] = [g.to(device) for g in entry["gradients"]]
        if entry["buffers"] is not None:
            entry["buffers"] = [b.to(device) for b in entry["buffers"]]
        if entry["metadata"]["labels"] is not None:
            entry["metadata"]["labels"] = entry["metadata"]["labels"].to(device)
    payload = honest_payload(model, data_cfg, public_buffers=True)
    return AttrDict(model=model, loss_fn=loss_fn, server_payload=payload, shared_data=shared,
                    true_user_data=dict(data=x_true, labels=labels), data_cfg=data_cfg)


def build_multi_query_case(queries=2, device="cpu", num_data_points=2, seed_model=0, seed_data=1, drift=0.01):
    """One user answering `queries` server queries (the reference's `num_queries`, servers.py:150-165 / users.py): the same
    private batch, a different model state per query (query 0 = the seeded ConvNet, query q > 0 = its parameters moved by
    seeded Gaussian noise of scale `drift`), one gradient list per query.  The attacker then sums the gradient-matching
    objective over the (model, gradient) pairs (optimization_based_attack.py:152-155)."""
    import copy

    base = build_case("convnet", "CIFAR10", num_data_points, device="cpu", seed_model=seed_model, seed_data=seed_data)
    x_true, labels = base.true_user_data["data"], base.true_user_data["labels"]
    device = torch.device(device)
    models, payloads, shared = [], [], []
    gen = torch.Generator().manual_seed(1000 + seed_model)
    for q in range(queries):
        model = copy.deepcopy(base.model)
        if q > 0:
            with torch.no_grad():
                for p in model.parameters():
                    p.add_(drift * torch.randn(p.shape, generator=gen))
        entry = single_step_update(model, base.loss_fn, x_true, labels, True, False)[0]
        entry["gradients"] = [g.to(device) for g in entry["gradients"]]
        entry["metadata"]["labels"] = entry["metadata"]["labels"].to(device)
        model = model.to(device)
        models.append(model)
        payloads.extend(honest_payload(model, base.data_cfg, public_buffers=True))
        shared.append(entry)
    return AttrDict(model=models[0].to(device), models=models, loss_fn=base.loss_fn, server_payload=payloads, shared_data=shared,
                    true_user_data=dict(data=x_true, labels=labels), data_cfg=base.data_cfg)


def psnr(reconstruction, truth, data_cfg):
    """Mean per-example PSNR on de-normalised, clamped images (analysis.py:228-229, metrics.py:122-130, factor=1)."""
    mean = torch.as_tensor(data_cfg.mean, dtype=torch.float32)[None, :, None, None]
    std = torch.as_tensor(data_cfg.std, dtype=torch.float32)[None, :, None, None]
    rec = torch.clamp(reconstruction.detach().float().cpu() * std + mean, 0, 1)
    ref = torch.clamp(truth.detach().float().cpu() * std + mean, 0, 1)
    mse = ((rec - ref) ** 2).reshape(rec.shape[0], -1).mean(dim=1)
    if bool((mse == 0).any()):
        return float("inf")
    return float((10 * torch.log10(1.0 / mse)).mean())


# -----------------------------------------------------------------------------------------------------------------
# text case (BASELINE config 5 family: BERT masked-LM + TAG joint attack)
# -----------------------------------------------------------------------------------------------------------------
class TokenModel(torch.nn.Module):
    """Uniform call interface around a HuggingFace masked-LM: integer inputs are token ids, floating inputs are taken as
    already-embedded tokens (what the attacker optimises after cutting off the embedding layer).  Same contract as the
    reference's ``HuggingFaceContainer`` (model_preparation.py:134-149)."""

    def __init__(self, model):
        super().__init__()
        self.model = model

    def forward(self, inputs):
        # The math backend of scaled_dot_product_attention is plain matmul/softmax ops: it supports the double backward
        # the attack needs, and (unlike transformers' "eager" mask helper, which builds a device scalar from a host value
        # on every call) it can be captured into a hipGraph.
        from torch.nn.attention import SDPBackend, sdpa_kernel

        with sdpa_kernel(SDPBackend.MATH):
            if inputs.dtype == torch.long:
                out = self.model(input_ids=inputs)
            else:
                out = self.model(inputs_embeds=inputs)
        return out["logits"]


class MaskedLMLoss(torch.nn.Module):
    """Cross entropy over flattened tokens for integer or soft (probability) targets (losses.py:29-42)."""

    def __init__(self, vocab_size):
        super().__init__()
        self.vocab_size = vocab_size
        self.ce = torch.nn.CrossEntropyLoss()

    def forward(self, outputs, labels):
        target = labels.view(-1) if labels.dtype == torch.long else labels.view(-1, self.vocab_size)
        return self.ce(outputs.view(-1, self.vocab_size), target)


def build_text_case(device="cpu", vocab_size=300, seq_len=8, hidden=64, layers=2, heads=2, seed_model=0, seed_data=1,
                    full_size=False, attention="sdpa"):
    """Random-init BERT masked-LM (tiny by default, bert-base sized with ``full_size``), one sequence of random tokens,
    labels = tokens, user labels withheld (the joint attacker optimises them)."""
    from transformers import BertConfig, BertForMaskedLM

    torch.manual_seed(seed_model)
    if full_size:
        cfg = BertConfig(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, attn_implementation=attention)
        vocab_size = cfg.vocab_size
    else:
        cfg = BertConfig(vocab_size=vocab_size, hidden_size=hidden, num_hidden_layers=layers, num_attention_heads=heads,
                         intermediate_size=2 * hidden, max_position_embeddings=max(32, seq_len), hidden_dropout_prob=0.0,
                         attention_probs_dropout_prob=0.0, attn_implementation=attention)
    model = TokenModel(BertForMaskedLM(cfg))
    model.eval()
    loss_fn = MaskedLMLoss(vocab_size)
    tokens = torch.randint(0, vocab_size, (1, seq_len), generator=torch.Generator().manual_seed(seed_data))
    loss = loss_fn(model(tokens), tokens)
    grads = torch.autograd.grad(loss, tuple(model.parameters()))
    device = torch.device(device)
    model = model.to(device)
    data_cfg = AttrDict(name="synthetic-text", modality="text", task="masked-lm", vocab_size=vocab_size, shape=[seq_len],
                        classes=vocab_size)
    payload = [dict(parameters=[p for p in model.parameters()], buffers=[b for b in model.buffers()], metadata=data_cfg)]
    shared = [dict(gradients=[g.detach().to(device) for g in grads], buffers=None,
                   metadata=dict(num_data_points=1, labels=None, local_hyperparams=None))]
    return AttrDict(model=model, loss_fn=loss_fn, server_payload=payload, shared_data=shared,
                    true_user_data=dict(data=tokens, labels=tokens), data_cfg=data_cfg)


def multi_step_update(model, loss_fn, x, labels, steps, data_per_step, lr):
    """FedAvg user (users.py:336-413, honest case): `steps` plain-SGD steps on consecutive slices of the batch; what
    is shared is the parameter *difference* p_local - p_server, together with the local hyper-parameters."""
    from torch.func import functional_call

    names = [n for n, _ in model.named_parameters()]
    params = [p.detach().clone() for p in model.parameters()]
    start = [p.clone() for p in params]
    buffers = dict(model.named_buffers())
    step_labels, seen = [], 0
    for _ in range(steps):
        xs, ys = x[seen : seen + data_per_step], labels[seen : seen + data_per_step]
        seen = (seen + data_per_step) % x.shape[0]
        live = [p.requires_grad_(True) for p in params]
        loss = loss_fn(functional_call(model, ({**dict(zip(names, live)), **buffers},), (xs,)), ys)
        grads = torch.autograd.grad(loss, live)
        params = [(p - lr * g).detach() for p, g in zip(live, grads)]
        step_labels.append(ys)
    update = [p - s for p, s in zip(params, start)]
    return [dict(gradients=update, buffers=None,
                 metadata=dict(num_data_points=x.shape[0], labels=labels,
                               local_hyperparams=dict(lr=lr, steps=steps, data_per_step=data_per_step, labels=step_labels)))]


def build_fedavg_case(device="cpu", num_data_points=4, steps=2, data_per_step=2, lr=0.05, seed_model=0, seed_data=1,
                      model_name="convnet", data_name="CIFAR10"):
    data_cfg = get_data_config(data_name)
    model = build_model(model_name, data_cfg.classes, seed_model)
    loss_fn = torch.nn.CrossEntropyLoss()
    x_true, labels = synthetic_user_data(data_cfg, num_data_points, seed_data)
    shared = multi_step_update(model, loss_fn, x_true, labels, steps, data_per_step, lr)
    device = torch.device(device)
    model = model.to(device)
    for entry in shared:
        entry["gradients"] = [g.to(device) for g in entry["gradients"]]
        entry["metadata"]["labels"] = entry["metadata"]["labels"].to(device)
        entry["metadata"]["local_hyperparams"]["labels"] = [l.to(device) for l in entry["metadata"]["local_hyperparams"]["labels"]]
    payload = honest_payload(model, data_cfg, public_buffers=True)
    return AttrDict(model=model, loss_fn=loss_fn, server_payload=payload, shared_data=shared,
                    true_user_data=dict(data=x_true, labels=labels), data_cfg=data_cfg)
