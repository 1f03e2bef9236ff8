"""Attack configurations as plain attribute dictionaries.

The reference composes these with Hydra/OmegaConf from ``breaching/config/attack/*.yaml`` (accessor
``breaching.get_attack_config``, breaching/__init__.py:24-29).  The attackers only use attribute access, ``[]``,
``.keys()``, ``.items()`` and ``**`` on the nodes (optimization_based_attack.py:29-48), so any duck-typed mapping
works -- an OmegaConf ``DictConfig`` passed by the unmodified ``simulate_breach.py`` included.  The hyper-parameters
below restate the reference YAML values (file:line given per entry); they are data, not code.
"""

import copy


class AttrDict(dict):
    """dict with attribute access, recursively applied."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    @staticmethod
    def _wrap(value):
        if isinstance(value, dict) and not isinstance(value, AttrDict):
            return AttrDict(value)
        return value

    def __setitem__(self, key, value):
        super().__setitem__(key, self._wrap(value))

    def __getattr__(self, item):
        try:
            return self[item]
        except KeyError:
            raise AttributeError(item) from None

    def __setattr__(self, key, value):
        self[key] = value

    def __deepcopy__(self, memo):
        return AttrDict({k: copy.deepcopy(v, memo) for k, v in self.items()})


def cfg_get(node, key, default=None):
    """`node[key]`, or `default` when the key is absent or the node is not a mapping (cfg objects are duck-typed: Hydra DictConfig
    in the reference's scripts, `AttrDict` here)."""
    try:
        value = node[key]
    except (KeyError, AttributeError, TypeError):
        return default
    return value


def deep_merge(base, override):
    """Hydra `defaults: [_default, _self_]` semantics for plain dicts: recursive, `override` wins."""
    out = copy.deepcopy(base)
    for k, v in override.items():
        if isinstance(v, dict) and isinstance(out.get(k), dict):
            out[k] = deep_merge(out[k], v)
        else:
            out[k] = copy.deepcopy(v)
    return out


# breaching/config/attack/_default_optimization_attack.yaml:1-43
_DEFAULT_OPTIMIZATION = dict(
    type="default",
    attack_type="optimization",
    label_strategy="bias-corrected",
    text_strategy="run-embedding",
    token_recovery="from-labels",
    objective=dict(type="euclidean", scale=1.0, task_regularization=0.0),
    restarts=dict(num_trials=1, scoring="euclidean"),
    init="randn",
    normalize_gradients=False,
    optim=dict(
        optimizer="Adam",
        signed=None,
        step_size=1.0,
        boxed=False,
        max_iterations=400,
        step_size_decay=None,
        langevin_noise=0.0,
        warmup=0,
        grad_clip=None,
        callback=100,
    ),
    augmentations=None,
    differentiable_augmentations=False,
    regularization=None,
    impl=dict(dtype="float", mixed_precision=False, JIT=None),
)

_ATTACKS = {
    # breaching/config/attack/invertinggradients.yaml:4-29
    "invertinggradients": dict(
        type="invertinggradients",
        objective=dict(type="cosine-similarity", scale=1.0),
        restarts=dict(num_trials=1, scoring="cosine-similarity"),
        optim=dict(
            optimizer="adam",
            signed="hard",
            step_size=0.1,
            boxed=True,
            max_iterations=24_000,
            step_size_decay="step-lr",
            callback=1000,
        ),
        regularization=dict(total_variation=dict(scale=0.2, inner_exp=1, outer_exp=1)),
    ),
    # breaching/config/attack/seethroughgradients.yaml:4-38
    "seethroughgradients": dict(
        type="see-through-gradients",
        label_strategy="yin",
        objective=dict(type="euclidean", scale=1e-4),
        restarts=dict(num_trials=1, scoring="euclidean"),
        optim=dict(
            optimizer="adam",
            signed=False,
            step_size=0.1,
            boxed=True,
            max_iterations=20_000,
            step_size_decay="cosine-decay",
            langevin_noise=0.01,
            warmup=50,
            callback=1000,
        ),
        regularization=dict(
            total_variation=dict(scale=1e-4, inner_exp=1, outer_exp=1),
            norm=dict(scale=1e-6, pnorm=2),
            deep_inversion=dict(scale=0.1),
        ),
    ),
    # breaching/config/attack/tag.yaml:7-31  (YAML `label_strategy: None` is the *string* "None")
    "tag": dict(
        type="tag",
        attack_type="joint-optimization",
        label_strategy="None",
        token_recovery="from-embedding",
        init="randn-trunc",
        objective=dict(type="tag-euclidean", scale=1.0, task_regularization=0.0, tag_scale=0.1, scale_scheme="linear"),
        optim=dict(
            optimizer="bert-adam",
            step_size=0.05,
            boxed=False,
            max_iterations=1000,
            grad_clip=1.0,
            warmup=50,
            step_size_decay="linear",
            callback=100,
        ),
    ),
    # breaching/config/attack/modern.yaml
    "modern": dict(
        type="invertinggradients",
        objective=dict(type="cosine-similarity", scale=1.0),
        init="patterned-4",
        restarts=dict(num_trials=1, scoring="cosine-similarity"),
        optim=dict(optimizer="adam", signed="soft", step_size=0.1, boxed=True, max_iterations=24_000,
                   step_size_decay="cosine-decay", warmup=50, callback=1000),
        regularization=dict(
            total_variation=dict(scale=0.1, inner_exp=2, outer_exp=0.5, double_opponents=True),
            features=dict(scale=0.1),
            deep_inversion=dict(scale=0.0),
        ),
    ),
    # breaching/config/attack/legacy.yaml
    "legacy": dict(
        type="invertinggradients",
        objective=dict(type="cosine-similarity", scale=1.0),
        init="zeros",
        restarts=dict(num_trials=1, scoring="cosine-similarity"),
        optim=dict(optimizer="adam", signed="soft", step_size=0.1, boxed=True, max_iterations=24_000,
                   step_size_decay="cosine-decay", callback=1000),
        regularization=dict(
            total_variation=dict(scale=0.1, inner_exp=2, outer_exp=0.5, double_opponents=True),
            features=dict(scale=0.1),
            deep_inversion=dict(scale=0.00005),
        ),
    ),
    # breaching/config/attack/clsattack.yaml
    "clsattack": dict(
        type="invertinggradients",
        objective=dict(type="cosine-similarity", scale=1.0),
        init="patterned-4-randn",
        restarts=dict(num_trials=1, scoring="cosine-similarity"),
        optim=dict(optimizer="adam", signed="soft", step_size=0.1, boxed=True, max_iterations=24_000,
                   step_size_decay="cosine-decay", warmup=50, callback=1000),
        regularization=dict(
            total_variation=dict(scale=0.2, inner_exp=2, outer_exp=0.5, double_opponents=True),
            features=dict(scale=0.0),
            deep_inversion=dict(scale=0.0),
        ),
    ),
    # breaching/config/attack/beyondinfering.yaml (L-BFGS: generic torch.optim loop)
    "beyondinfering": dict(
        type="beyond-infering",
        optim=dict(optimizer="L-BFGS", step_size=1.0, boxed=True, max_iterations=400),
        regularization=dict(total_variation=dict(scale=0.2352, inner_exp=2, outer_exp=1.25)),
    ),
    # breaching/config/attack/wei.yaml (L-BFGS: generic torch.optim loop)
    "wei": dict(
        type="beyond-infering",
        objective=dict(type="euclidean", scale=1.0, task_regularization=1.0),
        init="patterned-16",
        optim=dict(optimizer="L-BFGS", step_size=1.0, boxed=True, max_iterations=300),
    ),
    # breaching/config/attack/sanitycheck.yaml
    "sanitycheck": dict(
        type="sanitycheck",
        objective=dict(type="cosine-similarity", scale=1.0),
        optim=dict(optimizer="adam", signed=None, step_size=1, boxed=True, max_iterations=1, step_size_decay="none",
                   callback=0),
    ),
    # breaching/config/attack/deepleakage.yaml (L-BFGS joint attack) is served by the generic torch.optim loop.
    "deepleakage": dict(
        type="deep-leakage",
        attack_type="joint-optimization",
        label_strategy="None",
        token_recovery="from-embedding",
        optim=dict(optimizer="L-BFGS", step_size=1.0, boxed=False, max_iterations=1200, callback=100),
    ),
}


def _apply_overrides(cfg, overrides):
    """Hydra-style ``a.b.c=value`` overrides (values parsed with YAML scalar rules)."""
    import yaml

    for item in overrides or []:
        key, _, raw = item.partition("=")
        node = cfg
        parts = key.split(".")
        for part in parts[:-1]:
            if part not in node or node[part] is None:
                node[part] = AttrDict()
            node = node[part]
        node[parts[-1]] = yaml.safe_load(raw)
    return cfg


def get_attack_config(attack="invertinggradients", overrides=None):
    """Counterpart of ``breaching.get_attack_config`` (breaching/__init__.py:24-29) without Hydra."""
    if attack not in _ATTACKS:
        raise ValueError(f"Unknown attack configuration {attack!r}; available: {sorted(_ATTACKS)}")
    cfg = AttrDict(deep_merge(_DEFAULT_OPTIMIZATION, _ATTACKS[attack]))
    return _apply_overrides(cfg, overrides)


def available_attacks():
    return sorted(_ATTACKS)


# Dataset metadata the attackers read from ``server_payload[0]["metadata"]`` (base_attack.py:51-57).
# breaching/config/case/data/CIFAR10.yaml:1-22 and ImageNet.yaml:1-22
_DATA = {
    "CIFAR10": dict(
        name="CIFAR10", modality="vision", task="classification", classes=10, shape=(3, 32, 32), normalize=True,
        mean=(0.4914672374725342, 0.4822617471218109, 0.4467701315879822),
        std=(0.24703224003314972, 0.24348513782024384, 0.26158785820007324),
    ),
    "ImageNet": dict(
        name="ImageNet", modality="vision", task="classification", classes=1000, shape=(3, 224, 224), normalize=True,
        mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225),
    ),
}


def get_data_config(name):
    if name not in _DATA:
        raise ValueError(f"Unknown data configuration {name!r}; available: {sorted(_DATA)}")
    return AttrDict(copy.deepcopy(_DATA[name]))
