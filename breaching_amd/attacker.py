"""MI355X-native optimisation-based gradient-inversion attackers behind the reference's plugin API.

Drop-in boundary:
  * ``prepare_attack(model, loss, cfg_attack, setup)``                    reference: breaching/attacks/__init__.py:12-34
  * ``HipOptimizationAttacker.__init__(model, loss_fn, cfg_attack, setup)`` reference: base_attack.py:24-29,
                                                                           optimization_based_attack.py:27-48
  * ``reconstruct(server_payload, shared_data, server_secrets=None, initial_data=None, dryrun=False)
        -> (dict(data=..., labels=...), stats)``                          reference: optimization_based_attack.py:63-88

Host code stays Python; the victim model's forward / backward / double backward run on PyTorch-ROCm; the attack
arithmetic (gradient-matching reduction, priors, signed Adam step with projection and best tracking) runs in the
hand-written gfx950 kernels of libbreach_hip.so.  There is no CPU fallback: constructing an attacker for a non-ROCm
device raises.
"""

import copy
import logging
import time
from collections import defaultdict

import numpy as np
import torch

from . import _lib, schedules, streams as trial_streams, trials, workers
from .config import cfg_get as _cfg_get
from .gm import objective_lookup
from .priors import HipNormRegularization, HipTotalVariation, launch_tv_norm, regularizer_lookup
# the HIP layers of the private victim-model copy (kernels E / F) live in their own, frozen module; re-exported here for callers
from .victim_layers import (FusedEpilogueError, _EvalAffineBatchNorm2d, _EvalBNFunction, _EvalBNGradFunction, _HipLayerNorm,  # noqa: F401
                            _launch_eval_bn, _LayerNormFunction, _LayerNormGradFunction, _PendingBatchNorm, _under_functorch,
                            fast_eval_bn_enabled, fast_eval_bn_mode, fast_layer_norm_enabled, fuse_bn_relu_enabled, fuse_bn_relu_policy,
                            use_affine_eval_batchnorm, use_hip_layernorm)

log = logging.getLogger(__name__)

# base_attack.py:15
embedding_layer_names = ["encoder.weight", "word_embeddings.weight", "transformer.wte"]

_DEFAULT_SETUP = dict(dtype=torch.float, device=torch.device("cpu"))


class HipOptimizationAttacker:
    """Optimisation-based attacker (reference class ``OptimizationBasedAttacker``) running on libbreach_hip.so."""

    def __init__(self, model, loss_fn, cfg_attack, setup=_DEFAULT_SETUP):
        # base_attack.py:24-29
        self.cfg = cfg_attack
        self.memory_format = torch.channels_last if cfg_attack.impl.mixed_precision else torch.contiguous_format
        self.setup = dict(device=torch.device(setup["device"]), dtype=getattr(torch, cfg_attack.impl.dtype))
        if self.setup["device"].type != "cuda":
            raise RuntimeError(
                f"HipOptimizationAttacker needs a ROCm device (setup['device']={self.setup['device']}); the HIP hot "
                "path has no CPU fallback. Use the reference attacker for CPU runs."
            )
        if self.setup["dtype"] != torch.float32 or cfg_attack.impl.mixed_precision:
            raise NotImplementedError("The HIP hot path computes in fp32 (impl.dtype=float, mixed_precision=False).")
        _lib.load()  # fail loudly right here when the extension is missing
        self.model_template = copy.deepcopy(model)
        self.loss_fn = copy.deepcopy(loss_fn)

        # optimization_based_attack.py:29-33
        objective_cls = objective_lookup.get(self.cfg.objective.type)
        if objective_cls is None:
            raise ValueError(f"Unknown objective type {self.cfg.objective.type} given.")
        self.objective = objective_cls(**self.cfg.objective)

        # optimization_based_attack.py:34-40
        self.regularizers = []
        regs = _cfg_get(self.cfg, "regularization")
        if regs is not None and hasattr(regs, "keys"):
            for key in regs.keys():
                if regs[key].scale > 0:
                    self.regularizers.append(regularizer_lookup[key](self.setup, **regs[key]))
        for reg in self.regularizers:  # kernel D tuning (measurements only): launch arguments carried by the plan
            if hasattr(reg, "tuning"):
                reg.tuning = {name: int(_cfg_get(self.cfg.impl, f"bn_{name}") or 0) for name in ("grid_cap", "load_depth", "finalize_block")}

        # optimization_based_attack.py:42-48
        augs = _cfg_get(self.cfg, "augmentations")
        if augs is not None and hasattr(augs, "keys") and len(list(augs.keys())) > 0:
            self.augmentations = self._build_augmentations(augs, setup)
        else:
            self.augmentations = torch.nn.Sequential()

    @staticmethod
    def _build_augmentations(augs, setup):
        # Augmentations are out of the hot-path scope; when the reference package is
        # importable (the drop-in situation) its torch modules are used as they are.
        try:
            from breaching.attacks.auxiliaries.augmentations import augmentation_lookup
        except Exception as exc:  # pragma: no cover - depends on the environment
            raise NotImplementedError(
                "cfg.augmentations is set but the reference augmentation modules are not importable."
            ) from exc
        mods = [augmentation_lookup[key](**augs[key]) for key in augs.keys()]
        return torch.nn.Sequential(*mods).to(**setup)

    def __repr__(self):  # optimization_based_attack.py:50-61
        n = "\n"
        return f"""Attacker (of type {self.__class__.__name__}) with settings:
    Hyperparameter Template: {self.cfg.type}

    Objective: {repr(self.objective)}
    Regularizers: {(n + ' ' * 18).join([repr(r) for r in self.regularizers])}
    Augmentations: {(n + ' ' * 18).join([repr(r) for r in self.augmentations])}

    Optimization Setup:
        {(n + ' ' * 8).join([f'{key}: {val}' for key, val in self.cfg.optim.items()])}
        """

    # ==============================================================================================================
    # reconstruct
    # ==============================================================================================================
    def reconstruct(self, server_payload, shared_data, server_secrets=None, initial_data=None, dryrun=False):
        t_call = time.perf_counter()
        num_trials = self.cfg.restarts.num_trials
        preset = getattr(self, "_preset", None)  # a trial worker: starting points and labels come from rank 0
        pool = self._trial_worker_pool(num_trials) if preset is None else None
        self._trial_execution = {}
        timing = dict(prepare_s=0.0, trials_s=0.0, score_s=0.0, select_s=0.0)
        try:
            if pool is not None:
                "ClassAttack" in server_secrets  # noqa: B015 -- None raises TypeError here, in the caller, as in the reference (:82)
                # The caller's inputs go to the workers BEFORE prepare_attack rebinds / normalises them (base_attack.py:214-220,
                # :298-303 work in place), as one broadcast per dtype over the pool's communicator; nothing tensor-valued is
                # copied to the host or pickled (round 6; servers.py:138-147 / users.py:176-183 define what must arrive).
                pool.begin_job()
                pool.ship("inputs", dict(server_payload=list(server_payload), shared_data=list(shared_data),
                                         server_secrets=server_secrets, initial_data=initial_data), self.setup["device"])
            rec_models, labels, stats = self.prepare_attack(server_payload, shared_data)
            if preset is not None and preset["labels"] is not None:
                labels = preset["labels"].to(self.setup["device"])
            if pool is None and preset is None and workers.active_pool() is not None:
                # The default process group belongs to an idle worker pool (a one-trial call on this attacker, or another
                # attacker's pool): nobody would join a collective, so this call is a single rank.
                shard = trials.TrialShard(num_trials)
            else:
                shard = trials.TrialShard.current(num_trials)
            num_points = shared_data[0]["metadata"]["num_data_points"]
            # Device RNG order.  The reference draws trial t's starting point right before trial t runs, and its Langevin
            # noise draws (:169) sit between consecutive starting points.  Without noise the order of the starting-point draws
            # is all that matters, so they are drawn up front (a sharded or concurrent run then starts every trial exactly
            # where the sequential run does).  With noise on a single rank the reference's interleaving is kept: trials run
            # one at a time and each draws its start when its turn comes.  Only noise + several ranks deviates (documented).
            noisy = float(self.cfg.optim.langevin_noise or 0.0) > 0
            lazy_draws = noisy and shard.world == 1
            if preset is not None:
                inits = {t: self._adopt_initial_state(state) for t, state in preset["inits"].items()}
            else:
                inits = {} if lazy_draws else {t: self._draw_initial_state(num_points, labels) for t in range(num_trials)}
            if pool is not None:  # the starting points of the workers' trials, drawn here in the reference's order
                pool.ship("starts", dict(labels=labels, inits={t: tuple(inits[t]) for t in range(num_trials) if t % pool.world != 0}),
                          self.setup["device"])
                pool.submit([dict(dryrun=dryrun)] * (pool.world - 1))
            timing["prepare_s"] = time.perf_counter() - t_call

            local_scores, local_solutions = {}, {}
            group_runs = None  # the FusedTrial objects (device state + captured hipGraph) of the previous group, re-armed for the next
            mine = list(shard.local_trials())
            width = trials_in_flight(self.cfg) if (self._fused_loop_supported() and not noisy) else 1
            try:
                for start in range(0, len(mine), max(width, 1)):
                    group = mine[start : start + max(width, 1)]
                    t_group = time.perf_counter()
                    if len(group) == 1:
                        solutions = {group[0]: self._run_trial(rec_models, shared_data, labels, stats, group[0], initial_data,
                                                               dryrun, init_state=inits.get(group[0]))}
                    else:
                        solutions, group_runs = self._run_trial_group(rec_models, shared_data, labels, stats, group, initial_data, dryrun,
                                                                      {t: inits[t] for t in group}, reuse=group_runs)
                    t_score = time.perf_counter()
                    timing["trials_s"] += t_score - t_group
                    for trial, solution in solutions.items():
                        local_solutions[trial] = solution
                        local_scores[trial] = self._score_trial(self._solution_data(solution),
                                                                self._score_labels(solution, labels), rec_models, shared_data)
                    timing["score_s"] += time.perf_counter() - t_score
            except KeyboardInterrupt:
                print("Trial procedure manually interruped.")
            if pool is not None:
                t_wait = time.perf_counter()
                pool.expect("trials_done")  # a crashed worker raises here instead of hanging the selection collective
                pool.timing["trials_wait_s"] = round(time.perf_counter() - t_wait, 4)
                pool.broadcast(("go",))
            before_select = getattr(self, "_before_select", None)
            if before_select is not None:
                before_select()
            stats["execution_trials"] = dict(self._trial_execution)  # merged over the ranks with the loss histories
            t_select = time.perf_counter()
            optimal = self._select_optimal_reconstruction(local_solutions, local_scores, stats, shard)
            timing["select_s"] = time.perf_counter() - t_select
            if pool is not None:
                pool.timing["select_s"] = round(timing["select_s"], 4)
                pool.finish()
        except BaseException:
            # A failure on any rank between the first `ship` and the last `ok` -- a worker's error report, or this rank's own
            # trials raising -- must not leave healthy workers waiting for a `go` that never comes (the next call would read their
            # stale messages and enter the collective alone): cancel the job everywhere; a pool that cannot be drained is
            # closed, and the next call starts a new one.
            if pool is not None:
                pool.abort()
                if pool.closed:
                    self._pool = None
            raise
        # How this call was executed, on the channel callers already read (base_attack.py:45): per-trial launch mode of
        # this rank's trials, the pool that shared the trials (backend, world, devices) or why there was none, and where this
        # rank's wall time went (`timing`: preparation incl. shipping, the trial loops, rescoring :191-204, selection :206-218).
        timing["total_s"] = time.perf_counter() - t_call
        stats["execution"] = dict(trials=stats.pop("execution_trials"), pool=pool.describe() if pool is not None else None,
                                  fused_epilogue_fallback=getattr(self, "fused_epilogue_fallback", None),
                                  pool_fallback=getattr(self, "_pool_fallback", None), world=shard.world,
                                  trial_streams=trial_streams.calibration_report(self.setup["device"]),
                                  timing={k: round(v, 4) for k, v in timing.items()})
        reconstructed_data = self._package(optimal, labels)
        if server_payload[0]["metadata"].modality == "text":
            raw = reconstructed_data["data"]
            reconstructed_data = self._postprocess_text_data(reconstructed_data)
            self._attach_raw_embeddings(reconstructed_data, raw)
        if "ClassAttack" in server_secrets:  # optimization_based_attack.py:82-87 (None raises TypeError as there)
            true_num_data = server_secrets["ClassAttack"]["true_num_data"]
            full = torch.zeros([true_num_data, *self.data_shape], **self.setup)
            full[server_secrets["ClassAttack"]["target_indx"]] = optimal if torch.is_tensor(optimal) else optimal[0]
            reconstructed_data["data"] = full
            reconstructed_data["labels"] = server_secrets["ClassAttack"]["all_labels"]
        return reconstructed_data, stats

    # trial workers --------------------------------------------------------------------------
    def _trial_worker_pool(self, num_trials):
        """The pool of per-GPU worker processes that shares this call's trials, or None: one trial, one usable device, a
        process group that already exists (launched under torch.distributed.run: the ranks shard among themselves), or
        BREACH_HIP_TRIAL_DEVICES / cfg.impl.trial_devices naming a single device."""
        import torch.distributed as dist

        if getattr(self, "_is_trial_worker", False) or num_trials < 2:
            return None
        pool = getattr(self, "_pool", None)
        if pool is not None and not pool.closed:
            return pool
        self._pool_fallback = None
        devices = workers.requested_devices(self.cfg, self.setup["device"])[:num_trials]
        if len(devices) < 2:
            return None
        foreign = workers.active_pool()
        if foreign is not None and not foreign.busy:
            # Another attacker of this process left its pool idle (it was not closed or collected yet): its workers hold
            # that attacker's model.  Take the process group over instead of silently running on one GPU.
            log.info("Closing another attacker's idle trial worker pool.")
            foreign.close()
        if dist.is_available() and dist.is_initialized():
            return None  # pre-launched ranks (torch.distributed.run), or a busy pool of another attacker owns the group
        template = copy.deepcopy(self.model_template).to("cpu")
        loss_fn = copy.deepcopy(self.loss_fn).to("cpu") if isinstance(self.loss_fn, torch.nn.Module) else self.loss_fn
        try:
            self._pool = workers.TrialWorkerPool(devices, workers.attacker_runner_factory,
                                                 (type(self).__name__, template, loss_fn, self.cfg))
        except Exception as exc:  # e.g. a victim model class the workers cannot import
            if _cfg_get(self.cfg.impl, "trial_pool", "auto") == "required":
                raise
            # all trials stay on this GPU; callers see it in stats["execution"]["pool_fallback"]
            self._pool_fallback = f"could not start the trial workers on devices {devices}: {exc!r}"
            log.warning(f"{self._pool_fallback}; running every trial on {self.setup['device']}.")
            self._pool = None
        return self._pool

    def close(self):
        """Stop the trial workers (also happens when the attacker is garbage collected)."""
        pool = getattr(self, "_pool", None)
        if pool is not None:
            pool.close()
            self._pool = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _adopt_initial_state(self, state):
        out = []
        for tensor in state:
            t = tensor.detach().to(**self.setup).contiguous().requires_grad_(True)
            t.grad = torch.zeros_like(t)
            out.append(t)
        return tuple(out)

    # hooks the joint attacker overrides -------------------------------------------------------------------------
    def _draw_initial_state(self, num_points, labels):
        return (self._initialize_data([num_points, *self.data_shape]),)

    @staticmethod
    def _solution_data(solution):
        return solution

    @staticmethod
    def _score_labels(solution, labels):
        return labels

    def _package(self, optimal, labels):
        return dict(data=optimal, labels=labels)

    def _attach_raw_embeddings(self, reconstructed_data, raw):
        pass

    # ==============================================================================================================
    # preparation (base_attack.py:43-74)
    # ==============================================================================================================
    def prepare_attack(self, server_payload, shared_data):
        stats = defaultdict(list)
        shared_data = shared_data.copy()
        server_payload = server_payload.copy()

        metadata = server_payload[0]["metadata"]
        self.data_shape = metadata.shape
        if hasattr(metadata, "mean"):
            self.dm = torch.as_tensor(metadata.mean, **self.setup)[None, :, None, None]
            self.ds = torch.as_tensor(metadata.std, **self.setup)[None, :, None, None]
        else:
            self.dm, self.ds = torch.tensor(0, **self.setup), torch.tensor(1, **self.setup)

        rec_models = self._construct_models_from_payload_and_buffers(server_payload, shared_data)
        shared_data = self._cast_shared_data(shared_data)
        if metadata.modality == "text":
            rec_models, shared_data = self._prepare_for_text_data(shared_data, rec_models)
        self._rec_models = rec_models
        if shared_data[0]["metadata"]["labels"] is None:
            labels = self._recover_label_information(shared_data, server_payload, rec_models)
        else:
            labels = shared_data[0]["metadata"]["labels"].clone().to(self.setup["device"])
        if self.cfg.normalize_gradients:
            shared_data = self._normalize_gradients(shared_data)
        return rec_models, labels, stats

    def _construct_models_from_payload_and_buffers(self, server_payload, shared_data):  # base_attack.py:169-212
        models = []
        for idx, payload in enumerate(server_payload):
            new_model = copy.deepcopy(self.model_template)
            new_model.to(**self.setup, memory_format=self.memory_format)
            if shared_data[idx]["buffers"] is not None:  # the user sent its buffers: use them
                buffers = shared_data[idx]["buffers"]
                new_model.eval()
            elif payload["buffers"] is not None:  # public server buffers
                buffers = payload["buffers"]
                new_model.eval()
            else:  # no buffers anywhere: batch statistics of the candidate itself
                new_model.train()
                for module in new_model.modules():
                    if hasattr(module, "track_running_stats"):
                        module.reset_parameters()
                        module.track_running_stats = False
                buffers = []
            with torch.no_grad():
                for param, server_state in zip(new_model.parameters(), payload["parameters"]):
                    param.copy_(server_state.to(**self.setup))
                for buffer, server_state in zip(new_model.buffers(), buffers):
                    buffer.copy_(server_state.to(**self.setup))
            if self.cfg.impl.JIT is not None:
                raise NotImplementedError("impl.JIT (torch.jit script/trace of the victim model) is not supported.")
            bn_mode = fast_eval_bn_mode(self.cfg)
            if bn_mode is not None:
                use_affine_eval_batchnorm(new_model, bn_mode, fuse_epilogue=fuse_bn_relu_enabled(self.cfg))
            if fast_layer_norm_enabled(self.cfg):
                use_hip_layernorm(new_model)
            models.append(new_model)
        return models

    def _cast_shared_data(self, shared_data):  # base_attack.py:214-220 (rebinds the caller's dict entries, as there)
        for data in shared_data:
            data["gradients"] = [g.to(**self.setup) for g in data["gradients"]]
            if data["buffers"] is not None:
                data["buffers"] = [b.to(device=self.setup["device"], dtype=self.setup["dtype"]) for b in data["buffers"]]
        return shared_data

    def _normalize_gradients(self, shared_data, fudge_factor=1e-6):  # base_attack.py:298-303
        for data in shared_data:
            grad_norm = torch.stack([g.pow(2).sum() for g in data["gradients"]]).sum().sqrt()
            torch._foreach_div_(data["gradients"], max(grad_norm, fudge_factor))
        return shared_data

    def _prepare_for_text_data(self, shared_data, rec_models):  # base_attack.py:76-122
        if self.cfg.text_strategy == "run-embedding":
            self.embeddings = []
            for model, data in zip(rec_models, shared_data):
                names = [n for n, _ in model.named_parameters()]
                name_to_idx = dict(zip(names, range(len(data["gradients"]))))
                position = None
                for marker in embedding_layer_names:
                    for key in name_to_idx:
                        if marker in key:
                            position = name_to_idx[key]
                if position is None:
                    raise ValueError("Could not locate the token embedding among the model parameters.")
                weight = list(model.parameters())[position]
                self.embeddings.append(dict(weight=weight, grads=data["gradients"].pop(position)))

                def _disable(module, weight=weight):
                    for child_name, child in module.named_children():
                        if isinstance(child, torch.nn.Embedding):
                            if child.weight is weight:
                                setattr(module, child_name, torch.nn.Identity())
                        else:
                            _disable(child)

                _disable(model)
            _, token_embedding_dim = self.embeddings[0]["weight"].shape
            self.data_shape = [*self.data_shape, token_embedding_dim]
        elif self.cfg.text_strategy == "no-preprocessing":
            pass
        else:
            raise ValueError(f"Invalid text strategy {self.cfg.text_strategy} given.")
        return rec_models, shared_data

    def _postprocess_text_data(self, reconstructed_user_data, models=None):  # base_attack.py:124-167
        def _closest_token(recovered, table):
            recovered = recovered - recovered.mean(dim=-1, keepdim=True)
            table = table - table.mean(dim=-1, keepdim=True)
            norm_rec = recovered.pow(2).sum(dim=-1)
            norm_tab = table.pow(2).sum(dim=-1)
            cosim = recovered.matmul(table.T) / norm_rec[:, None] / norm_tab[None, :]
            return cosim.argmax(dim=1)

        if hasattr(self, "embeddings"):
            embedding_weight = self.embeddings[0]["weight"]
        else:
            raise NotImplementedError("Token recovery without a cut-off embedding layer is not supported.")
        strategy = self.cfg.token_recovery
        if strategy == "from-embedding":
            recovered = reconstructed_user_data["data"]
            base_shape = recovered.shape[0:2]
            tokens = _closest_token(recovered.view(-1, recovered.shape[-1]), embedding_weight).view(*base_shape)
        elif strategy == "from-labels":
            tokens = reconstructed_user_data["labels"]
        elif strategy == "from-limited-embedding":
            recovered = reconstructed_user_data["data"]
            base_shape = recovered.shape[0:2]
            active = reconstructed_user_data["labels"].unique()
            matches = _closest_token(recovered.view(-1, recovered.shape[-1]), embedding_weight[active, :])
            tokens = active[matches].view(*base_shape)
        else:
            raise ValueError(f"Invalid token recovery strategy {strategy} given.")
        reconstructed_user_data["data"] = tokens
        return reconstructed_user_data

    # label recovery (base_attack.py:305-475) ---------------------------------------------------------------------
    def _recover_label_information(self, user_data, server_payload, rec_models):
        """Labels read off the gradient of the last layer (weight = gradients[-2], bias = gradients[-1]) when the user
        withholds them.  One small function per strategy; the result is padded with random classes if a strategy infers
        fewer labels than data points, then sorted (order carries no information)."""
        strategy = self.cfg.label_strategy
        if strategy is None:
            return None
        n = user_data[0]["metadata"]["num_data_points"]
        num_classes = user_data[0]["gradients"][-1].shape[0]
        device = self.setup["device"]
        recover = {
            "iDLG": self._labels_idlg,
            "analytic": self._labels_analytic,
            "yin": self._labels_yin,
            "wainakh-simple": self._labels_wainakh_simple,
            "bias-corrected": self._labels_bias_corrected,
            "random": lambda data, count, classes: torch.randint(0, classes, (count,), device=device),
        }
        if strategy == "exhaustive":
            raise ValueError(f"Exhaustive label searching not implemented; it would need {num_classes ** n} attacks.")
        if strategy in ("wainakh-whitebox", "bias-text"):
            raise NotImplementedError(f"Label strategy {strategy} is outside the HIP hot-path scope.")
        if strategy not in recover:
            raise ValueError(f"Invalid label recovery strategy {strategy} given.")
        labels = recover[strategy](user_data, n, num_classes)
        if len(labels) < n:  # pad with random labels
            labels = torch.cat([labels, torch.randint(0, num_classes, (n - len(labels),), device=device)])
        labels = labels.sort()[0]
        log.info(f"Recovered labels {labels.tolist()} through strategy {strategy}.")
        return labels

    @staticmethod
    def _labels_idlg(user_data, n, num_classes):
        # Zhao et al. 2020: the class whose last-layer weight-gradient row sums lowest, per query
        picks = [torch.argmin(d["gradients"][-2].sum(dim=-1), dim=-1).detach() for d in user_data]
        return torch.stack(picks).unique()

    @staticmethod
    def _labels_analytic(user_data, n, num_classes):
        # classes with a negative bias gradient are present (exact while all labels are distinct)
        present = [(d["gradients"][-1] < 0).nonzero() for d in user_data]
        return torch.stack(present).unique()[:n]

    @staticmethod
    def _labels_yin(user_data, n, num_classes):
        # Yin et al. 2021: rank classes by the smallest entry of their weight-gradient row (summed over queries)
        row_minima = 0
        for d in user_data:
            row_minima = row_minima + d["gradients"][-2].min(dim=-1)[0]
        return row_minima.argsort()[:n]

    def _labels_wainakh_simple(self, user_data, n, num_classes):
        # Wainakh et al.: negative row sums mark present classes; every found label lowers the row sum by an "impact"
        device = self.setup["device"]
        impact = 0
        for d in user_data:
            row_sums = d["gradients"][-2].sum(dim=1)
            negative_mass = torch.where(row_sums < 0, row_sums, torch.zeros_like(row_sums)).sum()
            impact = impact + negative_mass * (1 + 1 / num_classes) / n / len(user_data)
        row_sums = torch.stack([d["gradients"][-2].sum(dim=1) for d in user_data]).mean(dim=0)
        found = []
        cls = 0
        for cls in range(num_classes):
            if row_sums[cls] < 0:
                found.append(torch.as_tensor(cls, device=device))
                row_sums[cls] -= impact
        while len(found) < n:
            found.append(torch.as_tensor(row_sums.argmin(), device=device))
            row_sums[cls] -= impact  # sic: the reference decrements the stale loop index here (base_attack.py:407)
        return torch.stack(found)

    @staticmethod
    def _labels_bias_corrected(user_data, n, num_classes):
        # the default of the optimisation attacks: analytic recovery, then repeated labels by bias-gradient mass
        bias = torch.stack([d["gradients"][-1] for d in user_data]).mean(dim=0)
        present = (bias < 0).nonzero()
        found = [*present.squeeze(dim=-1)]
        impact = bias[present].sum() / n
        bias[present] = bias[present] - impact
        while len(found) < n:
            nxt = bias.argmin()
            found.append(nxt)
            bias[nxt] -= impact
        return torch.stack(found)

    # candidate initialisation (base_attack.py:222-285) -----------------------------------------------------------
    def _initialize_data(self, data_shape):
        """Starting point of a trial, drawn inside the normalised data space (base_attack.py:222-285).

        Scheme names: ``randn``, ``randn-trunc``, ``rand``, ``zeros``; colour fills ``red|green|blue|dark|light[-true]``;
        tiled random patches ``[rand|randn]-patterned-K`` and ``[rand-]wei-K`` (K = patch width)."""
        scheme, setup, shape = self.cfg.init, self.setup, list(data_shape)

        def uniform_pm1(size):
            return torch.rand(size, **setup) * 2 - 1.0

        plain = {
            "randn": lambda: torch.randn(shape, **setup),
            "randn-trunc": lambda: (torch.randn(shape, **setup) * 0.1).clamp(-0.1, 0.1),
            "rand": lambda: uniform_pm1(shape),
            "zeros": lambda: torch.zeros(shape, **setup),
        }
        if scheme in plain:
            candidate = plain[scheme]()
        elif any(colour in scheme for colour in ("red", "green", "blue", "dark", "light")):
            if "light" in scheme:
                candidate = torch.ones(shape, **setup)
            else:  # "dark" falls through to blue's channel, as in the reference
                candidate = torch.zeros(shape, **setup)
                candidate[:, 0 if "red" in scheme else 1 if "green" in scheme else 2, :, :] = 1
            if "-true" in scheme:  # really RGB, not normalised RGB
                candidate = (candidate - self.dm) / self.ds
        elif "patterned" in scheme or "wei" in scheme:
            width = int("".join(ch for ch in scheme if ch.isdigit()))
            patch = [shape[0], 3, width, width]
            if "patterned" in scheme:  # uniform only when the name says "rand" without the "n"
                uniform = "rand" in scheme and "randn" not in scheme
            else:  # wei-K: any "rand" in the name (even "randn") selects the uniform draw -- reference behaviour
                uniform = "rand" in scheme
            seed = uniform_pm1(patch) if uniform else torch.randn(patch, **setup)
            reps = [int(torch.as_tensor(shape[d] / width).ceil()) for d in (2, 3)]
            candidate = torch.tile(seed, (1, 1, *reps))[:, :, : shape[2], : shape[3]].contiguous().clone()
        else:
            raise ValueError(f"Unknown initialization scheme {scheme} given.")
        candidate = candidate.contiguous()
        candidate.requires_grad = True
        candidate.grad = torch.zeros_like(candidate)
        return candidate

    def _init_optimizer(self, candidate):  # base_attack.py:287-296 -> common.py:5-40 (torch.optim, generic loop only)
        optim = self.cfg.optim
        name = str(optim.optimizer).lower()
        lr = optim.step_size
        if name == "adam":
            optimizer = torch.optim.Adam(candidate, lr=lr)
        elif name == "adam-safe":
            optimizer = torch.optim.Adam(candidate, lr=lr, betas=(0.5, 0.99), eps=1e-4)
        elif name == "bert-adam":
            optimizer = torch.optim.AdamW(candidate, lr=lr, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.01)
        elif name == "momgd":
            optimizer = torch.optim.SGD(candidate, lr=lr, momentum=0.9, nesterov=True)
        elif name == "gd":
            optimizer = torch.optim.SGD(candidate, lr=lr, momentum=0.0)
        elif name == "l-bfgs":
            optimizer = torch.optim.LBFGS(candidate, lr=lr)
        else:
            raise ValueError(f"Invalid optimizer {optim.optimizer} given.")
        lrs = schedules.lr_sequence(lr, optim.step_size_decay, optim.warmup, optim.max_iterations,
                                    length=optim.max_iterations + 1)
        return optimizer, _TableScheduler(optimizer, lrs)

    # ==============================================================================================================
    # one trial
    # ==============================================================================================================
    def _fused_loop_supported(self):
        optim = self.cfg.optim
        if schedules.optimizer_hparams(optim.optimizer) is None:  # raises ValueError for unknown names
            return False
        if self.cfg.differentiable_augmentations or len(self.augmentations) > 0:
            return False
        return True

    def _run_trial(self, rec_model, shared_data, labels, stats, trial, initial_data=None, dryrun=False, init_state=None):
        for regularizer in self.regularizers:
            regularizer.initialize(rec_model, shared_data, labels)
        self.objective.initialize(self.loss_fn, self.cfg.impl, shared_data[0]["metadata"]["local_hyperparams"])
        self.objective.prepare(rec_model, shared_data)

        if init_state is None:
            init_state = self._draw_initial_state(shared_data[0]["metadata"]["num_data_points"], labels)
        candidates = list(init_state)
        if initial_data is not None:  # optimization_based_attack.py:100-101
            candidates[0].data = initial_data.data.clone().to(**self.setup).contiguous()

        if self._fused_loop_supported():
            best = self._fused_loop(candidates, labels, rec_model, shared_data, stats, trial, dryrun)
        else:
            best = self._generic_loop(candidates, labels, rec_model, shared_data, stats, trial, dryrun)
        return best[0] if len(best) == 1 else tuple(best)

    def _run_trial_group(self, rec_model, shared_data, labels, stats, group, initial_data, dryrun, init_states, reuse=None):
        """Several independent trials in flight on one GPU, each on its own HIP stream (one trial does not fill an MI355X:
        two processes sharing a GPU reach 1.45x the throughput of one).  The iterations are enqueued round-robin; per trial
        the result is exactly what `_run_trial` produces -- the trials share only read-only state.

        `reuse`: the runs of the previous group of the same `reconstruct` call (returned next to the solutions).  A later group
        RE-ARMS them -- new starting point copied into the same candidate tensor, moments, best copy, history and the device state
        record reset -- instead of building new trials: their captured hipGraphs stay valid (same addresses), so trial 5 of 32
        starts on graph replays and skips three eager iterations and a capture (round 6: 32 x 1 000 ResNet-18 iterations on one
        GPU, 8 groups: 60.7 s -> 58.6 s, profiles/r6_bench_driver_style.json).  Graph replay is bit-identical to eager launches, so nothing else changes."""
        device = self.setup["device"]
        optim = self.cfg.optim
        max_iterations = int(optim.max_iterations)
        main = torch.cuda.current_stream(device)
        if not reuse:
            # (A re-armed group keeps the objective, the priors and their device buffers exactly as the captured graphs know them:
            # re-initialising a DeepInversion prior would free the plan buffers those graphs write to.)
            for regularizer in self.regularizers:
                regularizer.initialize(rec_model, shared_data, labels)
            self.objective.initialize(self.loss_fn, self.cfg.impl, shared_data[0]["metadata"]["local_hyperparams"])
            # Shared read-only state is produced on the caller's stream BEFORE the side streams fork from it (they
            # `wait_stream(main)` below): the packed observed gradients and the frozen statistics of the affine eval-BN layers.
            self.objective.prepare(rec_model, shared_data)
            for model in rec_model:
                for module in model.modules():
                    if isinstance(module, _EvalAffineBatchNorm2d) and not module.training and module.running_var is not None:
                        module._frozen_statistics()
            self._first_use_warm_up(rec_model, shared_data, labels, init_states[group[0]])
        # EVERY trial of the group gets a side stream of its own; none runs on the caller's stream.  Measured (round 3,
        # profiles/r3_stall_bisect.jsonl): once an earlier attack of the process has replayed a hipGraph on the caller's
        # stream (the legacy null stream in simulate_breach.py / benchmark_breaches.py), a group with one trial on that
        # stream and three on side streams runs at ~120 iterations/s instead of ~430 -- every replay then serialises against
        # the other trials' streams; with all four on side streams the rate is 428 whether or not an attack ran before
        # (and the same 430-440 in a fresh process).  GPU_MAX_HW_QUEUES=8 (breaching_amd/__init__.py) leaves a hardware
        # queue for each of them next to the caller's.  WHICH side streams matters as much: two busy streams on one of the four
        # hardware compute pipes run slower than one after the other (round 4, profiles/r4_inflight_pipes_probe.jsonl: four
        # trials 526 it/s on distinct pipes, 173 with one collision), so the streams are picked by measurement, once per process
        # and device, and reused by every later group (breaching_amd/streams.py).
        streams = dict(zip(group, trial_streams.side_streams(device, len(group))))
        runs = {}
        for idx, t in enumerate(group):
            candidates = list(init_states[t])
            if initial_data is not None:
                candidates[0].data = initial_data.data.clone().to(**self.setup).contiguous()
            streams[t].wait_stream(main)
            with torch.cuda.stream(streams[t]):
                if reuse is not None and idx < len(reuse) and reuse[idx].rearm(candidates):
                    runs[t] = reuse[idx]
                else:
                    runs[t] = FusedTrial(self, candidates, labels, rec_model, shared_data)
        current_wallclock = time.time()
        iterations_run = 0
        try:
            for iteration in range(max_iterations):
                for t in group:
                    with torch.cuda.stream(streams[t]):
                        runs[t].step()
                iterations_run = iteration + 1
                if iteration + 1 == max_iterations or iteration % optim.callback == 0:
                    timestamp = time.time()
                    alive = False
                    for t in group:
                        with torch.cuda.stream(streams[t]):
                            host = runs[t].read_state()
                        log.info(f"| Trial {t} | It: {iteration + 1} | Rec. loss: {host['total']:2.4f} | "
                                 f"T: {timestamp - current_wallclock:4.2f}s")
                        alive = alive or not host["dead"]
                    current_wallclock = timestamp
                    if not alive:
                        log.info("Recovery loss is non-finite in every trial of the group. Cancelling reconstruction!")
                        break
                if dryrun:
                    break
        except KeyboardInterrupt:
            print(f"Recovery interrupted manually in iteration {iterations_run}!")
        solutions = {}
        for t in group:
            with torch.cuda.stream(streams[t]):
                stats[f"Trial_{t}_Val"].extend(runs[t].loss_history(iterations_run))
                best = [b.clone() for b in runs[t].best()]  # the run (and its `best` buffer) may serve the next group
                self.last_trial_execution = runs[t].execution_mode()
                self._record_execution(t, self.last_trial_execution)
            main.wait_stream(streams[t])
            solutions[t] = best[0] if len(best) == 1 else tuple(best)
        return solutions, [runs[t] for t in group]

    def _first_use_warm_up(self, rec_model, shared_data, labels, init_state):
        """Once per attacker and process, before its first group of concurrent trials: one throw-away evaluation of the objective
        and its gradient on a copy of a starting point, on the caller's stream, then a device synchronisation.  Everything the
        libraries underneath do on first use of this model -- MIOpen's solver selection and kernel compilation for every
        convolution configuration of the three autograd orders, rocBLAS handles, lazily built plans -- is finished before four
        trials start issuing the same work from four streams.  Seen without it (round 6, a fresh box, four worker processes sharing
        one GPU): the FIRST trial of a worker's first group left its single-rank trajectory at iteration 1 in 2 of ~80 runs, by
        7e-5 and by 16 %, every other trial bit-identical.  Costs one iteration; changes no trial (no RNG draw, no state)."""
        if getattr(self, "_warmed_up", False):
            return
        scratch = [c.detach().clone().requires_grad_(True) for c in init_state]
        _, autograd_regs = self._split_regularizers()
        total, _ = self._autograd_objective(scratch, labels, rec_model, shared_data, autograd_regs)
        if torch.is_tensor(total) and total.requires_grad:
            torch.autograd.grad(total, scratch, allow_unused=True)
        for reg in autograd_regs:  # nothing of this evaluation may stay alive (an autograd graph would block the hipGraph capture)
            release = getattr(reg, "release_graph", None)
            if release is not None:
                release()
        del total, scratch
        torch.cuda.synchronize(self.setup["device"])
        self._warmed_up = True

    def _record_execution(self, trial, mode):
        record = getattr(self, "_trial_execution", None)
        if record is not None:
            record[trial] = mode

    # hooks the joint attacker overrides -------------------------------------------------------------------------
    def _labels_for_objective(self, candidates, labels):
        return labels

    def _boxed_flags(self, candidates):
        return [bool(self.cfg.optim.boxed)]

    # ---- the autograd part of the closure (optimization_based_attack.py:145-165) --------------------------------
    def _autograd_objective(self, candidates, labels, rec_model, shared_data, autograd_regularizers):
        candidate = candidates[0]
        total_objective = 0
        total_task_loss = 0
        obj_labels = self._labels_for_objective(candidates, labels)
        for model, data in zip(rec_model, shared_data):
            try:
                objective, task_loss = self.objective(model, data["gradients"], candidate, obj_labels)
            except FusedEpilogueError as exc:
                # fuse_bn_relu="auto": this model's forward cannot take the deferred BatchNorm launch.  Nothing of the iteration
                # has been committed yet (the error is raised inside the victim's forward pass): switch the fusion off on this
                # model copy and evaluate again -- more launches, same values.  Not possible while a hipGraph is being captured.
                if fuse_bn_relu_policy(self.cfg) == "required" or torch.cuda.is_current_stream_capturing():
                    raise
                switched = sum(1 for m in model.modules() if isinstance(m, _EvalAffineBatchNorm2d) and m.fuse_epilogue)
                for m in model.modules():
                    if isinstance(m, _EvalAffineBatchNorm2d):
                        m.fuse_epilogue = False
                self.fused_epilogue_fallback = f"BatchNorm -> ReLU fusion switched off on {switched} layers: {exc}"
                log.warning(f"{self.fused_epilogue_fallback}  (cfg.impl.fuse_bn_relu='required' turns this into an error.)")
                objective, task_loss = self.objective(model, data["gradients"], candidate, obj_labels)
            total_objective = total_objective + objective
            total_task_loss = total_task_loss + task_loss
        for regularizer in autograd_regularizers:
            total_objective = total_objective + regularizer(candidate)
        return total_objective, total_task_loss

    def _split_regularizers(self):
        """TV / norm priors have an analytic gradient that goes straight into kernel B; the rest needs autograd."""
        fused_terms, autograd_regs = {}, []
        for reg in self.regularizers:
            if isinstance(reg, HipTotalVariation) and "tv_scale" not in fused_terms:
                fused_terms.update(reg.fused_terms())
            elif isinstance(reg, HipNormRegularization) and "norm_scale" not in fused_terms:
                fused_terms.update(reg.fused_terms())
            else:
                autograd_regs.append(reg)
        return fused_terms, autograd_regs

    def _fused_loop(self, candidates, labels, rec_model, shared_data, stats, trial, dryrun):
        """Sync-free loop: PyTorch autograd for the model, HIP kernels for everything else.

        reference: optimization_based_attack.py:103-143.  Host synchronisation happens only at the logging cadence
        (``optim.callback``) and at the end; the loss history, best-so-far candidate and non-finite flag live on the GPU.
        """
        optim = self.cfg.optim
        max_iterations = int(optim.max_iterations)
        run = FusedTrial(self, candidates, labels, rec_model, shared_data)
        current_wallclock = time.time()
        iterations_run = 0
        try:
            for iteration in range(max_iterations):
                run.step()
                iterations_run = iteration + 1
                if iteration + 1 == max_iterations or iteration % optim.callback == 0:
                    host = run.read_state()  # the only host synchronisation of the loop
                    timestamp = time.time()
                    log.info(
                        f"| It: {iteration + 1} | Rec. loss: {host['total']:2.4f} | "
                        f" Task loss: {float(self.current_task_loss):2.4f} | T: {timestamp - current_wallclock:4.2f}s"
                    )
                    current_wallclock = timestamp
                    if host["dead"]:
                        log.info(f"Recovery loss is non-finite in iteration {host['first_bad']}. Cancelling reconstruction!")
                        break
                if dryrun:
                    break
        except KeyboardInterrupt:
            print(f"Recovery interrupted manually in iteration {iterations_run}!")
        stats[f"Trial_{trial}_Val"].extend(run.loss_history(iterations_run))
        self.last_trial_execution = run.execution_mode()
        self._record_execution(trial, self.last_trial_execution)
        return run.best()

    # ---- generic loop: any torch.optim optimiser, differentiable augmentations, L-BFGS closures ------------------
    def _generic_loop(self, candidates, labels, rec_model, shared_data, stats, trial, dryrun):
        """The reference loop shape (optimization_based_attack.py:103-143) with the HIP objective / priors as autograd
        nodes.  Used for SGD / L-BFGS and augmentations, where the fused candidate step does not apply."""
        optim = self.cfg.optim
        best = [c.detach().clone() for c in candidates]
        minimal_value_so_far = torch.as_tensor(float("inf"), **self.setup)
        optimizer, scheduler = self._init_optimizer(candidates)
        boxed_flags = self._boxed_flags(candidates)
        current_wallclock = time.time()
        iteration = 0
        try:
            for iteration in range(optim.max_iterations):
                closure = self._compute_objective(candidates, labels, rec_model, optimizer, shared_data, iteration)
                objective_value, task_loss = optimizer.step(closure), self.current_task_loss
                scheduler.step()
                with torch.no_grad():
                    for tensor, boxed in zip(candidates, boxed_flags):
                        if boxed:
                            tensor.data = torch.max(torch.min(tensor, (1 - self.dm) / self.ds), -self.dm / self.ds)
                    if objective_value < minimal_value_so_far:
                        minimal_value_so_far = objective_value.detach()
                        best = [c.detach().clone() for c in candidates]
                if iteration + 1 == optim.max_iterations or iteration % optim.callback == 0:
                    timestamp = time.time()
                    log.info(
                        f"| It: {iteration + 1} | Rec. loss: {objective_value.item():2.4f} | "
                        f" Task loss: {float(task_loss):2.4f} | T: {timestamp - current_wallclock:4.2f}s"
                    )
                    current_wallclock = timestamp
                if not torch.isfinite(objective_value):
                    log.info(f"Recovery loss is non-finite in iteration {iteration}. Cancelling reconstruction!")
                    break
                stats[f"Trial_{trial}_Val"].append(objective_value.item())
                if dryrun:
                    break
        except KeyboardInterrupt:
            print(f"Recovery interrupted manually in iteration {iteration}!")
        self.last_trial_execution = f"torch.optim loop ({optim.optimizer})"
        self._record_execution(trial, self.last_trial_execution)
        return [b.detach() for b in best]

    def _compute_objective(self, candidates, labels, rec_model, optimizer, shared_data, iteration):
        """Closure for torch.optim (optimization_based_attack.py:145-189)."""
        optim = self.cfg.optim

        def closure():
            optimizer.zero_grad()
            candidate = candidates[0]
            if self.cfg.differentiable_augmentations:
                augmented = self.augmentations(candidate)
            else:
                augmented = candidate
                augmented.data = self.augmentations(candidate.data)
            total_objective, total_task_loss = self._autograd_objective([augmented, *candidates[1:]], labels, rec_model,
                                                                        shared_data, self.regularizers)
            if total_objective.requires_grad:
                total_objective.backward(inputs=list(candidates), create_graph=False)
            with torch.no_grad():
                if optim.langevin_noise > 0:
                    step_size = optimizer.param_groups[0]["lr"]
                    for tensor in candidates:
                        tensor.grad += optim.langevin_noise * step_size * torch.randn_like(tensor.grad)
                if optim.grad_clip is not None:
                    for tensor in candidates:
                        grad_norm = tensor.grad.norm()
                        if grad_norm > optim.grad_clip:
                            tensor.grad.mul_(optim.grad_clip / (grad_norm + 1e-6))
                if optim.signed is not None:
                    if optim.signed == "soft":
                        scaling_factor = 1 - iteration / optim.max_iterations
                        for tensor in candidates:
                            tensor.grad.mul_(scaling_factor).tanh_().div_(scaling_factor)
                    elif optim.signed == "hard":
                        for tensor in candidates:
                            tensor.grad.sign_()
            self.current_task_loss = total_task_loss
            return total_objective

        return closure

    # ==============================================================================================================
    # scoring and selection (optimization_based_attack.py:191-218)
    # ==============================================================================================================
    def _score_trial(self, candidate, labels, rec_model, shared_data):
        scoring = self.cfg.restarts.scoring
        if scoring in ["euclidean", "cosine-similarity"]:
            # fresh objective at scale 1.0, no task regularisation -- reference quirk kept (:195)
            objective = objective_lookup[scoring]()
            objective.initialize(self.loss_fn, self.cfg.impl, shared_data[0]["metadata"]["local_hyperparams"])
            score = 0
            for model, data in zip(rec_model, shared_data):
                score = score + objective(model, data["gradients"], candidate, labels)[0].detach()
        elif scoring in ["TV", "total-variation"]:
            # The reference builds TotalVariation(scale=1.0) without its `setup` argument and raises TypeError (:201).
            score = HipTotalVariation(self.setup, scale=1.0)(candidate).detach()
        else:
            raise ValueError(f"Scoring mechanism {scoring} not implemented.")
        return score if score.isfinite() else float("inf")

    def _select_optimal_reconstruction(self, local_solutions, local_scores, stats, shard):
        optimal_val, optimal_solution = shard.select(local_solutions, local_scores, stats, self.setup["device"])
        stats["opt_value"] = optimal_val
        if np.isfinite(optimal_val):
            log.info(f"Optimal candidate solution with rec. loss {optimal_val:2.4f} selected.")
            return optimal_solution
        log.info("No valid reconstruction could be found.")
        if torch.is_tensor(optimal_solution):
            return torch.zeros_like(optimal_solution)
        return tuple(torch.zeros_like(s) for s in optimal_solution)


GRAPH_WARMUP_ITERATIONS = 3
DEFAULT_TRIALS_IN_FLIGHT = 4
MAX_TRIALS_IN_FLIGHT = 4




def trials_in_flight(cfg):
    """How many of a rank's trials run concurrently on separate streams: cfg.impl.trials_in_flight or
    BREACH_HIP_TRIALS_IN_FLIGHT, default 4, never more than MAX_TRIALS_IN_FLIGHT.  Measured on one MI355X, ResNet-18: 228 / 356 /
    ~450 / 526-545 iterations/s with 1 / 2 / 3 / 4 trials in flight, then a collapse -- 216 / 248 / 277 with 5 / 6 / 8.  Round 4
    found the reason (profiles/r4_inflight_pipes_probe.jsonl, breaching_amd/streams.py): streams are hardware queues dealt onto FOUR
    compute pipes, and two graph-replaying streams on one pipe run slower than one after the other (two trials: 80 vs 356 it/s);
    a fifth busy stream always shares a pipe.  Within the four, throughput is bounded by the chip-wide retirement rate of dependent
    dispatches, not by CUs (profiles/r4_cu_mask_probe.jsonl).  1 restores the reference's strictly sequential order; per-trial
    results do not depend on the width."""
    import os

    env = os.environ.get("BREACH_HIP_TRIALS_IN_FLIGHT")
    value = int(env) if env is not None else _cfg_get(cfg.impl, "trials_in_flight", DEFAULT_TRIALS_IN_FLIGHT)
    value = max(int(value or DEFAULT_TRIALS_IN_FLIGHT), 1)
    if value > MAX_TRIALS_IN_FLIGHT:
        log.warning(f"trials_in_flight={value} lowered to {MAX_TRIALS_IN_FLIGHT}: the GPU has four hardware compute pipes, and a fifth "
                    "graph-replaying stream shares one -- slower than four (measured, see attacker.trials_in_flight).")
        value = MAX_TRIALS_IN_FLIGHT
    return value


def graph_replay_enabled(cfg):
    """hipGraph replay is on unless BREACH_HIP_GRAPH=0 or cfg.impl.hip_graph is false."""
    return graph_replay_policy(cfg) != "off"


def graph_replay_policy(cfg):
    """"auto" (default: a failed capture logs a warning, the trial continues with eager HIP launches -- the reference attacks
    arbitrary models, and one with a host synchronisation or a data-dependent shape in its forward cannot be captured --
    and `stats["execution"]` / `attacker.last_trial_execution` say so), "required" (a failed capture raises: eager launches
    are 2.4x slower; the test suite and bench.py run in this mode so that a fall-back can never pass silently), or "off".
    Set by cfg.impl.hip_graph (True / "auto" / "required" / False) or BREACH_HIP_GRAPH (1 / auto / required / 0);
    BREACH_HIP_GRAPH_STRICT=1 turns every "auto" into "required" and leaves "off" alone (test suites)."""
    import os

    flag = os.environ.get("BREACH_HIP_GRAPH")
    if flag is None:
        flag = _cfg_get(cfg.impl, "hip_graph", True)
    if flag is None:
        policy = "auto"
    elif isinstance(flag, str):
        flag = flag.strip().lower()
        if flag in ("required", "require", "strict"):
            policy = "required"
        else:
            policy = "off" if flag in ("0", "false", "off", "no") else "auto"
    else:
        policy = "auto" if bool(flag) else "off"
    if policy == "auto" and os.environ.get("BREACH_HIP_GRAPH_STRICT", "0").strip().lower() not in ("", "0", "false", "off", "no"):
        policy = "required"
    return policy


class FusedTrial:
    """Device-resident state of one trial of the fused loop plus ``step()``, the body of one attack iteration.

    One ``step()`` = victim forward/backward/double-backward on PyTorch-ROCm + kernel A (fwd, finalize, bwd) + kernel C +
    loss commit + [gradient norm] + kernel B.  Nothing in it reads device memory from the host.
    reference: the body of the loop at optimization_based_attack.py:110-121 with its closure :145-189.
    """

    def __init__(self, attacker, candidates, labels, rec_model, shared_data):
        lib = _lib.load()
        self.lib = lib
        self.attacker = attacker
        self.candidates, self.labels, self.rec_model, self.shared_data = list(candidates), labels, rec_model, shared_data
        cfg = attacker.cfg
        optim = cfg.optim
        device = attacker.setup["device"]
        self.device = device
        self.max_iterations = int(optim.max_iterations)
        hp = schedules.optimizer_hparams(optim.optimizer)
        self.lrs = schedules.lr_sequence(optim.step_size, optim.step_size_decay, optim.warmup, self.max_iterations)
        table = schedules.adam_schedule_table(self.lrs, hp["betas"][0], hp["betas"][1], hp["weight_decay"])
        self.sched_dev = torch.from_numpy(table).to(device)
        self.state = torch.zeros(_lib.BH_STATE_WORDS, dtype=torch.int32, device=device)
        self.history = torch.zeros(self.max_iterations, dtype=torch.float32, device=device)
        self.norm_ws = torch.empty(_lib.BH_PRIOR_MAX_GRID, dtype=torch.float64, device=device)
        self.list_norm_ws = None  # partial rows of the per-tensor gradient norms of a multi-tensor (joint) attack, sized on first use

        self.fused_terms, self.autograd_regs = attacker._split_regularizers()
        first = self.candidates[0]
        self.use_prior = len(self.fused_terms) > 0
        if self.use_prior and not (first.dim() == 4 and first.shape[1] == 3):
            raise ValueError(f"Total variation / norm priors expect [B,3,H,W] candidates, got {tuple(first.shape)}.")
        self.prior_grad = torch.empty_like(first) if self.use_prior else None
        self.prior_partials = (
            torch.empty(_lib.BH_PRIOR_MAX_GRID * _lib.BH_PRIOR_PARTIAL_STRIDE, dtype=torch.float64, device=device)
            if self.use_prior else None
        )
        signed = optim.signed
        sign_mode = _lib.SIGN_HARD if signed == "hard" else _lib.SIGN_SOFT if signed == "soft" else _lib.SIGN_NONE
        self.langevin = float(optim.langevin_noise or 0.0)
        # impl.langevin_noise=host: draw the noise from torch's default CPU generator and copy it over -- what a CPU run of the
        # reference draws from the same seed, so parity tests can put identical noise on both sides.  The copy synchronises,
        # hence no graph replay in this mode; the default draws on the device.
        self.host_noise = self.langevin > 0 and _cfg_get(cfg.impl, "langevin_noise", "device") == "host"
        self.grad_clip = optim.grad_clip
        self.slots = []  # per optimised tensor: params struct + moment buffers + best copy
        for tensor, boxed in zip(self.candidates, attacker._boxed_flags(self.candidates)):
            P = _lib.StepParams()
            P.n = tensor.numel()
            P.max_iterations = self.max_iterations
            P.sign_mode = sign_mode
            P.beta1, P.beta2, P.eps = hp["betas"][0], hp["betas"][1], hp["eps"]
            P.decoupled_wd = int(hp["decoupled"] and hp["weight_decay"] != 0)
            P.langevin = self.langevin
            # negative = no clipping; 0 is a legal threshold (the reference then scales the gradient to ~0, :171-174)
            P.grad_clip = float(self.grad_clip) if self.grad_clip is not None else -1.0
            P.boxed = int(boxed)
            P.channels, P.plane = 1, max(tensor.numel(), 1)
            if boxed:
                lo = (-attacker.dm / attacker.ds).flatten().tolist()
                hi = ((1 - attacker.dm) / attacker.ds).flatten().tolist()
                if len(lo) > 4:
                    raise NotImplementedError("Box projection supports at most 4 channels.")
                if len(lo) > 1:
                    P.channels = len(lo)
                    P.plane = tensor.numel() // (tensor.shape[0] * tensor.shape[1])
                for c, (l, h) in enumerate(zip(lo, hi)):
                    P.lo[c], P.hi[c] = l, h
            self.slots.append(dict(x=tensor, P=P, m=torch.zeros_like(tensor), v=torch.zeros_like(tensor),
                                   best=tensor.detach().clone()))
        with torch.cuda.device(device):
            _lib.check(lib.bh_state_reset(_lib.ptr(self.state), _lib.current_stream_handle(device)), "bh_state_reset")
        self.iterations = 0
        self.tickets = {}  # zeroed device words for kernel D's layer-total ticket, one per model
        # hipGraph replay of the whole iteration: the loop is launch-bound (~1.2k kernels per ResNet-18 iteration), and
        # nothing in step() needs the host, so after a few eager iterations the body is captured once and replayed.
        self.graph = None
        self.graph_failed = None
        self.capture_after = GRAPH_WARMUP_ITERATIONS
        self.graph_policy = graph_replay_policy(cfg)
        self.use_graph = self.graph_policy != "off" and not self.host_noise

    def rearm(self, candidates):
        """Start another trial on this object's device state and captured hipGraph: the new starting point goes INTO the candidate
        tensors the graph was captured on; moments, best copy, loss history and the state record are reset on the current stream.
        Returns False (the caller builds a new trial) when the shapes differ or this run never captured a graph."""
        if self.graph is None or len(candidates) != len(self.slots):
            return False
        if any(c.shape != slot["x"].shape or c.dtype != slot["x"].dtype for c, slot in zip(candidates, self.slots)):
            return False
        with torch.no_grad(), torch.cuda.device(self.device):
            for slot, start in zip(self.slots, candidates):
                slot["x"].copy_(start.detach())
                slot["m"].zero_()
                slot["v"].zero_()
                slot["best"].copy_(slot["x"])
            self.history.zero_()
            _lib.check(self.lib.bh_state_reset(_lib.ptr(self.state), _lib.current_stream_handle(self.device)), "bh_state_reset")
        self.iterations = 0
        return True

    def step(self):
        """One attack iteration: a graph replay when captured, the eager body otherwise."""
        if self.graph is not None:
            self.graph.replay()
        else:
            self._enqueue_iteration()
            if self.use_graph and self.iterations + 1 == self.capture_after:
                self._capture()
        self.iterations += 1

    def execution_mode(self):
        """How the iterations of this trial were issued: "hipGraph replay", or "eager launches" with the reason (switched off,
        fewer iterations than the warm-up, host-side noise, or the capture error).  `attacker.last_trial_execution` keeps the
        last trial's answer, so a silent fall-back to eager launches is visible to callers and tests."""
        if self.graph is not None:
            return "hipGraph replay"
        if self.graph_failed is not None:
            return f"eager launches (capture failed: {self.graph_failed})"
        if self.host_noise:
            return "eager launches (host-side Langevin noise)"
        if not self.use_graph:
            return "eager launches (graph replay switched off)"
        return "eager launches (run shorter than the capture warm-up)"

    def disable_graph(self):
        """Back to eager launches (bench.py uses this to time individual kernels with events)."""
        self.graph, self.use_graph = None, False

    def _capture(self):
        device = self.device
        try:
            # Nothing from the eager iterations may keep an autograd graph alive: a surviving AccumulateGrad node is bound
            # to the eager stream and capture would have to synchronise with it (ROCm 7.2 crashes in hipStreamEndCapture).
            import gc

            for reg in self.autograd_regs:
                release = getattr(reg, "release_graph", None)
                if release is not None:
                    release()
            gc.collect()
            torch.cuda.synchronize(device)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.device(device), torch.cuda.graph(graph):
                self._enqueue_iteration()
            self.graph = graph
        except Exception as exc:
            self.graph, self.use_graph, self.graph_failed = None, False, repr(exc)
            torch.cuda.synchronize(device)
            if self.graph_policy == "required":
                raise RuntimeError(
                    f"hipGraph capture of the attack iteration failed ({exc!r}) and cfg.impl.hip_graph / BREACH_HIP_GRAPH is "
                    "'required'. With 'auto' (the default) the trial continues with eager launches; False never captures."
                ) from exc
            # "auto": stay on the eager HIP path; never leave the GPU.  Visible to callers in stats["execution"]["trials"].
            log.warning(f"hipGraph capture of the attack iteration failed ({exc!r}); continuing with eager launches "
                        "(about 2.4x slower).  cfg.impl.hip_graph='required' turns this into an error.")

    def _enqueue_iteration(self):
        lib, att, device = self.lib, self.attacker, self.device
        with torch.cuda.device(device):
            stream = _lib.current_stream_handle(device)
            for tensor in self.candidates:
                tensor.grad = None
            scoped = [reg for reg in self.autograd_regs if hasattr(reg, "ticket_scope")]
            for owner in scoped:  # this trial's launches all run on this stream: one re-zeroed ticket word per BN plan
                owner.ticket_scope = self.tickets
            try:
                total_objective, task_loss = att._autograd_objective(self.candidates, self.labels, self.rec_model,
                                                                     self.shared_data, self.autograd_regs)
            finally:
                for owner in scoped:
                    owner.ticket_scope = None
            grads = torch.autograd.grad(total_objective, self.candidates, create_graph=False)
            n_reg = 0
            if self.use_prior:
                t = self.fused_terms
                _, _, grid = launch_tv_norm(self.candidates[0].detach(), t.get("tv_scale", 0.0), t.get("inner_exp", 1),
                                            t.get("outer_exp", 1), t.get("eps", 1e-8), t.get("double_opponents", False),
                                            t.get("norm_scale", 0.0), t.get("norm_p", 2.0), grad_out=self.prior_grad,
                                            partials=self.prior_partials)
                n_reg = grid * _lib.BH_PRIOR_PARTIAL_STRIDE
            objective_value = total_objective.detach().reshape(-1)
            if objective_value.dtype != torch.float32 or not objective_value.is_contiguous():
                objective_value = objective_value.to(torch.float32).contiguous()
            _lib.check(
                lib.bh_loss_commit(_lib.ptr(self.state), _lib.ptr(self.history), self.max_iterations,
                                   _lib.ptr(objective_value), _lib.ptr(self.prior_partials), n_reg, None, None, stream),
                "bh_loss_commit",
            )
            if len(self.slots) > 1:
                # Joint data + label attack (optimization_with_label_attack.py:177-190): noise, clipping by each tensor's OWN norm,
                # sign and the Adam step of ALL optimised tensors in ONE sum-of-squares launch and ONE kernel-B launch (SURVEY
                # section 8 a17: "kernel B launched over a 2-tensor list"; round 5: three launches per tensor).
                entries = (_lib.StepSlot * len(self.slots))()
                keep = []  # the operands of the launches, alive until they are enqueued
                for idx, (slot, grad) in enumerate(zip(self.slots, grads)):
                    grad = grad.contiguous()
                    reg_grad = self.prior_grad if (self.use_prior and idx == 0) else None
                    noise = None
                    if self.langevin > 0:  # :177-180, tensor by tensor like the reference's loop
                        noise = torch.randn(grad.shape, dtype=grad.dtype).to(grad.device) if self.host_noise else torch.randn_like(grad)
                    keep.append((grad, noise))
                    entry = entries[idx]
                    entry.params = slot["P"]
                    entry.x, entry.g, entry.g_reg, entry.noise = slot["x"].data_ptr(), grad.data_ptr(), _lib.ptr(reg_grad).value, _lib.ptr(noise).value
                    entry.m, entry.v, entry.best = slot["m"].data_ptr(), slot["v"].data_ptr(), slot["best"].data_ptr()
                if self.grad_clip is not None:
                    if self.list_norm_ws is None:
                        rows = lib.bh_step_list_norm_rows(len(self.slots), entries)
                        _lib.check(min(rows, 0), "bh_step_list_norm_rows")
                        self.list_norm_ws = torch.empty(max(rows, 1), dtype=torch.float64, device=device)
                    _lib.check(lib.bh_grad_norm_list(_lib.ptr(self.state), len(self.slots), entries, _lib.ptr(self.sched_dev),
                                                     _lib.ptr(self.list_norm_ws), stream), "bh_grad_norm_list")
                _lib.check(lib.bh_candidate_step_list(_lib.ptr(self.state), _lib.ptr(self.sched_dev), len(self.slots), entries,
                                                      _lib.ptr(self.list_norm_ws), stream), "bh_candidate_step_list")
                att.current_task_loss = task_loss
                return
            for idx, (slot, grad) in enumerate(zip(self.slots, grads)):
                grad = grad.contiguous()
                reg_grad = self.prior_grad if (self.use_prior and idx == 0) else None
                noise = None
                if self.langevin > 0:  # optimization_based_attack.py:169
                    noise = torch.randn(grad.shape, dtype=grad.dtype).to(grad.device) if self.host_noise else torch.randn_like(grad)
                if self.grad_clip is not None:
                    _lib.check(
                        lib.bh_grad_norm(_lib.ptr(self.state), _lib.ptr(grad), _lib.ptr(reg_grad), _lib.ptr(noise),
                                         grad.numel(), _lib.ptr(self.sched_dev), self.langevin, _lib.ptr(self.norm_ws), stream),
                        "bh_grad_norm",
                    )
                _lib.check(
                    lib.bh_candidate_step(_lib.ptr(self.state), _lib.ptr(self.sched_dev), slot["P"], _lib.ptr(slot["x"]),
                                          _lib.ptr(grad), _lib.ptr(reg_grad), _lib.ptr(noise), _lib.ptr(slot["m"]),
                                          _lib.ptr(slot["v"]), _lib.ptr(slot["best"]), stream),
                    "bh_candidate_step",
                )
        att.current_task_loss = task_loss

    def read_state(self):
        """Synchronising read of the trial record."""
        host = self.state.cpu()
        return dict(
            it=host[_lib.STATE_IT].item(), dead=host[_lib.STATE_DEAD].item() != 0, first_bad=host[_lib.STATE_FIRST_BAD].item(),
            total=host[_lib.STATE_TOTAL : _lib.STATE_TOTAL + 1].view(torch.float32).item(),
            minimum=host[_lib.STATE_MIN : _lib.STATE_MIN + 1].view(torch.float32).item(),
        )

    def loss_history(self, iterations_run):
        """Objective values the reference would have appended to stats (stops before the first non-finite one, :131-135)."""
        first_bad = self.read_state()["first_bad"]
        kept = iterations_run if first_bad < 0 else min(first_bad, iterations_run)
        return self.history[:kept].cpu().tolist()

    def best(self):
        return [slot["best"].detach() for slot in self.slots]


class _TableScheduler:
    """`scheduler.step()` for the generic loop: writes the precomputed rate into the optimiser (common.py:22-38)."""

    def __init__(self, optimizer, lrs):
        self.optimizer, self.lrs, self.k = optimizer, lrs, 0
        self._apply()

    def _apply(self):
        lr = self.lrs[min(self.k, len(self.lrs) - 1)]
        for group in self.optimizer.param_groups:
            group["lr"] = lr

    def step(self):
        self.k += 1
        self._apply()


class HipOptimizationJointAttacker(HipOptimizationAttacker):
    """Joint data + label optimisation (reference ``OptimizationJointAttacker``, optimization_with_label_attack.py)."""

    def _recover_label_information(self, user_data, server_payload, rec_models, embedding_grads=None):  # :42-49
        num_data_points = user_data[0]["metadata"]["num_data_points"]
        metadata = server_payload[0]["metadata"]
        if metadata["task"] == "classification":
            return self._initialize_data([num_data_points, metadata.classes])
        return self._initialize_data([num_data_points, self.data_shape[0], metadata.vocab_size])

    def reconstruct(self, server_payload, shared_data, server_secrets=None, initial_data=None, dryrun=False):
        if shared_data[0]["metadata"]["labels"] is not None:  # :54-58 (checked before any work here)
            raise ValueError(
                "Joint optimization only makes sense if no labels are provided. "
                "Switch to attack.attack_type=optimization instead"
            )
        return super().reconstruct(server_payload, shared_data, server_secrets, initial_data, dryrun)

    def _run_trial(self, rec_model, shared_data, labels, stats, trial, initial_data=None, dryrun=False, init_state=None):
        if len(self.regularizers) > 0:
            # optimization_with_label_attack.py:94 references an undefined name as soon as a regulariser is configured
            raise NameError("name 'labels' is not defined (reference behaviour: joint attack with regularisers is broken)")
        return super()._run_trial(rec_model, shared_data, labels, stats, trial, initial_data, dryrun, init_state)

    def _draw_initial_state(self, num_points, label_template):  # :98-99
        data = self._initialize_data([num_points, *self.data_shape])
        label_candidate = self._initialize_data(label_template.shape)
        return (data, label_candidate)

    def _labels_for_objective(self, candidates, labels):  # :168-170
        return candidates[1].softmax(dim=-1)

    def _boxed_flags(self, candidates):  # only the data tensor is projected (:124-128)
        return [bool(self.cfg.optim.boxed), False]

    @staticmethod
    def _solution_data(solution):
        return solution[0]

    @staticmethod
    def _score_labels(solution, labels):
        # The reference scores with argmax of the label *template*, not of the optimised labels (:65-67).
        return labels.argmax(dim=-1)

    def _package(self, optimal, labels):
        return dict(data=optimal[0], labels=labels.argmax(dim=-1))

    def _attach_raw_embeddings(self, reconstructed_data, raw):  # :78-80
        reconstructed_data["raw_embeddings"] = raw


def prepare_attack(model, loss, cfg_attack, setup=_DEFAULT_SETUP):
    """Factory with the reference's signature (breaching/attacks/__init__.py:12-34) for the two optimisation branches."""
    if cfg_attack.attack_type == "optimization":
        return HipOptimizationAttacker(model, loss, cfg_attack, setup)
    if cfg_attack.attack_type == "joint-optimization":
        return HipOptimizationJointAttacker(model, loss, cfg_attack, setup)
    if cfg_attack.attack_type in (
        "multiscale", "analytic", "april-analytic", "imprint-readout", "decepticon-readout", "recursive",
        "permutation-optimization",
    ):
        raise NotImplementedError(
            f"attack_type={cfg_attack.attack_type} is outside the MI355X hot-path scope; use the reference attacker."
        )
    raise ValueError(f"Invalid type of attack {cfg_attack.attack_type} given.")
