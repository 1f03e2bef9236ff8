"""Gradient-matching objectives on the HIP kernels (kernel A), behind the reference's objective interface.

reference: breaching/attacks/auxiliaries/objectives.py
  * ``GradientLoss`` (:9-72): ``initialize(loss_fn, cfg_impl, local_hyperparams)`` and
    ``forward(model, gradient_data, candidate, labels) -> (objective, task_loss.detach())``
  * the list reductions ``Euclidean`` :75-95, ``EuclideanTag`` :98-141, ``L1Loss`` :144-166, ``CosineSimilarity``
    :169-196, ``AngularSimilarity`` :199-217, ``MaskedCosineSimilarity`` :220-244, ``FastCosineSimilarity`` :247-276
  * ``objective_lookup`` :496-506

  * the FedAvg unroll ``_grad_fn_multi_step`` :48-72 and the Pearlmutter finite-difference objectives :279-493

The victim model's forward / backward / double backward stay on PyTorch-ROCm; the reduction over the per-parameter
gradient list and its derivative (kernel A), the parameter-list updates of the FedAvg local steps and the offset
parameters of the Pearlmutter objectives (multi-tensor kernels, csrc/mt_kernels.hip) run here.
"""

import ctypes
from ctypes import c_int32, c_int64, c_void_p

import torch
from torch.autograd.function import once_differentiable

from . import _lib


def _require_cuda(t, what):
    if not t.is_cuda:
        raise RuntimeError(
            f"{what} lives on {t.device}; the HIP gradient-matching path needs tensors on a ROCm device "
            "(there is no CPU fallback)."
        )


class GradientMatchPlan:
    """Static state for one (observed gradient list, objective) pair: chunk table, packed data, launch bounds.

    Built once per attack (reference keeps the Python list, base_attack.py:214-220); pairs beyond the shorter list are
    dropped like ``zip`` does at objectives.py:190.
    """

    def __init__(self, gradient_data, n_pairs=None, rows_cap=0, cache_policy=0, cache_policy_bwd=None):
        lib = _lib.load()
        self.rows_cap = int(rows_cap)  # workgroups per forward launch group (0 = BH_GM_DEFAULT_ROWS): a plan field, not library state
        # cache policy of the streaming accesses (0 = by size: non-temporal loads when the two lists exceed the Infinity Cache;
        # 1 plain; 2 non-temporal loads; 3 non-temporal loads and backward stores) -- _lib.GM_CACHE_*
        self.cache_policy = int(cache_policy)                                                      # forward launch
        self.cache_policy_bwd = int(cache_policy if cache_policy_bwd is None else cache_policy_bwd)  # backward launch
        tensors = list(gradient_data)
        if n_pairs is not None:
            tensors = tensors[:n_pairs]
        if len(tensors) == 0:
            raise ValueError("Gradient matching needs at least one gradient tensor.")
        for t in tensors:
            _require_cuda(t, "observed gradient")
            if t.dtype != torch.float32:
                raise NotImplementedError(f"HIP gradient matching computes in fp32; got {t.dtype} (impl.dtype must be float).")
        # Identity of the list this plan was packed from: strong references to the caller's tensors (plain data, no
        # autograd graph) plus their version counters.  An address can be recycled by the caching allocator once a list is
        # freed (one attacker reused for user after user, benchmark_breaches.py:60-70), so addresses prove nothing;
        # object identity of tensors that are kept alive does, and the version counter catches in-place edits.
        self._sources = tuple(tensors)
        self._versions = tuple(t._version for t in tensors)
        # packed in logical (row-major) order whatever the caller's strides are: the reconstructed gradients are compared
        # element by element in that order (a channels_last user gradient must not be packed in its physical order)
        tensors = [t if t.is_contiguous() else t.contiguous() for t in tensors]
        self.device = tensors[0].device
        self.n_tensors = len(tensors)
        self.shapes = [tuple(t.shape) for t in tensors]
        self.numels = [t.numel() for t in tensors]
        self.total_elements = sum(self.numels)

        numel_arr = (c_int64 * self.n_tensors)(*self.numels)
        n_chunks, flat_elems = c_int64(0), c_int64(0)
        _lib.check(lib.bh_gm_table_size(self.n_tensors, numel_arr, ctypes.byref(n_chunks), ctypes.byref(flat_elems)), "bh_gm_table_size")
        self.n_chunks, self.flat_elems = n_chunks.value, flat_elems.value
        if self.n_chunks == 0:
            raise ValueError("Gradient list holds no elements.")
        self._chunks_host = (_lib.GmChunk * self.n_chunks)()
        flat_off = (c_int64 * self.n_tensors)()
        _lib.check(lib.bh_gm_build_table(self.n_tensors, numel_arr, self._chunks_host, self.n_chunks, flat_off), "bh_gm_build_table")
        self.flat_offsets = list(flat_off)
        n_groups = lib.bh_gm_num_groups(self.n_tensors)
        self._group_bounds = (c_int32 * (n_groups + 1))()
        _lib.check(lib.bh_gm_group_bounds(self.n_tensors, self._chunks_host, self.n_chunks, self._group_bounds), "bh_gm_group_bounds")
        # rows of the forward workspace: one per persistent workgroup (<= 2048 per launch group)
        self.n_rows = _lib.check(lib.bh_gm_fwd_rows(self.n_tensors, self._group_bounds, self.rows_cap), "bh_gm_fwd_rows")

        # device copies
        raw = torch.frombuffer(bytearray(bytes(self._chunks_host)), dtype=torch.uint8)
        self.chunks_dev = raw.to(self.device)
        self.data_flat = torch.empty(max(self.flat_elems, 4), dtype=torch.float32, device=self.device)
        self._dummy = torch.zeros(4, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            srcs = self._pointer_array(tensors)
            _lib.check(
                lib.bh_gm_pack(self.n_tensors, srcs, _lib.ptr(self.chunks_dev), self.n_chunks, self._group_bounds,
                               _lib.ptr(self.data_flat), _lib.current_stream_handle(self.device)),
                "bh_gm_pack",
            )
        # optional per-launch timing (bench.py): lists of (start, end) hipEvent pairs on the launch stream
        self.timers = None
        # device-side running [sum of forward-launch spans in wall-clock ticks, launches]: bench.py switches it on
        # (`span_enabled`), zeroes and reads it; off by default -- it costs the finalize kernel two dependent loads
        self.span_accum = torch.zeros(2, dtype=torch.float64, device=self.device)
        self.span_enabled = False

    def forward_span_us(self, reset=False):
        """Average span of the forward launches since the last reset, from the device wall clock (synchronises)."""
        total, count = self.span_accum.cpu().tolist()
        if reset:
            self.span_accum.zero_()
        khz = _lib.load().bh_wall_clock_khz()
        return (total / count) / (khz / 1e3) if count > 0 and khz > 0 else None, int(count)

    def enable_timing(self):
        self.timers = dict(fwd=[], fin=[], bwd=[])

    def _timed(self, key):
        """A fresh (start, stop) pair of hipEvent handles, recorded by the C side right around the launch."""
        if self.timers is None or torch.cuda.is_current_stream_capturing():
            return None, None
        lib = _lib.load()
        pair = (c_void_p(), c_void_p())
        for ev in pair:
            _lib.check(lib.bh_event_create(ctypes.byref(ev)), "bh_event_create")
        self.timers[key].append(pair)
        return pair

    def drain_timers(self):
        """Elapsed microseconds per recorded launch, per direction; destroys the events."""
        lib = _lib.load()
        out = {}
        for key, pairs in (self.timers or {}).items():
            vals = []
            for a, b in pairs:
                ms = ctypes.c_float(0)
                _lib.check(lib.bh_event_elapsed_ms(a, b, ctypes.byref(ms)), "bh_event_elapsed_ms")
                vals.append(ms.value * 1e3)
                lib.bh_event_destroy(a)
                lib.bh_event_destroy(b)
            out[key] = vals
        self.timers = None
        return out

    # -- helpers ---------------------------------------------------------------------------------------------------
    def _prepare(self, tensors):
        """Contiguous, 16-byte aligned fp32 views of this iteration's gradients (copies only when unavoidable)."""
        out = []
        for i in range(self.n_tensors):
            t = tensors[i]
            if t.dtype != torch.float32:
                raise NotImplementedError(f"HIP gradient matching computes in fp32; got {t.dtype}.")
            if t.numel() != self.numels[i]:
                raise ValueError(
                    f"Gradient {i} has {t.numel()} elements but the observed gradient has {self.numels[i]}."
                )
            if not t.is_contiguous():
                t = t.contiguous()
            if t.numel() and t.data_ptr() % 16:
                t = t.clone(memory_format=torch.contiguous_format)
            out.append(t)
        return out

    def _pointer_array(self, tensors):
        dummy = self._dummy.data_ptr()
        addrs = [(t.data_ptr() if t.numel() else dummy) for t in tensors]
        for i, a in enumerate(addrs):
            if a % 16:
                # only reachable for the observed gradient at pack time; make an aligned copy
                tensors[i] = tensors[i].clone(memory_format=torch.contiguous_format)
                addrs[i] = tensors[i].data_ptr()
        # No reference to `tensors` is kept: launches are stream ordered, and holding this iteration's gradients (which
        # carry their autograd graph) would keep stale AccumulateGrad nodes alive and break hipGraph capture.
        return (c_void_p * len(addrs))(*addrs)

    def matches(self, gradient_data, n_pairs):
        """True when this plan was packed from exactly these tensor objects and none was written to since."""
        if n_pairs != self.n_tensors or len(gradient_data) < n_pairs:
            return False
        for i in range(n_pairs):
            t = gradient_data[i]
            if t is not self._sources[i] or t._version != self._versions[i]:
                return False
        return True

    # -- launches --------------------------------------------------------------------------------------------------
    def forward(self, kind, rec, scale, tag_scale=0.0, fudge=1e-7, weights=None, fd_eps=0.0):
        """Enqueue forward reduction + finalize; returns the fresh fp32 statistics record [BH_GM_STAT_WORDS].
        ``fd_eps`` > 0 also fills the finite-difference words the Pearlmutter objectives read."""
        lib = _lib.load()
        stats = torch.empty(_lib.BH_GM_STAT_WORDS, dtype=torch.float32, device=self.device)
        # per-call workspace (16 KB): trials that run concurrently on different streams share this plan
        partials = torch.empty(self.n_rows * _lib.BH_GM_PARTIAL_STRIDE, dtype=torch.float64, device=self.device)
        stream = _lib.current_stream_handle(self.device)
        ptrs = self._pointer_array(rec)
        ev0, ev1 = self._timed("fwd")
        _lib.check(
            lib.bh_gm_fwd(kind, self.n_tensors, ptrs, _lib.ptr(self.data_flat), _lib.ptr(self.chunks_dev), self.n_chunks,
                          self._group_bounds, _lib.ptr(weights), float(tag_scale), _lib.ptr(partials), self.rows_cap, self.cache_policy, stream, ev0, ev1),
            "bh_gm_fwd",
        )
        ev0, ev1 = self._timed("fin")
        _lib.check(
            lib.bh_gm_finalize(kind, _lib.ptr(partials), self.n_rows, float(scale), float(tag_scale), float(fudge), float(fd_eps),
                               _lib.ptr(stats), _lib.ptr(self.span_accum if self.span_enabled else None), stream, ev0, ev1),
            "bh_gm_finalize",
        )
        return stats

    def backward(self, kind, rec, stats, gout, weights=None):
        """Enqueue the backward launch; returns the flat gradient buffer (views are cut by the caller)."""
        lib = _lib.load()
        grad_flat = torch.empty(max(self.flat_elems, 4), dtype=torch.float32, device=self.device)
        stream = _lib.current_stream_handle(self.device)
        ptrs = self._pointer_array(rec)
        ev0, ev1 = self._timed("bwd")
        _lib.check(
            lib.bh_gm_bwd(kind, self.n_tensors, ptrs, _lib.ptr(self.data_flat), _lib.ptr(self.chunks_dev), self.n_chunks,
                          self._group_bounds, _lib.ptr(weights), _lib.ptr(stats), _lib.ptr(gout), _lib.ptr(grad_flat), self.cache_policy_bwd, stream, ev0, ev1),
            "bh_gm_bwd",
        )
        return grad_flat

    def split(self, grad_flat):
        return [grad_flat[o : o + n].view(s) for o, n, s in zip(self.flat_offsets, self.numels, self.shapes)]

    def patched_parameters(self, params, grads, stats, mult):
        """theta + mult * eps_n * v(grad, data) for the whole list in one launch (bh_mt_patch), v's coefficients read from
        the statistics record on the device.  Returns T views of one packed buffer."""
        lib = _lib.load()
        if getattr(self, "_mt_bounds", None) is None:
            self._mt_bounds = (c_int32 * (lib.bh_mt_num_groups(self.n_tensors) + 1))()
            _lib.check(lib.bh_mt_group_bounds(self.n_tensors, self._chunks_host, self.n_chunks, self._mt_bounds), "bh_mt_group_bounds")
        out = torch.empty(max(self.flat_elems, 4), dtype=torch.float32, device=self.device)
        coef = c_void_p(stats.data_ptr() + _lib.GM_STAT_PATCH_D * 4)
        _lib.check(
            lib.bh_mt_patch(self.n_tensors, self._pointer_array(params), self._pointer_array(grads), _lib.ptr(self.data_flat), coef,
                            float(mult), _lib.ptr(self.chunks_dev), self.n_chunks, self._mt_bounds, _lib.ptr(out),
                            _lib.current_stream_handle(self.device)),
            "bh_mt_patch",
        )
        return self.split(out)


class ListLayout:
    """Chunk table and packed layout for a list of tensors of given shapes (the multi-tensor elementwise kernels)."""

    def __init__(self, shapes, device):
        lib = _lib.load()
        self.device = torch.device(device)
        self.shapes = [tuple(s) for s in shapes]
        self.n_tensors = len(self.shapes)
        self.numels = [int(torch.Size(s).numel()) for s in self.shapes]
        numel_arr = (c_int64 * self.n_tensors)(*self.numels)
        n_chunks, flat_elems = c_int64(0), c_int64(0)
        _lib.check(lib.bh_gm_table_size(self.n_tensors, numel_arr, ctypes.byref(n_chunks), ctypes.byref(flat_elems)), "bh_gm_table_size")
        self.n_chunks, self.flat_elems = n_chunks.value, flat_elems.value
        if self.n_chunks == 0:
            raise ValueError("Tensor list holds no elements.")
        chunks = (_lib.GmChunk * self.n_chunks)()
        flat_off = (c_int64 * self.n_tensors)()
        _lib.check(lib.bh_gm_build_table(self.n_tensors, numel_arr, chunks, self.n_chunks, flat_off), "bh_gm_build_table")
        self.flat_offsets = list(flat_off)
        self.mt_bounds = (c_int32 * (lib.bh_mt_num_groups(self.n_tensors) + 1))()
        _lib.check(lib.bh_mt_group_bounds(self.n_tensors, chunks, self.n_chunks, self.mt_bounds), "bh_mt_group_bounds")
        self.chunks_dev = torch.frombuffer(bytearray(bytes(chunks)), dtype=torch.uint8).to(self.device)
        self._dummy = torch.zeros(4, dtype=torch.float32, device=self.device)

    def matches(self, tensors):
        return len(tensors) == self.n_tensors and all(tuple(t.shape) == s for t, s in zip(tensors, self.shapes))

    def prepare(self, tensors, what):
        """Contiguous, 16-byte aligned fp32 tensors (copies only when unavoidable)."""
        out = []
        for t in tensors:
            _require_cuda(t, what)
            if t.dtype != torch.float32:
                raise NotImplementedError(f"HIP multi-tensor kernels compute in fp32; got {t.dtype}.")
            if not t.is_contiguous():
                t = t.contiguous()
            if t.numel() and t.data_ptr() % 16:
                t = t.clone(memory_format=torch.contiguous_format)
            out.append(t)
        return out

    def pointers(self, tensors, nullable=False):
        dummy = self._dummy.data_ptr()
        return (c_void_p * self.n_tensors)(*[(0 if (t is None and nullable) else (t.data_ptr() if t.numel() else dummy)) for t in tensors])

    def empty_flat(self):
        return torch.empty(max(self.flat_elems, 4), dtype=torch.float32, device=self.device)

    def split(self, flat):
        return [flat[o : o + n].view(s) for o, n, s in zip(self.flat_offsets, self.numels, self.shapes)]


class _LocalStepFunction(torch.autograd.Function):
    """One local SGD step of the FedAvg unroll as a single node: (params, grads) -> params + alpha * grads [- base], all T
    tensors in one launch (views of one packed buffer).  Backward: d/d params = identity, d/d grads = alpha (one launch)."""

    @staticmethod
    def forward(ctx, layout, alpha, with_base, *tensors):
        lib = _lib.load()
        T = layout.n_tensors
        params = layout.prepare(tensors[:T], "parameter")
        grads = layout.prepare(tensors[T : 2 * T], "step gradient")
        base = layout.prepare(tensors[2 * T : 3 * T], "server parameter") if with_base else None
        out = layout.empty_flat()
        with torch.cuda.device(layout.device):
            _lib.check(
                lib.bh_mt_axpy(T, layout.pointers(params), layout.pointers(grads), layout.pointers(base) if with_base else None,
                               float(alpha), _lib.ptr(layout.chunks_dev), layout.n_chunks, layout.mt_bounds, _lib.ptr(out),
                               _lib.current_stream_handle(layout.device)),
                "bh_mt_axpy",
            )
        ctx.layout, ctx.alpha, ctx.with_base = layout, float(alpha), with_base
        return tuple(layout.split(out))

    @staticmethod
    @once_differentiable
    def backward(ctx, *gouts):
        lib = _lib.load()
        layout, T = ctx.layout, ctx.layout.n_tensors
        present = iter(layout.prepare([g for g in gouts if g is not None], "upstream gradient"))
        gouts = [None if g is None else next(present) for g in gouts]
        grad_params = list(gouts)  # identity
        grad_grads = [None] * T
        if any(ctx.needs_input_grad[3 + T : 3 + 2 * T]):
            flat = layout.empty_flat()
            with torch.cuda.device(layout.device):
                _lib.check(
                    lib.bh_mt_scale(T, layout.pointers(gouts, nullable=True), ctx.alpha, _lib.ptr(layout.chunks_dev), layout.n_chunks,
                                    layout.mt_bounds, _lib.ptr(flat), _lib.current_stream_handle(layout.device)),
                    "bh_mt_scale",
                )
            grad_grads = layout.split(flat)
        grad_base = [None] * T if ctx.with_base else []
        if ctx.with_base and any(ctx.needs_input_grad[3 + 2 * T :]):
            grad_base = [None if g is None else -g for g in gouts]
        return (None, None, None, *grad_params, *grad_grads, *grad_base)


class _GradMatchFunction(torch.autograd.Function):
    """objective(rec_0, ..., rec_{T-1}) as one differentiable node; backward hands autograd T views of one buffer."""

    @staticmethod
    def forward(ctx, plan, kind, scale, tag_scale, fudge, weights, *rec):
        rec = plan._prepare(rec)
        with torch.cuda.device(plan.device):
            stats = plan.forward(kind, rec, scale, tag_scale, fudge, weights)
        ctx.plan, ctx.kind, ctx.weights, ctx.stats = plan, kind, weights, stats
        ctx.n_inputs = len(rec)
        ctx.save_for_backward(*rec)
        return stats[0:1]  # shape (1,) like `gradient_rec[0].new_zeros(1,)` at objectives.py:91

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        plan = ctx.plan
        rec = list(ctx.saved_tensors)
        gout = gout.contiguous().to(torch.float32)
        with torch.cuda.device(plan.device):
            grad_flat = plan.backward(ctx.kind, rec, ctx.stats, gout, ctx.weights)
        grads = plan.split(grad_flat)
        return (None, None, None, None, None, None, *grads)


class HipGradientLoss(torch.nn.Module):
    """Counterpart of ``GradientLoss`` (objectives.py:9-72) whose list reduction runs in kernel A."""

    kind_name = None

    def __init__(self, scale=1.0, task_regularization=0.0, **kwargs):
        super().__init__()
        self.scale = scale
        self.task_regularization = task_regularization
        self._plans = []

    @property
    def _plan(self):
        """The most recently used plan (bench.py reads its timers); None before the first objective evaluation."""
        return self._plans[0] if self._plans else None

    # objectives.py:16-24
    def initialize(self, loss_fn, cfg_impl, local_hyperparams=None):
        self.loss_fn = loss_fn
        self.local_hyperparams = local_hyperparams
        self.cfg_impl = cfg_impl
        if getattr(cfg_impl, "mixed_precision", False):
            raise NotImplementedError("The HIP gradient-matching path is fp32 only (impl.mixed_precision must be False).")
        self._grad_fn = self._single_step_gradient if local_hyperparams is None else self._multi_step_update
        # A new trial (or a new user: the reference builds the attacker once and calls reconstruct per user,
        # benchmark_breaches.py:60-70) starts from freshly packed observed gradients -- 20 us -- never from a cached plan.
        self._plans = []

    def prepare(self, models, shared_data):
        """Pack every observed gradient list now, on the current stream (one plan per model / query).  Called before the
        loop so that no plan is ever built inside it (a build does a blocking host-to-device copy, which a hipGraph
        capture cannot contain) and before trials fork onto side streams (they only read the plans)."""
        for model, data in zip(models, shared_data):
            n_rec = sum(1 for _ in model.parameters())
            self._plan_for_count(n_rec, data["gradients"])

    # objectives.py:26-32
    def forward(self, model, gradient_data, candidate, labels):
        gradient, task_loss = self._grad_fn(model, candidate, labels)
        objective = self.gradient_based_loss(gradient, gradient_data)
        if self.task_regularization != 0:
            objective = objective + self.task_regularization * task_loss
        return objective, task_loss.detach()

    def _plan_for(self, gradient_rec, gradient_data):
        return self._plan_for_count(len(gradient_rec), gradient_data)

    def _plan_for_count(self, n_rec, gradient_data):
        n_pairs = min(n_rec, len(gradient_data))
        for plan in self._plans:
            if plan.matches(gradient_data, n_pairs):
                return plan
        tuning = dict(rows_cap=0, cache_policy=0, cache_policy_bwd=0)  # measurements only: cfg.impl.gm_rows_cap / gm_cache_policy[_bwd]
        impl = getattr(self, "cfg_impl", None)
        for key in tuning:
            try:
                tuning[key] = int(impl[f"gm_{key}"] or 0)
            except (KeyError, AttributeError, TypeError):
                pass
        plan = GradientMatchPlan(gradient_data, n_pairs, **tuning)
        self._plans.append(plan)  # one plan per observed list: multi-query / multi-model payloads alternate between them
        return plan

    def _weights(self, plan, n_rec):
        return None

    def _extra(self):
        return 0.0, 1e-7  # tag_scale, fudge

    def gradient_based_loss(self, gradient_rec, gradient_data):
        gradient_rec = list(gradient_rec)
        for t in gradient_rec[:1]:
            _require_cuda(t, "reconstructed gradient")
        plan = self._plan_for(gradient_rec, gradient_data)
        tag_scale, fudge = self._extra()
        kind = _lib.GM_KINDS[self.kind_name]
        return _GradMatchFunction.apply(plan, kind, float(self.scale), tag_scale, fudge,
                                        self._weights(plan, len(gradient_rec)), *gradient_rec[: plan.n_tensors])

    # objectives.py:40-46
    def _single_step_gradient(self, model, candidate, labels):
        model.zero_grad()
        task_loss = self.loss_fn(model(candidate), labels)
        gradient = torch.autograd.grad(task_loss, tuple(model.parameters()), create_graph=True)
        return gradient, task_loss

    # objectives.py:48-72 (FedAvg unroll).  The functional forward / backward of each local step stay PyTorch; the
    # parameter update of every step (`param - lr * grad` over the whole list) and the final `p_local - p_server` are one
    # multi-tensor launch each (bh_mt_axpy), the last step fused with the difference.
    def _multi_step_update(self, model, candidate, labels):
        from torch.func import functional_call

        model.zero_grad()
        names = [n for n, _ in model.named_parameters()]
        server = [p for _, p in model.named_parameters()]
        buffers = dict(model.named_buffers())
        hp = self.local_hyperparams
        layout = getattr(self, "_step_layout", None)
        if layout is None or not layout.matches(server) or layout.device != server[0].device:
            layout = self._step_layout = ListLayout([p.shape for p in server], server[0].device)
        params = server
        seen = 0
        task_loss = None
        steps = int(hp["steps"])
        if steps < 1:
            raise ValueError("local_hyperparams['steps'] must be at least 1.")
        for i in range(steps):
            data = candidate[seen : seen + hp["data_per_step"]]
            seen = (seen + hp["data_per_step"]) % candidate.shape[0]
            step_labels = hp["labels"][i]
            task_loss = self.loss_fn(functional_call(model, ({**dict(zip(names, params)), **buffers},), (data,)), step_labels)
            step_grad = torch.autograd.grad(task_loss, params, create_graph=True)
            last = i + 1 == steps
            params = _LocalStepFunction.apply(layout, -float(hp["lr"]), last, *params, *step_grad, *(server if last else ()))
        return list(params), task_loss  # after the last step: p_local - p_server


class HipEuclidean(HipGradientLoss):
    kind_name = "euclidean"

    def __repr__(self):
        return f"Euclidean loss with scale={self.scale} and task reg={self.task_regularization} [HIP gfx950]"


class HipL1Loss(HipGradientLoss):
    kind_name = "l1"

    def __repr__(self):
        return f"L1 loss with scale={self.scale} and task reg={self.task_regularization} [HIP gfx950]"


class HipCosineSimilarity(HipGradientLoss):
    kind_name = "cosine-similarity"

    def __repr__(self):
        return f"Cosine Similarity with scale={self.scale} and task reg={self.task_regularization} [HIP gfx950]"


class HipAngularSimilarity(HipGradientLoss):
    kind_name = "angular"

    def __init__(self, scale=1.0, task_regularization=0.0, fudge_factor=1e-7, **kwargs):
        super().__init__(scale, task_regularization)
        self.fudge_factor = 1e-7  # the reference ignores its ctor argument (objectives.py:208)

    def _extra(self):
        return 0.0, self.fudge_factor

    def __repr__(self):
        return f"Angular Similarity with scale={self.scale} and task reg={self.task_regularization} [HIP gfx950]"


class HipMaskedCosineSimilarity(HipGradientLoss):
    kind_name = "masked-cosine-similarity"

    def __init__(self, scale=1.0, mask_value=1e-6, task_regularization=0.0, **kwargs):
        super().__init__(scale, task_regularization)
        self.mask_value = 1e-6  # the reference ignores its ctor argument (objectives.py:228)

    def __repr__(self):
        return (
            f"Masked Cosine Similarity with scale={self.scale} and task reg={self.task_regularization}. "
            f"Mask val={self.mask_value} [HIP gfx950]"
        )


class HipFastCosineSimilarity(HipGradientLoss):
    kind_name = "fast-cosine-similarity"

    def __repr__(self):
        return f"Fast Cosine Similarity with scale={self.scale} and task reg={self.task_regularization} [HIP gfx950]"


class HipEuclideanTag(HipGradientLoss):
    """objectives.py:98-141; the per-tensor weights ride along in a small device vector."""

    kind_name = "tag-euclidean"

    def __init__(self, scale=1.0, task_regularization=0.0, tag_scale=0.1, scale_scheme="linear", **kwargs):
        super().__init__(scale, task_regularization)
        self.tag_scale = tag_scale
        self.scale_scheme = scale_scheme
        self._weight_cache = None

    def _extra(self):
        return float(self.tag_scale), 1e-7

    def _weights(self, plan, n_rec):
        # The reference sizes the weight ramp by len(gradient_rec) and lets zip() drop the tail (objectives.py:117-125,
        # :139): with a tied decoder weight the reconstructed list is one longer than the observed one.
        cached = self._weight_cache
        if cached is not None and cached[0] is plan and cached[2] == n_rec:
            return cached[1]
        n = n_rec
        setup = dict(dtype=torch.float32, device=plan.device)
        if self.scale_scheme == "linear":  # objectives.py:117-118
            weights = torch.arange(n, 0, -1, **setup) / n
        elif self.scale_scheme == "exp":  # :119-122
            weights = torch.arange(n, 0, -1, **setup).softmax(dim=0)
            weights = weights / weights[0]
        else:  # :123-124
            weights = torch.ones(n, **setup)
        weights = weights[: plan.n_tensors].contiguous()
        self._weight_cache = (plan, weights, n_rec)
        return weights

    def __repr__(self):
        return (
            f"Tag loss with scale={self.scale}, weight scheme {self.scale_scheme}, L1 scale {self.tag_scale} "
            f"and task reg={self.task_regularization} [HIP gfx950]"
        )


class _AttachGradient(torch.autograd.Function):
    """value with a prescribed derivative: d value / d candidate := `grad` (the finite-difference estimate)."""

    @staticmethod
    def forward(ctx, candidate, value, grad):
        ctx.save_for_backward(grad)
        return value.clone()

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        (grad,) = ctx.saved_tensors
        return grad * gout, None, None


class HipPearlmutterEuclidean(HipGradientLoss):
    """objectives.py:279-460 -- the double backward is replaced by a finite difference of first-order gradients: with
    g = dL/dtheta, v = d objective / d g and eps_n = eps / |g|,
        d objective / d x  ~  scale * ( dL/dx (theta + eps_n v) - dL/dx (theta) ) / eps_n        ("forward"; also "backward",
    "central", "upwind").  The reference patches the live parameters in place and restores them afterwards; here the
    offset parameters are written out of place (bh_mt_patch, coefficients and eps_n stay on the device) and the second pass
    runs through `torch.func.functional_call`, so nothing needs restoring.  The estimate is attached to the objective value as
    its derivative with respect to the candidate (the reference adds it to `candidate.grad` directly, :354-356)."""

    kind_name = "pearlmutter-loss"

    def __init__(self, scale=1.0, eps=1e-3, level_gradients=False, fudge_factor=1e-6, task_regularization=0.0,
                 implementation="forward", **kwargs):
        super().__init__(scale, task_regularization)
        self.eps = eps
        self.level_gradients = level_gradients
        self.fudge_factor = fudge_factor
        self.implementation = implementation

    def initialize(self, loss_fn, cfg_impl, local_hyperparams=None):
        if local_hyperparams is not None:  # :304-305
            raise ValueError("This loss is only implemented for local gradients so far.")
        if self.implementation not in ("forward", "backward", "central", "upwind"):  # :316-317
            raise ValueError(f"Invalid finite difference implementation {self.implementation} given.")
        super().initialize(loss_fn, cfg_impl, None)

    def _offset_gradient(self, model, names, buffers, params, candidate, labels):
        from torch.func import functional_call

        loss = self.loss_fn(functional_call(model, ({**dict(zip(names, params)), **buffers},), (candidate,)), labels)
        (grad,) = torch.autograd.grad(loss, (candidate,), create_graph=False)
        return grad

    def forward(self, model, gradient_data, candidate, labels):
        model.zero_grad()
        names = [n for n, _ in model.named_parameters()]
        params = [p for _, p in model.named_parameters()]
        buffers = dict(model.named_buffers())
        task_loss = self.loss_fn(model(candidate), labels)
        *gradients, dLdx = torch.autograd.grad(task_loss, (*params, candidate), create_graph=False)
        if self.level_gradients and self.implementation in ("forward", "backward"):
            # :345-348, :373-376 -- the reference levels the gradients only in its forward / backward variants; its central
            # and upwind variants (:394-452) ignore the flag, and so do these
            grad_norm = torch.stack([g.pow(2).sum() for g in gradients]).sum().sqrt()
            torch._foreach_div_(gradients, max(grad_norm, self.fudge_factor))
        plan = self._plan_for(gradients, gradient_data)
        if plan.n_tensors != len(gradients):
            raise ValueError("Pearlmutter objectives need one observed gradient per model parameter.")  # foreach ops of :455
        gradients = plan._prepare(gradients)
        kind = _lib.GM_KINDS[self.kind_name]
        with torch.cuda.device(plan.device):
            stats = plan.forward(kind, gradients, float(self.scale), 0.0, 1e-7, None, fd_eps=float(self.eps))
            theta = plan._prepare([p.detach() for p in params])
            inv_step = stats[_lib.GM_STAT_FD_SCALE : _lib.GM_STAT_FD_SCALE + 1]  # scale / eps_n, on the device
            if self.implementation == "forward":  # :347-354
                shifted = self._offset_gradient(model, names, buffers, plan.patched_parameters(theta, gradients, stats, 1.0), candidate, labels)
                estimate = (shifted - dLdx) * inv_step
            elif self.implementation == "backward":  # :375-382
                shifted = self._offset_gradient(model, names, buffers, plan.patched_parameters(theta, gradients, stats, -1.0), candidate, labels)
                estimate = (dLdx - shifted) * inv_step
            else:
                plus = self._offset_gradient(model, names, buffers, plan.patched_parameters(theta, gradients, stats, 0.5), candidate, labels)
                minus = self._offset_gradient(model, names, buffers, plan.patched_parameters(theta, gradients, stats, -0.5), candidate, labels)
                if self.implementation == "central":  # :401-413
                    estimate = (plus - minus) * inv_step
                else:  # "upwind" :436-444 -- the reference takes torch.max / torch.min ALONG DIM 0 of dL/dx here; kept
                    step = stats[_lib.GM_STAT_FD_STEP : _lib.GM_STAT_FD_STEP + 1]
                    d_plus, d_minus = (plus - dLdx) / step, (dLdx - minus) / step
                    estimate = (torch.max(dLdx, 0)[0] * d_minus + torch.min(dLdx, 0)[0] * d_plus) * self.scale
        if self.task_regularization != 0:
            estimate = estimate + self.task_regularization * dLdx  # :356
        objective = _AttachGradient.apply(candidate, stats[0:1], estimate)
        return objective, task_loss.detach()

    def __repr__(self):
        return (
            f"Pearlmutter-type Finite Differences Loss with scale={self.scale} and task reg={self.task_regularization}."
            f"Finite Difference Eps: {self.eps}. Level gradients: {self.level_gradients}. "
            f"{f'Fudge-factor: {self.fudge_factor}' if self.level_gradients else ''} [HIP gfx950]"
        )


class HipPearlmutterCosine(HipPearlmutterEuclidean):
    """objectives.py:463-493 -- the same finite difference along the first-order direction of the cosine objective."""

    kind_name = "cosine-similarity"


# objectives.py:496-506
objective_lookup = {
    "euclidean": HipEuclidean,
    "cosine-similarity": HipCosineSimilarity,
    "masked-cosine-similarity": HipMaskedCosineSimilarity,
    "fast-cosine-similarity": HipFastCosineSimilarity,
    "angular": HipAngularSimilarity,
    "l1": HipL1Loss,
    "tag-euclidean": HipEuclideanTag,
    "pearlmutter-loss": HipPearlmutterEuclidean,
    "pearlmutter-cosine": HipPearlmutterCosine,
}
