"""One optimisation step of unsupervised OGC segmentation training — the body of the reference's
``Trainer._train_it`` (train_seg.py:47-86) — factored out so that the benchmark, the smoke test and the
training driver share it.  Semantics kept: views are flattened into the batch for the network, the loss is
called with ``step_w=True, it=it*b``, and the optimiser step is skipped when any gradient contains NaN
(train_seg.py:81-83; here one fused check and one host sync instead of one per parameter)."""
import torch

from .losses.seg_loss_unsup import (DynamicLoss, EntropyLoss, InvarianceLoss, RankLoss, SmoothLoss,
                                    UnsupervisedOGCLossSingleFrame,
                                    UnsupervisedOGCLoss)

KITTI_LOSS = dict(  # config/seg/kittisf/kittisf_unsup.yaml:40-56
    weights=[10.0, 0.1, 0.1], start_steps=[0, 100, 1000],
    dynamic_loss_params=dict(loss_norm=2),
    smooth_loss_params=dict(w_knn=3., w_ball_q=1., knn_loss_params=dict(k=32, radius=1., loss_norm=1),
                            ball_q_loss_params=dict(k=64, radius=2., loss_norm=1)),
    invariance_loss_params=dict(loss_norm=2))

SAPIEN_LOSS = dict(  # config/seg/sapien/sapien_unsup.yaml
    weights=[10.0, 0.1, 0.1], start_steps=[0, 0, 0],
    dynamic_loss_params=dict(loss_norm=2),
    smooth_loss_params=dict(w_knn=3., w_ball_q=1., knn_loss_params=dict(k=8, radius=0.1, loss_norm=1),
                            ball_q_loss_params=dict(k=16, radius=0.2, loss_norm=1)),
    invariance_loss_params=dict(loss_norm=2))


def build_criterion(cfg, single_frame=False):
    """single_frame: the Waymo variant of the loss (one frame per sample, train_seg_waymo.py:244-334)."""
    cls = UnsupervisedOGCLossSingleFrame if single_frame else UnsupervisedOGCLoss
    return cls(DynamicLoss(**cfg["dynamic_loss_params"]), SmoothLoss(**cfg["smooth_loss_params"]),
                               InvarianceLoss(**cfg["invariance_loss_params"]), EntropyLoss(), RankLoss(),
                               weights=cfg["weights"], start_steps=cfg["start_steps"])


class PendingStep:
    """Outcome of a train_step whose scalars are still on their way to the host.  ``result()`` -> (loss_dict, stepped)."""

    def __init__(self, pending_losses, bad_scalar):
        self._losses, self._bad, self._out = pending_losses, bad_scalar, None
        self.prefetched = None  # PrefetchedGeometry of the next batch, when train_step was given one

    def refresh(self):
        """Forget the values read so far: the device scalars were rewritten (a graph replay of the step)."""
        self._out = None
        if hasattr(self._losses, "_dict"):
            self._losses._dict = None
        return self

    def result(self):
        if self._out is None:
            losses = self._losses.resolve() if hasattr(self._losses, "resolve") else self._losses
            self._out = (losses, not bool(self._bad.get()[0]))
        return self._out


def make_optimizer(params, lr, weight_decay=0.0, capturable=False):
    """Adam as the reference builds it (train_seg.py:320).  On the GPU the fused implementation is used: the same
    update, and it can skip itself on the device when handed a `found_inf` flag, which is what lets train_step apply
    the reference's NaN-gradient rule (train_seg.py:81-83) without stopping to read the flag on the host."""
    params = list(params)
    fused = bool(params) and all(p.is_cuda for p in params)
    # capturable: step counts live on the device, so optimizer.step() can be part of a HIP graph (graph_step.py)
    return torch.optim.Adam(params, lr=lr, weight_decay=weight_decay, fused=fused, capturable=bool(capturable and fused))


ADAM_KERNEL = True   # the NaN rule + the Adam update as two launches of this library (csrc/adam.hip); False: torch's fused step


def prepare_adam_kernel(optimizer):
    """Build (or re-validate) the pointer tables of _adam_kernel_step without launching anything: what a HIP-graph capture of the
    step needs done beforehand.  True when the optimizer is in the configuration the kernel covers and its state exists."""
    return _adam_kernel_step(optimizer, tables_only=True) is True


def _adam_kernel_step(optimizer, tables_only=False):
    """The reference's NaN-gradient rule and optimizer.step() of a torch.optim.Adam (one parameter group, fp32 parameters on one
    GPU, L2 weight decay, no amsgrad) as ogc_adam_step: two launches over a chunk table instead of the eleven of
    _foreach_norm / stack / sum / isnan / _foreach_add_ / 3 x multi_tensor_apply / _foreach_sub_ (0.3 ms of a C4 step's main
    queue, 0.5 ms of its launch thread).  Works on the tensors of optimizer.state (exp_avg, exp_avg_sq, step), so state_dict() /
    load_state_dict() are unaffected.  Returns the flag (int32, 1 where a gradient held a NaN and the step was skipped on the
    device), or None when the optimizer is not in the configuration this covers (the caller then takes the torch path)."""
    import ctypes
    from . import _lib
    if not ADAM_KERNEL or type(optimizer) is not torch.optim.Adam or len(optimizer.param_groups) != 1:
        return None
    group = optimizer.param_groups[0]
    if not group["params"] or not group["params"][0].is_cuda:
        return None
    cache = getattr(optimizer, "_ogc_adam_tables", None)
    if cache is not None:  # load_state_dict() replaces the group dictionaries and the state tensors; .to() the parameters
        params = cache["params"]
        if cache["group"] is not group or len(group["params"]) != len(params):
            cache = None
        else:  # every pointer of the table must still be the live tensor's (any parameter or state tensor may have moved)
            state = optimizer.state
            live = [p.data_ptr() for p in group["params"]]
            for key in ("exp_avg", "exp_avg_sq", "step"):
                live += [state[p][key].data_ptr() if key in state.get(p, ()) else 0 for p in group["params"]]
            if live != cache["ptrs"]:
                cache = None
    if cache is None:
        if torch.cuda.is_current_stream_capturing():
            # the tables are uploaded with host-to-device copies: not inside a capture (graph_step.py builds them right before it
            # begins — prepare_adam_kernel — so a captured step does take this path)
            return None
        if (group.get("amsgrad") or group.get("maximize") or group.get("differentiable") or group.get("decoupled_weight_decay")
                or isinstance(group["lr"], torch.Tensor)):
            return None
        params = list(group["params"])
        L = _lib.load()
        if not params or len(params) > L.ogc_adam_max_tensors():
            return None
        states = [optimizer.state.get(p) for p in params]
        if any(not st or "exp_avg" not in st or not torch.is_tensor(st.get("step")) for st in states):
            return None  # first step: torch builds the state
        dev = params[0].device
        tensors = [[p for p in params], [st["exp_avg"] for st in states], [st["exp_avg_sq"] for st in states],
                   [st["step"] for st in states]]
        ok = (dev.type == "cuda" and all(t.device == dev and t.dtype == torch.float32 and t.is_contiguous()
                                          for row in tensors for t in row)
              and all(st["step"].numel() == 1 for st in states))
        if not ok:
            return None
        chunk = L.ogc_adam_chunk()
        rows = [[t.data_ptr() for t in row] for row in tensors] + [[p.numel() for p in params]]
        pieces = [(i, off) for i, p in enumerate(params) for off in range(0, p.numel(), chunk)]
        cache = {"group": group, "params": params, "ptrs": [v for row in rows[:4] for v in row],
                 "table": torch.tensor(rows, dtype=torch.int64, device=dev),
                 "chunks": torch.tensor(pieces, dtype=torch.int32, device=dev), "n_chunks": len(pieces),
                 "snapshot": torch.empty(len(params), dtype=torch.float32, device=dev),
                 "array": ctypes.c_void_p * len(params)}
        optimizer._ogc_adam_tables = cache
    if tables_only:
        return True
    params = cache["params"]
    ptrs = []
    for p in params:
        g = p.grad
        if g is None or g.dtype != torch.float32 or g.device != p.device or not g.is_contiguous() or g.shape != p.shape:
            return None  # parameters without a (plain) gradient this step: torch's own filtering
        ptrs.append(g.data_ptr())
    flag = torch.zeros(1, dtype=torch.int32, device=params[0].device)
    beta1, beta2 = group["betas"]
    with torch.cuda.device(params[0].device):  # (the launch goes to the current device's stream)
        _lib.call("ogc_adam_step", len(params), cache["n_chunks"], cache["table"].data_ptr(), cache["chunks"].data_ptr(),
                  cache["array"](*ptrs), cache["snapshot"].data_ptr(), flag.data_ptr(), float(group["lr"]), float(beta1),
                  float(beta2), float(group["eps"]), float(group["weight_decay"]),
                  torch.cuda.current_stream(params[0].device).cuda_stream)
    optimizer._opt_called = True  # (what torch's LR schedulers look at to warn about the call order)
    return flag


def _fused_adam_step(optimizer, found_inf):
    """optimizer.step() of a fused torch.optim.Adam without its per-parameter Python — state lookup, list building and device
    grouping for ~190 parameters cost the launch thread 0.9 ms per step, and at C4 that thread is level with the GPU.  The same
    three calls torch makes (step counts + 1, torch._fused_adam_ with found_inf, step counts - found_inf) on lists cached on the
    optimizer; the state tensors are those of optimizer.state, so state_dict() / load_state_dict() are unaffected.  Returns
    False when the optimizer is not in the plain configuration this covers (the caller then calls optimizer.step())."""
    if type(optimizer) is not torch.optim.Adam or not hasattr(torch, "_fused_adam_"):
        return False
    cache = getattr(optimizer, "_ogc_fused_lists", None)
    if cache is not None:  # load_state_dict() replaces the group dictionaries and the state tensors: start over then
        groups = optimizer.param_groups
        if len(groups) != len(cache) or any(g is not c[0] or optimizer.state[c[1][0]].get("step") is not c[4][0]
                                            for g, c in zip(groups, cache)):
            cache = None
    if cache is None:
        cache = []
        for group in optimizer.param_groups:
            if (not group.get("fused") or group.get("amsgrad") or group.get("maximize") or group.get("differentiable")
                    or group.get("decoupled_weight_decay") or isinstance(group["lr"], torch.Tensor)):
                return False
            params = list(group["params"])
            if not params:
                return False
            states = [optimizer.state.get(p) for p in params]
            if any(not st or "exp_avg" not in st for st in states):
                return False  # first step: torch builds the state
            if any(p.device != params[0].device or p.dtype != params[0].dtype or p.is_complex() for p in params):
                return False
            cache.append((group, params, [st["exp_avg"] for st in states], [st["exp_avg_sq"] for st in states],
                          [st["step"] for st in states]))
        optimizer._ogc_fused_lists = cache
    work = []
    for group, params, exp_avgs, exp_avg_sqs, steps in cache:
        if len(group["params"]) != len(params):
            optimizer._ogc_fused_lists = None
            return False
        grads = [p.grad for p in params]
        if any(g is None for g in grads):
            return False  # parameters without a gradient this step: torch's own filtering
        work.append((group, params, grads, exp_avgs, exp_avg_sqs, steps))
    for group, params, grads, exp_avgs, exp_avg_sqs, steps in work:
        beta1, beta2 = group["betas"]
        torch._foreach_add_(steps, 1)
        torch._fused_adam_(params, grads, exp_avgs, exp_avg_sqs, [], steps, amsgrad=False, lr=group["lr"], beta1=beta1,
                           beta2=beta2, weight_decay=group["weight_decay"], eps=group["eps"], maximize=False,
                           grad_scale=None, found_inf=found_inf)
        if found_inf is not None:
            torch._foreach_sub_(steps, [found_inf] * len(steps))
    optimizer._opt_called = True  # (what torch's LR schedulers look at to warn about the call order)
    return True


def _step_with_flag(optimizer, bad):
    """Fused optimizer: the kernel itself skips the update (and the step count) when found_inf == 1."""
    found_inf = bad.float().reshape(())
    if _fused_adam_step(optimizer, found_inf):
        return
    optimizer.grad_scale = None
    optimizer.found_inf = found_inf
    try:
        optimizer.step()
    finally:
        del optimizer.grad_scale
        del optimizer.found_inf


def _views(batch):
    """(clouds of all views for the network (b t, n, 3); per-view clouds and flows [(b, n, 3)] x t; the same view-major as ONE
    tensor each (t b, n, 3)).  One transposing copy per tensor: the per-view lists are slices of it (the loss works on the
    views stacked along the batch, which is the view-major tensor itself — no re-concatenation)."""
    pcs, segms, flows, _ = batch
    b, t, n = segms.size()
    flat = pcs.view(b * t, n, -1).contiguous()
    pcs_s, flows_s = pcs.transpose(0, 1).contiguous(), flows.transpose(0, 1).contiguous()       # (t, b, n, 3)
    return flat, list(pcs_s.unbind(0)), list(flows_s.unbind(0)), pcs_s.view(t * b, n, -1), flows_s.view(t * b, n, -1)


class PrefetchedGeometry:
    """Coordinate-only work of a step (FPS / kNN / 3-NN of every encoder level) queued ahead of time for `batch`.  None
    of it depends on the weights, so a trainer that already holds the next batch can let it run on a side stream
    underneath the current step's dense kernels (the sampling kernels keep one workgroup per cloud busy — 16 of 256
    CUs — for milliseconds, each launch waiting on the last).
    loss_ahead: also queue the smooth term's neighbour searches (kNN, ball query, transposed lists) ahead.  Off by
    default: those kernels fill the chip for ~0.55 ms, so underneath the dense kernels they take as much from them as
    they cost when run in line before the loss (measured at C4: 18.5 ms per step ahead, 18.2 ms in line; round 3: 11.54 against
    11.36 ms)."""

    def __init__(self, segnet, criterion, batch, aug_transform, loss_ahead=False):
        from .utils.streams import launch_on_side, side_stream
        net = segnet.module if hasattr(segnet, "module") else segnet
        self.batch, self.aug = batch, aug_transform
        self.flat, self.pcs_l, self.flows_l, self.pcs_s, self.flows_s = _views(batch)
        ready = torch.cuda.Event()
        ready.record()
        self.model = net.plan_geometry_async(self.flat, after=ready) if hasattr(net, "plan_geometry_async") else None
        self.loss = None
        if loss_ahead and hasattr(criterion, "plan_geometry"):
            for p in self.pcs_l:
                p.record_stream(side_stream(p.device, "loss-geometry"))
            self.loss = launch_on_side(side_stream(self.flat.device, "loss-geometry"),
                                       lambda: criterion.plan_geometry(self.pcs_l, aug_transform), after=ready)


class _SplitViews(torch.autograd.Function):
    """(b, t, n, k) -> t contiguous (b, n, k) tensors, one per view.  As `masks[:, tt]` or `masks.unbind(1)` the backward pass is
    a zero-fill of the whole tensor, a copy and an add PER VIEW (twelve launches for four views in the step's trace); here it is
    one stack, and the forward pass one transposing copy instead of one per view."""

    @staticmethod
    def forward(ctx, masks):
        parts = masks.transpose(0, 1).contiguous()
        ctx.shape = parts.shape[1:]
        ctx.set_materialize_grads(False)
        # the views, and the view-major tensor they are slices of (t b, n, k): a loss that works on the stacked views takes
        # that one and needs neither a concatenation forward nor a stack of per-view gradients backward
        return tuple(parts.unbind(0)) + (parts.view((-1,) + tuple(parts.shape[2:])),)

    @staticmethod
    def backward(ctx, *grads):
        views, stacked = grads[:-1], grads[-1]
        total = None
        if any(g is not None for g in views):
            ref = next(g for g in views if g is not None)
            total = torch.stack([g if g is not None else ref.new_zeros(ctx.shape) for g in views], 1)
        if stacked is not None:
            g = stacked.reshape((len(views),) + tuple(ctx.shape)).transpose(0, 1)  # (reshape: the gradient may be strided)
            total = g if total is None else total + g
        return total


def train_step(segnet, criterion, optimizer, batch, it, aug_transform, sync=True, prefetched=None, next_batch=None):
    """batch = (pcs (b,t,n,3), segms (b,t,n), flows (b,t,n,3), valids), already on the device.
    Returns (loss_dict, stepped); with sync=False a PendingStep whose result() gives the same pair later, so the
    host can queue the next step while this one still runs (no host synchronisation inside the step when the
    optimizer is fused — see make_optimizer).
    next_batch: the batch of the following step, if the caller already has it: its geometry is queued on side
    streams during this step and handed back as PendingStep.prefetched, to be passed as `prefetched=` next time."""
    from .utils.streams import HostScalars
    from .utils.zero_arena import zero_arena
    segnet.train()
    optimizer.zero_grad(set_to_none=True)   # (also drops last step's gradients that live in the zero arena, before it is refilled)
    b = batch[1].size(0)
    on_gpu = batch[0].is_cuda
    if on_gpu:
        # forward, loss and backward take their atomically accumulated buffers (statistics, weight gradients, moments, counters)
        # from ONE region zeroed by one launch here (utils/zero_arena.py) instead of ~40 fills of their own
        with zero_arena(batch[0].device):
            return _train_step(segnet, criterion, optimizer, batch, it, aug_transform, sync, prefetched, next_batch, b, on_gpu)
    return _train_step(segnet, criterion, optimizer, batch, it, aug_transform, sync, prefetched, next_batch, b, on_gpu)


def _train_step(segnet, criterion, optimizer, batch, it, aug_transform, sync, prefetched, next_batch, b, on_gpu):
    from .utils.streams import HostScalars
    if prefetched is not None and (prefetched.batch is not batch or prefetched.aug != aug_transform):
        prefetched = None
    if prefetched is None and on_gpu:
        prefetched = PrefetchedGeometry(segnet, criterion, batch, aug_transform)  # queued now, behind nothing
    if prefetched is not None:
        flat, pcs_l, flows_l, pcs_s, flows_s = prefetched.flat, prefetched.pcs_l, prefetched.flows_l, prefetched.pcs_s, prefetched.flows_s
        masks = segnet(flat, flat, geometry=prefetched.model) if prefetched.model is not None else segnet(flat, flat)
        kw = {"geometry": prefetched.loss} if prefetched.loss is not None else {}
    else:
        flat, pcs_l, flows_l, pcs_s, flows_s = _views(batch)
        masks = segnet(flat, flat)
        kw = {}
    t, n = batch[1].size(1), batch[1].size(2)
    masks = masks.view(b, t, n, -1)
    *masks_l, masks_s = _SplitViews.apply(masks)
    if getattr(criterion, "takes_stacked_views", False):
        kw["stacked"] = (pcs_s, masks_s, flows_s)
    loss, losses = criterion(pcs_l, masks_l, flows_l, step_w=True, it=it * b, aug_transform=aug_transform, sync=False,
                             **kw)
    # The next batch's coordinate-only work goes to its side stream HERE, between the loss and the backward pass: queued before
    # the loss it ran underneath the loss's own neighbour searches (latency-bound launches that want the whole chip: the ball
    # query took 37 us there against 35.5 here), underneath the backward pass's 7 ms of dense kernels it costs nobody anything
    # (step time unchanged: 11.30 against 11.35 ms in an A/B of five rounds).
    upcoming = None
    if next_batch is not None and on_gpu:
        upcoming = PrefetchedGeometry(segnet, criterion, next_batch, aug_transform)
    try:
        loss.backward()
    except RuntimeError as err:  # the reference returns without stepping (train_seg.py:75-78)
        if _is_distributed(segnet) or _must_surface(err):
            # a rank that leaves the step alone would hang the others in the gradient collective; and a failing operator of
            # THIS library, an out-of-memory condition or a runtime error of the device is a fault to be seen, not a numerical
            # accident of the batch
            raise
        pending = PendingStep(losses, HostScalars(torch.tensor([True])))
        pending.prefetched = upcoming
        return pending.result() if sync else pending
    _average_gradients(segnet)
    # under DDP the all-reduced gradients make the NaN decision identical on all ranks
    flag = _adam_kernel_step(optimizer) if on_gpu else None
    if flag is not None:
        pending = PendingStep(losses, HostScalars(flag))
        pending.prefetched = upcoming
        return pending.result() if sync else pending
    grads = [p.grad for p in segnet.parameters() if p.grad is not None]
    bad = torch.isnan(torch.stack(torch._foreach_norm(grads)).sum())  # NaN anywhere -> NaN norm
    if getattr(optimizer, "_step_supports_amp_scaling", False) and bad.is_cuda:
        _step_with_flag(optimizer, bad)
        pending = PendingStep(losses, HostScalars(bad.reshape(1)))
    else:
        skip = bool(bad)  # host sync
        if not skip:
            optimizer.step()
        pending = PendingStep(losses, HostScalars(torch.tensor([skip])))
    pending.prefetched = upcoming
    return pending.result() if sync else pending


def _must_surface(err):
    """Failures of backward() that are NOT the numerical accidents the reference's `except RuntimeError` is there for: errors of
    this library's operators, out-of-memory, HIP runtime errors.  Skipping the step on those would leave a run that never
    updates its weights and never says why (the step is asynchronous: nobody reads the flag in time)."""
    from ._lib import OgcOpsError
    if isinstance(err, (OgcOpsError, torch.cuda.OutOfMemoryError)):
        return True
    text = str(err)
    if any(tag in text for tag in ("HIP error", "hipError", "CUDA error", "out of memory", "rocBLAS", "MIOpen")):
        return True
    # What the reference's handler was written for is a decomposition that fails on a degenerate batch (torch.svd / eigh in the
    # backward pass of the rigid fit).  Those arrive as torch.linalg.LinAlgError (a RuntimeError subclass) or as a RuntimeError
    # naming the solver routine ("... gesvd ... failed to converge", "hipsolver error ... syevd", "... is singular").  Anything
    # else — a shape or stride error inside one of this package's autograd Functions, a type error — is a bug, and a silently
    # skipped step would hide it until MAX_CONSECUTIVE_SKIPS stops the run.
    if not (isinstance(err, _LINALG_ERRORS) or _SOLVER_WORDS.search(text)):
        return True
    global _first_skip_reported
    if not _first_skip_reported:
        _first_skip_reported = True
        import warnings
        warnings.warn("training step skipped after a RuntimeError in backward() (further ones are not reported): %s" % text[:500])
    return False


_first_skip_reported = False
_LINALG_ERRORS = tuple(t for t in (getattr(torch.linalg, "LinAlgError", None), getattr(torch._C, "_LinAlgError", None))
                       if isinstance(t, type))
# whole words only ("inf" / "nan" as bare substrings match "info", "inference", "nanoseconds": dropped)
# autograd node names of anomaly mode ("Function 'SvdBackward0' returned nan values", 'LinalgEighBackward0', 'LinalgSvdBackward0')
# carry the solver's name glued to "Backward<n>": matched by the second alternative
_SOLVER_WORDS = __import__("re").compile(
    r"\b(gesvdj?|gesdd|syevd?j?|heevd?|geqrf|getrf|potrf|hipsolver|cusolver|rocsolver|lapack|magma|svd|eigh?|linalg|"
    r"converge[ds]?|convergence|singular|ill-conditioned)\b"
    r"|\b(linalg)?_?(svd|eigh?|eigvalsh?|qr|cholesky|inv(erse)?(_ex)?|solve(_ex)?|det|slogdet|lstsq|pinv)\w*backward\d*\b"
    r"|returned nan values", __import__("re").IGNORECASE)


def _is_distributed(model):
    return hasattr(model, "average_gradients") or isinstance(model, torch.nn.parallel.DistributedDataParallel)


def _average_gradients(model):
    """Data-parallel gradient mean when the model is wrapped by utils/dist_util.FlatDataParallel (one collective per
    step); DistributedDataParallel has already averaged inside backward(), a bare module has nothing to do."""
    average = getattr(model, "average_gradients", None)
    if average is not None:
        average()


def _nan_safe_step(params, optimizer):
    """The reference's NaN-gradient rule (train_seg.py:81-83, train_flow.py:84-86) without a host round trip when the
    optimizer is fused; returns a HostScalars holding the 'skipped' flag."""
    from .utils.streams import HostScalars
    flag = _adam_kernel_step(optimizer)
    if flag is not None:
        return HostScalars(flag)
    grads = [p.grad for p in params if p.grad is not None]
    bad = torch.isnan(torch.stack(torch._foreach_norm(grads)).sum())  # NaN anywhere -> NaN norm
    if getattr(optimizer, "_step_supports_amp_scaling", False) and bad.is_cuda:
        _step_with_flag(optimizer, bad)
        return HostScalars(bad.reshape(1))
    skip = bool(bad)  # host sync
    if not skip:
        optimizer.step()
    return HostScalars(torch.tensor([skip]))


class PrefetchedFlowGeometry:
    """The local encoder's sampling chains of `batch`, queued ahead of time on a side stream (FlowStep3D.plan_geometry_async)."""

    def __init__(self, flownet, batch):
        net = flownet.module if hasattr(flownet, "module") else flownet
        pcs = batch[0]
        self.batch = batch
        self.pc1, self.pc2 = pcs[:, 0].contiguous(), pcs[:, 1].contiguous()
        self.plan = None
        if pcs.is_cuda and hasattr(net, "plan_geometry_async"):
            ready = torch.cuda.Event()
            ready.record()
            self.plan = net.plan_geometry_async(self.pc1, self.pc2, after=ready)


def flow_train_step(flownet, criterion, optimizer, batch, model_iters, sync=True, prefetched=None, next_batch=None):
    """One unsupervised FlowStep3D step (body of the reference's train_flow.py:62-88).
    batch = (pcs (b,t,n,3), segms, flows, valids) on the device; the pair is (pcs[:, 0], pcs[:, 1]).
    Returns (loss_dict, stepped), or a PendingStep with sync=False.
    next_batch: the following step's batch, if the caller holds it: its sampling chains are queued on a side stream between this
    step's loss and backward pass and handed back as PendingStep.prefetched, to be passed as `prefetched=` next time."""
    flownet.train()
    optimizer.zero_grad(set_to_none=True)
    pcs = batch[0]
    if pcs.is_cuda:
        # the step's ~330 small accumulators (BatchNorm sums, weight gradients, counters) out of ONE region zeroed by one launch
        # (utils/zero_arena.py), as in the segmentation step: 1.4 ms of fills per step at 4 x 8192-point pairs
        from .utils.zero_arena import zero_arena
        with zero_arena(pcs.device):
            return _flow_train_step(flownet, criterion, optimizer, batch, model_iters, sync, prefetched, next_batch)
    return _flow_train_step(flownet, criterion, optimizer, batch, model_iters, sync, prefetched, next_batch)


def _flow_train_step(flownet, criterion, optimizer, batch, model_iters, sync, prefetched=None, next_batch=None):
    pcs = batch[0]
    if prefetched is not None and prefetched.batch is batch and prefetched.plan is not None:
        pc1, pc2 = prefetched.pc1, prefetched.pc2
        flow_preds = flownet(pc1, pc2, pc1, pc2, iters=model_iters, geometry=prefetched.plan)
    else:
        pc1, pc2 = pcs[:, 0].contiguous(), pcs[:, 1].contiguous()
        flow_preds = flownet(pc1, pc2, pc1, pc2, iters=model_iters)
    extra = None
    if batch[2] is not None:  # ground-truth flow of the first frame: EPE per iteration, monitored (train_flow.py:75-76)
        from .metrics.flow_metric import epe_terms
        extra = epe_terms(batch[2][:, 0], flow_preds)
    loss, losses = criterion(pc1, pc2, flow_preds, sync=False, extra=extra)
    upcoming = None
    if next_batch is not None and pcs.is_cuda:   # (underneath the backward pass's dense kernels: one workgroup per cloud)
        upcoming = PrefetchedFlowGeometry(flownet, next_batch)
    try:
        loss.backward()
    except RuntimeError as err:  # train_flow.py:80-83: the step is skipped
        if _is_distributed(flownet) or _must_surface(err):
            raise
        from .utils.streams import HostScalars
        pending = PendingStep(losses, HostScalars(torch.tensor([True])))
        pending.prefetched = upcoming
        return pending.result() if sync else pending
    _average_gradients(flownet)
    net = flownet.module if hasattr(flownet, "module") else flownet
    pending = PendingStep(losses, _nan_safe_step(list(net.parameters()), optimizer))
    pending.prefetched = upcoming
    return pending.result() if sync else pending
