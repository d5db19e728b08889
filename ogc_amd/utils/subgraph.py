"""Dense sub-graphs of a training step as forward / backward pairs of HIP graphs.

A C4 step is ~600 launches of which the GPU needs 11 ms and Python ~9: the launch thread is within reach of being
the bound, and on a loaded host it is.  The parts of the network that are fixed chains of kernels on tensors of fixed shapes —
the slot branch, the feature-propagation modules, the set-abstraction MLPs — can be recorded once with
torch.cuda.make_graphed_callables and replayed with one launch each way: the same kernels in the same order on the same
stream, Python out of the loop.  (Used for the slot branch.  The feature-propagation modules were tried as well — interpolation,
concatenation and MLP with the neighbour lists as arguments — and dropped: the C4 step went from 11.2 to 12.2 ms.  The attempt
is what exposed the memset nodes of this stack: DESIGN.md §4b, csrc/ogc_common.h ogc_zero_async.)  What varies from step to step (features, coordinates, neighbour lists) enters as tensor
arguments, copied into the graph's static inputs; everything the body touches besides its arguments and the owner's
parameters must be constant.

run(owner, name, body, tensors, parts) evaluates body(*tensors) that way when it may and eagerly otherwise:
  * training mode with gradients on, on the GPU, not inside an enclosing capture (graph_step.py captures the whole step);
  * no process group, or the wrapper is utils/dist_util.FlatDataParallel (which marks the modules: DistributedDataParallel's
    gradient hooks and a captured backward pass crashed a rank in tests/test_ddp_gpu.py);
  * the previous graphed call of this body has had its backward pass (or its output is gone): the graphs work on STATIC
    buffers, so a second forward pass before the first one's backward pass would overwrite what that pass needs.

Lifetime of gradients: as with torch.cuda.make_graphed_callables itself, the parameter gradients a graphed backward pass hands
to autograd are its static buffers — ``param.grad`` of the modules in `parts` is valid until the NEXT replay of that body and
is overwritten by it.  train_step consumes gradients inside the step that produced them (optimizer.zero_grad(set_to_none=True)
comes first), so nothing there depends on more; code that holds or accumulates these gradients across steps sets
``subgraph.ENABLED = False`` (or OGC_SUBGRAPHS=0 in the environment) and gets ordinary tensors (tests/test_zero_arena_gpu.py).
"""
import gc
import warnings
import weakref

import torch
import torch.nn as nn

import os as _os

ENABLED = _os.environ.get("OGC_SUBGRAPHS", "1") != "0"


def allowed(owner, ref):
    if not (ENABLED and owner.training and torch.is_grad_enabled() and ref.is_cuda
            and not torch.cuda.is_current_stream_capturing() and not owner.__dict__.get("_no_subgraphs", False)):
        return False
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        return owner.__dict__.get("_graphs_allowed_under_dp", False)
    return True


class _Body(nn.Module):
    """body(*tensors) as a module whose parameters are the owner's (make_graphed_callables takes a module's parameters as graph
    inputs that require a gradient).  Never registered anywhere: the owner's state_dict is untouched."""

    def __init__(self, parts, body):
        super().__init__()
        self.parts = nn.ModuleList(parts)
        self._body = body

    def forward(self, *tensors):
        return self._body(*tensors)


def _key(name, tensors, parts):
    """(The graphs replay against the parameter STORAGES they were captured with: net.to(), .float() or a re-wrap that moves a
    parameter must not find the old graph.)"""
    from ..pointnet2 import pointnet2 as _api
    prec = getattr(_api._native, "get_matmul_precision", lambda: "fp32")()
    where = tuple(p.data_ptr() for m in parts for p in m.parameters())
    return (name, prec, tuple((tuple(t.shape), t.dtype, bool(t.requires_grad)) for t in tensors), where)


def _make(owner, name, body, tensors, parts):
    module = _Body(parts, body)
    if not all(p.requires_grad for p in module.parameters()):
        return None
    module.train()
    sample = tuple(t.detach().clone().requires_grad_(t.requires_grad) for t in tensors)
    # (the capture's warm-up runs on a stream of its own, which is where autograd then expects these parameters' gradients
    # to be accumulated; it synchronises the streams itself and says so — silenced for the capture only)
    quiet = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
    # No collector pass inside the capture: a cycle collected there may free a tensor or an older graph of another network,
    # and a runtime call of that kind on a capturing stream aborts the process (seen in tests/test_fallbacks_gpu.py, which
    # builds one network after another).
    collecting = gc.isenabled()
    try:
        if quiet is not None:
            quiet(False)
        gc.collect()
        gc.disable()
        return torch.cuda.make_graphed_callables(module, sample)
    except Exception as err:  # a capture that does not work on this stack must not take training down with it
        warnings.warn("'%s' of %s not captured as a HIP graph (%s): running eagerly" % (name, type(owner).__name__, str(err)[:300]))
        return None
    finally:
        if collecting:
            gc.enable()
        if quiet is not None:
            quiet(True)


def run(owner, name, body, tensors, parts=None):
    """body(*tensors) -> tensor, through the graphs where allowed() says so (see the module docstring).  parts: the modules whose
    parameters the body uses — ALL of them and no others (default: the owner itself)."""
    tensors = tuple(tensors)
    if not allowed(owner, tensors[0]):
        return body(*tensors)
    store = owner.__dict__.setdefault("_subgraphs", {})
    parts = [owner] if parts is None else list(parts)
    key = _key(name, tensors, parts)
    if key not in store:
        for stale in [k for k in store if k[:3] == key[:3]]:  # the same body on parameters that have moved since
            del store[stale]
        store[key] = [_make(owner, name, body, tensors, parts), None]
    entry = store[key]
    graphed, last = entry
    if graphed is None or (last is not None and last[0]() is not None and not last[1][0]):
        return body(*tensors)
    out = graphed(*tensors)
    done = [False]
    if out.requires_grad:
        out.register_hook(lambda g, d=done: d.__setitem__(0, True))
    entry[1] = (weakref.ref(out), done)
    return out


def forbid(root):
    """No sub-graphs under `root` any more (the whole step is about to be captured as one graph)."""
    for m in root.modules():
        m.__dict__["_no_subgraphs"] = True


def allow_under_data_parallel(root):
    """The wrapper around `root` puts no hooks on the parameters: sub-graphs stay allowed under its process group."""
    for m in root.modules():
        m.__dict__["_graphs_allowed_under_dp"] = True
