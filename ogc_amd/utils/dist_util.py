"""Data parallelism for the training steps: one process per GPU, ONE gradient collective per step.

The reference trains on a single GPU (train_seg.py / train_flow.py have no distributed path); SURVEY.md §8e scales
its step the only way it shards: every rank steps on its own batch and the gradients are averaged.  The segmentation
net has 2.4 MB of parameters in ~190 tensors, so the all-reduce itself is tens of microseconds over xGMI — what costs
is per-tensor bookkeeping.  `torch.nn.parallel.DistributedDataParallel` runs an autograd hook per parameter and
several Python layers per forward: measured on the C4 step (tools/ddp_cost.py, one process, RCCL) it adds 2.6 ms of
launch-thread time to a 14.9 ms step whose launch thread is only ~3 ms ahead of the GPU, and the step becomes
16.1 ms.  `FlatDataParallel` does the same arithmetic with three launches after the backward pass:

    flat = cat(all gradients)  ->  all_reduce(flat, SUM) / world  ->  multi-tensor copy back into the .grad tensors

(no overlap with the backward pass is attempted: there is nothing worth hiding), and broadcasts rank 0's parameters
and buffers once at construction, as DistributedDataParallel does.  Results equal DistributedDataParallel's: the mean
of the ranks' gradients in every .grad (tests/test_ddp_cpu.py, tests/test_ddp_gpu.py)."""
import torch
import torch.distributed as dist
import torch.nn as nn


class FlatDataParallel(nn.Module):
    """Wraps `module` (kept as ``.module``, like DistributedDataParallel, so checkpoints and ``hasattr(m, 'module')``
    unwrapping work unchanged).  The training steps call ``average_gradients()`` between backward() and the
    optimizer; gradients that are None on this rank take part as zeros (every rank must own the same parameters)."""

    def __init__(self, module, process_group=None, broadcast=True):
        super().__init__()
        self.module = module
        self.process_group = process_group
        self.world_size = dist.get_world_size(process_group)
        self._params = [p for p in module.parameters() if p.requires_grad]
        self._events = None   # [(start, end)] around every collective when time_collectives() was called
        from . import subgraph
        subgraph.allow_under_data_parallel(module)  # this wrapper puts no hooks on the parameters
        if broadcast and self.world_size > 1:
            self._broadcast([p.data for p in module.parameters()])
            bufs = [b.data for b in module.buffers() if b.is_floating_point()]
            if bufs:
                self._broadcast(bufs)

    def _broadcast(self, tensors):
        by_dtype = {}
        for t in tensors:
            by_dtype.setdefault(t.dtype, []).append(t)
        for group in by_dtype.values():
            flat = torch.cat([t.reshape(-1) for t in group])
            dist.broadcast(flat, src=dist.get_global_rank(self.process_group, 0) if self.process_group else 0,
                           group=self.process_group)
            torch._foreach_copy_(group, [v.view_as(t) for v, t in zip(flat.split([t.numel() for t in group]), group)])

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    def average_gradients(self):
        """.grad <- mean over the ranks of .grad, for every parameter; one collective."""
        if self.world_size == 1 and not _always_reduce:
            return
        # fixed parameter order on every rank, zeros in place of a missing gradient: the flat layouts agree even when
        # a data-dependent branch left different parameters unused on different ranks
        parts = [(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in self._params]
        flat = torch.cat(parts)
        timed = self._events is not None and flat.is_cuda
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.process_group)
        if timed:
            e1.record()
            self._events.append((e0, e1))
        if self.world_size > 1:
            flat.mul_(1.0 / self.world_size)
        views = [v.view_as(p) for v, p in zip(flat.split([p.numel() for p in self._params]), self._params)]
        owners = [(p, v) for p, v in zip(self._params, views) if p.grad is not None]
        for p, v in zip(self._params, views):
            if p.grad is None:
                p.grad = v.clone()
        if owners:
            torch._foreach_copy_([p.grad for p, _ in owners], [v for _, v in owners])


    def payload_bytes(self):
        return sum(p.numel() * p.element_size() for p in self._params)

    def time_collectives(self, on=True):
        """Bracket every gradient all-reduce with events on the launch stream (benchmark reporting)."""
        self._events = [] if on else None

    def collective_ms(self):
        """Mean duration of the timed all-reduces (call after a device synchronisation); None if none were timed."""
        if not self._events:
            return None
        return sum(a.elapsed_time(b) for a, b in self._events) / len(self._events)


_always_reduce = False


def always_reduce(flag=True):
    """Run the collective even in a one-process group (bench.py under a one-process launcher exercises RCCL)."""
    global _always_reduce
    _always_reduce = bool(flag)


def data_parallel(module, process_group=None):
    """`module` wrapped for gradient averaging when a process group is initialised, the module itself otherwise."""
    if dist.is_available() and dist.is_initialized():
        return FlatDataParallel(module, process_group)
    return module


def core_block(allowed, local_rank, local_world):
    """The CPUs rank `local_rank` of `local_world` ranks on this host keeps: a contiguous block of the sorted allowed set
    (at least one CPU; every rank a different block while there are enough of them)."""
    cpus = sorted(allowed)
    per = max(1, len(cpus) // max(1, local_world))
    start = (local_rank * per) % len(cpus)
    return set(cpus[start:start + per])


def pin_rank_to_cores():
    """One process per GPU, each with ONE hot thread: the step's ~600 launches take the launch thread 10-12 ms, about what
    the GPU needs for the kernels of a C4 step, so a launch thread that is migrated between sockets or shares a core with
    another rank's becomes the bottleneck of that rank (and, through the all-reduce, of all of them).  With more than one
    rank on the host every rank keeps to its own block of the allowed CPUs (blocks in rank order: on the usual two-socket
    hosts that is also the socket its GPU hangs on).  OGC_PIN_CORES=0 leaves the affinity alone.  Returns the block or None."""
    import os
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
    if local_world <= 1 or os.environ.get("OGC_PIN_CORES", "1") == "0" or not hasattr(os, "sched_setaffinity"):
        return None
    allowed = os.sched_getaffinity(0)
    if len(allowed) < 4 * local_world:
        return None   # a rank keeps three threads busy (launch thread, autograd's backward thread, a runtime helper): blocks of
                      # fewer than four CPUs would make them take turns — better left to the scheduler
    block = core_block(allowed, int(os.environ.get("LOCAL_RANK", "0")), local_world)
    try:
        os.sched_setaffinity(0, block)
    except OSError:
        return None
    return block
