"""Multi-GPU data parallelism for render_step: one process per GPU, rays / frames sharded
across ranks with replicated parameters, and ONE exchange step per training iteration -- an all-reduce(sum) of
the gradients over RCCL/xGMI (`torch.distributed` backend "nccl" on ROCm; "gloo" in the CPU tests).

Gradient payload = two 50.4 MB hash tables + <0.5 MB of MLP / beta parameters.  xGMI is point-to-point
(7 links x ~153 GB/s per GPU), so the two big tables go as two large asynchronous all-reduces (RCCL splits each
over all links) and the ~20 small tensors are flattened into a single bucket instead of 20 latency-bound calls.
"""
from typing import Iterable, List, Tuple

import contextlib

import torch
import torch.distributed as dist

BIG = 1 << 20


def _parse_cpulist(text: str):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        cpus.update(range(int(a), int(b or a) + 1))
    return cpus


def pin_to_gpu_numa_node(device_index: int = 0):
    """one process per GPU, pinned to the CPUs of the GPU's NUMA node: the step issues ~330 launches and 17 size read-backs,
    so doorbell / read-back latency across the socket interconnect shows up directly in the step time (a rank scheduled
    on the far socket of a 2-socket host runs host-bound).  Linux only; returns the node id, or None if the topology is
    not exposed (no change then).  Never widens an affinity mask that the launcher already narrowed."""
    import os
    try:
        pr = torch.cuda.get_device_properties(device_index)
        bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        node = int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read())
        if node < 0:
            return None
        cpus = _parse_cpulist(open(f"/sys/devices/system/node/node{node}/cpulist").read())
        allowed = os.sched_getaffinity(0) & cpus
        if not allowed:
            return None
        os.sched_setaffinity(0, allowed)
        return node
    except (OSError, ValueError, AttributeError, RuntimeError):
        return None


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """contiguous, balanced [start, end) of `n_items` rays/frames for `rank` (first n % world ranks get one more)."""
    q, r = divmod(n_items, world)
    start = rank * q + min(rank, r)
    return start, start + q + (1 if rank < r else 0)


def allreduce_gradients(params: Iterable[torch.nn.Parameter], average: bool = False) -> int:
    """in-place all-reduce(sum) of .grad over the default process group; returns bytes reduced.
    Parameters without a gradient on this rank (e.g. unused branch) contribute zeros, like DDP's
    find_unused_parameters=True (launch.py:89-98)."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return 0
    params = [p for p in params if p.requires_grad]
    for p in params:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    big = [p.grad for p in params if p.grad.numel() >= BIG]
    small = [p.grad for p in params if p.grad.numel() < BIG]
    handles = [dist.all_reduce(t, async_op=True) for t in big]
    nbytes = sum(t.numel() * t.element_size() for t in big)
    if small:
        flat = torch.cat([t.reshape(-1) for t in small])
        dist.all_reduce(flat)
        off = 0
        for t in small:
            t.copy_(flat[off:off + t.numel()].view_as(t))
            off += t.numel()
        nbytes += flat.numel() * flat.element_size()
    for h in handles:
        h.wait()
    if average:
        w = dist.get_world_size()
        for p in params:
            p.grad.div_(w)
    return nbytes


def allreduce_scalars(values: List[float], device) -> List[float]:
    """global sums of a few python scalars (ray / sample counts for loss normalisation, metrics)."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return list(values)
    t = torch.tensor(values, dtype=torch.float64, device=device)
    dist.all_reduce(t)
    return t.tolist()


class OverlappedGradientAllReduce:
    """all-reduce of the big (hash-table) gradients started from autograd's post-accumulate hooks, i.e. while the rest
    of the backward pass is still running: the radiance table's gradient is complete half-way through backward, its
    50 MB all-reduce over xGMI then hides behind the SDF-head backward.  `finish()` (after loss.backward()) launches
    whatever is left, waits, and reduces the small tensors as one bucket.  Same result as allreduce_gradients().

    Collectives must be issued in the SAME order on every rank, also when a rank produced no gradient for some table
    (no ray hit anything): the launch order is therefore fixed up front -- reverse parameter order, i.e. the order in
    which autograd normally completes them -- and a ready gradient waits for its predecessors in that order."""

    def __init__(self, params: Iterable[torch.nn.Parameter], single_rank_too: bool = False):
        """single_rank_too: issue the collectives in a one-rank group as well (a sum over one rank: same values) -- how the
        RCCL code path (hooks, async handles on device tensors, the flat bucket) is exercised on a one-GPU box."""
        self.params = [p for p in params if p.requires_grad]
        self.active = dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or single_rank_too)
        self.order = [p for p in reversed(self.params) if p.numel() >= BIG]
        self._ready = set()
        self._next = 0
        self._handles = []
        self._hooks = []
        self._hold = False
        if self.active:
            for p in self.order:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_ready))

    def _pump(self):
        while self._next < len(self.order) and id(self.order[self._next]) in self._ready:
            self._handles.append(dist.all_reduce(self.order[self._next].grad, async_op=True))
            self._next += 1

    @contextlib.contextmanager
    def no_sync(self):
        """gradient accumulation over ray chunks (DDP.no_sync analogue): backward passes inside this context only
        accumulate into .grad; the hooks launch the all-reduces from the first backward pass OUTSIDE of it (the last chunk)."""
        self._hold = True
        try:
            yield
        finally:
            self._hold = False

    def _on_ready(self, p):
        if self._hold:
            return
        if id(p) in self._ready:
            raise RuntimeError("OverlappedGradientAllReduce: a table received a second gradient in the same backward pass "
                               "(e.g. the curvature term evaluates the SDF grid twice); its all-reduce is already in "
                               "flight -- use allreduce_gradients() after backward for such steps")
        self._ready.add(id(p))
        self._pump()

    def finish(self) -> int:
        if not self.active:
            return 0
        nbytes = 0
        small = []
        for p in self.params:
            if p.grad is None:                      # no gradient arrived on this rank: contribute zeros
                p.grad = torch.zeros_like(p)
            if p.numel() >= BIG:
                self._ready.add(id(p))
                nbytes += p.grad.numel() * p.grad.element_size()
            else:
                small.append(p.grad)
        self._pump()
        if small:
            flat = torch.cat([t.reshape(-1) for t in small])
            dist.all_reduce(flat)
            off = 0
            for t in small:
                t.copy_(flat[off:off + t.numel()].view_as(t))
                off += t.numel()
            nbytes += flat.numel() * flat.element_size()
        for h in self._handles:
            h.wait()
        self._handles, self._next = [], 0
        self._ready.clear()
        return nbytes

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
