"""Multi-GPU data parallelism for render_step: one process per GPU, rays / frames sharded
across ranks with replicated parameters, and ONE exchange step per training iteration -- an all-reduce(sum) of
the gradients over RCCL/xGMI (`torch.distributed` backend "nccl" on ROCm; "gloo" in the CPU tests).

Gradient payload = two 50.4 MB hash tables + <0.5 MB of MLP / beta parameters.  xGMI is point-to-point
(7 links x ~153 GB/s per GPU), so the two big tables go as two large asynchronous all-reduces (RCCL splits each
over all links) and the ~20 small tensors are flattened into a single bucket instead of 20 latency-bound calls.
"""
from typing import Iterable, List, Tuple

import torch
import torch.distributed as dist

BIG = 1 << 20


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
