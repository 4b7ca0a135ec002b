"""Optimiser step of the training iteration behind the interface the reference's trainer uses.

The reference builds `torch.optim.Adam(param_groups, lr=1e-3, betas=(0.9, 0.99), eps=1e-15)` with per-group `lr` /
`weight_decay` (configs/config.yaml:110-136, systems/utils.py:314-325) and drives it with `torch.optim.lr_scheduler`
objects (systems/utils.py:328-346).  `Adam` below is a `torch.optim.Optimizer` subclass with the same constructor
arguments, `param_groups`, `state` layout (`step`, `exp_avg`, `exp_avg_sq` -> `state_dict()` is interchangeable with
torch.optim.Adam's for the same parameter groups) and scheduler compatibility; `step()` updates every
parameter tensor with ONE `ia_adam_step` launch instead of ~10 elementwise kernels per tensor.

No CPU fallback: parameters must live on the GPU (`_lib.ptr` raises otherwise).
"""
import ctypes as C
import math
from typing import Optional

import torch

from . import _lib as L


class Adam(torch.optim.Optimizer):
    """torch.optim.Adam semantics (no amsgrad / maximize / capturable), fused over all tensors.

    grad_scale: multiplied into every gradient inside the kernel -- 1/world_size turns the summed all-reduce of
    parallel.allreduce_gradients() into DDP's mean without another pass over the 100 MB of table gradients."""

    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                 grad_scale: float = 1.0):
        if lr < 0.0 or eps < 0.0 or weight_decay < 0.0:
            raise ValueError("lr, eps and weight_decay must be non-negative")
        if not (0.0 <= betas[0] < 1.0 and 0.0 <= betas[1] < 1.0):
            raise ValueError(f"invalid betas {betas}")
        # the extra keys are torch.optim.Adam's own defaults, so that a state_dict written here loads into torch's Adam
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, amsgrad=False,
                                      maximize=False, foreach=None, capturable=False, differentiable=False, fused=None,
                                      decoupled_weight_decay=False))
        self.grad_scale = float(grad_scale)
        # parameter -> (the state's `step` tensor, the 0-d numpy array it is a view of): the count is bumped by writing the array --
        # 27 host-tensor increments + .item() per step otherwise, and the 4096-ray step is bound by the host's launch rate
        self._t = {}

    def _init_state(self, p):
        st = self.state[p]
        if len(st) == 0:
            st["step"] = torch.tensor(0.0, dtype=torch.float32)          # host scalar tensor, like torch (capturable=False)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        # launches are grouped by (betas, eps, step): bias_correction2 is a kernel-wide scalar
        batches = {}
        for group in self.param_groups:
            b1, b2 = group["betas"]
            if group.get("amsgrad") or group.get("maximize") or group.get("decoupled_weight_decay"):
                raise NotImplementedError("amsgrad / maximize / decoupled weight decay are not used by the reference")
            lr, wd, eps = float(group["lr"]), float(group["weight_decay"]), float(group["eps"])
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.grad.is_sparse:
                    raise RuntimeError("Adam does not support sparse gradients")
                if p.dtype != torch.float32 or p.grad.dtype != torch.float32:
                    raise RuntimeError("intrinsicavatar_amd.optim.Adam: fp32 parameters only")
                st = self._init_state(p)
                ent = self._t.get(p)
                if ent is None or ent[0] is not st["step"]:          # first step, or the state was replaced (load_state_dict)
                    import numpy as np
                    arr = np.array(float(st["step"]), dtype=np.float32)
                    ent = self._t[p] = (torch.from_numpy(arr), arr)
                    st["step"] = ent[0]
                t = int(ent[1]) + 1
                ent[1][...] = t
                bc1 = 1.0 - b1 ** t
                bc2_sqrt = math.sqrt(1.0 - b2 ** t)
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                if not p.is_contiguous():
                    raise RuntimeError("intrinsicavatar_amd.optim.Adam: parameters must be contiguous")
                batches.setdefault((b1, b2, eps, bc2_sqrt), []).append((p, g, st, lr / bc1, wd))
        for (b1, b2, eps, bc2_sqrt), items in batches.items():
            n = len(items)
            P, G, M, V = ((C.c_void_p * n)() for _ in range(4))
            N = (C.c_int64 * n)()
            S, W = (C.c_float * n)(), (C.c_float * n)()
            for i, (p, g, st, step_size, wd) in enumerate(items):
                P[i], G[i] = L.ptr(p).value, L.ptr(g).value
                M[i], V[i] = L.ptr(st["exp_avg"]).value, L.ptr(st["exp_avg_sq"]).value
                N[i], S[i], W[i] = p.numel(), step_size, wd
            L.check(L.lib().ia_adam_step(L.i32(n), P, G, M, V, N, S, W, L.f32(b1), L.f32(b2), L.f32(eps), L.f32(bc2_sqrt),
                                         L.f32(self.grad_scale), L.stream()), "ia_adam_step")
        if batches:
            from . import fields
            fields.params_changed()          # the kernel wrote the parameters through raw pointers: Tensor._version did not move
        return loss


def reference_param_groups(rs, lr: float = 1e-3, color_grid_wd: float = 0.0, material=None, emitter=None):
    """the parameter groups of configs/config.yaml:116-136 for a RenderStep (names as in the reference config):
    geometry (hash grid + SDF head), radiance.network, radiance.xyz_encoding (with `color_grid_wd` L2 decay), density
    (lr 1e-3), and, for the PBR branch, material / emitter."""
    geo, rad = rs.geometry, rs.radiance
    rad_grid = [rad.grid_params]
    rad_net = [p for p in rad.parameters() if p is not rad.grid_params and p.requires_grad]
    groups = [dict(params=list(geo.parameters()), name="geometry", lr=lr),
              dict(params=rad_net, name="radiance.network", lr=lr),
              dict(params=rad_grid, name="radiance.xyz_encoding", lr=lr, weight_decay=color_grid_wd),
              dict(params=list(rs.density.parameters()), name="density", lr=1e-3)]
    if material is not None:
        groups.append(dict(params=list(material.parameters()), name="material", lr=lr))
    if emitter is not None:
        groups.append(dict(params=list(emitter.parameters()), name="emitter", lr=lr))
    return [g for g in groups if g["params"]]


def reference_optimizer(rs, lr: float = 1e-3, color_grid_wd: float = 0.0, material=None, emitter=None,
                        grad_scale: float = 1.0, warmup_steps: Optional[int] = 1000,
                        milestones=(12500, 18750, 22500, 23750), gamma: float = 0.3):
    """Adam + the scheduler of configs/config.yaml:137-155: SequentialLR(LinearLR 0.01 -> 1 over `warmup_steps`,
    then MultiStepLR(milestones, gamma) -- milestones count from the END of the warm-up, as SequentialLR restarts the
    second scheduler's step counter).  returns (optimizer, scheduler or None).

    The parameter groups are the reference's for the modules a RenderStep owns (geometry, radiance.network,
    radiance.xyz_encoding, density, material, emitter); the reference's pose_correction / pose_encoder / deformer groups
    belong to modules outside the render_step path, so an optimiser checkpoint of the full reference model does NOT load
    here -- only same-group torch.optim.Adam state does."""
    opt = Adam(reference_param_groups(rs, lr, color_grid_wd, material, emitter), lr=lr, betas=(0.9, 0.99), eps=1e-15,
               grad_scale=grad_scale)
    sched = None
    if warmup_steps:
        S = torch.optim.lr_scheduler
        warm = S.LinearLR(opt, start_factor=0.01, end_factor=1.0, total_iters=warmup_steps)
        if milestones:
            decay = S.MultiStepLR(opt, milestones=list(milestones), gamma=gamma)
            sched = S.SequentialLR(opt, schedulers=[warm, decay], milestones=[warmup_steps])
        else:
            sched = warm
    elif milestones:
        sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=list(milestones), gamma=gamma)
    return opt, sched
