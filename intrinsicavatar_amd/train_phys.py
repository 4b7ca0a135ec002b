"""Training step WITH the physically based branch (BASELINE config 4): differentiable form of forward_ with enable_phys
(models/intrinsic_avatar.py:1066-1156 rgb_normal_mats_alpha_fn, :1290-1470 volume scattering) on the MI355X kernels.

On top of train.py (SDF field with first+second order terms, alpha, compositing) this adds, each a torch.autograd.Function
whose forward AND backward are HIP kernels behind the C ABI:

    _HashEncode   hash grid #2 once, shared by the radiance and the material heads        ia_hashgrid_fwd / _bwd_binned
    _MLP2         radiance 67->64->64->3 and Lipschitz material 48->64->64->5             ia_mlp_fwd / ia_mlp_bwd_fused
    pbr._PbrShade light / uniform_light estimator: BRDF + env lookup + Lo                 ia_pbr_shade / ia_pbr_shade_bwd
    volume-interaction re-sampling (K1, no grad) + differentiable gathers / weights       ia_ray_resampling + torch indexing

What is constant w.r.t. the parameters follows the reference: sample positions, the candidate search, the re-sampled
interval indices, the secondary rays (directions, transmittance, indirect radiance) and the sampling weights are all
computed under no_grad there (:672-703, :772-795).
"""
from typing import Dict, Optional

import torch
from torch import Tensor
from torch.autograd import Function

from . import _lib as L
from . import fields, lib_nerfacc, nerfacc, pbr, render, train
from .tinycudann import _SH4


class _HashEncode(Function):
    @staticmethod
    def forward(ctx, xp, table):
        xp = xp.contiguous()
        ctx.save_for_backward(xp, table)
        return fields.hashgrid_forward(xp, table)

    @staticmethod
    def backward(ctx, g_enc):
        xp, table = ctx.saved_tensors
        g_table = torch.zeros_like(table)
        fields.hashgrid_backward(xp, g_enc.contiguous(), g_table)
        return None, g_table


class _MLP2(Function):
    """kind 1 (radiance: enc, xp, feat, sh, normal) / kind 2 (material: enc, xp, feat); sigmoid output.
    Gradients for every input segment (views of one [n, IN_PAD] buffer) and the six effective weights."""

    @staticmethod
    def forward(ctx, kind, out_dim, W1, b1, W2, b2, W3, b3, *segs):
        spec = {1: ((32, 1.0, 0.0), (3, 2.0, -1.0), (13, 1.0, 0.0), (16, 1.0, 0.0), (3, 1.0, 0.0)),
                2: ((32, 1.0, 0.0), (3, 2.0, -1.0), (13, 1.0, 0.0))}[kind]
        segs = [s.contiguous().float() for s in segs]
        ws = [t.contiguous().float() for t in (W1, b1, W2, b2, W3, b3)]
        y = fields.mlp_forward(kind, [(s, w, m, a) for s, (w, m, a) in zip(segs, spec)], *ws, out_dim)
        ctx.kind, ctx.out_dim, ctx.spec = kind, out_dim, spec
        ctx.save_for_backward(*ws, *segs)
        return y

    @staticmethod
    def backward(ctx, g_y):
        saved = ctx.saved_tensors
        ws, segs = saved[:6], saved[6:]
        n, dev = segs[0].shape[0], segs[0].device
        in_dim = sum(w for w, _, _ in ctx.spec)
        pad = (in_dim + 1) // 2 * 2
        g_x = torch.empty((n, pad), device=dev)
        ns, ptrs, strides, widths, muls, adds = train._segs([(s, w, m, a) for s, (w, m, a) in zip(segs, ctx.spec)])
        d = [torch.zeros((64, in_dim), device=dev), torch.zeros(64, device=dev), torch.zeros((64, 64), device=dev),
             torch.zeros(64, device=dev), torch.zeros((ctx.out_dim, 64), device=dev), torch.zeros(ctx.out_dim, device=dev)]
        L.check(L.lib().ia_mlp_bwd_fused(L.i32(ctx.kind), L.i64(n), L.i32(ns), ptrs, strides, widths, muls, adds,
                                         *[L.ptr(w) for w in ws], L.ptr(g_y.contiguous().float()), L.ptr(g_x), L.i32(pad),
                                         *[L.ptr(t) for t in d], L.stream()), "ia_mlp_bwd_fused")
        g_segs, c0 = [], 0
        for w, m, _ in ctx.spec:
            g_segs.append(g_x[:, c0:c0 + w] * m if m != 1.0 else g_x[:, c0:c0 + w])
            c0 += w
        return (None, None, *d, *g_segs)


def shade_differentiable_phys(rs, material, emitter, rays_o: Tensor, rays_d: Tensor, ray_indices: Tensor, t_starts: Tensor,
                              t_ends: Tensor, packed_info: Tensor, spp: int, light_u: Tensor, shuffle_u: Tensor,
                              render_mode: str = "uniform_light", env_base: Optional[Tensor] = None,
                              background_color: Optional[Tensor] = None, global_illumination: bool = False,
                              jitter_n: Optional[Tensor] = None, light_sampling: str = "shared") -> Dict[str, Tensor]:
    """differentiable rgb_normal_mats_alpha_fn + rendering_with_normals_mats_sdf + volume scattering.
    jitter_n [n_samples,3]: standard-normal noise of the material jitter pass (torch.randn_like in the reference,
    :1116-1140): materials are re-evaluated at x_cano + 0.01 * jitter_n for the relative smoothness maps (:1546-1597).
    light_sampling (render_mode 'light' only): 'shared' = the eval form of pbr_light_forward (:777-786): ONE set of spp
    light directions per frame (light_u [spp,3]) permuted per ray (shuffle_u [n_rays,spp]); 'per_point' = the training form
    (:772-776, `self.training`): an independent emitter.sample() per foreground point -- light_u is then [>= n_fg, 3]
    uniforms (or None: drawn on the device) and shuffle_u is not used."""
    dfm, geo, rad = rs.deformer, rs.geometry, rs.radiance
    n_rays = packed_info.shape[0]
    dev = rays_o.device
    pts = render.ray_points(rays_o, rays_d, ray_indices, t_starts, t_ends)
    with torch.no_grad():
        d = dfm.deform(pts, geo, with_grad=False, with_feature=False, want_fwd=True)
        valid = d["valid"]
        sel = d["sel"].long().clamp(min=0)
        c2w = d["fwd_J"].reshape(-1, 3, 3)[d["cand_src"].long()[sel]] if d["n_candidates"] > 0 else \
            torch.zeros((pts.shape[0], 3, 3), device=dev)
    W1k, b1, W2, b2 = geo.effective_weights()
    out, grad_c = train._SDFField.apply(d["pts_cano"], geo.grid_params, W1k, b1, W2, b2, geo.center, geo.scale, 0xFFFFFFFF,
                                            geo.inv_scale_host())
    vf = valid[:, None].float()
    feat = out * vf
    sdf = torch.where(valid, out[:, 0], torch.full_like(out[:, 0], 1e5))
    dflt_g = torch.nn.functional.pad(torch.ones((1, 1), device=dev), (2, 0))       # [0, 0, 1] without a host -> device scalar copy (a sync)
    sdf_grad = torch.where(valid[:, None], (c2w * grad_c[:, None, :]).sum(-1), dflt_g)
    w2s_rot = dfm.w2s[:3, :3].contiguous()
    normal_smpl, normal_world, refl01 = train._ShadePrep.apply(sdf_grad, rays_d, ray_indices, w2s_rot)
    alphas = train._Alpha.apply(sdf, t_ends - t_starts, rs.density.get_beta())
    # hash grid #2 once; radiance and material heads
    xp2 = ((d["pts_cano"] - rad.center) / rad.scale + 0.5).detach().contiguous()
    enc2 = _HashEncode.apply(xp2, rad.grid_params)
    sh = _SH4.apply(refl01)
    rgbs = _MLP2.apply(1, 3, *rad.effective_weights(), enc2, xp2, feat, sh, normal_world)
    mask = rad.prog.mask(rad.global_step, dev)
    mraw = _MLP2.apply(2, 5, *material.effective_weights(mask), enc2, xp2, feat)
    albedo = mraw[:, :3] * material.albedo_scale + material.albedo_bias
    rough = mraw[:, 3:4] * material.roughness_scale + material.roughness_bias
    metal = mraw[:, 4:5] * material.metallic_scale + material.metallic_bias
    weights, trans = nerfacc._WeightFromAlpha.apply(alphas, packed_info)
    acc = lambda v: nerfacc._Accumulate.apply(weights, v, ray_indices, packed_info)      # noqa: E731
    extra_maps = {}
    if jitter_n is not None:
        # material jitter pass: geometry feature + radiance embedding + material head at the jittered canonical points
        # (material_feature = hybrid); no deformer, no validity mask, exactly as the reference evaluates it
        x_j = (d["pts_cano"] + 0.01 * jitter_n[:pts.shape[0]]).detach().contiguous()
        out_j, _ = train._SDFField.apply(x_j, geo.grid_params, W1k, b1, W2, b2, geo.center, geo.scale, 0xFFFFFFFF, geo.inv_scale_host())
        xp2_j = ((x_j - rad.center) / rad.scale + 0.5).contiguous()
        enc2_j = _HashEncode.apply(xp2_j, rad.grid_params)
        mraw_j = _MLP2.apply(2, 5, *material.effective_weights(mask), enc2_j, xp2_j, out_j)
        alb_j = mraw_j[:, :3] * material.albedo_scale + material.albedo_bias
        rough_j = mraw_j[:, 3:4] * material.roughness_scale + material.roughness_bias
        metal_j = mraw_j[:, 4:5] * material.metallic_scale + material.metallic_bias

        def rel(v, vj):            # compute_relative_smoothness_loss (:383-388)
            base = torch.maximum(v, vj).clamp_min(1e-6)
            return (((v - vj) / base) ** 2).sum(-1, keepdim=True)
        orient = (rays_d[ray_indices] * normal_smpl).sum(-1, keepdim=True).clamp_min(0.0)
        extra_maps = dict(normals_orientation_loss_map=acc(orient.contiguous()),
                          albedo_smoothness_loss_map=acc(rel(albedo, alb_j).contiguous()),
                          roughness_smoothness_loss_map=acc(rel(rough, rough_j).contiguous()),
                          metallic_smoothness_loss_map=acc(rel(metal, metal_j).contiguous()))
    res = dict(comp_rgb=acc(rgbs), comp_normal=acc(normal_world), opacity=acc(None), albedo=acc(albedo.contiguous()),
               roughness=acc(rough.contiguous()), metallic=acc(metal.contiguous()), weights=weights, alphas=alphas, sdf=sdf,
               sdf_grad=sdf_grad, valid=valid, n_samples=pts.shape[0], **extra_maps)
    # ---- volume scattering (enable_phys)
    if background_color is None:
        background_color = torch.ones(3, device=dev)
    rgb_phys = background_color[None].expand(n_rays, 3)
    stats = dict(n_resampled=0, n_fg=0, n_secondary=0)
    if ray_indices.numel() > 0:
        vi = pbr.VolumeInteraction(ray_indices, t_starts, t_ends, n_rays, spp, weights, sdf)
        stats["n_resampled"], stats["n_fg"] = vi.R, vi.F
        if vi.F > 0:
            F_ = vi.F
            # differentiable gathers of the per-interval attributes + re-sampled weights w[s] / count[s] (ia_vi_gather / _bwd)
            w_fg, nrm, alb, rgh, mtl = vi.gather(rays_o, rays_d, weights, normal_smpl, albedo, rough, metal)
            with torch.no_grad():
                inv_pdf = None
                if render_mode == "light" and light_sampling == "per_point":
                    u = light_u[:F_].contiguous() if light_u is not None else None
                    dirs = emitter.sample(F_, u, w2s_rot=w2s_rot)                      # emitter.sample + transform_dirs_w2s
                    ro, rd, src, out_dirs = pbr.secondary_rays(nrm, vi.positions, dirs)
                else:
                    if render_mode == "light":
                        dirs_smpl = emitter.sample(spp, light_u, w2s_rot=w2s_rot)
                        inv_pdf_all = None
                    else:
                        assert spp == 512, "uniform_light asserts samples_per_pixel == 512 (:1392)"
                        dirs_smpl, inv_pdf_all = pbr.uniform_sphere_stratified(16, 32, light_u[:, :2])
                    shuffled = vi.shuffle(shuffle_u)
                    ro, rd, src, out_dirs = pbr.secondary_rays(nrm, vi.positions, dirs_smpl, dir_index=shuffled)
                    inv_pdf = inv_pdf_all[shuffled.long()] if inv_pdf_all is not None else None
                stats["n_secondary"] = int(ro.shape[0])
                t_, c_ = rs.compute_indirect_radiance(ro, rd)
                sec_tr, sec_rgb = pbr.scatter_secondary(F_, src, t_, c_)
            fg_Lo, fg_Ld, fg_Ls = pbr.pbr_shade_differentiable(
                render_mode, nrm, alb, rgh, mtl, vi.view_dirs, out_dirs, sec_tr, sec_rgb if global_illumination else None, emitter,
                w2s_rot, inv_pdf=inv_pdf, env_base=env_base)
            # background re-samples carry the background colour (Lo.scatter_(0, bg_indices, background_color), :1335-1342),
            # their weights sum to the ray's transmittance; rays without samples show the background (:1452-1466)
            rgb_phys = vi.composite(w_fg, fg_Lo, 1.0 - res["opacity"], background_color)
            res.update(fg_Lo=fg_Lo, fg_Lo_diff=fg_Ld, fg_Lo_spec=fg_Ls, fg_weights=w_fg, secondary_tr=sec_tr, secondary_rgb=sec_rgb,
                       out_dirs=out_dirs, inv_pdf=inv_pdf, volume_interaction=vi)
    res.update(comp_rgb_phys=rgb_phys, stats=stats)
    return res


def training_loss_phys(out: Dict[str, Tensor], target_rgb: Tensor, target_mask: Optional[Tensor] = None,
                       lambda_phys: float = 1.0, lambda_smooth: float = 0.0, lambda_orient: float = 0.0, **kw) -> Tensor:
    """train.training_loss + the L1 term on the physically based image (systems/intrinsic_avatar.py:180-190) and, when
    the jitter pass ran, the material smoothness / normal orientation maps."""
    loss = train.training_loss(out, target_rgb, target_mask, **kw) + lambda_phys * (out["comp_rgb_phys"] - target_rgb).abs().mean()
    if "albedo_smoothness_loss_map" in out:
        if lambda_smooth > 0.0:
            loss = loss + lambda_smooth * (out["albedo_smoothness_loss_map"].mean() + out["roughness_smoothness_loss_map"].mean()
                                           + out["metallic_smoothness_loss_map"].mean())
        if lambda_orient > 0.0:
            loss = loss + lambda_orient * out["normals_orientation_loss_map"].mean()
    return loss


def forward_backward_phys_pipelined(rs, views, material, emitter, spp: int, n_workers: int = 2, **kw) -> Dict[str, int]:
    """Several ray chunks of ONE frame in flight: `views` = [(rays, target_rgb, target_mask, loss_scale), ...] as a caller would pass
    them to RenderStep.forward_backward_phys one after the other (gradient accumulation over the chunks of a frame).  Here worker k
    takes the chunks k, k + n_workers, ... on its own HIP stream and host thread, and -- what makes it pay -- worker k + 1 starts when
    worker k ENTERS its secondary march: the primary sampling / shading / backward of one half-frame (many small kernels and the
    size read-backs, which leave most of the device idle) then run under the other half-frame's march (search, hash gather, SDF head:
    the part that fills the device).  Two processes on one GPU showed the head-room (1.16 x); symmetric workers that
    start together stay in the same phase and gain nothing.
    Results: every chunk's forward is what forward_backward_phys computes for it (ray-batch sharding invariance); the parameter
    gradients are the sum of the chunks' gradients -- with two chunks a two-term sum, which is the same bits in either order.
    Each worker marches its secondary rays on ONE stream (its own)."""
    import threading
    dev = views[0][0].device
    main = torch.cuda.current_stream(dev)
    n_workers = max(1, min(n_workers, len(views)))
    # the workers' streams live as long as the RenderStep: the caching allocator keeps one pool per stream, and a fresh stream per
    # step would strand the previous step's blocks in dead pools
    streams = getattr(rs, "_pipeline_streams", None)
    if streams is None or len(streams) != n_workers or streams[0].device != dev:
        streams = rs._pipeline_streams = [torch.cuda.Stream(device=dev) for _ in range(n_workers)]
    entered = [threading.Event() for _ in range(n_workers)]
    errors, stats = [], [None] * len(views)
    _ = rs.grid_bits, rs._sort_grid_params()            # lazily cached host-side state: made before the threads start

    def worker(k):
        try:
            if k > 0:
                entered[k - 1].wait()                   # start when the previous worker's first chunk reaches its march
            rs._march_hooks.streams = 1
            rs._march_hooks.enter = entered[k].set
            with torch.cuda.device(dev), torch.cuda.stream(streams[k]):
                for ci in range(k, len(views), n_workers):
                    r, t, m, frac = views[ci]
                    o = rs.forward_backward_phys(r, t, material, emitter, spp, None, None, target_mask=m, loss_scale=frac, **kw)
                    stats[ci] = {a: int(b) for a, b in o["stats"].items() if isinstance(b, (int, float))}
                    del o
        except BaseException as e:                      # noqa: B902 -- re-raised by the caller
            errors.append(e)
        finally:
            entered[k].set()                            # never leave the next worker waiting
            rs._march_hooks.enter = None
            rs._march_hooks.streams = None
    threads = [threading.Thread(target=worker, args=(k,), daemon=True) for k in range(n_workers)]
    for s_ in streams:
        s_.wait_stream(main)
    for t_ in threads:
        t_.start()
    for t_ in threads:
        t_.join()
    for s_ in streams:
        main.wait_stream(s_)
    if errors:
        raise errors[0]
    tot: Dict[str, int] = {}
    for st in stats:
        for a, b in (st or {}).items():
            tot[a] = tot.get(a, 0) + b
    return tot
