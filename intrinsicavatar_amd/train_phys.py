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
        g_table = L.zeros_like(table)
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
        shapes = [(64, in_dim), (64,), (64, 64), (64,), (ctx.out_dim, 64), (ctx.out_dim,)]
        d = train.zeros_like_shapes(shapes, dev)          # ONE fill for the six accumulated weight gradients
        L.check(L.lib().ia_mlp_bwd_fused(L.i32(ctx.kind), L.i64(n), L.i32(ns), ptrs, strides, widths, muls, adds,
                                         *[L.ptr(w) for w in ws], L.ptr(g_y.contiguous().float()), L.ptr(g_x), L.i32(pad),
                                         *[L.ptr(t) for t in d], L.stream()), "ia_mlp_bwd_fused")
        g_segs, c0 = [], 0
        for k, (w, m, _) in enumerate(ctx.spec):
            if not ctx.needs_input_grad[8 + k]:          # e.g. the grid coordinates: constants of the step
                g_segs.append(None)
            else:
                g_segs.append(g_x[:, c0:c0 + w] * m if m != 1.0 else g_x[:, c0:c0 + w])
            c0 += w
        return (None, None, *d, *g_segs)


class _MaterialAffine(Function):
    """sigmoid outputs [n,5] -> albedo [n,3], roughness [n,1], metallic [n,1] in the ranges of VolumeMaterial.forward
    (models/pbr/material.py:44-50): ia_material_affine / _bwd instead of three slices, six element-wise launches and their backward."""

    @staticmethod
    def forward(ctx, mraw, material):
        mraw = mraw.contiguous()
        n, dev = mraw.shape[0], mraw.device
        alb, rgh, mtl = torch.empty((n, 3), device=dev), torch.empty((n, 1), device=dev), torch.empty((n, 1), device=dev)
        ctx.scales = (float(material.albedo_scale), float(material.roughness_scale), float(material.metallic_scale))
        L.check(L.lib().ia_material_affine(L.i64(n), L.ptr(mraw), L.f32(material.albedo_scale), L.f32(material.albedo_bias),
                                           L.f32(material.roughness_scale), L.f32(material.roughness_bias), L.f32(material.metallic_scale),
                                           L.f32(material.metallic_bias), L.ptr(alb), L.ptr(rgh), L.ptr(mtl), L.stream()), "ia_material_affine")
        ctx.n = n
        ctx.dev = dev
        ctx.set_materialize_grads(False)
        return alb, rgh, mtl

    @staticmethod
    def backward(ctx, g_alb, g_rgh, g_mtl):
        cg = lambda t: t.contiguous() if t is not None else None      # noqa: E731
        g = torch.empty((ctx.n, 5), device=ctx.dev)
        a, r, m = ctx.scales
        L.check(L.lib().ia_material_affine_bwd(L.i64(ctx.n), L.ptr(cg(g_alb)), L.ptr(cg(g_rgh)), L.ptr(cg(g_mtl)), L.f32(a), L.f32(r), L.f32(m),
                                               L.ptr(g), L.stream()), "ia_material_affine_bwd")
        return g, None


class _PhysLoss(Function):
    """the default loss composition of IntrinsicAvatarSystem.training_step (systems/intrinsic_avatar.py:165-252) on the composited maps:
    L1 rgb + lambda_eik * eikonal mean + lambda_mask * BCE(opacity, mask) + lambda_phys * L1 rgb_phys -- ia_phys_loss / _bwd, one launch
    each way instead of ~15 element-wise / reduce launches and their backward.  Returns (loss, terms [5])."""

    @staticmethod
    def forward(ctx, comp_rgb, comp_rgb_phys, opacity, eik_part, target_rgb, target_mask, lambda_phys, lambda_mask, lambda_eik, eik_denom):
        c = lambda t: t.contiguous().float() if t is not None else None      # noqa: E731
        comp_rgb, comp_rgb_phys, opacity, target_rgb, target_mask = c(comp_rgb), c(comp_rgb_phys), c(opacity), c(target_rgb), c(target_mask)
        n, dev = comp_rgb.shape[0], comp_rgb.device
        terms = torch.empty(5, device=dev)
        k = int(eik_part.shape[0]) if eik_part is not None else 0
        nb = int(L.lib().ia_phys_loss_tmp_bytes(L.i64(n)))
        tmp = torch.empty(nb, dtype=torch.uint8, device=dev) if nb else None
        L.check(L.lib().ia_phys_loss(L.i64(n), L.ptr(comp_rgb), L.ptr(comp_rgb_phys), L.ptr(opacity), L.ptr(target_rgb), L.ptr(target_mask),
                                     L.ptr(eik_part), L.i32(k), L.f32(lambda_phys), L.f32(lambda_mask), L.f32(lambda_eik), L.f32(eik_denom),
                                     L.ptr(terms), L.ptr(tmp), L.stream()), "ia_phys_loss")
        ctx.save_for_backward(comp_rgb, comp_rgb_phys, opacity, target_rgb, target_mask)
        ctx.cfg = (float(lambda_phys), float(lambda_mask), float(lambda_eik), float(eik_denom), tuple(eik_part.shape) if eik_part is not None else None)
        ctx.mark_non_differentiable(terms)
        ctx.set_materialize_grads(False)
        return terms[4], terms

    @staticmethod
    def backward(ctx, g_loss, _g_terms):
        comp_rgb, comp_rgb_phys, opacity, target_rgb, target_mask = ctx.saved_tensors
        lp, lm, le, den, has_eik = ctx.cfg
        n, dev = comp_rgb.shape[0], comp_rgb.device
        g_rgb = torch.empty_like(comp_rgb)
        g_phys = torch.empty_like(comp_rgb_phys) if comp_rgb_phys is not None else None
        g_op = torch.empty_like(opacity) if target_mask is not None else None
        g_eik = torch.empty(1, device=dev) if has_eik else None
        L.check(L.lib().ia_phys_loss_bwd(L.i64(n), L.ptr(comp_rgb), L.ptr(comp_rgb_phys), L.ptr(opacity), L.ptr(target_rgb), L.ptr(target_mask),
                                         L.ptr(g_loss.reshape(1).float().contiguous()), L.f32(lp), L.f32(lm), L.f32(le), L.f32(den),
                                         L.ptr(g_rgb), L.ptr(g_phys), L.ptr(g_op), L.ptr(g_eik), L.stream()), "ia_phys_loss_bwd")
        # (d loss / d eikonal sum is ONE scalar: it goes back as a stride-0 view of the partials' shape, train._EikonalPartials reads element 0)
        return g_rgb, g_phys, g_op, (g_eik.expand(has_eik) if has_eik else None), None, None, None, None, None, None


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
    W1k, b1, W2, b2 = geo.effective_weights()
    out, grad_c = train._SDFField.apply(d["pts_cano"], geo.grid_params, W1k, b1, W2, b2, geo.center, geo.scale, 0xFFFFFFFF,
                                            geo.inv_scale_host())
    # invalid points: sdf 1e5, feature 0, gradient [0,0,1] (snarf_deformer.py:192-231); valid ones: the normal push-forward with the
    # winning candidate's blended rotation -- one kernel each way (ia_select_push / _bwd)
    feat, sdf, sdf_grad, _c2w = train._SelectPush.apply(out, grad_c, valid, d["fwd_J"] if d["n_candidates"] > 0 else None, d["cand_src"], d["sel"])
    w2s_rot = dfm.w2s[:3, :3].contiguous()
    normal_smpl, normal_world, refl01 = train._ShadePrep.apply(sdf_grad, rays_d, ray_indices, w2s_rot)
    alphas = train._Alpha.apply(sdf, t_ends - t_starts, rs.density.get_beta())
    # hash grid #2 once; radiance and material heads
    with torch.no_grad():
        xp2 = fields.normalize_points(d["pts_cano"], rad.center, rad.scale)
    enc2 = _HashEncode.apply(xp2, rad.grid_params)
    sh = _SH4.apply(refl01)
    rgbs = _MLP2.apply(1, 3, *rad.effective_weights(), enc2, xp2, feat, sh, normal_world)
    mask = rad.prog.mask(rad.global_step, dev)
    mat_w = material.effective_weights(mask)             # once per step: the jitter pass reads the same matrices
    mraw = _MLP2.apply(2, 5, *mat_w, enc2, xp2, feat)
    albedo, rough, metal = _MaterialAffine.apply(mraw, material)
    weights, trans = nerfacc._WeightFromAlpha.apply(alphas, packed_info)
    acc = lambda v: nerfacc._Accumulate.apply(weights, v, ray_indices, packed_info)      # noqa: E731
    extra_maps = {}
    if jitter_n is not None:
        # material jitter pass: geometry feature + radiance embedding + material head at the jittered canonical points
        # (material_feature = hybrid); no deformer, no validity mask, exactly as the reference evaluates it
        with torch.no_grad():
            x_j = (d["pts_cano"] + 0.01 * jitter_n[:pts.shape[0]]).contiguous()
        out_j, _ = train._SDFField.apply(x_j, geo.grid_params, W1k, b1, W2, b2, geo.center, geo.scale, 0xFFFFFFFF, geo.inv_scale_host())
        xp2_j = fields.normalize_points(x_j, rad.center, rad.scale)
        enc2_j = _HashEncode.apply(xp2_j, rad.grid_params)
        mraw_j = _MLP2.apply(2, 5, *mat_w, enc2_j, xp2_j, out_j)
        alb_j, rough_j, metal_j = _MaterialAffine.apply(mraw_j, material)

        def rel(v, vj):            # compute_relative_smoothness_loss (:383-388)
            base = torch.maximum(v, vj).clamp_min(1e-6)
            return (((v - vj) / base) ** 2).sum(-1, keepdim=True)
        orient = (rays_d[ray_indices] * normal_smpl).sum(-1, keepdim=True).clamp_min(0.0)
        extra_maps = dict(normals_orientation_loss_map=acc(orient.contiguous()),
                          albedo_smoothness_loss_map=acc(rel(albedo, alb_j).contiguous()),
                          roughness_smoothness_loss_map=acc(rel(rough, rough_j).contiguous()),
                          metallic_smoothness_loss_map=acc(rel(metal, metal_j).contiguous()))
    res = dict(comp_rgb=acc(rgbs), comp_normal=acc(normal_world), opacity=acc(None), albedo=acc(albedo),
               roughness=acc(rough), metallic=acc(metal), weights=weights, alphas=alphas, sdf=sdf,
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
                       out_dirs=out_dirs, inv_pdf=inv_pdf, volume_interaction=vi, fg_normals=nrm)
    res.update(comp_rgb_phys=rgb_phys, stats=stats)
    return res


FUSED_LOSS = __import__("os").environ.get("IA_FUSED_LOSS", "1") != "0"


def training_loss_phys(out: Dict[str, Tensor], target_rgb: Tensor, target_mask: Optional[Tensor] = None,
                       lambda_phys: float = 1.0, lambda_smooth: float = 0.0, lambda_orient: float = 0.0, **kw) -> Tensor:
    """train.training_loss + the L1 term on the physically based image (systems/intrinsic_avatar.py:180-190) and, when
    the jitter pass ran, the material smoothness / normal orientation maps."""
    if FUSED_LOSS and not kw.get("lambda_curv"):
        # one kernel each way (ia_phys_loss); the eikonal partial sums of ia_eikonal go in as they are
        lam_eik, lam_mask = kw.get("lambda_eik", 0.1), kw.get("lambda_mask", 0.1)
        part = train._EikonalPartials.apply(out["sdf_grad"], out["valid"])
        denom = kw.get("eik_denominator")
        denom = float(denom) if denom is not None else float(max(out["sdf_grad"].shape[0], 1))
        op = out["opacity"] if target_mask is not None else None
        loss, terms = _PhysLoss.apply(out["comp_rgb"], out["comp_rgb_phys"], op, part, target_rgb, target_mask, lambda_phys, lam_mask, lam_eik, denom)
        out["loss_terms"] = terms
    else:
        loss = train.training_loss(out, target_rgb, target_mask, **kw) + lambda_phys * (out["comp_rgb_phys"] - target_rgb).abs().mean()
    if "albedo_smoothness_loss_map" in out:
        if lambda_smooth > 0.0:
            loss = loss + lambda_smooth * (out["albedo_smoothness_loss_map"].mean() + out["roughness_smoothness_loss_map"].mean()
                                           + out["metallic_smoothness_loss_map"].mean())
        if lambda_orient > 0.0:
            loss = loss + lambda_orient * out["normals_orientation_loss_map"].mean()
    return loss


def reference_training_loss(out: Dict[str, Tensor], rgb: Tensor, alpha: Optional[Tensor], lam: Dict[str, float], material):
    """IntrinsicAvatarSystem.training_step (systems/intrinsic_avatar.py:160-301) on the model's output dict (RenderStep._training_dict /
    forward_train_): every term the reference logs, weighted by `lam` (the `system.loss` section of configs/config.yaml:87-109 with the
    schedules already evaluated: plain floats; a term whose weight is 0 or missing is not added).  -> (loss, {term: value}).
      rgb_l1 / rgb_mse            :165-178   sRGB image over rays_valid_full
      rgb_phys_l1 / rgb_phys_mse  :181-212   physically based image over rays_valid_phys_full (add_emitter False)
      rgb_demodulated             :217-224   luma of the demodulated image against the target's channel maximum
      eikonal                     :235-239   mean over ALL samples of (|grad sdf| - 1)^2
      mask_bce / mask_mse, opaque :242-257   clamped opacity against the batch's alpha
      sparsity                    :260-264
      material regularisers       :285-290   models/pbr/material.py:53-87 (smoothness / orientation means, albedo entropy)"""
    F_ = torch.nn.functional
    v, vp = out["rays_valid_full"][..., 0], out["rays_valid_phys_full"][..., 0]
    t = {}
    t["rgb_mse"] = F_.mse_loss(out["comp_rgb_full"][v], rgb[v])
    t["rgb_l1"] = F_.l1_loss(out["comp_rgb_full"][v], rgb[v])
    t["rgb_phys_mse"] = F_.mse_loss(out["comp_rgb_phys_full"][vp], rgb[vp])
    t["rgb_phys_l1"] = F_.l1_loss(out["comp_rgb_phys_full"][vp], rgb[vp])
    t["rgb_demodulated"] = F_.l1_loss(pbr.luma(out["comp_demod_phys_full"][vp]), pbr.max_value(rgb[vp]))
    t["eikonal"] = ((torch.linalg.norm(out["sdf_grad_samples"], ord=2, dim=-1) - 1.0) ** 2).mean()
    if alpha is not None:
        op = torch.clamp(out["opacity"].squeeze(-1), 1.0e-3, 1.0 - 1.0e-3)

        def bce(x, y):            # systems/criterions.py:229-233
            return -(y * torch.log(x) + (1 - y) * torch.log(1 - x)).mean()
        t["mask_mse"] = F_.mse_loss(op, alpha)
        t["mask_bce"] = bce(op, alpha)
        t["opaque"] = bce(op, op)
    t["sparsity"] = torch.exp(-float(lam.get("sparsity_scale", 1.0)) * out["sdf_samples"].abs()).mean()
    reg = material.regularizations(out)
    for k in ("normal_orientation", "albedo_smoothness", "roughness_smoothness", "metallic_smoothness", "albedo_entropy"):
        if k in reg:
            t[k] = reg[k]
    loss = 0.0
    for k, val in t.items():
        w = float(lam.get("lambda_" + k, 0.0) or 0.0)
        if w != 0.0:
            loss = loss + w * val
    return loss, t


def forward_backward_phys_pipelined(rs, views, material, emitter, spp: int, n_workers: int = 2, **kw) -> Dict[str, int]:
    """EXPERIMENTAL (not the default of any entry point; bench.py takes it only with IA_FRAME_PIPELINE=n; measured 317-325 ms against
    310-322 ms for the headline step: no gain, profiles/r05_frame_pipeline_ab.jsonl).  Covered by
    tests/test_gpu_train.py::test_pipelined_half_frames_match_the_sequential_chunk_loop (same gradients as the sequential chunk loop to
    2e-5 of a group's largest entry); shared per-call diagnostics of the RenderStep (last_secondary_streams, the deformer's finite-voxel
    flag) are written by both workers and only meaningful after the call.
    Several ray chunks of ONE frame in flight: `views` = [(rays, target_rgb, target_mask, loss_scale), ...] as a caller would pass
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
