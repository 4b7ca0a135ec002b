"""Differentiable (training) form of the shading pass of render_step on the MI355X kernels.

The reference trains through `rgb_normal_alpha_fn` + `rendering_with_normals_sdf`
(models/intrinsic_avatar.py:1032-1064,1272-1287; models/volrend.py:638-807) with torch autograd,
including the double backward through the analytic normal (models/rf/geometry.py:165-172).  Here
each stage is a torch.autograd.Function whose forward AND backward are HIP kernels behind the C ABI:

    _SDFField      hash grid + SDF MLP -> (feature[13], d sdf/d x)      ia_hashgrid_fwd/bwd, ia_mlp_fwd, ia_sdf_mlp_bwd
    _ShadePrep     normals / reflected direction                         ia_shade_prep(_bwd)
    _Alpha         Laplace density -> alpha                              ia_laplace_alpha(_bwd)
    _Radiance      hash grid #2 + SH + radiance MLP                      ia_hashgrid_fwd/bwd, ia_sh4_fwd/bwd, ia_mlp_fwd/bwd
    nerfacc._WeightFromAlpha / _Accumulate                               ia_render_weight_from_alpha(_bwd), ia_accumulate_*(_bwd)

Sample positions, the candidate search and the winner selection are not differentiated, exactly as in the
reference (resampling under no_grad; torch.gather passes gradients to the winning candidate only).  The tiny
weight-gradient reductions dW = G^T A ([64 x n] x [n x <=68]) use the split-K MFMA kernel ia_wgrad (rocBLAS maps this
shape to a single output tile walking K = millions of points).
"""
import ctypes as C
import math
import os
from typing import Dict, Optional

import torch
from torch import Tensor
from torch.autograd import Function

from . import _lib as L
from . import fields, lib_nerfacc, nerfacc, render


# fused data + weight backward (csrc/mlp_train.hip); False = operand pairs through HBM + ia_wgrad (csrc/mlp_bwd.hip)
FUSED_WGRAD = os.environ.get("IA_FUSED_WGRAD", "1") != "0"


def wgrad(G: Tensor, M: int, A: Tensor, N: int, want_bias: bool = True):
    """dW [M,N] = G[:, :M]^T A[:, :N], db [M] = column sums of G -- split-K MFMA kernel (ia_wgrad)."""
    dev = G.device
    dW = torch.zeros((M, N), device=dev)
    db = torch.zeros(M, device=dev) if want_bias else None
    L.check(L.lib().ia_wgrad(L.i64(G.shape[0]), L.ptr(G), L.i32(G.stride(0)), L.i32(M), L.ptr(A), L.i32(A.stride(0)), L.i32(N),
                             L.ptr(dW), L.i32(N), L.ptr(db), L.stream()), "ia_wgrad")
    return dW, db


def _jac_contract_T(jac: Tensor, v: Tensor) -> Tensor:
    """out[n,3] = sum_k v[n,k] * jac[n,k,:]   (d L / d x' from d L / d enc; ia_hashgrid_jac_contract mode 0)."""
    n, K = jac.shape[0], jac.shape[1]
    out = torch.empty((n, 3), device=jac.device)
    L.check(L.lib().ia_hashgrid_jac_contract(L.i32(0), L.i64(n), L.i32(K), L.ptr(jac), C.c_void_p(v.data_ptr()),
                                             L.i32(v.stride(0)), L.ptr(out), L.i32(3), L.stream()), "ia_hashgrid_jac_contract")
    return out


def _segs(segs):
    ns = len(segs)
    ptrs = (C.c_void_p * ns)()
    strides = (C.c_int * ns)()
    widths = (C.c_int * ns)()
    muls = (C.c_float * ns)()
    adds = (C.c_float * ns)()
    for i, (t, w, m, a) in enumerate(segs):
        assert t.is_cuda and t.dtype == torch.float32 and t.stride(-1) == 1
        ptrs[i], strides[i], widths[i], muls[i], adds[i] = t.data_ptr(), t.stride(0), w, m, a
    return ns, ptrs, strides, widths, muls, adds


class _SDFField(Function):
    """(x_cano, table, W1k, b1, W2, b2) -> (out[n,13], grad[n,3]); grads w.r.t. table and weights (1st + 2nd order)."""

    @staticmethod
    def forward(ctx, x, table, W1k, b1, W2, b2, center, scale, level_bits=0xFFFFFFFF, inv_scale_host=None):
        ctx.level_bits = int(level_bits)
        xp = fields.normalize_points(x, center, scale) if not x.requires_grad else ((x - center) / scale + 0.5).contiguous()
        enc, jac = fields.hashgrid_forward(xp, table, with_jac=True)
        inv = inv_scale_host if inv_scale_host is not None else (1.0 / scale).tolist()      # the fallback is a host sync
        y, grad = fields.mlp_forward(0, [(enc, 32, 1.0, 0.0), (xp, 3, 2.0, -1.0)], W1k, b1, None, None, W2, b2, 13,
                                     jac=jac, xyz_col=32, inv_scale=inv, want_grad=True)
        ctx.save_for_backward(xp, enc, jac, table, W1k.contiguous(), b1.contiguous(), W2.contiguous(), b2.contiguous(), scale)
        return y, grad

    @staticmethod
    def backward(ctx, g_y, g_grad):
        xp, enc, jac, table, W1k, b1, W2, b2, scale = ctx.saved_tensors
        n, dev = xp.shape[0], xp.device
        g_y = g_y.contiguous().float()
        q = (g_grad / scale).contiguous().float()
        gE, gG = torch.empty((n, 32), device=dev), torch.empty((n, 32), device=dev)
        ns, ptrs, strides, widths, muls, adds = _segs([(enc, 32, 1.0, 0.0), (xp, 3, 2.0, -1.0)])
        g_table = L.zeros_like(table)
        if FUSED_WGRAD:
            dW1k, db1, dW2, db2 = zeros_like_shapes([(64, 35), (64,), (13, 64), (13,)], dev)
            want_x = ctx.needs_input_grad[0]
            g_xyz = torch.empty((n, 3), device=dev) if want_x else None
            L.check(L.lib().ia_sdf_mlp_bwd_fused(L.i64(n), L.i32(ns), ptrs, strides, widths, muls, adds, L.ptr(W1k), L.ptr(b1),
                                                 L.ptr(W2), L.ptr(b2), L.ptr(jac), L.ptr(g_y), L.ptr(q), L.ptr(gE), L.ptr(gG),
                                                 L.ptr(dW1k), L.ptr(db1), L.ptr(dW2), L.ptr(db2), L.ptr(g_xyz), L.stream()),
                    "ia_sdf_mlp_bwd_fused")
            fields.hashgrid_backward(xp, gE, g_table, g_jac=gG, q=q, level_mask=ctx.level_bits)
            g_x = None
            if want_x:
                # first-order d L / d x (pose gradients, SNARF's implicit differentiation): J_enc^T gE + 2 g_xyz, / scale.
                # The dependence of the analytic normal on x (a Hessian term; tiny-cuda-nn returns zero for it too) is dropped.
                g_x = (_jac_contract_T(jac, gE) + 2.0 * g_xyz) / scale
            return g_x, g_table, dW1k, db1, dW2, db2, None, None, None, None
        if ctx.needs_input_grad[0]:
            raise NotImplementedError("pose gradients (d L / d x_cano) are only produced by the fused backward (FUSED_WGRAD)")
        Hh, U = torch.empty((n, 36), device=dev), torch.empty((n, 36), device=dev)
        DZ, GZ, A, DGS = (torch.empty((n, 64), device=dev) for _ in range(4))
        L.check(L.lib().ia_sdf_mlp_bwd(L.i64(n), L.i32(ns), ptrs, strides, widths, muls, adds, L.ptr(W1k), L.ptr(b1),
                                       L.ptr(W2), L.ptr(b2), L.ptr(jac), L.ptr(g_y), L.ptr(q), L.ptr(gE), L.ptr(gG),
                                       L.ptr(Hh), L.ptr(U), L.ptr(DZ), L.ptr(GZ), L.ptr(A), L.ptr(DGS), L.stream()),
                "ia_sdf_mlp_bwd")
        fields.hashgrid_backward(xp, gE, g_table, g_jac=gG, q=q, level_mask=ctx.level_bits)
        dW1k, db1 = wgrad(DZ, 64, Hh, 35)
        dW1k = dW1k + wgrad(GZ, 64, U, 35, want_bias=False)[0]
        dW2, db2 = wgrad(g_y, 13, A, 64)
        dW2[0] += wgrad(DGS, 64, DGS, 1)[1]          # column sums of DGS
        return None, g_table, dW1k, db1, dW2, db2, None, None, None, None


class _ShadePrep(Function):
    @staticmethod
    def forward(ctx, sdf_grad, rays_d, ray_indices, w2s_rot):
        sdf_grad = sdf_grad.contiguous()
        ns, nw, rf = render.shade_prep(sdf_grad, rays_d, ray_indices, w2s_rot)
        ctx.save_for_backward(sdf_grad, rays_d, ray_indices, w2s_rot, ns)
        ctx.set_materialize_grads(False)
        return ns, nw, rf

    @staticmethod
    def backward(ctx, g_ns, g_nw, g_rf):
        sdf_grad, rays_d, ray_indices, w2s_rot, ns = ctx.saved_tensors
        n = sdf_grad.shape[0]
        cg = lambda t: t.contiguous() if t is not None else None      # noqa: E731  (a missing gradient is NULL: zero inside the kernel)
        out = torch.empty_like(sdf_grad)
        # g_ns: normal_smpl = g / max(|g|, 1e-6), read by the BRDF of the PBR branch only
        L.check(L.lib().ia_shade_prep_bwd(L.i64(n), L.ptr(sdf_grad), L.ptr(rays_d), L.ptr(ray_indices), L.ptr(w2s_rot),
                                          L.ptr(cg(g_nw)), L.ptr(cg(g_rf)), L.ptr(cg(g_ns)), L.ptr(out), L.stream()), "ia_shade_prep_bwd")
        return out, None, None, None


class _SelectPush(Function):
    """(out[n,13], grad_c[n,3]) -> (feat, sdf, sdf_grad, c2w): masking by `valid`, defaults for points without a canonical
    correspondence and the normal push-forward with the winning candidate's blended rotation, in one kernel each way."""

    @staticmethod
    def forward(ctx, out, grad_c, valid, fwd_J, cand_src, sel):
        n, dev = out.shape[0], out.device
        out, grad_c = out.contiguous(), grad_c.contiguous()
        feat, sdf = torch.empty((n, 13), device=dev), torch.empty(n, device=dev)
        sdf_grad, c2w = torch.empty((n, 3), device=dev), torch.empty((n, 3, 3), device=dev)
        L.check(L.lib().ia_select_push(L.i64(n), L.ptr(out), L.ptr(grad_c), L.ptr(valid), L.ptr(fwd_J), L.ptr(cand_src),
                                       L.ptr(sel), L.ptr(feat), L.ptr(sdf), L.ptr(sdf_grad), L.ptr(c2w), L.stream()),
                "ia_select_push")
        ctx.save_for_backward(valid, c2w)
        ctx.mark_non_differentiable(c2w)
        ctx.set_materialize_grads(False)
        return feat, sdf, sdf_grad, c2w

    @staticmethod
    def backward(ctx, g_feat, g_sdf, g_sdf_grad, _g_c2w):
        valid, c2w = ctx.saved_tensors
        n, dev = valid.shape[0], valid.device
        cg = lambda t: t.contiguous() if t is not None else None      # noqa: E731
        g_out, g_gc = torch.empty((n, 13), device=dev), torch.empty((n, 3), device=dev)
        L.check(L.lib().ia_select_push_bwd(L.i64(n), L.ptr(valid), L.ptr(c2w), L.ptr(cg(g_feat)), L.ptr(cg(g_sdf)),
                                           L.ptr(cg(g_sdf_grad)), L.ptr(g_out), L.ptr(g_gc), L.stream()), "ia_select_push_bwd")
        return g_out, g_gc, None, None, None, None


class _Eikonal(Function):
    """sdf_grad[n,3], valid[n] -> (sum over valid of (|g| - 1)^2, number of valid samples)."""

    @staticmethod
    def forward(ctx, sdf_grad, valid):
        n = sdf_grad.shape[0]
        sdf_grad = sdf_grad.contiguous()
        k = int(L.lib().ia_eikonal_partials(L.i64(n)))
        part = torch.zeros((max(k, 1), 2), device=sdf_grad.device)
        L.check(L.lib().ia_eikonal(L.i64(n), L.ptr(sdf_grad), L.ptr(valid), L.ptr(part), L.stream()), "ia_eikonal")
        ctx.save_for_backward(sdf_grad, valid)
        tot = part.sum(0)
        ctx.mark_non_differentiable(tot[1])
        return tot[0], tot[1]

    @staticmethod
    def backward(ctx, g_sum, _g_cnt):
        sdf_grad, valid = ctx.saved_tensors
        out = torch.empty_like(sdf_grad)
        w = g_sum.reshape(1).float().contiguous()
        L.check(L.lib().ia_eikonal_bwd(L.i64(sdf_grad.shape[0]), L.ptr(sdf_grad), L.ptr(valid), L.ptr(w), L.ptr(out), L.stream()),
                "ia_eikonal_bwd")
        return out, None


class _EikonalPartials(Function):
    """sdf_grad [n,3], valid [n] -> the per-workgroup partial sums [k,2] of ia_eikonal as they are (column 0: sum over valid of
    (|g| - 1)^2): train_phys._PhysLoss adds them up inside its own kernel.  The gradient that comes back is the SAME scalar for every
    partial (d loss / d sum), read from the first element."""

    @staticmethod
    def forward(ctx, sdf_grad, valid):
        n = sdf_grad.shape[0]
        sdf_grad = sdf_grad.contiguous()
        k = int(L.lib().ia_eikonal_partials(L.i64(n)))
        part = torch.empty((k, 2), device=sdf_grad.device) if k > 0 else torch.zeros((1, 2), device=sdf_grad.device)
        L.check(L.lib().ia_eikonal(L.i64(n), L.ptr(sdf_grad), L.ptr(valid), L.ptr(part), L.stream()), "ia_eikonal")
        ctx.save_for_backward(sdf_grad, valid)
        return part

    @staticmethod
    def backward(ctx, g_part):
        sdf_grad, valid = ctx.saved_tensors
        out = torch.empty_like(sdf_grad)
        w = g_part.as_strided((1,), (1,))                     # d loss / d (sum): one scalar, whatever the view it arrived in
        L.check(L.lib().ia_eikonal_bwd(L.i64(sdf_grad.shape[0]), L.ptr(sdf_grad), L.ptr(valid), L.ptr(w), L.ptr(out), L.stream()),
                "ia_eikonal_bwd")
        return out, None


def zeros_like_shapes(shapes, dev):
    """zero tensors of the given shapes as views of ONE zero-filled buffer (one fill launch instead of one per tensor; every view
    starts at a multiple of 64 floats)."""
    sizes = [int(math.prod(s)) for s in shapes]
    offs, tot = [], 0
    for n in sizes:
        offs.append(tot)
        tot += (n + 63) // 64 * 64
    buf = L.zeros(tot, dev)
    return [buf[o:o + n].view(s) for o, n, s in zip(offs, sizes, shapes)]


class _Alpha(Function):
    @staticmethod
    def forward(ctx, sdf, dists, beta):
        sdf, dists = sdf.contiguous(), dists.contiguous()
        b = beta.detach().reshape(1).float().contiguous()
        ctx.save_for_backward(sdf, dists, b)
        return render.laplace_alpha(sdf, dists, b)

    @staticmethod
    def backward(ctx, g):
        sdf, dists, b = ctx.saved_tensors
        g = g.contiguous()
        g_sdf = torch.empty_like(sdf)
        g_beta = L.zeros(1, sdf.device)
        L.check(L.lib().ia_laplace_alpha_bwd(L.i64(sdf.shape[0]), L.ptr(sdf), L.ptr(dists), L.f32(0.0), L.ptr(b), L.ptr(g),
                                             L.ptr(g_sdf), L.ptr(g_beta), L.stream()), "ia_laplace_alpha_bwd")
        return g_sdf, None, g_beta.reshape(())


class _Radiance(Function):
    """(x_cano, table2, feat, refl01, normal_world, weights...) -> rgb[n,3] (sigmoid)."""

    @staticmethod
    def forward(ctx, x, table, feat, refl01, normal_world, W1k, b1, W2, b2, W3, b3, center, scale, level_bits=0xFFFFFFFF):
        ctx.level_bits = int(level_bits)
        ctx.scale = scale
        xp = ((x - center) / scale + 0.5).contiguous()
        enc = fields.hashgrid_forward(xp, table)
        refl01 = refl01.contiguous()
        sh = fields.sh4(refl01)
        feat, normal_world = feat.contiguous(), normal_world.contiguous()
        segs = [(enc, 32, 1.0, 0.0), (xp, 3, 2.0, -1.0), (feat, 13, 1.0, 0.0), (sh, 16, 1.0, 0.0), (normal_world, 3, 1.0, 0.0)]
        ws = [t.contiguous() for t in (W1k, b1, W2, b2, W3, b3)]
        rgb = fields.mlp_forward(1, segs, *ws, 3)
        ctx.save_for_backward(xp, enc, feat, refl01, sh, normal_world, table, *ws)
        return rgb

    @staticmethod
    def backward(ctx, g_rgb):
        xp, enc, feat, refl01, sh, normal_world, table, W1k, b1, W2, b2, W3, b3 = ctx.saved_tensors
        n, dev = xp.shape[0], xp.device
        g_rgb = g_rgb.contiguous().float()
        g_x = torch.empty((n, 68), device=dev)
        segs = [(enc, 32, 1.0, 0.0), (xp, 3, 2.0, -1.0), (feat, 13, 1.0, 0.0), (sh, 16, 1.0, 0.0), (normal_world, 3, 1.0, 0.0)]
        ns, ptrs, strides, widths, muls, adds = _segs(segs)
        if FUSED_WGRAD:
            dW1, db1, dW2, db2, dW3, db3 = zeros_like_shapes([(64, 67), (64,), (64, 64), (64,), (3, 64), (3,)], dev)
            L.check(L.lib().ia_mlp_bwd_fused(L.i32(1), L.i64(n), L.i32(ns), ptrs, strides, widths, muls, adds, L.ptr(W1k),
                                             L.ptr(b1), L.ptr(W2), L.ptr(b2), L.ptr(W3), L.ptr(b3), L.ptr(g_rgb), L.ptr(g_x),
                                             L.i32(68), L.ptr(dW1), L.ptr(db1), L.ptr(dW2), L.ptr(db2), L.ptr(dW3), L.ptr(db3),
                                             L.stream()), "ia_mlp_bwd_fused")
        else:
            X = torch.empty((n, 68), device=dev)
            A1, A2, G1, G2 = (torch.empty((n, 64), device=dev) for _ in range(4))
            G3 = torch.empty((n, 16), device=dev)
            L.check(L.lib().ia_mlp_bwd(L.i32(1), L.i64(n), L.i32(ns), ptrs, strides, widths, muls, adds, L.ptr(W1k), L.ptr(b1),
                                       L.ptr(W2), L.ptr(b2), L.ptr(W3), L.ptr(b3), L.ptr(g_rgb), L.ptr(g_x), L.i32(68),
                                       L.ptr(X), L.ptr(A1), L.ptr(A2), L.ptr(G1), L.ptr(G2), L.ptr(G3), L.stream()), "ia_mlp_bwd")
            dW1, db1 = wgrad(G1, 64, X, 67)
            dW2, db2 = wgrad(G2, 64, A1, 64)
            dW3, db3 = wgrad(G3, 3, A2, 64)
        g_table = L.zeros_like(table)
        fields.hashgrid_backward(xp, g_x, g_table, level_mask=ctx.level_bits)      # columns 0..31, row stride 68
        g_feat = g_x[:, 35:48]
        g_nw = g_x[:, 64:67]
        g_sh = g_x[:, 48:64]
        g_refl01 = torch.empty((n, 3), device=dev)
        L.check(L.lib().ia_sh4_bwd(L.i64(n), L.ptr(refl01), C.c_void_p(g_sh.data_ptr()), L.i32(68), L.ptr(g_refl01),
                                   L.stream()), "ia_sh4_bwd")
        g_pos = None
        if ctx.needs_input_grad[0]:       # pose gradients: the hash Jacobian of grid #2 is recomputed here (not kept in forward)
            scale = ctx.scale
            _, jac2 = fields.hashgrid_forward(xp, table, with_jac=True)
            g_pos = (_jac_contract_T(jac2, g_x) + 2.0 * g_x[:, 32:35]) / scale
        return (g_pos, g_table, g_feat, g_refl01, g_nw, dW1, db1, dW2, db2, dW3, db3, None, None, None)


def curvature_laplace(geo, pts_cano: Tensor, grad_c: Tensor, rand_u: Tensor, eps: float = 1e-4) -> Tensor:
    """curvature term of VolumeSDF.forward(with_laplace=True) (models/rf/geometry.py:173-203, from PermutoSDF): angle / pi
    between the analytic normal at x and at x + eps * tangent, tangent = normal x normalize(rand_u).  rand_u [n,3] in
    [0,1) is an explicit input (torch.rand_like in the reference).  Both normals come out of _SDFField, so the loss
    back-propagates through the second-order path of ia_sdf_mlp_bwd_fused / ia_hashgrid_bwd like the eikonal term.
    Deviation: the dependence of the probe position x + eps * tangent on the parameters (an eps-scaled term that needs
    d^2 enc / d x^2, which tiny-cuda-nn does not provide either) is dropped: the probe point is a constant."""
    W1k, b1, W2, b2 = geo.effective_weights()
    nrm = torch.nn.functional.normalize
    with torch.no_grad():
        tangent = torch.cross(nrm(grad_c, dim=-1, eps=1e-6), nrm(rand_u, dim=-1, eps=1e-6), dim=-1)
        x_d = (pts_cano + eps * tangent).contiguous()
    _, grad_d = _SDFField.apply(x_d, geo.grid_params, W1k, b1, W2, b2, geo.center, geo.scale, geo.prog.level_bits(),
                                geo.inv_scale_host())
    dot = (nrm(grad_c, dim=-1, eps=1e-6) * nrm(grad_d, dim=-1, eps=1e-6)).sum(-1)
    return torch.acos(dot.clamp(-1.0 + 1e-6, 1.0 - 1e-6)) / math.pi


def shade_differentiable(rs, rays_o: Tensor, rays_d: Tensor, ray_indices: Tensor, t_starts: Tensor, t_ends: Tensor,
                         packed_info: Tensor, curv_u: Optional[Tensor] = None) -> Dict[str, Tensor]:
    """differentiable rgb_normal_alpha_fn + rendering_with_normals_sdf for the samples found by the no-grad pass.
    curv_u [n_samples,3]: uniforms for the curvature probe directions (enables out["sdf_laplace"])."""
    dfm, geo, rad = rs.deformer, rs.geometry, rs.radiance
    n_rays = packed_info.shape[0]
    pts = render.ray_points(rays_o, rays_d, ray_indices, t_starts, t_ends)
    pose_grad = dfm.tfs.requires_grad and torch.is_grad_enabled()
    with torch.no_grad():
        d = dfm.deform(pts, geo, with_grad=False, with_feature=False, want_fwd=True, want_jinv=pose_grad)      # candidate search + winner selection (the gradient is evaluated once, on the winners, by _SDFField)
        valid = d["valid"]
        win = c2w = None         # win: flat (point, init) index of each sample's winning candidate (pose mode), or a flag
        if d["n_candidates"] > 0:
            win = True
            if pose_grad:
                win = d["cand_src"].long()[d["sel"].long().clamp(min=0)]
                c2w = d["fwd_J"].reshape(-1, 3, 3)[win]
    pts_cano = d["pts_cano"]
    if pose_grad and win is not None:
        # pose_correction / SMPL parameters are being optimised (configs/config.yaml: pose_correction.enable_pose_correction):
        # the roots get their implicit-function derivative and the normal push-forward its blended-rotation derivative;
        # the forward VALUES stay the ones the search kernels produced.
        J_inv_win = d["J_inv"].reshape(-1, 3, 3)[win]
        pts_cano, R = dfm.implicit_pose_terms(pts_cano, J_inv_win, valid)
        c2w = c2w + (R - R.detach()) * valid[:, None, None].to(R.dtype)
    W1k, b1, W2, b2 = geo.effective_weights()
    out, grad_c = _SDFField.apply(pts_cano, geo.grid_params, W1k, b1, W2, b2, geo.center, geo.scale, geo.prog.level_bits(),
                                  geo.inv_scale_host())
    # invalid points: sdf 1e5, feature 0, gradient [0,0,1] (snarf_deformer.py:192-231)
    if pose_grad and win is not None:          # c2w carries the pose graph: plain torch expressions
        vf = valid[:, None].float()
        dflt_g = torch.tensor([0.0, 0.0, 1.0], device=pts.device)
        feat = out * vf
        sdf = torch.where(valid, out[:, 0], torch.full_like(out[:, 0], 1e5))
        sdf_grad = torch.where(valid[:, None], (c2w * grad_c[:, None, :]).sum(-1), dflt_g[None])
    else:
        feat, sdf, sdf_grad, c2w = _SelectPush.apply(out, grad_c, valid, d["fwd_J"] if win is not None else None,
                                                     d["cand_src"], d["sel"])
    w2s_rot = dfm.w2s[:3, :3].contiguous()
    normal_smpl, normal_world, refl01 = _ShadePrep.apply(sdf_grad, rays_d, ray_indices, w2s_rot)
    alphas = _Alpha.apply(sdf, t_ends - t_starts, rs.density.get_beta())
    rgbs = _Radiance.apply(pts_cano, rad.grid_params, feat, refl01, normal_world, *rad.effective_weights(),
                           rad.center, rad.scale, rad.prog.level_bits())
    weights, trans = nerfacc._WeightFromAlpha.apply(alphas, packed_info)
    acc = lambda v: nerfacc._Accumulate.apply(weights, v, ray_indices, packed_info)      # noqa: E731
    res = dict(comp_rgb=acc(rgbs), comp_normal=acc(normal_world), opacity=acc(None),
               depth=acc(((t_starts + t_ends) / 2.0)[:, None]), weights=weights, alphas=alphas, rgbs=rgbs, sdf=sdf,
               sdf_grad=sdf_grad, valid=valid, n_samples=pts.shape[0], pts_cano=d["pts_cano"], c2w=c2w)
    if pose_grad and win is not None:
        res["J_inv"] = J_inv_win
    if curv_u is not None:       # laplace = 0 for points without a valid candidate (snarf_deformer.py:233-235)
        res["sdf_laplace"] = curvature_laplace(geo, d["pts_cano"], grad_c, curv_u) * valid.float()
    return res


def training_loss(out: Dict[str, Tensor], target_rgb: Tensor, target_mask: Optional[Tensor] = None,
                  lambda_eik: float = 0.1, lambda_mask: float = 0.1, lambda_curv: float = 0.0,
                  eik_denominator: Optional[float] = None) -> Tensor:
    """the rgb / eikonal / mask terms of systems/intrinsic_avatar.py:167-251 (L1 rgb, (|grad|-1)^2, BCE opacity)."""
    loss = (out["comp_rgb"] - target_rgb).abs().mean()
    # eikonal term: the reference takes .mean() over ALL samples of out["sdf_grad_samples"]
    # (systems/intrinsic_avatar.py:235-237); samples without a valid root carry the default gradient [0,0,1]
    # (snarf_deformer.py:233-235), i.e. they add 0 to the sum but count in the denominator.  Written as a masked sum over
    # the valid samples (no boolean-mask indexing: that costs a host sync forward and a 1.3 ms index_put in backward for
    # 4.4 M samples) divided by the TOTAL sample count -- `eik_denominator` overrides it with a global count under ray
    # sharding (sum of shard gradients == full-batch gradient).
    eik_sum, _n_valid = _Eikonal.apply(out["sdf_grad"], out["valid"])
    denom = float(eik_denominator) if eik_denominator is not None else float(max(out["sdf_grad"].shape[0], 1))
    loss = loss + lambda_eik * eik_sum / denom
    if target_mask is not None:
        op = out["opacity"][:, 0].clamp(1e-3, 1 - 1e-3)
        loss = loss + lambda_mask * torch.nn.functional.binary_cross_entropy(op, target_mask)
    if lambda_curv > 0.0 and "sdf_laplace" in out:            # systems/intrinsic_avatar.py:265-268
        loss = loss + lambda_curv * out["sdf_laplace"].abs().mean()
    return loss
