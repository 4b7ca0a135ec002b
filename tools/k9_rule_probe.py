#!/usr/bin/env python3
"""Design tool for a K9-CONSISTENT early filter of the Broyden search.

The exact search (fuse_cuda_kernel_fast.cu:252-452) is emulated vectorised in torch with the whole trajectory of every
(point, init) item kept; retirement rules are then evaluated offline, because retiring one search never changes the trajectory
of another.  The rule family (one lane per point, inits in REVERSE order, as broyden_spec_kernel):

  before fetch k of init i (at x_k; k = 0 is the start point) the search is RETIRED when a recorded root r of a later init
  has  |x_k - r|_inf < eps,  r is TIGHT (Frobenius norm of Broyden's final J_inv at r <= tau: every search that stops inside
  {|g| < cvg} around the same true root then ends within ~2 cvg tau of r),  and -- for k >= 1 -- the step that led to x_k
  is short,  |x_k - x_{k-1}|_inf < kappa eps  (the search is settling there, not flying by);  k = 0 retirement is a switch.
  A search that COMPLETES valid is dropped when it lies within K9's 1e-4 (L2) of a recorded root, recorded when it is farther
  than 2e-4 from all of them, and otherwise -- or when the row's three slots are full -- the POINT is flagged and redone by
  the exact search (cost: all its fetches again).

Per rule: fetches per point (flagged points pay twice), flagged fraction, fraction of points whose candidate SET differs from
K9 after the exact search (flagged points count as exact), fraction that lose a distinct root (> 1 mm from every kept one).

  python tools/k9_rule_probe.py            # GPU: the secondary-march points of the headline frame (IA_NSEC rays)
  python tools/k9_rule_probe.py --cpu N    # no GPU: N points of tools/cluster_emul.py's synthetic scene
"""
import itertools
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ITERS = 10
CVG, DVG = 1e-5, 1e-1


def sample_J(vJ, g):
    """trilinear, align_corners, zero padding.  vJ [D,H,W,12], g [n,3] in [-1,1] -> [n,12]"""
    D, H, W, _ = vJ.shape
    ix = (g[:, 0] + 1) / 2 * (W - 1)
    iy = (g[:, 1] + 1) / 2 * (H - 1)
    iz = (g[:, 2] + 1) / 2 * (D - 1)
    def clean(v):
        return torch.where(torch.isfinite(v) & (v.abs() <= 2147483648.0), v, torch.full_like(v, -100.0))
    ix, iy, iz = clean(ix), clean(iy), clean(iz)
    fx, fy, fz = ix.floor(), iy.floor(), iz.floor()
    x0, y0, z0 = fx.long(), fy.long(), fz.long()
    out = torch.zeros((g.shape[0], 12), dtype=torch.float32, device=g.device)
    flat = vJ.reshape(-1, 12)
    for c in range(8):
        cx, cy, cz = c & 1, (c >> 1) & 1, (c >> 2) & 1
        xx, yy, zz = x0 + cx, y0 + cy, z0 + cz
        wgt = ((ix - fx) if cx else (fx + 1 - ix)) * ((iy - fy) if cy else (fy + 1 - iy)) * ((iz - fz) if cz else (fz + 1 - iz))
        ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H) & (zz >= 0) & (zz < D)
        lin = (zz.clamp(0, D - 1) * H + yy.clamp(0, H - 1)) * W + xx.clamp(0, W - 1)
        out += torch.where(ok[:, None], flat[lin] * wgt[:, None], torch.zeros((), device=g.device))
    return out


def true_jinv_norm(vJ, x, offk, sck, with_det=False):
    """Frobenius norm of the inverse of the TRUE Jacobian of g(x) = A(x) x + b(x) - xd at x [n,3]  ([A | b] = the trilinear voxel_J):
    dg_i/dx_a = A_ia + sum_c dw_c/dx_a (A_c x + b_c)_i -- the weight-gradient term Broyden's estimate only learns along its own steps."""
    D, H, W, _ = vJ.shape
    g = (x + offk) * sck
    ix = (g[:, 0] + 1) / 2 * (W - 1)
    iy = (g[:, 1] + 1) / 2 * (H - 1)
    iz = (g[:, 2] + 1) / 2 * (D - 1)
    dcoord = torch.stack([sck[0] * (W - 1) / 2, sck[1] * (H - 1) / 2, sck[2] * (D - 1) / 2]) if sck.dim() == 1 else None
    sc3 = sck.reshape(-1)[:3]
    dco = torch.stack([sc3[0] * (W - 1) / 2, sc3[1] * (H - 1) / 2, sc3[2] * (D - 1) / 2])
    fx, fy, fz = ix.floor(), iy.floor(), iz.floor()
    x0, y0, z0 = fx.long(), fy.long(), fz.long()
    flat = vJ.reshape(-1, 12)
    n = x.shape[0]
    A = torch.zeros((n, 3, 3), device=x.device)
    Gd = torch.zeros((n, 3, 3), device=x.device)               # sum_c dw_c/dx_a p_c  -> [n, i, a]
    tx, ty, tz = ix - fx, iy - fy, iz - fz
    for c in range(8):
        cx, cy, cz = c & 1, (c >> 1) & 1, (c >> 2) & 1
        xx, yy, zz = x0 + cx, y0 + cy, z0 + cz
        wx, wy, wz = (tx if cx else 1 - tx), (ty if cy else 1 - ty), (tz if cz else 1 - tz)
        sx, sy, sz = (1.0 if cx else -1.0), (1.0 if cy else -1.0), (1.0 if cz else -1.0)
        ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H) & (zz >= 0) & (zz < D)
        lin = (zz.clamp(0, D - 1) * H + yy.clamp(0, H - 1)) * W + xx.clamp(0, W - 1)
        v = torch.where(ok[:, None], flat[lin], torch.zeros((), device=x.device)).reshape(n, 3, 4)
        p_c = torch.einsum("nij,nj->ni", v[:, :, :3], x) + v[:, :, 3]
        A += v[:, :, :3] * (wx * wy * wz)[:, None, None]
        dw = torch.stack([sx * wy * wz * dco[0], wx * sy * wz * dco[1], wx * wy * sz * dco[2]], -1)      # [n, a]
        Gd += p_c[:, :, None] * dw[:, None, :]
    J = A + Gd
    det = torch.linalg.det(J)
    Jinv = torch.linalg.inv(torch.where((det.abs() > 1e-12)[:, None, None], J, torch.eye(3, device=x.device).expand(n, 3, 3) * 1e-6))
    if with_det:
        return Jinv.reshape(n, 9).norm(dim=-1), det
    return Jinv.reshape(n, 9).norm(dim=-1)


def cell_table(vJ, offk, sck, m=3):
    """per voxel cell (iz, iy, ix), ix in [0, W - 2] ...: max over m^3 sample points inside the cell of |J_true^-1|_F, +inf where det J changes
    sign among the samples (a fold of the skinning map crosses the cell) -> [D-1, H-1, W-1].  What a precompute pass would store per pose."""
    D, H, W, _ = vJ.shape
    dev = vJ.device
    sc3, of3 = sck.reshape(-1)[:3], offk.reshape(-1)[:3]
    fr = (torch.arange(m, device=dev, dtype=torch.float32) + 0.5) / m if m > 1 else torch.tensor([0.5], device=dev)
    if m > 1:
        fr = torch.linspace(0.02, 0.98, m, device=dev)
    out = torch.zeros((D - 1, H - 1, W - 1), device=dev)
    sign_min = torch.full((D - 1, H - 1, W - 1), 1e30, device=dev)
    sign_max = torch.full((D - 1, H - 1, W - 1), -1e30, device=dev)
    iz, iy, ix = torch.meshgrid(torch.arange(D - 1, device=dev), torch.arange(H - 1, device=dev), torch.arange(W - 1, device=dev), indexing="ij")
    for a in fr.tolist():
        for b in fr.tolist():
            for c in fr.tolist():
                gx = (ix.float() + c) / (W - 1) * 2 - 1
                gy = (iy.float() + b) / (H - 1) * 2 - 1
                gz = (iz.float() + a) / (D - 1) * 2 - 1
                g = torch.stack([gx, gy, gz], -1).reshape(-1, 3)
                x = g / sc3 - of3
                nrm, det = true_jinv_norm(vJ, x, offk, sck, with_det=True)
                out = torch.maximum(out, nrm.reshape(out.shape))
                sign_min = torch.minimum(sign_min, det.reshape(out.shape))
                sign_max = torch.maximum(sign_max, det.reshape(out.shape))
    out = torch.where((sign_min > 0) | (sign_max < 0), out, torch.full_like(out, float("inf")))
    cell_table.sign = torch.where(sign_min > 0, 1, torch.where(sign_max < 0, -1, 0)).to(torch.int8)
    return out


def dilate_table(ctab, sign, tau):
    """tight3 [D-1,H-1,W-1] bool: the cell AND its 26 neighbours are tight (<= tau) with ONE sign of det (cells outside the grid count as
    not tight): the skinning map is coherently oriented and steep on the whole neighbourhood, so no second root exists within a cell width."""
    import torch.nn.functional as F
    t = (ctab <= tau)
    pos = (t & (sign > 0)).float()[None, None]
    neg = (t & (sign < 0)).float()[None, None]
    allpos = -F.max_pool3d(-F.pad(pos, (1, 1, 1, 1, 1, 1), value=0.0), 3, 1) > 0.5
    allneg = -F.max_pool3d(-F.pad(neg, (1, 1, 1, 1, 1, 1), value=0.0), 3, 1) > 0.5
    return (allpos | allneg)[0, 0]


def search(xd, vJ, tfs, bones, offk, sck):
    """all inits of all points -> traj [P,I,ITERS+1,3] (x_k = position of fetch k), nfetch [P,I], valid [P,I], xfin [P,I,3],
    jn [P,I] (Frobenius norm of J_inv at the end)."""
    P, I = xd.shape[0], bones.shape[0]
    dev = xd.device
    T = tfs[bones.long()]                                              # [I,4,4]
    xt = xd[:, None, :].expand(P, I, 3).reshape(-1, 3)
    Tt = T[None].expand(P, I, 4, 4).reshape(-1, 4, 4)
    x = torch.einsum("nji,nj->ni", Tt[:, :3, :3], xt - Tt[:, :3, 3])
    n = x.shape[0]
    traj = torch.full((n, ITERS + 1, 3), float("nan"), device=dev)
    traj[:, 0] = x
    jtraj = torch.zeros((n, ITERS + 1), device=dev)          # |J_inv|_F the step to x_k was made with (k >= 1)
    gtraj = torch.zeros((n, ITERS + 1), device=dev)          # |g(x_k)|_2
    Jl = sample_J(vJ, (x + offk) * sck).reshape(n, 3, 4)
    Ji = Jl[:, :, :3].transpose(1, 2).contiguous()
    g = torch.einsum("nij,nj->ni", Jl[:, :, :3], x) + Jl[:, :, 3] - xt
    gtraj[:, 0] = g.norm(dim=-1)
    live = torch.ones(n, dtype=torch.bool, device=dev)
    nfetch = torch.ones(n, dtype=torch.int32, device=dev)
    valid = torch.zeros(n, dtype=torch.bool, device=dev)
    xfin = torch.zeros((n, 3), device=dev)
    for it in range(ITERS):
        idx = torch.nonzero(live)[:, 0]
        if idx.numel() == 0:
            break
        u = -torch.einsum("nij,nj->ni", Ji[idx], g[idx])
        jtraj[idx, it + 1] = Ji[idx].reshape(-1, 9).norm(dim=-1)
        xn = x[idx] + u
        gg = (xn + offk) * sck
        Jl = sample_J(vJ, gg).reshape(-1, 3, 4)
        gn = torch.einsum("nij,nj->ni", Jl[:, :, :3], xn) + Jl[:, :, 3] - xt[idx]
        nrm = (gn * gn).sum(-1)
        x[idx] = xn
        traj[idx, it + 1] = xn
        gtraj[idx, it + 1] = nrm.sqrt()
        nfetch[idx] += 1
        conv = nrm < CVG * CVG
        div = ~conv & ~(nrm <= DVG * DVG)
        inbox = (gg.abs() <= 1).all(-1)
        valid[idx[conv & inbox]] = True
        xfin[idx[conv]] = xn[conv]
        cont = ~conv & ~div
        ci = idx[cont]
        dx, dg = u[cont], gn[cont] - g[ci]
        Jc = Ji[ci]
        c_ = torch.einsum("nji,nj->ni", Jc, dx)
        s = (c_ * dg).sum(-1, keepdim=True)
        r = -torch.einsum("nij,nj->ni", Jc, dg) + dx
        Ji[ci] = Jc + r[:, :, None] * c_[:, None, :] / s[:, :, None]
        g[ci] = gn[cont]
        live[idx[~cont]] = False
    jn = Ji.reshape(n, 9).norm(dim=-1)
    search.gtraj = gtraj.reshape(P, I, ITERS + 1)            # (side channel: keeps the 6-tuple of the callers)
    return traj.reshape(P, I, ITERS + 1, 3), nfetch.reshape(P, I), valid.reshape(P, I), xfin.reshape(P, I, 3), jn.reshape(P, I), jtraj.reshape(P, I, ITERS + 1)


def k9(x, valid):
    P, I = valid.shape
    keep = valid.clone()
    for i in range(I):
        for j in range(i + 1, I):
            d = ((x[:, i] - x[:, j]) ** 2).sum(-1)
            keep[:, i] &= ~(valid[:, j] & (d < 1e-8))
    return keep


def cell_id(x, offk, sck, dims):
    """voxel cell of canonical point x (the trilinear fetch's floor indices), packed into one integer."""
    D, H, W = dims
    g = (x + offk) * sck
    ix = ((g[..., 0] + 1) / 2 * (W - 1)).floor().clamp(-2, W + 1).long() + 2
    iy = ((g[..., 1] + 1) / 2 * (H - 1)).floor().clamp(-2, H + 1).long() + 2
    iz = ((g[..., 2] + 1) / 2 * (D - 1)).floor().clamp(-2, D + 1).long() + 2
    return (iz * 1024 + iy) * 1024 + ix


def evaluate(traj, nfetch, valid, xfin, jn, keep_exact, eps, tau, kappa, k0, samecell=0, cells=None, tau2=1e9, jtraj=None, slots=3, zone=2e-4, examples=None,
             steep=0.0, gtraj=None, tautrue=0.0, jn_true=None, taucell=0.0, jn_cell=None, jn_free3=None):
    """-> dict of SUMS over the chunk's points."""
    P, I = valid.shape
    dev = valid.device
    roots = torch.zeros((P, slots, 3), device=dev)
    rcell = torch.zeros((P, slots), dtype=torch.long, device=dev)
    tight = torch.zeros((P, slots), dtype=torch.bool, device=dev)
    free3 = torch.zeros((P, slots), dtype=torch.bool, device=dev)
    n_roots = torch.zeros(P, dtype=torch.long, device=dev)
    flagged = torch.zeros(P, dtype=torch.bool, device=dev)
    keep = torch.zeros((P, I), dtype=torch.bool, device=dev)
    fetches = torch.zeros(P, dtype=torch.long, device=dev)
    ks = torch.arange(ITERS + 1, device=dev)
    slot_ids = torch.arange(slots, device=dev)
    for i in range(I - 1, -1, -1):
        xk = traj[:, i]                                                # [P,K,3]
        nf = nfetch[:, i].long()
        have = slot_ids[None, :] < n_roots[:, None]                    # [P,S]
        d = (xk[:, :, None, :] - roots[:, None, :, :]).abs().amax(-1)  # [P,K,S]
        okr = (d < eps) & (have & tight)[:, None, :]
        if samecell == 1:
            okr &= cells[0][:, i, :, None] == rcell[:, None, :]
        elif samecell == 2:              # hybrid: no cut for a root whose 27-cell neighbourhood is tight with one orientation (free3)
            okr &= (cells[0][:, i, :, None] == rcell[:, None, :]) | free3[:, None, :]
        if steep > 0:
            # g must be as steep around r as a tight root promises, seen from the last point whose g is known: |g(x_{k-1})| >= steep |x_{k-1} - r|
            # (a flat valley -- a fold of the skinning map -- holds two roots 1e-4 ... 1e-3 apart that both look tight to Broyden's estimate)
            dprev = (xk[:, :-1, None, :] - roots[:, None, :, :]).norm(dim=-1)          # [P,K-1,S]
            okp = gtraj[:, i, :-1, None] >= steep * dprev
            okr[:, 1:] &= okp
        near = okr.any(-1)                                             # [P,K]
        if jtraj is not None:
            near &= jtraj[:, i] <= tau2
        step = torch.full((P, ITERS + 1), 0.0, device=dev)
        step[:, 1:] = (xk[:, 1:] - xk[:, :-1]).abs().amax(-1)
        cond = near & (step < kappa * eps) & (ks[None, :] < nf[:, None])
        if not k0:
            cond[:, 0] = False
        ret_k = torch.where(cond.any(-1), cond.float().argmax(-1), torch.full((P,), -1, device=dev))
        retired = ret_k >= 0
        fetches += torch.where(retired, ret_k, nf)
        done = ~retired & valid[:, i]
        xf = xfin[:, i]
        d2 = ((xf[:, None, :] - roots) ** 2).sum(-1)                   # [P,S]
        d2 = torch.where(have, d2, torch.full_like(d2, 1e30))
        dmin = d2.amin(-1)
        dropped = done & (dmin < 1e-8)
        inzone = done & ~dropped & (dmin < zone * zone)
        rec = done & ~dropped & ~inzone
        full = rec & (n_roots >= slots)
        flagged |= inzone | full
        rec &= ~full
        keep[:, i] = rec
        pi = torch.nonzero(rec)[:, 0]
        roots[pi, n_roots[pi]] = xf[pi]
        tight[pi, n_roots[pi]] = (jn[pi, i] <= tau) & ((jn_true[pi, i] <= tautrue) if tautrue > 0 else True) & \
            ((jn_cell[pi, i] <= taucell) if taucell > 0 else True)
        if samecell:
            rcell[pi, n_roots[pi]] = cells[1][pi, i]
        if samecell == 2:
            free3[pi, n_roots[pi]] = jn_free3[pi, i]
        n_roots[pi] += 1
    total_exact = nfetch.long().sum(-1)
    fetches = torch.where(flagged, fetches + total_exact, fetches)
    keep = torch.where(flagged[:, None], keep_exact, keep)
    mism = (keep != keep_exact).any(-1)
    lost = torch.zeros(P, dtype=torch.bool, device=dev)
    xs1 = torch.where(keep[..., None], xfin, torch.full_like(xfin, -1e9))
    for i in range(I):
        dmin = (xfin[:, i:i + 1, :] - xs1).abs().amax(-1).amin(-1)
        lost |= keep_exact[:, i] & (dmin > 1e-3)
    extra = (keep & ~keep_exact).any(-1)
    if examples is not None:
        for p_ in torch.nonzero(mism)[:, 0][:8].tolist():
            examples.append(dict(valid=valid[p_].int().tolist(), keep_exact=keep_exact[p_].int().tolist(), keep=keep[p_].int().tolist(),
                                 jn=[round(v, 3) for v in jn[p_].tolist()], nfetch=nfetch[p_].tolist(),
                                 x=[[round(c, 6) for c in row] for row in xfin[p_].tolist()]))
    return dict(fetches=int(fetches.sum()), flagged=int(flagged.sum()), mismatch=int(mism.sum()), lost=int(lost.sum()), extra=int(extra.sum()))


def rules():
    out = []
    if os.environ.get("IA_RULES") == "hybrid":
        out.append(dict(eps=1e-3, tau=2.5, kappa=1e9, k0=1, samecell=1, tau2=3.0, taucell=2.5))
        for tau3 in (2.5, 2.0):
            out.append(dict(eps=1e-3, tau=2.5, kappa=1e9, k0=1, samecell=2, tau2=3.0, taucell=2.5, tau3=tau3))
        for eps in (1.5e-3, 2e-3, 3e-3):
            out.append(dict(eps=eps, tau=2.5, kappa=1e9, k0=1, samecell=2, tau2=3.0, taucell=2.5, tau3=2.5))
            out.append(dict(eps=eps, tau=2.0, kappa=1e9, k0=1, samecell=2, tau2=3.0, taucell=2.0, tau3=2.0))
        return out
    if os.environ.get("IA_RULES") == "nocell":
        out.append(dict(eps=1e-3, tau=2.5, kappa=1e9, k0=1, samecell=1, tau2=3.0, taucell=2.5))
        for eps, tau2, taucell, dil in itertools.product((1e-3, 2e-3), (3.0, 1e9), (2.5, 2.0), (1, 0)):
            out.append(dict(eps=eps, tau=2.5, kappa=1e9, k0=1, samecell=0, tau2=tau2, taucell=taucell, dilate=dil))
        return out
    if os.environ.get("IA_RULES") == "eps":
        for eps, tau2, taucell in itertools.product((1e-3, 2e-3, 5e-3, 2e-2), (3.0, 1e9), (2.5, 2.0)):
            out.append(dict(eps=eps, tau=2.5, kappa=1e9, k0=1, samecell=1, tau2=tau2, taucell=taucell))
        return out
    if os.environ.get("IA_RULES") == "cell":
        out.append(dict(eps=1e-3, tau=2.5, kappa=1e9, k0=1, samecell=1, tau2=3.0))
        for taucell, tau in itertools.product((2.0, 2.25, 2.5, 3.0), (2.5, 1e9)):
            out.append(dict(eps=1e-3, tau=tau, kappa=1e9, k0=1, samecell=1, tau2=3.0, taucell=taucell))
        return out
    if os.environ.get("IA_RULES") == "true":
        for tautrue, tau in itertools.product((0.0, 2.5, 3.0, 4.0, 6.0), (2.5, 1e9)):
            out.append(dict(eps=1e-3, tau=tau, kappa=1e9, k0=1, samecell=1, tau2=3.0, tautrue=tautrue))
        return out
    if os.environ.get("IA_RULES") == "steep":
        for steep, k0, tau, tau2 in itertools.product((0.0, 0.1, 0.2, 0.3, 0.5), (1, 0), (2.5,), (3.0,)):
            out.append(dict(eps=1e-3, tau=tau, kappa=1e9, k0=k0, samecell=1, tau2=tau2, steep=steep))
        return out
    for eps, tau, sc, tau2 in itertools.product((1e-3, 2e-3), (2.2, 2.5), (0, 1), (3.0, 6.0, 1e9)):
        out.append(dict(eps=eps, tau=tau, kappa=1e9, k0=1, samecell=sc, tau2=tau2))
    return out


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--cpu":
        import cluster_emul as CE                                     # tools/ is on sys.path when run as a script
        n = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
        vJ, tfs, offk, sck, wgrid, rig = CE.scene()
        xd = torch.from_numpy(CE.points(n, vJ, tfs, wgrid, offk, sck, rig))
        from intrinsicavatar_amd import synthetic as S
        vJ, tfs, offk, sck = torch.from_numpy(vJ), torch.from_numpy(tfs), torch.from_numpy(offk), torch.from_numpy(sck)
        bones = torch.tensor(S.INIT_BONES)
        chunk = 100000
    else:
        import spec_search_probe as SP
        from intrinsicavatar_amd import synthetic as S
        n_sec = int(os.environ.get("IA_NSEC", str(1 << 21)))
        rs, rays, _ = S.build_frame(SP.dev, 540, 540, pose_seed=0, beta=0.01, pose=os.environ.get("IA_POSE") or None)
        xd = SP.march_points(rs, rays, n_sec)
        dfm = rs.deformer
        vJ, tfs, offk, sck, bones = dfm.voxel_J_cl[0], dfm.tfs[0], dfm.offset_kernel, dfm.scale_kernel, dfm.init_bones
        chunk = 1 << 20
    P = xd.shape[0]
    rl = rules()
    acc = [dict(fetches=0, flagged=0, mismatch=0, lost=0, extra=0) for _ in rl]
    tot = dict(points=P, fetches_exact=0, survivors=0, valid=0, jn_hist=None, dup_dist_hist=None)
    jn_edges = torch.tensor([0, 1.5, 1.8, 2.0, 2.5, 3.0, 4.0, 6.0, 10.0, 1e30])
    dd_edges = torch.tensor([0, 1e-5, 2e-5, 5e-5, 1e-4, 2e-4, 5e-4, 1e-3, 1e30])
    jn_hist = torch.zeros(len(jn_edges) - 1, dtype=torch.long)
    dd_hist = torch.zeros((len(jn_edges) - 1, len(dd_edges) - 1), dtype=torch.long)
    examples = []
    ctab = None
    dtabs = {}
    for c0 in range(0, P, chunk):
        x = xd[c0:c0 + chunk]
        traj, nfetch, valid, xfin, jn, jtraj = search(x, vJ, tfs, bones, offk, sck)
        keep_exact = k9(xfin, valid)
        cells = (cell_id(traj, offk, sck, vJ.shape[:3]), cell_id(xfin, offk, sck, vJ.shape[:3]))
        jn_cell = None
        if any(r.get("taucell", 0) > 0 for r in rl):
            if ctab is None:
                ctab = cell_table(vJ, offk, sck, int(os.environ.get("IA_CELL_M", "3")))
                print("# cell table: finite", float(torch.isfinite(ctab).float().mean()), "<=2.5", float((ctab <= 2.5).float().mean()), file=sys.stderr)
            D_, H_, W_ = vJ.shape[:3]
            gq = (xfin.reshape(-1, 3) + offk) * sck
            cx = ((gq[:, 0] + 1) / 2 * (W_ - 1)).floor().long()
            cy = ((gq[:, 1] + 1) / 2 * (H_ - 1)).floor().long()
            cz = ((gq[:, 2] + 1) / 2 * (D_ - 1)).floor().long()
            inside = (cx >= 0) & (cx < W_ - 1) & (cy >= 0) & (cy < H_ - 1) & (cz >= 0) & (cz < D_ - 1)
            v = ctab[cz.clamp(0, D_ - 2), cy.clamp(0, H_ - 2), cx.clamp(0, W_ - 2)]
            jn_cell = torch.where(inside, v, torch.full_like(v, float("inf"))).reshape(xfin.shape[:2])
            jn_cell_dil = {}
            jn_free3_by = {}
            for t3 in sorted({r["tau3"] for r in rl if r.get("tau3")}):
                if ("f", t3) not in dtabs:
                    dtabs[("f", t3)] = dilate_table(ctab, cell_table.sign, t3)
                    print("# free3 table tau3", t3, float(dtabs[("f", t3)].float().mean()), file=sys.stderr)
                vf = dtabs[("f", t3)][cz.clamp(0, D_ - 2), cy.clamp(0, H_ - 2), cx.clamp(0, W_ - 2)]
                jn_free3_by[t3] = (inside & vf).reshape(xfin.shape[:2])
            for tc in sorted({r.get("taucell", 0) for r in rl if r.get("dilate")}):
                if tc not in dtabs:
                    dtabs[tc] = dilate_table(ctab, cell_table.sign, tc)
                    print("# dilated table tau", tc, "tight3", float(dtabs[tc].float().mean()), "tight", float((ctab <= tc).float().mean()), file=sys.stderr)
                vd = dtabs[tc][cz.clamp(0, D_ - 2), cy.clamp(0, H_ - 2), cx.clamp(0, W_ - 2)]
                jn_cell_dil[tc] = torch.where(inside & vd, torch.zeros_like(v), torch.full_like(v, float("inf"))).reshape(xfin.shape[:2])
        jn_true = None
        if any(r.get("tautrue", 0) > 0 for r in rl):
            jn_true = true_jinv_norm(vJ, xfin.reshape(-1, 3), offk, sck).reshape(xfin.shape[:2])
        tot["fetches_exact"] += int(nfetch.long().sum())
        tot["survivors"] += int(keep_exact.sum())
        tot["valid"] += int(valid.sum())
        # diagnostics: distance of a dropped duplicate to the nearest LATER valid root, by the norm of that root's J_inv
        I = valid.shape[1]
        for i in range(I):
            best = torch.full((x.shape[0],), 1e30, device=x.device)
            bjn = torch.zeros(x.shape[0], device=x.device)
            for j in range(i + 1, I):
                d = (xfin[:, i] - xfin[:, j]).norm(dim=-1)
                d = torch.where(valid[:, j], d, torch.full_like(d, 1e30))
                bjn = torch.where(d < best, jn[:, j], bjn)
                best = torch.minimum(best, d)
            sel = valid[:, i] & (best < 1e-2)
            a = torch.bucketize(bjn[sel].cpu(), jn_edges[1:-1])
            b = torch.bucketize(best[sel].cpu(), dd_edges[1:-1])
            dd_hist.view(-1).index_add_(0, a * (len(dd_edges) - 1) + b, torch.ones_like(a))
        jn_hist += torch.histc(torch.bucketize(jn[keep_exact].cpu(), jn_edges[1:-1]).float(), bins=len(jn_edges) - 1, min=0, max=len(jn_edges) - 1).long()
        for r, a in zip(rl, acc):
            e = evaluate(traj, nfetch, valid, xfin, jn, keep_exact, r["eps"], r["tau"], r["kappa"], r["k0"], r["samecell"], cells, r["tau2"], jtraj,
                         examples=(examples if (r["eps"] == 1e-3 and r["tau"] == 2.5 and r["samecell"] == 1 and r["tau2"] == 3.0 and not r.get("steep")) else None),
                         steep=r.get("steep", 0.0), gtraj=search.gtraj, tautrue=r.get("tautrue", 0.0), jn_true=jn_true,
                         taucell=r.get("taucell", 0.0), jn_cell=(jn_cell_dil[r["taucell"]] if r.get("dilate") else jn_cell),
                         jn_free3=(jn_free3_by[r["tau3"]] if r.get("tau3") else None))
            for k in a:
                a[k] += e[k]
        del traj, jtraj
        search.gtraj = None
    res = dict(points=P, fetches_per_point_exact=tot["fetches_exact"] / P, survivors_per_point=tot["survivors"] / P,
               valid_per_point=tot["valid"] / P, jn_edges=jn_edges.tolist(), jn_hist_survivors=jn_hist.tolist(),
               dup_dist_edges=dd_edges.tolist(), dup_dist_hist_by_jn=dd_hist.tolist(), rules=[], examples=examples[:40])
    for r, a in zip(rl, acc):
        res["rules"].append(dict(**r, fetches_per_point=a["fetches"] / P, saved=1 - a["fetches"] / tot["fetches_exact"],
                                 flagged=a["flagged"] / P, set_mismatch=a["mismatch"] / P, lost_root=a["lost"] / P, extra=a["extra"] / P))
    print(json.dumps(res))
    for r in res["rules"]:
        print(f"# eps={r['eps']:g} tau={r['tau']:g} kappa={r['kappa']:g} k0={r['k0']} samecell={r['samecell']} tau2={r['tau2']:g} steep={r.get('steep', 0):g} tautrue={r.get('tautrue', 0):g} taucell={r.get('taucell', 0):g} dilate={r.get('dilate', 0)} tau3={r.get('tau3', 0):g}: fetches {r['fetches_per_point']:.2f} (-{100 * r['saved']:.1f} %) "
              f"flagged {r['flagged']:.2e} mismatch {r['set_mismatch']:.2e} lost {r['lost_root']:.2e} extra {r['extra']:.2e}", file=sys.stderr)


if __name__ == "__main__":
    main()
