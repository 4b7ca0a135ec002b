#!/usr/bin/env python3
"""CPU emulation of the speculative early filter for the Broyden search -- design tool, no GPU.

Runs the 13-init search of fuse_cuda_kernel_fast.cu:252-452 vectorised in numpy (fp32) on the synthetic rig, keeps the whole
trajectory of every (point, init) item, and then evaluates retirement rules offline:

    after fetch k (k = 2, 3): retire init i when a LATER live init j is within eps of it and both residuals contract;
    the last member of such a group (the one K9 would keep, filter.cu:10-54) runs on; when it ends invalid the retired
    members it blocked are searched after all.

For each rule: trilinear fetches per point (exact search: all of them) and the fraction of points whose post-K9 candidate
set differs from the exact one.

    python tools/cluster_emul.py [n_points]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from intrinsicavatar_amd import synthetic as S      # noqa: E402

F32 = np.float32


def scene(D=32, H=128, W=128):
    w, offk, sck, bbox = S.skinning_weight_grid(D, H, W)
    rig = S.make_rig(S.make_pose(0))
    tfs = rig["tfs"][0].astype(F32)                                   # [24,4,4]
    vJ = np.einsum("jv,jc->vc", w[0].reshape(24, -1), tfs[:, :3, :].reshape(24, 12)).astype(F32).reshape(D, H, W, 12)
    return vJ, tfs, offk.reshape(3).astype(F32), sck.reshape(3).astype(F32), w[0], rig


def sample_J(vJ, g):
    """trilinear, align_corners, zero padding; g [n,3] in [-1,1]; returns [n,12]."""
    D, H, W, _ = vJ.shape
    ix = (g[:, 0] + 1) / 2 * (W - 1)
    iy = (g[:, 1] + 1) / 2 * (H - 1)
    iz = (g[:, 2] + 1) / 2 * (D - 1)
    bad = ~np.isfinite(ix) | ~np.isfinite(iy) | ~np.isfinite(iz)
    ix, iy, iz = (np.where(bad | (np.abs(v) > 1e9), -100.0, v).astype(F32) for v in (ix, iy, iz))
    x0, y0, z0 = np.floor(ix).astype(np.int64), np.floor(iy).astype(np.int64), np.floor(iz).astype(np.int64)
    out = np.zeros((g.shape[0], 12), F32)
    for c in range(8):
        xx, yy, zz = x0 + (c & 1), y0 + ((c >> 1) & 1), z0 + ((c >> 2) & 1)
        wgt = (np.where(c & 1, ix - x0, x0 + 1 - ix) * np.where(c & 2, iy - y0, y0 + 1 - iy) * np.where(c & 4, iz - z0, z0 + 1 - iz)).astype(F32)
        ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H) & (zz >= 0) & (zz < D)
        v = vJ[np.clip(zz, 0, D - 1), np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)]
        out += np.where(ok[:, None], v * wgt[:, None], 0).astype(F32)
    return out


def points(n, vJ, tfs, wgrid, offk, sck, rig, seed=0):
    """posed-space query points shaped like the secondary march of the bench scene: canonical points in the model's blob
    (sphere-initialised SDF: radius ~0.5 of the canonical box), pushed forward by LBS, plus up to 6 cm of noise."""
    rng = np.random.default_rng(seed)
    D, H, W, _ = vJ.shape
    c = rng.normal(size=(n, 3)).astype(F32)
    c = c / np.linalg.norm(c, axis=1, keepdims=True) * (rng.random((n, 1)) ** (1 / 3)).astype(F32)
    half = np.array([1 / sck[0], 1 / sck[1], 1 / sck[2]], F32)
    xc = c * half * 0.55 - offk
    J = sample_J(vJ, ((xc + offk) * sck).astype(F32)).reshape(n, 3, 4)
    xd = np.einsum("nij,nj->ni", J[:, :, :3], xc) + J[:, :, 3]
    return (xd + rng.normal(size=(n, 3)).astype(F32) * 0.03).astype(F32)


def search(xd, vJ, tfs, offk, sck, cvg=1e-5, dvg=1e-1, iters=10):
    """all inits of all points.  returns traj [P,I,iters+1,3] (x after each fetch's step: traj[...,0,:] = x0, [...,k,:] = x_k),
    res [P,I,iters+1] (|g(x_k)|^2, nan where not evaluated), nfetch [P,I], valid [P,I], xfin [P,I,3]."""
    bones = S.INIT_BONES
    P, I = xd.shape[0], len(bones)
    T = tfs[bones]                                                    # [I,4,4]
    xt = np.repeat(xd[:, None, :], I, 1).reshape(-1, 3)
    Tt = np.tile(T, (P, 1, 1))
    x = np.einsum("nji,nj->ni", Tt[:, :3, :3], xt - Tt[:, :3, 3]).astype(F32)      # R^T (xd - t)
    n = x.shape[0]
    traj = np.full((n, iters + 1, 3), np.nan, F32)
    res = np.full((n, iters + 1), np.nan, F32)
    traj[:, 0] = x
    Jl = sample_J(vJ, ((x + offk) * sck).astype(F32)).reshape(n, 3, 4)
    Ji = np.transpose(Jl[:, :, :3], (0, 2, 1)).copy()
    g = (np.einsum("nij,nj->ni", Jl[:, :, :3], x) + Jl[:, :, 3] - xt).astype(F32)
    res[:, 0] = (g * g).sum(-1)
    live = np.ones(n, bool)
    nfetch = np.ones(n, np.int32)
    valid = np.zeros(n, bool)
    xfin = np.zeros((n, 3), F32)
    for it in range(iters):
        idx = np.nonzero(live)[0]
        if idx.size == 0:
            break
        u = -np.einsum("nij,nj->ni", Ji[idx], g[idx]).astype(F32)
        xn = (x[idx] + u).astype(F32)
        gg = ((xn + offk) * sck).astype(F32)
        Jl = sample_J(vJ, gg).reshape(-1, 3, 4)
        gn = (np.einsum("nij,nj->ni", Jl[:, :, :3], xn) + Jl[:, :, 3] - xt[idx]).astype(F32)
        nrm = (gn * gn).sum(-1)
        x[idx] = xn
        traj[idx, it + 1] = xn
        res[idx, it + 1] = nrm
        nfetch[idx] += 1
        with np.errstate(invalid="ignore"):
            conv = nrm < cvg * cvg
            div = ~conv & ~(nrm <= dvg * dvg)                         # nan counts as diverged
        inbox = (np.abs(gg) <= 1).all(-1)
        valid[idx[conv & inbox]] = True
        xfin[idx[conv]] = xn[conv]
        cont = ~conv & ~div
        # Broyden update for the continuing items
        ci = idx[cont]
        dx, dg = u[cont], (gn[cont] - g[ci]).astype(F32)
        Jc = Ji[ci]
        c_ = np.einsum("nji,nj->ni", Jc, dx)                         # J^T dx  (columns)
        s = (c_ * dg).sum(-1, keepdims=True)
        r = -np.einsum("nij,nj->ni", Jc, dg) + dx
        with np.errstate(all="ignore"):
            Ji[ci] = (Jc + r[:, :, None] * c_[:, None, :] / s[:, :, None]).astype(F32)
        g[ci] = gn[cont]
        live[idx[~cont]] = False
    return (traj.reshape(P, I, iters + 1, 3), res.reshape(P, I, iters + 1), nfetch.reshape(P, I), valid.reshape(P, I),
            xfin.reshape(P, I, 3))


def k9(x, valid):
    """filter.cu:10-54: drop i when a LATER valid j lies within 1e-4."""
    P, I = valid.shape
    keep = valid.copy()
    for i in range(I):
        for j in range(i + 1, I):
            d = ((x[:, i] - x[:, j]) ** 2).sum(-1)
            keep[:, i] &= ~(valid[:, j] & (d < 1e-8))
    return keep


def evaluate(traj, res, nfetch, valid, xfin, k, eps, rho):
    """rule: after fetch k (x_{k-1} evaluated), item i is RETIRED if it is live, contracting (res[k-1] < rho * res[k-2]) and a
    later live contracting item j has |x_{k-1}^i - x_{k-1}^j|_inf < eps.  Blocker = the largest such j.  Items whose blocker
    ends invalid are re-run (cost: their full exact search), recursively.  Returns (fetches per point, mismatch fraction)."""
    P, I = valid.shape
    live = nfetch > k                                                  # still running after fetch k
    with np.errstate(invalid="ignore"):
        contr = live & (res[:, :, k - 1] < rho * res[:, :, k - 2]) if k >= 2 else live
    xk = traj[:, :, k - 1]
    blocker = np.full((P, I), -1, np.int32)
    for i in range(I):
        for j in range(i + 1, I):
            near = contr[:, i] & contr[:, j] & (np.abs(xk[:, i] - xk[:, j]).max(-1) < eps)
            blocker[:, i] = np.where(near, j, blocker[:, i])
    # resolve from the last init backwards: item runs iff not blocked, or its blocker ended (ran and) invalid / (was itself retired and) ...
    runs = np.zeros((P, I), bool)
    val_s = np.zeros((P, I), bool)
    for i in range(I - 1, -1, -1):
        b = blocker[:, i]
        has = b >= 0
        bi = np.clip(b, 0, I - 1)
        b_runs = np.take_along_axis(runs, bi[:, None], 1)[:, 0]
        b_valid = np.take_along_axis(val_s, bi[:, None], 1)[:, 0]
        # blocked item stays retired when its blocker did not run (retired itself: same group) or ran and is valid
        run_i = ~has | (b_runs & ~b_valid)
        runs[:, i] = run_i
        val_s[:, i] = run_i & valid[:, i]
    fetches = np.where(runs, nfetch, np.minimum(nfetch, k)).sum(1)    # retired items paid k fetches (re-runs: +k more, ignored: rare)
    rerun = (runs & (blocker >= 0)).sum()
    keep_exact = k9(xfin, valid)
    keep_spec = k9(xfin, val_s)
    mism = (keep_exact != keep_spec).any(1)
    return fetches.mean(), mism.mean(), rerun / P, keep_exact.sum(1).mean()


def evaluate_progressive(traj, res, nfetch, valid, xfin, k0, eps, rho, k1=10):
    """the same rule applied after EVERY fetch k0 <= k <= k1 to the items still running."""
    P, I = valid.shape
    retired_at = np.zeros((P, I), np.int32)                            # 0 = never retired
    blocker = np.full((P, I), -1, np.int32)
    for k in range(k0, k1 + 1):
        running = (nfetch > k) & (retired_at == 0)
        with np.errstate(invalid="ignore"):
            contr = running & (res[:, :, k - 1] < rho * res[:, :, k - 2])
        xk = traj[:, :, k - 1]
        for i in range(I):
            for j in range(i + 1, I):
                near = contr[:, i] & contr[:, j] & (retired_at[:, i] == 0) & (np.abs(xk[:, i] - xk[:, j]).max(-1) < eps)
                blocker[:, i] = np.where(near, j, blocker[:, i])
            newly = (blocker[:, i] >= 0) & (retired_at[:, i] == 0)
            retired_at[:, i] = np.where(newly, k, retired_at[:, i])
            contr[:, i] &= ~newly                                      # a retired item blocks nobody from now on
    runs = np.zeros((P, I), bool)
    val_s = np.zeros((P, I), bool)
    for i in range(I - 1, -1, -1):
        b = blocker[:, i]
        has = b >= 0
        bi = np.clip(b, 0, I - 1)
        b_runs = np.take_along_axis(runs, bi[:, None], 1)[:, 0]
        b_valid = np.take_along_axis(val_s, bi[:, None], 1)[:, 0]
        run_i = ~has | (b_runs & ~b_valid)
        runs[:, i] = run_i
        val_s[:, i] = run_i & valid[:, i]
    cost = np.where(retired_at > 0, np.minimum(nfetch, retired_at), nfetch) + np.where(runs & (retired_at > 0), nfetch, 0)
    keep_exact = k9(xfin, valid)
    keep_spec = k9(xfin, val_s)
    mism = (keep_exact != keep_spec).any(1)
    # severity: an exact survivor with no speculative survivor within 1 mm = a distinct root was lost
    lost = np.zeros(P, bool)
    for i in range(I):
        dmin = np.full(P, np.inf)
        for j in range(I):
            d = np.abs(xfin[:, i] - xfin[:, j]).max(-1)
            dmin = np.where(keep_spec[:, j], np.minimum(dmin, d), dmin)
        lost |= keep_exact[:, i] & (dmin > 1e-3)
    extra = (keep_spec & ~keep_exact).any(1)
    return cost.sum(1).mean(), mism.mean(), lost.mean(), extra.mean()


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    vJ, tfs, offk, sck, wgrid, rig = scene()
    xd = points(n, vJ, tfs, wgrid, offk, sck, rig)
    t0 = time.time()
    traj, res, nfetch, valid, xfin = search(xd, vJ, tfs, offk, sck)
    print(f"{n} points x 13 inits: {time.time() - t0:.1f} s; fetches / item {nfetch.mean():.3f}; / point {nfetch.sum(1).mean():.2f}; "
          f"valid {valid.mean():.3f}; ended after 2 fetches {np.mean(nfetch == 2):.3f}; survivors / point {k9(xfin, valid).sum(1).mean():.3f}")
    print("hist of fetches:", np.bincount(nfetch.reshape(-1), minlength=12)[1:] / nfetch.size)
    if os.environ.get("SINGLE"):
        for k in (2, 3):
            for eps in (1e-3, 3e-3, 1e-2, 3e-2):
                for rho in (0.25, 1.0):
                    f, m, rr, surv = evaluate(traj, res, nfetch, valid, xfin, k, eps, rho)
                    print(f"k={k} eps={eps:g} rho={rho}: fetches/point {f:.2f}  mismatch {m:.2e}  re-runs/point {rr:.4f}")
    for k0 in (2, 3):
        for eps in (1e-3, 2e-3, 5e-3, 1e-2):
            for rho in (0.25, 1.0):
                f, m, lost, extra = evaluate_progressive(traj, res, nfetch, valid, xfin, k0, eps, rho)
                print(f"progressive from k={k0} eps={eps:g} rho={rho}: fetches/point {f:.2f}  set mismatch {m:.2e}  distinct root lost {lost:.2e}  extra survivor {extra:.2e}")


if __name__ == "__main__":
    main()
