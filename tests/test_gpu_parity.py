"""GPU (MI355X): the HIP path, called through the C ABI (libia_amd.so via ctypes), against
 (a) the CPU oracle on the same seeded inputs and (b) the committed golden vectors of the reference.

Bit-exact for every integer / bool / index output AND for the float outputs of K1..K10 and
traverse_grids (identical IEEE op order, no fma contraction on either side)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def T(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    return t if dtype is None else t.to(dtype)


def N(t):
    return t.detach().cpu().numpy()


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    from intrinsicavatar_amd import build
    build.build()
    from intrinsicavatar_amd import nerfacc, lib_nerfacc, fast_snarf
    return dict(nerfacc=nerfacc, lib=lib_nerfacc, snarf=fast_snarf)


# ----------------------------------------------------------------------------- traverse_grids
@pytest.mark.parametrize("hw,seed,step", [(128, 0, 4.3301 / 64), (96, 3, 0.0338), (64, 5, 0.011)])
def test_traverse_grids_vs_oracle(ops, oracle, hw, seed, step):
    from intrinsicavatar_amd import synthetic as S
    sc = S.make_scene(hw, hw, pose_seed=seed)
    rays = sc["rays"]
    n = rays.shape[0]
    rng = np.random.default_rng(seed)
    near = np.zeros(n, np.float32)
    far = np.full(n, 1e10, np.float32)
    if seed == 3:       # stratified jitter (temporal_occ_grid.py:162-163) and a far clip
        near = (rng.random(n) * step).astype(np.float32)
        far = np.full(n, 5.6, np.float32)
    ref = oracle.traverse_grids(rays[:, :3], rays[:, 3:6], sc["binaries"], sc["aabb"], near, far, step)
    iv, sm, term = ops["nerfacc"].traverse_grids(
        T(rays[:, :3]), T(rays[:, 3:6]), T(sc["binaries"])[None], T(sc["aabb"])[None],
        near_planes=T(near), far_planes=T(far), step_size=step, cone_angle=0.0)
    assert ref["samples"]["vals"].shape[0] > 1000
    np.testing.assert_array_equal(N(iv.packed_info), ref["intervals"]["packed_info"])
    np.testing.assert_array_equal(N(sm.packed_info), ref["samples"]["packed_info"])
    np.testing.assert_array_equal(N(iv.ray_indices), ref["intervals"]["ray_indices"])
    np.testing.assert_array_equal(N(iv.is_left), ref["intervals"]["is_left"])
    np.testing.assert_array_equal(N(iv.is_right), ref["intervals"]["is_right"])
    np.testing.assert_array_equal(N(sm.ray_indices), ref["samples"]["ray_indices"])
    np.testing.assert_array_equal(N(iv.vals), ref["intervals"]["vals"])      # bit-exact t values
    np.testing.assert_array_equal(N(sm.vals), ref["samples"]["vals"])
    np.testing.assert_array_equal(N(term), ref["termination_planes"])


def test_traverse_grids_edge_cases(ops, oracle):
    """empty grid, full grid, rays missing the box, rays starting inside, axis-parallel rays, zero rays."""
    rng = np.random.default_rng(1)
    aabb = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    n = 4096
    o = rng.uniform(-3, 3, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    o[:256] = rng.uniform(-0.9, 0.9, (256, 3))                 # inside the box
    d[256:512] = np.eye(3, dtype=np.float32)[rng.integers(0, 3, 256)] * rng.choice([-1, 1], (256, 1))   # axis-parallel
    near = np.zeros(n, np.float32)
    far = np.full(n, 1e10, np.float32)
    for name, grid in (("empty", np.zeros((16, 16, 16), bool)), ("full", np.ones((16, 16, 16), bool)),
                       ("random", rng.random((16, 24, 8)) < 0.3)):
        ref = oracle.traverse_grids(o, d, grid, aabb, near, far, 0.05)
        iv, sm, term = ops["nerfacc"].traverse_grids(T(o), T(d), T(grid)[None], T(aabb)[None], T(near), T(far), 0.05, 0.0)
        np.testing.assert_array_equal(N(iv.packed_info), ref["intervals"]["packed_info"], err_msg=name)
        np.testing.assert_array_equal(N(iv.vals), ref["intervals"]["vals"], err_msg=name)
        np.testing.assert_array_equal(N(iv.is_left), ref["intervals"]["is_left"], err_msg=name)
        np.testing.assert_array_equal(N(sm.vals), ref["samples"]["vals"], err_msg=name)
        if name == "empty":
            assert iv.vals.numel() == 0 and sm.vals.numel() == 0
    iv, sm, term = ops["nerfacc"].traverse_grids(T(o[:0]), T(d[:0]), T(np.ones((8, 8, 8), bool))[None], T(aabb)[None],
                                                 T(near[:0]), T(far[:0]), 0.05, 0.0)
    assert iv.vals.numel() == 0 and iv.packed_info.shape == (0, 2)


@pytest.fixture
def worst_case_capacity(ops, monkeypatch):
    """the fused traversal sizes its outputs for FUSED_CAP_PER_RAY samples per ray on average and falls back to the two-phase protocol
    beyond that; tests of the fused kernels on dense grids ask for the worst-case capacity so that the fused path is what runs."""
    monkeypatch.setattr(ops["nerfacc"], "FUSED_CAP_PER_RAY", 1 << 20)


def test_traverse_fused_equals_two_pass(ops, oracle, worst_case_capacity):
    """the single-launch traversal (ticketed tiles + look-back scan + element-parallel expansion) returns exactly the
    two-phase outputs: many-run rays (> 8 runs: in-order slow path), secondary-march style rays starting inside the box
    with a far clip, a ragged last tile, and the capacity-overflow fallback."""
    rng = np.random.default_rng(11)
    aabb = np.array([-1.25, -1.55, -1.25, 1.25, 0.95, 1.25], np.float32)
    n = 50_000 + 37
    o = rng.uniform(-1.2, 0.9, (n, 3)).astype(np.float32)
    o[: n // 4] = rng.uniform(-4, 4, (n // 4, 3))
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    near = np.zeros(n, np.float32)
    far = np.full(n, 1.5, np.float32)
    tr = ops["nerfacc"].traverse_grids
    for name, grid, step in (("checker", (np.indices((32, 32, 32)).sum(0) % 2).astype(bool), 1.5 / 63),
                             ("blob", rng.random((64, 64, 64)) < 0.08, 1.5 / 63),
                             ("dense", rng.random((64, 64, 64)) < 0.7, 0.011)):
        a = tr(T(o), T(d), T(grid)[None], T(aabb)[None], T(near), T(far), step, 0.0, method="two_pass")
        b = tr(T(o), T(d), T(grid)[None], T(aabb)[None], T(near), T(far), step, 0.0, method="fused", max_extent=1.5)
        assert a[1].vals.numel() > 10_000, name
        for k in ("vals", "packed_info", "ray_indices", "is_left", "is_right"):
            assert torch.equal(getattr(a[0], k), getattr(b[0], k)), (name, "intervals", k)
        for k in ("vals", "packed_info", "ray_indices"):
            assert torch.equal(getattr(a[1], k), getattr(b[1], k)), (name, "samples", k)
        assert torch.equal(a[2], b[2]), name
        # the per-sample interval ends the fused traversal writes == the boolean-mask gathers on the edge list
        assert a[1].t_starts is None and b[1].t_starts is not None
        assert torch.equal(b[1].t_starts, a[0].vals[a[0].is_left]) and torch.equal(b[1].t_ends, a[0].vals[a[0].is_right]), name
        ts, te = b[1].interval_ends(b[0])
        assert ts is b[1].t_starts and torch.equal(a[1].interval_ends(a[0])[1], te)
        if name == "checker":
            assert int((a[0].packed_info[:, 1] - a[1].packed_info[:, 1]).max()) > 8      # rays with more than 8 runs
    # capacity overflow (a deliberately wrong extent hint) -> transparent fallback to the two-phase protocol
    c = tr(T(o), T(d), T(grid)[None], T(aabb)[None], T(near), T(far), step, 0.0, method="fused", max_extent=0.02)
    assert torch.equal(a[0].vals, c[0].vals) and torch.equal(a[1].ray_indices, c[1].ray_indices)
    ref = oracle.traverse_grids(o[:3000], d[:3000], grid, aabb, near[:3000], far[:3000], step)
    e = tr(T(o[:3000]), T(d[:3000]), T(grid)[None], T(aabb)[None], T(near[:3000]), T(far[:3000]), step, 0.0, max_extent=1.5)
    np.testing.assert_array_equal(N(e[0].vals), ref["intervals"]["vals"])
    np.testing.assert_array_equal(N(e[1].vals), ref["samples"]["vals"])
    np.testing.assert_array_equal(N(e[0].is_left), ref["intervals"]["is_left"])
    np.testing.assert_array_equal(N(e[0].is_right), ref["intervals"]["is_right"])


@pytest.mark.parametrize("n", [65_536 + 77, 300_001])
def test_traverse_span_sorted_tiles_equal_the_ray_order_kernel(ops, monkeypatch, n, worst_case_capacity):
    """the fused traversal with the rays of a 1024-ray tile walked in order of their box-crossing span (traverse_sorted_kernel,
    big batches) against the ray-order kernel (traverse_fused_kernel) and the two-pass protocol: every output tensor equal --
    only which lane walks which ray changes.  Grids with rays of more than 4 runs (in-order re-walk), rays that miss the box,
    a ragged last tile."""
    rng = np.random.default_rng(n)
    aabb = np.array([-1.0, -1.1, -0.9, 1.0, 0.9, 1.1], np.float32)
    o = (rng.random((n, 3)).astype(np.float32) * 2.6 - 1.3)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    near = (rng.random(n) * 0.05).astype(np.float32)
    far = (0.2 + rng.random(n) * 1.5).astype(np.float32)
    tr = ops["nerfacc"].traverse_grids
    for name, grid, step in (("checker", (np.indices((64, 64, 64)).sum(0) % 2).astype(bool), 1.7 / 63),
                             ("blob", rng.random((64, 64, 64)) < 0.08, 1.7 / 63), ("dense", rng.random((64, 64, 64)) < 0.7, 0.02)):
        args = (T(o), T(d), T(grid)[None], T(aabb)[None], T(near), T(far), step, 0.0)
        a = tr(*args, method="fused", max_extent=1.75)                                  # ray-order tiles
        b = tr(*args, method="fused", max_extent=1.75, incoherent=True)                 # span-sorted tiles (what the secondary march asks for)
        monkeypatch.setenv("IA_TRAVERSE_TILES", "span")
        c = tr(*args, method="fused", max_extent=1.75)                                  # the A / B override
        monkeypatch.delenv("IA_TRAVERSE_TILES")
        assert a[1].vals.numel() > 50_000, name
        for other in (b, c):
            for k in ("vals", "packed_info", "ray_indices", "is_left", "is_right"):
                assert torch.equal(getattr(a[0], k), getattr(other[0], k)), (name, "intervals", k)
            for k in ("vals", "packed_info", "ray_indices", "t_starts", "t_ends"):
                assert torch.equal(getattr(a[1], k), getattr(other[1], k)), (name, "samples", k)
            assert torch.equal(a[2], other[2]), name
        if name == "checker":
            assert int((a[0].packed_info[:, 1] - a[1].packed_info[:, 1]).max()) > 4      # rays with more than 4 runs


def test_traverse_average_sized_capacity_falls_back_when_a_batch_needs_more(ops):
    """default capacity = FUSED_CAP_PER_RAY samples per ray on average: a dense grid needs more, the kernel raises its overflow flag and the
    call goes through the two-phase protocol -- same intervals / samples as with the worst-case capacity."""
    rng = np.random.default_rng(4)
    n = 600_000
    aabb = np.array([-1.0, -1.0, -1.0, 1.0, 1.0, 1.0], np.float32)
    o = (rng.random((n, 3)).astype(np.float32) * 1.6 - 0.8)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    grid = rng.random((64, 64, 64)) < 0.7
    nf = ops["nerfacc"]
    args = (T(o), T(d), T(grid)[None], T(aabb)[None], torch.zeros(n, device=DEV), torch.full((n,), 1.5, device=DEV), 0.011, 0.0)
    a = nf.traverse_grids(*args, method="fused", max_extent=1.5)
    assert a[1].vals.numel() > nf.FUSED_CAP_PER_RAY * n and a[1].t_starts is None          # fell back
    saved = nf.FUSED_CAP_PER_RAY
    try:
        nf.FUSED_CAP_PER_RAY = 1 << 20
        b = nf.traverse_grids(*args, method="fused", max_extent=1.5)
    finally:
        nf.FUSED_CAP_PER_RAY = saved
    assert b[1].t_starts is not None
    for k in ("vals", "packed_info", "ray_indices", "is_left", "is_right"):
        assert torch.equal(getattr(a[0], k), getattr(b[0], k)), k
    for k in ("vals", "packed_info", "ray_indices"):
        assert torch.equal(getattr(a[1], k), getattr(b[1], k)), k


@pytest.mark.parametrize("incoherent", [False, True])
def test_traverse_without_termination_planes_stops_at_the_occupied_box(ops, incoherent, worst_case_capacity):
    """termination_planes=False (what render_step asks for): the fused kernels end a ray's walk where it leaves the cell box of the
    occupied cells.  Every interval / sample tensor must equal the full walk's: grids whose occupied cells fill a small part of the
    box, touch its faces, are a single cell, are empty; rz not a multiple of 32 (whole-grid box); rays starting inside, beyond and
    beside the occupied region, axis-parallel rays (d == 0 on two axes)."""
    rng = np.random.default_rng(11)
    n = 200_003
    aabb = np.array([-1.0, -1.1, -0.9, 1.0, 0.9, 1.1], np.float32)
    o = (rng.random((n, 3)).astype(np.float32) * 2.6 - 1.3)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[:3000] = np.eye(3, dtype=np.float32)[rng.integers(0, 3, 3000)] * rng.choice([-1.0, 1.0], (3000, 1)).astype(np.float32)
    near = (rng.random(n) * 0.05).astype(np.float32)
    far = (0.2 + rng.random(n) * 2.5).astype(np.float32)
    tr = ops["nerfacc"].traverse_grids

    def region(shape, lo, hi, p):
        g = np.zeros(shape, bool)
        sl = tuple(slice(a, b + 1) for a, b in zip(lo, hi))
        g[sl] = rng.random(g[sl].shape) < p
        g[tuple(lo)] = g[tuple(hi)] = True
        return g

    grids = (("body", region((64, 64, 64), (20, 8, 25), (44, 57, 40), 0.3)),
             ("corner", region((64, 64, 64), (0, 0, 0), (9, 63, 5), 0.5)),
             ("far_corner", region((64, 64, 64), (50, 60, 33), (63, 63, 63), 0.5)),
             ("one_cell", region((64, 64, 64), (31, 32, 33), (31, 32, 33), 1.0)),
             ("empty", np.zeros((64, 64, 64), bool)),
             ("full", np.ones((32, 32, 32), bool)),
             ("rz48", region((32, 32, 48), (5, 6, 7), (20, 21, 40), 0.4)))
    for name, grid in grids:
        args = (T(o), T(d), T(grid)[None], T(aabb)[None], T(near), T(far), 2.0 / 63, 0.0)
        a = tr(*args, method="fused", max_extent=2.8, incoherent=incoherent)
        b = tr(*args, method="fused", max_extent=2.8, incoherent=incoherent, termination_planes=False)
        assert a[2] is not None and b[2] is None
        assert (a[1].vals.numel() > 0) == (name != "empty") and (a[1].vals.numel() > 2_000 or name in ("one_cell", "empty")), name
        for k in ("vals", "packed_info", "ray_indices", "is_left", "is_right"):
            assert torch.equal(getattr(a[0], k), getattr(b[0], k)), (name, "intervals", k)
        for k in ("vals", "packed_info", "ray_indices", "t_starts", "t_ends"):
            assert torch.equal(getattr(a[1], k), getattr(b[1], k)), (name, "samples", k)


def test_traverse_properties_full_size(ops):
    """540x540 (BASELINE config 2) -- size-independent properties instead of the (slow) oracle."""
    from intrinsicavatar_amd import synthetic as S
    sc = S.make_scene(540, 540, pose_seed=0)
    rays = sc["rays"]
    n = rays.shape[0]
    step = 4.3301 / 128
    iv, sm, term = ops["nerfacc"].traverse_grids(
        T(rays[:, :3]), T(rays[:, 3:6]), T(sc["binaries"])[None], T(sc["aabb"])[None],
        near_planes=torch.zeros(n, device=DEV), far_planes=torch.full((n,), 1e10, device=DEV), step_size=step)
    assert n == 291600
    pi = iv.packed_info
    assert int(pi[:, 1].sum()) == iv.vals.numel()
    assert torch.equal(pi[:, 0], torch.cumsum(pi[:, 1], 0) - pi[:, 1])
    assert int(iv.is_left.sum()) == int(iv.is_right.sum()) == sm.vals.numel()
    ts, te = iv.vals[iv.is_left], iv.vals[iv.is_right]
    assert torch.all(te > ts)
    torch.testing.assert_close(te - ts, torch.full_like(ts, step), rtol=0, atol=2e-6)
    assert torch.equal((ts + te) * 0.5, sm.vals)
    # sorted by ray, increasing t within a ray
    assert torch.all(sm.ray_indices[1:] >= sm.ray_indices[:-1])
    same = sm.ray_indices[1:] == sm.ray_indices[:-1]
    assert torch.all(sm.vals[1:][same] > sm.vals[:-1][same])
    # every sample mid-point lies in an occupied cell of the grid
    ro, rd = T(rays[:, :3])[sm.ray_indices], T(rays[:, 3:6])[sm.ray_indices]
    p = ro + rd * sm.vals[:, None]
    aabb = T(sc["aabb"])
    cell = ((p - aabb[:3]) / (aabb[3:] - aabb[:3]) * 64).long().clamp(0, 63)
    occ = T(sc["binaries"])[cell[:, 0], cell[:, 1], cell[:, 2]]
    assert occ.float().mean() > 0.995      # (mid-points within 1e-6 of a face may round to the neighbour)


# ----------------------------------------------------------------------------- compositing
def _random_packed(rng, n_rays, max_steps):
    steps = rng.integers(0, max_steps + 1, n_rays)
    steps[rng.random(n_rays) < 0.3] = 0
    cum = np.cumsum(steps)
    return np.stack([cum - steps, steps], -1).astype(np.int32), int(cum[-1])


def test_render_weight_and_accumulate_vs_oracle(ops, oracle):
    rng = np.random.default_rng(7)
    pi, S_ = _random_packed(rng, 5000, 48)
    alphas = rng.uniform(0, 0.9, S_).astype(np.float32)
    alphas[rng.random(S_) < 0.1] = 0.0
    ray_idx = oracle.unpack_info(pi, S_)
    w_ref, t_ref = oracle.render_weight_from_alpha(alphas, pi)
    w, t = ops["nerfacc"].render_weight_from_alpha(T(alphas), ray_indices=T(ray_idx), n_rays=pi.shape[0])
    np.testing.assert_array_equal(N(w), w_ref)
    np.testing.assert_array_equal(N(t), t_ref)
    w2, _ = ops["nerfacc"].render_weight_from_alpha(T(alphas), packed_info=T(pi))
    np.testing.assert_array_equal(N(w2), w_ref)
    for dim in (1, 3, 5):
        vals = rng.normal(size=(S_, dim)).astype(np.float32)
        ref = oracle.accumulate_along_rays(w_ref, vals, ray_idx, pi.shape[0])
        out = ops["nerfacc"].accumulate_along_rays(T(w_ref), T(vals), T(ray_idx), pi.shape[0])
        np.testing.assert_array_equal(N(out), ref)
    ref = oracle.accumulate_along_rays(w_ref, None, ray_idx, pi.shape[0])
    out = ops["nerfacc"].accumulate_along_rays(T(w_ref), None, T(ray_idx), pi.shape[0])
    np.testing.assert_array_equal(N(out), ref)


def test_compositing_backward_vs_torch_fp32(ops, oracle):
    """floating-point kernel => plain torch fp32/fp64 autograd reference (tolerance 1e-5 rel)."""
    rng = np.random.default_rng(11)
    pi, S_ = _random_packed(rng, 700, 20)
    alphas = rng.uniform(0.01, 0.8, S_).astype(np.float32)
    vals = rng.normal(size=(S_, 3)).astype(np.float32)
    ray_idx = oracle.unpack_info(pi, S_)
    a = T(alphas).requires_grad_(True)
    v = T(vals).requires_grad_(True)
    w, tr = ops["nerfacc"].render_weight_from_alpha(a, ray_indices=T(ray_idx), n_rays=pi.shape[0])
    col = ops["nerfacc"].accumulate_along_rays(w, v, T(ray_idx), pi.shape[0])
    acc = ops["nerfacc"].accumulate_along_rays(w, None, T(ray_idx), pi.shape[0])
    gcol = T(rng.normal(size=(pi.shape[0], 3)).astype(np.float32))
    gacc = T(rng.normal(size=(pi.shape[0], 1)).astype(np.float32))
    gtr = T(rng.normal(size=S_).astype(np.float32))
    ((col * gcol).sum() + (acc * gacc).sum() + (tr * gtr).sum()).backward()
    # reference in float64 on CPU
    a64 = torch.from_numpy(alphas).double().requires_grad_(True)
    v64 = torch.from_numpy(vals).double().requires_grad_(True)
    ri = torch.from_numpy(ray_idx)
    trs = []
    for b, s in pi:
        one = torch.ones(1, dtype=torch.float64)
        trs.append(torch.cat([one, torch.cumprod(1 - a64[b:b + s], 0)[:-1]]) if s > 0 else a64[:0])
    tr64 = torch.cat(trs)
    w64 = tr64 * a64
    col64 = torch.zeros(pi.shape[0], 3, dtype=torch.float64).index_add(0, ri, w64[:, None] * v64)
    acc64 = torch.zeros(pi.shape[0], 1, dtype=torch.float64).index_add(0, ri, w64[:, None])
    ((col64 * gcol.cpu().double()).sum() + (acc64 * gacc.cpu().double()).sum() + (tr64 * gtr.cpu().double()).sum()).backward()
    np.testing.assert_allclose(N(a.grad), a64.grad.numpy(), rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(N(v.grad), v64.grad.numpy(), rtol=1e-5, atol=1e-6)


# ----------------------------------------------------------------------------- resampling / pack
def test_resampling_vs_golden(ops, golden_dir):
    g = np.load(os.path.join(golden_dir, "golden_resampling.npz"))
    lib = ops["lib"]
    for c in range(int(g["n_cases"])):
        k = f"c{c}_"
        pi, st, en = T(g[k + "packed_info"]), T(g[k + "starts"])[:, None], T(g[k + "ends"])[:, None]
        w, al, sd, n = T(g[k + "weights"]), T(g[k + "alphas"]), T(g[k + "sdfs"]), int(g[k + "n"])
        r = lib.ray_resampling(pi, st, en, w, sd, n)
        for nm, v in zip(("rpi", "ts", "offsets", "indices", "fg_counts", "bg_counts", "surface_idx"), r):
            ref = g[k + "k1_" + nm]
            assert tuple(v.shape) == ref.shape and N(v).dtype == ref.dtype, (c, nm)
            np.testing.assert_array_equal(N(v), ref, err_msg=f"K1 case {c} {nm}")
        # K1 with capacity-sized outputs and its total left on the device (no size read-back of its own): the first `total` slots are the
        # fixture's, the per-ray outputs are the fixture's
        total = torch.empty(1, dtype=torch.int32, device=DEV)
        rc = lib.ray_resampling_capacity(pi, st, en, w, sd, n, total)
        Tn = int(total.item())
        assert Tn == g[k + "k1_ts"].shape[0] and rc[1].shape[0] == n * pi.shape[0] >= Tn
        for nm, v in zip(("rpi", "ts", "offsets", "indices", "fg_counts", "bg_counts", "surface_idx"), rc):
            vv = v[:Tn] if nm in ("ts", "offsets", "indices") else v
            np.testing.assert_array_equal(N(vv), g[k + "k1_" + nm], err_msg=f"K1 (capacity) case {c} {nm}")
        r = lib.ray_resampling_fine(pi, st, en, w, n)
        for nm, v in zip(("rpi", "starts", "ends", "is_fg"), r):
            np.testing.assert_array_equal(N(v), g[k + "k3_" + nm], err_msg=f"K3 case {c} {nm}")
        r = lib.ray_resampling_sdf_fine(pi, st, en, al, sd, n)
        for nm, v in zip(("rpi", "starts", "ends", "is_fg"), r):
            np.testing.assert_array_equal(N(v), g[k + "k4_" + nm], err_msg=f"K4 case {c} {nm}")
    for c in range(int(g["n_edge_cases"])):
        k = f"e{c}_"
        r = lib.ray_resampling_merge(T(g[k + "packed_info"]), T(g[k + "vals"]), T(g[k + "is_left"]),
                                     T(g[k + "is_right"]), T(g[k + "weights"]), int(g[k + "n"]))
        for nm, v in zip(("rpi", "vals", "dists", "is_left", "is_right", "is_resample", "is_fg"), r):
            np.testing.assert_array_equal(N(v), g[k + "k2_" + nm], err_msg=f"K2 case {c} {nm}")
        # K2 fused with its caller's selection of the reached edges (models/intrinsic_avatar.py:1221-1226) == the op sequence on K2's output
        rpi, rvals, _, ril, rir, _, is_fg = r
        fg_idx = torch.nonzero(is_fg)[:, 0]
        ray_idx = lib.unpack_info(rpi, rvals.shape[0])[fg_idx]
        want = (rvals[fg_idx], ril[fg_idx], rir[fg_idx], ray_idx, lib.pack_info(ray_idx, rpi.shape[0]))
        got = lib.ray_resampling_merge_compact(T(g[k + "packed_info"]), T(g[k + "vals"]), T(g[k + "is_left"]), T(g[k + "is_right"]),
                                               T(g[k + "weights"]), int(g[k + "n"]))
        for a_, b_, nm in zip(got, want, ("vals", "is_left", "is_right", "ray_indices", "packed_info")):
            assert torch.equal(a_, b_), (c, nm)
        # ... and the samples forward_ forms from the kept edges (models/intrinsic_avatar.py:1242-1247) == the boolean-mask op sequence
        _check_interval_samples(lib, got[4], got[0], got[1], got[2], got[3], (c, "merge"))
        # ... and both in one call with ONE read-back for the two sizes (capacity-sized edge buffers, the edge count left on the device):
        # identical to the two calls, element for element
        one = lib.ray_resampling_merge_compact_samples(T(g[k + "packed_info"]), T(g[k + "vals"]), T(g[k + "is_left"]), T(g[k + "is_right"]),
                                                       T(g[k + "weights"]), int(g[k + "n"]))
        for a_, b_, nm in zip(one[:5], got, ("vals", "is_left", "is_right", "ray_indices", "packed_info")):
            assert torch.equal(a_, b_) and a_.is_contiguous(), (c, nm)
        two = lib.interval_samples(got[4], got[0], got[1], got[3])
        for nm in ("is_left", "pos", "t_starts", "t_ends", "ray_indices", "packed_info"):
            assert torch.equal(getattr(one[5], nm), getattr(two, nm)), (c, "samples", nm)
        x = torch.rand(two.t_starts.shape[0], device=DEV)
        assert torch.equal(one[5].to_edges(x, 1e10), two.to_edges(x, 1e10)), c


def _check_interval_samples(lib, pinfo, vals, is_left, is_right, ray_indices, tag):
    smp = lib.interval_samples(pinfo, vals, is_left, ray_indices)
    assert int(is_left.sum()) == int(is_right.sum())
    assert torch.equal(smp.t_starts, vals[is_left]), tag
    assert torch.equal(smp.t_ends, vals[is_right]), tag
    assert torch.equal(smp.ray_indices, ray_indices[is_left]), tag
    assert torch.equal(smp.packed_info, lib.pack_info(ray_indices[is_left], pinfo.shape[0])), tag
    x = torch.rand(smp.t_starts.shape[0], device=vals.device)
    want = torch.full_like(vals, 1e10)
    want[is_left] = x
    assert torch.equal(smp.to_edges(x, 1e10), want), tag


def test_interval_samples_equal_the_mask_ops_on_a_march(ops):
    """lib_nerfacc.interval_samples on the edge list of a primary march (rays without edges, rays whose edges hold several runs, the
    last ray ending the list) and on an empty list."""
    from intrinsicavatar_amd import synthetic as S
    sc = S.make_scene(160, 160, pose_seed=1)
    rays = sc["rays"]
    n = rays.shape[0]
    iv, sm, _ = ops["nerfacc"].traverse_grids(T(rays[:, :3]), T(rays[:, 3:6]), T(sc["binaries"])[None], T(sc["aabb"])[None],
                                              torch.zeros(n, device=DEV), torch.full((n,), 1e10, device=DEV), 4.3301 / 128)
    assert sm.vals.numel() > 20_000 and int((iv.packed_info[:, 1] == 0).sum()) > 100
    _check_interval_samples(ops["lib"], iv.packed_info, iv.vals, iv.is_left, iv.is_right, iv.ray_indices, "march")
    smp = ops["lib"].interval_samples(iv.packed_info, iv.vals, iv.is_left, iv.ray_indices)
    assert torch.equal(smp.packed_info.long(), sm.packed_info) and torch.equal((smp.t_starts + smp.t_ends) * 0.5, sm.vals)
    # no edges at all: empty outputs, (0, 0) rows
    e = ops["lib"].interval_samples(torch.zeros((7, 2), dtype=torch.int32, device=DEV), iv.vals[:0], iv.is_left[:0], iv.ray_indices[:0])
    assert e.t_starts.numel() == 0 and e.ray_indices.numel() == 0 and int(e.packed_info.abs().sum()) == 0


def test_pack_unpack_vs_golden(ops, golden_dir):
    g = np.load(os.path.join(golden_dir, "golden_pack.npz"))
    lib = ops["lib"]
    pi, data = T(g["packed_info"]), T(g["data"])
    ri = lib.unpack_info(pi, data.shape[0])
    np.testing.assert_array_equal(N(ri), g["unpack_info"])
    np.testing.assert_array_equal(N(lib.pack_info(ri, pi.shape[0])), g["packed_info"])
    d = data.clone().requires_grad_(True)
    un = lib.unpack_data(pi, d, 32)
    np.testing.assert_array_equal(N(un), g["unpack_data"])
    gr = torch.randn_like(un)
    (un * gr).sum().backward()
    np.testing.assert_array_equal(N(d.grad), N(gr)[g["unpack_mask"]])
    pd, pinfo = lib.pack_data(T(g["pack_data_in"]), T(g["pack_data_mask"]))
    np.testing.assert_array_equal(N(pd), g["pack_data_out"])
    np.testing.assert_array_equal(N(pinfo), g["pack_data_info"])
    # int64 payload (spp shuffle indices, models/intrinsic_avatar.py:1368-1377)
    idx = torch.arange(data.shape[0], device=DEV)[:, None]
    un = lib.unpack_data(pi, idx, 32)
    assert un.dtype == torch.int64
    ref = g["unpack_mask"]
    assert torch.equal(un[..., 0][T(ref)], idx[:, 0])


def test_resampling_vs_oracle_large(ops, oracle):
    """bigger, seeded, ragged case against the oracle (n = spp = 256)."""
    rng = np.random.default_rng(5)
    pi, S_ = _random_packed(rng, 3000, 64)
    st = np.zeros(S_, np.float32)
    en = np.zeros(S_, np.float32)
    for b, s in pi:
        if s:
            dts = rng.uniform(0.005, 0.05, s).astype(np.float32)
            t = rng.uniform(3, 5) + np.cumsum(dts) - dts
            st[b:b + s] = t
            en[b:b + s] = t + dts
    al = rng.uniform(0, 0.4, S_).astype(np.float32)
    sd = rng.normal(0.1, 0.2, S_).astype(np.float32)
    w, _ = oracle.render_weight_from_alpha(al, pi)
    ref = oracle.ray_resampling(pi, st, en, w, sd, 256)
    out = ops["lib"].ray_resampling(T(pi), T(st)[:, None], T(en)[:, None], T(w), T(sd), 256)
    for a, b in zip(out, ref):
        np.testing.assert_array_equal(N(a), b)
    ref = oracle.ray_resampling_sdf_fine(pi, st, en, al, sd, 4)
    out = ops["lib"].ray_resampling_sdf_fine(T(pi), T(st)[:, None], T(en)[:, None], T(al), T(sd), 4)
    for a, b in zip(out, ref):
        np.testing.assert_array_equal(N(a), b)


# ----------------------------------------------------------------------------- fast-SNARF
@pytest.mark.parametrize("schedule", ["simple", "persistent"])
def test_fast_snarf_vs_golden(ops, golden_dir, schedule, monkeypatch):
    monkeypatch.setenv("IA_BROYDEN_SCHEDULE", schedule)     # both Broyden schedules must be bit-identical
    g = np.load(os.path.join(golden_dir, "golden_snarf.npz"))
    sn = ops["snarf"]
    vw = T(g["voxel_w"].astype(np.float32))
    tfs, off, sc = T(g["tfs"]), T(g["offset"]), T(g["scale"])
    _, _, D, H, W = vw.shape
    vd = torch.zeros(1, 3, D, H, W, device=DEV)
    vJ = torch.zeros(1, 12, D, H, W, device=DEV)
    vJcl = torch.zeros(1, D, H, W, 12, device=DEV)
    sn.precompute(vw, tfs, vd, vJ, off, sc, voxel_J_cl=vJcl)
    np.testing.assert_array_equal(N(vd), g["voxel_d"])
    np.testing.assert_array_equal(N(vJ), g["voxel_J"])
    np.testing.assert_array_equal(N(vJcl.permute(0, 4, 1, 2, 3)), g["voxel_J"])
    xd, bones = T(g["xd"]), T(g["bones"])
    n = xd.shape[1]
    for layout in ("ncdhw", "ndhwc"):
        x = torch.zeros(1, n, 13, 3, device=DEV)
        Ji = torch.zeros(1, n, 13, 3, 3, device=DEV)
        valid = torch.zeros(1, n, 13, dtype=torch.bool, device=DEV)
        grid = vJ if layout == "ncdhw" else sn.ChannelLastVoxelJ(vJcl)
        ret = sn.fuse_broyden(x, xd, vd, grid, tfs, bones, True, Ji, valid, off, sc, 1e-5, 1e-1)
        assert ret is None
        np.testing.assert_array_equal(N(valid), g["valid"], err_msg=layout)
        np.testing.assert_array_equal(N(x), g["x"], err_msg=layout)
        np.testing.assert_array_equal(N(Ji), g["J_inv"], err_msg=layout)
        np.testing.assert_array_equal(N(sn.filter(x, valid)), g["filtered"], err_msg=layout)


def test_fast_snarf_bit_exact_vs_the_oracle_on_a_large_batch(ops):
    """K8 on 40 k random points x 13 inits against oracle/ia_oracle.c (plain IEEE divisions in the rank-1 update): x, J_inv, valid
    bit for bit -- the kernel's update shares the reciprocal of the nine divisions' common denominator (snarf.hip J_inv_update)."""
    from intrinsicavatar_amd import synthetic as S
    from oracle import oracle as O
    sn = ops["snarf"]
    w, offk, sck, bbox = S.skinning_weight_grid(D=16, H=64, W=64, smooth_iters=5)
    rig = S.make_rig(S.make_pose(3, 0.25))
    vw, tfs, off, sc = T(w), T(rig["tfs"]), T(offk), T(sck)
    _, _, D, H, W = vw.shape
    vd = torch.zeros(1, 3, D, H, W, device=DEV)
    vJ = torch.zeros(1, 12, D, H, W, device=DEV)
    vJcl = torch.zeros(1, D, H, W, 12, device=DEV)
    sn.precompute(vw, tfs, vd, vJ, off, sc, voxel_J_cl=vJcl)
    n = 40_000
    g = torch.Generator(device="cpu").manual_seed(5)
    lo, hi = torch.from_numpy(S.body_aabb(rig["joints_posed"], 1.0)).split(3)
    xd = (torch.rand(1, n, 3, generator=g) * (hi - lo) + lo).to(DEV)
    x = torch.zeros(1, n, 13, 3, device=DEV)
    Ji = torch.zeros(1, n, 13, 3, 3, device=DEV)
    valid = torch.zeros(1, n, 13, dtype=torch.bool, device=DEV)
    sn.fuse_broyden(x, xd, vd, sn.ChannelLastVoxelJ(vJcl), tfs, T(S.INIT_BONES), True, Ji, valid, off, sc, 1e-5, 1e-1)
    xr, Jr, vr = O.fuse_broyden(N(xd), N(vJ), N(tfs), S.INIT_BONES, N(off), N(sc))
    vr = vr.astype(bool)
    assert 0.02 < vr.mean() < 0.98
    np.testing.assert_array_equal(N(valid), vr)
    m = vr[..., None]
    np.testing.assert_array_equal(np.where(m, N(x), 0), np.where(m, xr, 0))
    np.testing.assert_array_equal(np.where(m[..., None], N(Ji), 0), np.where(m[..., None], Jr, 0))


def test_fast_snarf_roundtrip_property(ops):
    """size-independent property at a large N: forward-skinning every valid root lands on the query point."""
    from intrinsicavatar_amd import synthetic as S
    sn = ops["snarf"]
    w, offk, sck, bbox = S.skinning_weight_grid(D=16, H=64, W=64, smooth_iters=5)
    rig = S.make_rig(S.make_pose(2, 0.2))
    vw, tfs, off, sc = T(w), T(rig["tfs"]), T(offk), T(sck)
    _, _, D, H, W = vw.shape
    vd = torch.zeros(1, 3, D, H, W, device=DEV)
    vJcl = torch.zeros(1, D, H, W, 12, device=DEV)
    sn.precompute(vw, tfs, vd, None, off, sc, voxel_J_cl=vJcl)
    n = 200_000
    g = torch.Generator(device="cpu").manual_seed(0)
    lo, hi = torch.from_numpy(S.body_aabb(rig["joints_posed"], 1.0)).split(3)
    xd = (torch.rand(1, n, 3, generator=g) * (hi - lo) + lo).to(DEV)
    x = torch.zeros(1, n, 13, 3, device=DEV)
    Ji = torch.zeros(1, n, 13, 3, 3, device=DEV)
    valid = torch.zeros(1, n, 13, dtype=torch.bool, device=DEV)
    sn.fuse_broyden(x, xd, vd, sn.ChannelLastVoxelJ(vJcl), tfs, T(S.INIT_BONES), True, Ji, valid, off, sc, 1e-5, 1e-1)
    assert 0.02 < valid.float().mean() < 0.98
    # forward skinning of the roots with torch grid_sample on the channel-first grid
    xc = x[valid]
    norm = (xc + off.reshape(1, 3)) * sc.reshape(1, 3)
    J = torch.nn.functional.grid_sample(vJcl.permute(0, 4, 1, 2, 3), norm[None, :, None, None, :], align_corners=True,
                                        padding_mode="zeros")[0, :, :, 0, 0].T.reshape(-1, 3, 4)
    xd_rec = (J[:, :, :3] @ xc[:, :, None])[:, :, 0] + J[:, :, 3]
    tgt = xd[0][:, None, :].expand(-1, 13, -1)[valid[0]]
    assert (xd_rec - tgt).norm(dim=-1).max() < 2e-5      # cvg threshold 1e-5 (+ fp32 interpolation noise)
    mask = sn.filter(x, valid)
    assert mask.sum() <= valid.sum() and torch.all(valid | ~mask)


def test_density_and_visibility_wrappers(ops):
    """nerfacc.render_weight_from_density / render_visibility_from_alpha / _from_density (imported by the reference,
    off the SDF path): definitions checked against a per-ray fp64 loop."""
    rng = np.random.default_rng(12)
    n_rays = 300
    steps = rng.integers(0, 20, n_rays)
    packed = np.stack([np.cumsum(steps) - steps, steps], -1).astype(np.int64)
    S = int(steps.sum())
    ts = np.sort(rng.random(S)).astype(np.float32)
    te = (ts + rng.random(S) * 0.05).astype(np.float32)
    sig = (rng.random(S) * 30).astype(np.float32)
    ri = np.repeat(np.arange(n_rays), steps)
    na = ops["nerfacc"]
    w, tr, al = na.render_weight_from_density(T(ts), T(te), T(sig), ray_indices=T(ri), n_rays=n_rays)
    vis_a = na.render_visibility_from_alpha(al, ray_indices=T(ri), n_rays=n_rays, early_stop_eps=1e-2, alpha_thre=0.05)
    vis_d = na.render_visibility_from_density(T(ts), T(te), T(sig), ray_indices=T(ri), n_rays=n_rays, early_stop_eps=1e-2,
                                              alpha_thre=0.05)
    a64 = 1 - np.exp(-sig.astype(np.float64) * (te.astype(np.float64) - ts))
    t64 = np.ones(S)
    for b, s in packed:
        if s > 0:
            t64[b:b + s] = np.concatenate([[1.0], np.cumprod(1 - a64[b:b + s])[:-1]])
    np.testing.assert_allclose(N(al), a64, rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(N(tr), t64, rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(N(w), t64 * a64, rtol=1e-4, atol=1e-6)
    ref_vis = (t64 >= 1e-2) & (a64 >= 0.05)
    safe = (np.abs(t64 - 1e-2) > 1e-5) & (np.abs(a64 - 0.05) > 1e-5)
    assert np.array_equal(N(vis_a)[safe], ref_vis[safe]) and torch.equal(vis_a, vis_d)


# ----------------------------------------------------------------------------- deformer candidate packing
@pytest.mark.parametrize("P,I", [(1, 13), (127, 13), (128, 13), (129, 13), (255, 13), (256, 13), (257, 13), (100_003, 13), (6_000_001, 13), (50_000, 5), (70_001, 16)])
@pytest.mark.parametrize("in_place", [False, True])
def test_filter_compact_single_pass_equals_the_three_step_path(P, I, in_place):
    """ia_deform_filter_compact (filter + count + chained scan + packed list in one pass, optionally in place over x)
    == ia_deform_filter_count -> ia_exclusive_scan_i32 -> ia_deform_compact, bit for bit, and == fast_snarf.filter (K9)."""
    import ctypes as C
    from intrinsicavatar_amd import _lib as L, fast_snarf
    lib, st = L.lib(), L.stream()
    g = torch.Generator(device="cpu").manual_seed(P * 31 + I)
    x = torch.rand(P, I, 3, generator=g)
    # clusters of near-identical candidates (|dx| around the 1e-4 threshold), as converged searches produce
    dup = torch.rand(P, I, generator=g) < 0.5
    x = torch.where(dup[..., None], x[:, :1] + (torch.rand(P, I, 3, generator=g) - 0.5) * 1.6e-4, x)
    valid = torch.rand(P, I, generator=g) < 0.6
    x, valid = x.to(DEV), valid.to(DEV)
    # three-step path
    mask0 = torch.empty((P, I), dtype=torch.bool, device=DEV)
    cnt0 = torch.empty(P, dtype=torch.int32, device=DEV)
    start0 = torch.empty(P, dtype=torch.int32, device=DEV)
    tot0 = torch.zeros(1, dtype=torch.int32, device=DEV)
    L.check(lib.ia_deform_filter_count(L.i64(P), L.i32(I), L.ptr(x), L.ptr(valid), L.ptr(mask0), L.ptr(cnt0), st), "fc")
    tmp = L.scan_tmp(P, DEV)
    L.check(lib.ia_exclusive_scan_i32(L.ptr(cnt0), L.ptr(start0), L.ptr(tot0), L.i64(P), L.ptr(tmp), st), "scan")
    Q = int(tot0.item())
    cx0 = torch.empty((Q, 3), device=DEV)
    cs0 = torch.empty(Q, dtype=torch.int32, device=DEV)
    L.check(lib.ia_deform_compact(L.i64(P), L.i32(I), L.ptr(x), L.ptr(mask0), L.ptr(start0), L.ptr(cx0), L.ptr(cs0), st), "compact")
    assert torch.equal(mask0, fast_snarf.filter(x[None], valid[None])[0])
    # single pass
    xin = x.clone()
    cnt = torch.full((P,), -1, dtype=torch.int32, device=DEV)
    start = torch.full((P,), -1, dtype=torch.int32, device=DEV)
    tot = torch.full((1,), -1, dtype=torch.int32, device=DEV)
    mask = torch.zeros((P, I), dtype=torch.bool, device=DEV)
    src = torch.full((P * I,), -1, dtype=torch.int32, device=DEV)
    out = xin if in_place else torch.full((P * I, 3), -1.0, device=DEV)
    nb = int(lib.ia_deform_filter_compact_tmp_bytes(L.i64(P)))
    tmp2 = torch.full(((nb + 7) // 8,), -1, dtype=torch.int64, device=DEV)        # the entry point clears its own scratch
    for _ in range(2):                                                             # twice on the same scratch
        if in_place:
            xin.copy_(x)
        L.check(lib.ia_deform_filter_compact(L.i64(P), L.i32(I), L.ptr(xin), L.ptr(valid), L.ptr(cnt), L.ptr(start), L.ptr(out),
                                             L.ptr(src), L.ptr(mask), L.ptr(tot), L.ptr(tmp2), C.c_size_t(tmp2.numel() * 8), st), "fcc")
    assert int(tot.item()) == Q and 0 < Q < int(valid.sum())
    assert torch.equal(cnt, cnt0) and torch.equal(start, start0) and torch.equal(mask, mask0)
    assert torch.equal(out.reshape(-1, 3)[:Q], cx0) and torch.equal(src[:Q], cs0)
    if not in_place:
        assert torch.equal(xin, x) and bool((out.reshape(-1, 3)[Q:] == -1).all())
    # the two-step form: tile-local packing in place, then the segmented copy
    xin = x.clone()
    cnt.fill_(-1); start.fill_(-1); tot.fill_(-1); mask.zero_()
    src_local = torch.full((P * I,), -1, dtype=torch.int32, device=DEV) if in_place else None       # with / without cand_src
    nb = int(lib.ia_deform_filter_tiles_tmp_bytes(L.i64(P)))
    tmp3 = torch.full(((nb + 7) // 8,), -1, dtype=torch.int64, device=DEV)
    L.check(lib.ia_deform_filter_tiles(L.i64(P), L.i32(I), L.ptr(xin), L.ptr(valid), L.ptr(cnt), L.ptr(start), L.ptr(src_local),
                                       L.ptr(mask), L.ptr(tot), L.ptr(tmp3), C.c_size_t(tmp3.numel() * 8), st), "tiles")
    assert int(tot.item()) == Q and torch.equal(cnt, cnt0) and torch.equal(mask, mask0)
    cx = torch.full((Q, 3), -1.0, device=DEV)
    cs = torch.full((Q,), -1, dtype=torch.int32, device=DEV) if in_place else None
    L.check(lib.ia_deform_pack_tiles(L.i64(P), L.i32(I), L.ptr(xin), L.ptr(src_local), L.ptr(start), L.ptr(cx), L.ptr(cs), L.ptr(tmp3), st),
            "pack")
    assert torch.equal(start, start0) and torch.equal(cx, cx0) and (cs is None or torch.equal(cs, cs0))


def test_foreground_compaction_equals_the_reference_op_sequence(ops):
    """lib_nerfacc.compact_foreground (ia_fg_count / ia_fg_compact) == unpack_info + three boolean-mask gathers + pack_info
    (models/intrinsic_avatar.py:516-528) on K4-shaped data: rays with 0 or 4 re-samples, any subset of them foreground."""
    LN = ops["lib"]
    rng = np.random.default_rng(4)
    n = 50_000
    cnt = np.where(rng.random(n) < 0.7, 4, 0).astype(np.int32)
    rpi = np.stack([np.cumsum(cnt) - cnt, cnt], -1).astype(np.int32)
    Tn = int(cnt.sum())
    starts, ends = rng.random((Tn, 1), dtype=np.float32), rng.random((Tn, 1), dtype=np.float32)
    is_fg = rng.random(Tn) < 0.4
    is_fg[: 4 * 10] = False                                              # leading rays without any foreground interval
    ri, ts, te, pinfo = LN.compact_foreground(T(rpi), T(starts), T(ends), T(is_fg))
    rri = LN.unpack_info(T(rpi), Tn)
    m = T(is_fg)
    assert torch.equal(ri, rri[m]) and torch.equal(ts, T(starts)[m, 0]) and torch.equal(te, T(ends)[m, 0])
    assert torch.equal(pinfo, LN.pack_info(rri[m], n))
    e = LN.compact_foreground(T(rpi), T(starts), T(ends), T(np.zeros(Tn, bool)))
    assert e[0].numel() == 0 and int(e[3][:, 1].sum()) == 0


def test_unpack_info_segmented_fill_on_long_unordered_and_empty_segments():
    """K5 as a wave-level segmented fill (csrc/resample.hip): the shape lib.nerfacc.unpack_info sees from the volume-interaction
    re-sampling (1024 entries per non-empty ray, models/pbr/utils.py:113-135), empty rays, ragged counts, and segments that are not in
    ray order -- against the definition (pack.cu:7-28: ray i owns [start_i, start_i + count_i))."""
    from intrinsicavatar_amd import lib_nerfacc as lib
    rng = np.random.default_rng(11)
    for n_rays, make in ((5000, lambda r: np.where(r.random(5000) < 0.4, 1024, 0)), (70001, lambda r: r.integers(0, 40, 70001)),
                         (129, lambda r: r.integers(0, 3000, 129)), (1, lambda r: np.array([77]))):
        cnt = make(rng).astype(np.int64)
        perm = rng.permutation(n_rays)                                  # segment order != ray order
        start = np.zeros(n_rays, np.int64)
        start[perm] = np.cumsum(cnt[perm]) - cnt[perm]
        S_ = int(cnt.sum())
        want = np.full(S_, -1, np.int64)
        for i in np.nonzero(cnt)[0]:
            want[start[i]:start[i] + cnt[i]] = i
        pi = torch.from_numpy(np.stack([start, cnt], 1).astype(np.int32)).to(DEV)
        got = lib.unpack_info(pi, S_)
        assert got.dtype == torch.int64 and np.array_equal(N(got), want), n_rays
        m = lib._unpack_info_to_mask(pi, int(cnt.max()))
        assert np.array_equal(N(m), np.arange(int(cnt.max()))[None, :] < cnt[:, None])
