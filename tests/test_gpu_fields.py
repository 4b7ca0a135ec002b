"""GPU: neural-field kernels (hash grid, SH, fused MFMA MLPs) vs the CPU oracle / reference modules.
Floating-point kernels => tolerance (stated per test); fp32 everywhere (reference: trainer.precision 32)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def N(t):
    return t.detach().cpu().numpy()


@pytest.fixture(scope="module")
def fields():
    from intrinsicavatar_amd import build
    build.build()
    from intrinsicavatar_amd import fields
    return fields


def _params(oracle, seed=0, amp=1e-1):
    total, offs, res, sc = oracle.hashgrid_offsets()
    rng = np.random.default_rng(seed)
    return (rng.uniform(-amp, amp, total * 2)).astype(np.float32), total


def test_hashgrid_layout_matches_instant_ngp(oracle, fields):
    total, offs, res, sc = oracle.hashgrid_offsets()
    assert list(res[:5]) == [16, 24, 34, 49, 71]
    assert total == fields.hash_n_entries() and abs(total * 2 * 4 / 1e6 - 50.4) < 0.3
    assert all(int(offs[l + 1] - offs[l]) == 1 << 19 for l in range(5, 16))


def test_hashgrid_fwd_and_jacobian_vs_oracle(oracle, fields):
    params, _ = _params(oracle)
    rng = np.random.default_rng(1)
    x = rng.random((20000, 3)).astype(np.float32)
    x[:8] = [[0, 0, 0], [1, 1, 1], [0.5, 0.5, 0.5], [0, 1, 0], [1, 0, 0], [0.999999, 0.5, 0.25], [1e-7, 0.3, 0.9], [0.25, 0.25, 0.25]]
    ref, jref = oracle.hashgrid_fwd(x, params, with_jac=True)
    enc, jac = fields.hashgrid_forward(T(x), T(params), with_jac=True)
    np.testing.assert_allclose(N(enc), ref, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(N(jac), jref, rtol=1e-4, atol=2e-4)      # |J| ~ scale(4096) * amp
    # known-answer: dense level 0 == trilinear interpolation of its 16^3 lattice
    total, offs, res, sc = oracle.hashgrid_offsets()
    lat = params[: 16 ** 3 * 2].reshape(16, 16, 16, 2)                     # index = x + y*16 + z*256
    sel = np.where((x < 0.9).all(1))[0][:100]
    p = x[sel] * sc[0] + 0.5
    i0 = np.floor(p).astype(int)
    f = p - i0
    tri = np.zeros((100, 2), np.float32)
    for c in range(8):
        o = np.array([(c >> 0) & 1, (c >> 1) & 1, (c >> 2) & 1])
        w = np.prod(np.where(o, f, 1 - f), axis=1)
        idx = i0 + o
        tri += w[:, None] * lat[idx[:, 2], idx[:, 1], idx[:, 0]]
    np.testing.assert_allclose(N(enc)[sel, :2], tri, rtol=1e-4, atol=1e-6)


def test_hashgrid_level_major_gather_equals_the_flat_kernel_bit_for_bit(oracle, fields, monkeypatch):
    """ia_hashgrid_fwd_xcd (one table at a time, straight-line two-point gather: aligned 16-byte pair loads + unconditional 8-byte
    loads) against the flat kernel (one lane per (point, level), plain 8-byte gathers through grid_index): features and Jacobian
    bit for bit, including points ON the cube's faces and points OUTSIDE the unit cube, whose wrapped cell coordinates need
    tiny-cuda-nn's full `index % hashmap_size` on the dense levels; ragged size (odd point count)."""
    params, _ = _params(oracle)
    rng = np.random.default_rng(7)
    n = fields.HASH_FWD_XCD_MIN + 4321
    x = rng.random((n, 3)).astype(np.float32)
    x[:8] = [[0, 0, 0], [1, 1, 1], [0.5, 0.5, 0.5], [0, 1, 0], [1, 0, 0], [0.999999, 0.5, 0.25], [1e-7, 0.3, 0.9], [0.25, 0.25, 0.25]]
    x[1000:6000] = (rng.random((5000, 3)) * 1.8 - 0.4).astype(np.float32)          # outside the cube on some axes
    x[6000:6100] = (rng.random((100, 3)) * 40.0 - 20.0).astype(np.float32)         # far outside
    out = {}
    for m in ("flat", "xcd"):
        monkeypatch.setenv("IA_HASH_FWD", m)
        e, j = fields.hashgrid_forward(T(x), T(params), with_jac=True)
        e2 = fields.hashgrid_forward(T(x), T(params), with_jac=False)
        assert torch.equal(e, e2)
        out[m] = (e, j)
    monkeypatch.delenv("IA_HASH_FWD")
    assert torch.equal(out["flat"][0], out["xcd"][0])
    assert torch.equal(out["flat"][1], out["xcd"][1])


def test_hashgrid_bwd_vs_oracle(oracle, fields):
    params, total = _params(oracle)
    rng = np.random.default_rng(2)
    x = rng.random((5000, 3)).astype(np.float32)
    g = rng.normal(size=(5000, 32)).astype(np.float32)
    ref = oracle.hashgrid_bwd_params(x, g, total * 2)
    gp = torch.zeros(total * 2, device=DEV)
    fields.hashgrid_backward(T(x), T(g), gp)
    np.testing.assert_allclose(N(gp), ref, rtol=1e-4, atol=1e-5)            # atomics: order-dependent rounding


@pytest.mark.parametrize("second", [False, True])
def test_hashgrid_bwd_binned_equals_atomic_scatter(oracle, fields, second):
    """the multisplit + LDS-reduction backward computes the same table gradient as the atomic scatter (and as the
    oracle): ray-ordered points (run merging active), masked levels, accumulation into a non-zero table, ragged n."""
    params, total = _params(oracle)
    rng = np.random.default_rng(7)
    n_r, per = 1237, 37
    o = rng.random((n_r, 1, 3)) * 0.6 + 0.2
    d = rng.normal(size=(n_r, 1, 3)); d /= np.linalg.norm(d, axis=-1, keepdims=True)
    x = np.clip(o + d * (np.arange(per)[None, :, None] * 0.004), 0, 1).reshape(-1, 3).astype(np.float32)
    n = x.shape[0]
    g = rng.normal(size=(n, 32)).astype(np.float32)
    g[:, 24:] = 0.0                                                          # progressive mask: finest 4 levels off
    gj = rng.normal(size=(n, 32)).astype(np.float32) if second else None
    q = rng.normal(size=(n, 3)).astype(np.float32) if second else None
    init = rng.normal(size=total * 2).astype(np.float32) * 0.1
    ga, gb = T(init.copy()), T(init.copy())
    kw = dict(g_jac=T(gj), q=T(q)) if second else {}
    fields.hashgrid_backward(T(x), T(g), ga, method="atomic", **kw)
    fields.hashgrid_backward(T(x), T(g), gb, method="binned", **kw)
    gc, gd = T(init.copy()), T(init.copy())
    fields.hashgrid_backward(T(x), T(g), gc, method="binned", level_mask=0x0FFF, **kw)      # caller-side band mask: same result
    fields.hashgrid_backward(T(x), T(g), gd, method="binned", **kw)
    assert torch.equal(gb, gd)                 # integer LDS accumulation: bit-reproducible run to run
    if not second:
        assert torch.equal(gb, gc)
    da, db = N(ga) - init, N(gb) - init
    scale = np.abs(da).max()
    assert np.abs(da - db).max() < 2e-5 * scale + 1e-5, np.abs(da - db).max()
    assert np.all((db != 0) == (da != 0)) or np.abs(db[(db != 0) != (da != 0)]).max() < 1e-6
    if not second:
        ref = oracle.hashgrid_bwd_params(x, g, total * 2)
        np.testing.assert_allclose(db, ref, rtol=1e-3, atol=2e-5 * scale)


def test_hashgrid_second_order_bwd_vs_autograd(oracle, fields):
    """d/dparams of <g_jac, J q> must equal the scatter the kernel does (double-backward path);
    checked against finite-difference-free autograd on a torch restatement of ONE dense level."""
    params, total = _params(oracle, seed=4)
    rng = np.random.default_rng(3)
    n = 300
    x = rng.random((n, 3)).astype(np.float32)
    gj = rng.normal(size=(n, 32)).astype(np.float32)
    q = rng.normal(size=(n, 3)).astype(np.float32)
    gp = torch.zeros(total * 2, device=DEV)
    fields.hashgrid_backward(T(x), None, gp, g_jac=T(gj), q=T(q))
    # linearity in params: L(params) = sum_{i,k} gj[i,k] * (J(params)[i,k,:] . q[i]);  dL/dparams . params == L
    _, jac = fields.hashgrid_forward(T(x), T(params), with_jac=True)
    Lval = float((T(gj).double() * (jac.double() * T(q).double()[:, None, :]).sum(-1)).sum())
    assert abs(float((gp.double() * T(params).double()).sum()) - Lval) < 1e-3 * max(1.0, abs(Lval))


def test_sh4_vs_oracle_and_closed_form(oracle, fields):
    rng = np.random.default_rng(5)
    d = rng.normal(size=(4096, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d01 = (d + 1) / 2
    out = N(fields.sh4(T(d01)))
    np.testing.assert_allclose(out, oracle.sh4(d01), rtol=1e-5, atol=1e-6)
    # orthonormality of the 16 real SH over the sphere (Monte-Carlo): E[Y_i Y_j] * 4pi ~ delta_ij
    G = out.T.astype(np.float64) @ out.astype(np.float64) / d.shape[0] * 4 * np.pi
    assert np.abs(G - np.eye(16)).max() < 0.15


def _weight_norm(g, v):
    return g * v / np.linalg.norm(v.astype(np.float64), axis=1, keepdims=True).astype(np.float32)


def test_mlps_vs_reference_modules(fields, golden_dir):
    """the three MLP shapes against outputs of the reference's own VanillaMLP / LipshitzMLP modules."""
    g = np.load(os.path.join(golden_dir, "golden_mlp.npz"))
    # SDF net: reference input order [xyz(3) | hash(32)]; kernel order [hash | xyz]
    W1 = _weight_norm(g["sdf_sd_layers.0.weight_g"], g["sdf_sd_layers.0.weight_v"])
    W2 = _weight_norm(g["sdf_sd_layers.2.weight_g"], g["sdf_sd_layers.2.weight_v"])
    x = g["sdf_x"]
    W1k = np.concatenate([W1[:, 3:], W1[:, :3]], 1)
    enc, xyz = T(x[:, 3:]), T(x[:, :3])
    y = fields.mlp_forward(0, [(enc, 32, 1.0, 0.0), (xyz, 3, 1.0, 0.0)], T(W1k), T(g["sdf_sd_layers.0.bias"]), None, None,
                           T(W2), T(g["sdf_sd_layers.2.bias"]), 13)
    np.testing.assert_allclose(N(y), g["sdf_y"], rtol=2e-5, atol=5e-6)
    # radiance net (sigmoid applied by the kernel; reference applies it outside: radiance.py:132-133)
    x = T(g["rad_x"])
    # the kernel takes the 67-wide row as its five sources (as the render path does)
    segs = [(x[:, :32].contiguous(), 32, 1.0, 0.0), (x[:, 32:35].contiguous(), 3, 1.0, 0.0), (x[:, 35:48].contiguous(), 13, 1.0, 0.0),
            (x[:, 48:64].contiguous(), 16, 1.0, 0.0), (x[:, 64:67].contiguous(), 3, 1.0, 0.0)]
    y = fields.mlp_forward(1, segs, *[T(g[f"rad_sd_layers.{i}.{k}"]) for i in (0, 2, 4) for k in ("weight", "bias")], 3)
    ref = 1 / (1 + np.exp(-g["rad_y"].astype(np.float64)))
    np.testing.assert_allclose(N(y), ref, rtol=2e-5, atol=2e-6)
    # strided (non-vectorisable) sources give the same result as the 16-byte-load path
    xs = torch.zeros((x.shape[0], 71), device=DEV)
    xs[:, 1:68] = x
    segs2 = [(xs[:, 1:33], 32, 1.0, 0.0), (xs[:, 33:36], 3, 1.0, 0.0), (xs[:, 36:49], 13, 1.0, 0.0), (xs[:, 49:65], 16, 1.0, 0.0),
             (xs[:, 65:68], 3, 1.0, 0.0)]
    y2 = fields.mlp_forward(1, segs2, *[T(g[f"rad_sd_layers.{i}.{k}"]) for i in (0, 2, 4) for k in ("weight", "bias")], 3)
    assert torch.equal(y, y2)
    # material net (LipshitzMLP normalisation folded on the host: network_utils.py:396-403)
    Ws, bs = [], []
    for i in range(3):
        w = g[f"mat_sd_weights_per_layer.{i}"]
        c = float(g[f"mat_sd_lipshitz_bound_per_layer.{i}"][0])
        sp = np.log1p(np.exp(c)) if c < 20 else c
        Ws.append((w * np.minimum(sp / np.abs(w).sum(1), 1.0)[:, None]).astype(np.float32))
        bs.append(g[f"mat_sd_biases_per_layer.{i}"])
    xm = T(g["mat_x"])
    msegs = [(xm[:, :32].contiguous(), 32, 1.0, 0.0), (xm[:, 32:35].contiguous(), 3, 1.0, 0.0), (xm[:, 35:48].contiguous(), 13, 1.0, 0.0)]
    y = fields.mlp_forward(2, msegs, T(Ws[0]), T(bs[0]), T(Ws[1]), T(bs[1]), T(Ws[2]), T(bs[2]), 5)
    ref = 1 / (1 + np.exp(-g["mat_y"].astype(np.float64)))
    np.testing.assert_allclose(N(y), ref, rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("n", [1, 63, 64, 65, 1000, 70001])
def test_sdf_field_vs_oracle(oracle, fields, n):
    """VolumeSDF (hash grid + xyz -> MLP) incl. the analytic normal, ragged sizes around the 64-point tile."""
    geo = fields.VolumeSDF(seed=0).to(DEV)
    with torch.no_grad():
        geo.grid_params.copy_((torch.rand(geo.grid_params.shape, generator=torch.Generator().manual_seed(1)) * 2 - 1) * 0.05)
        for p in geo.network.parameters():
            p.add_(torch.randn(p.shape, generator=torch.Generator().manual_seed(2)).to(DEV) * 0.02)
    bbox = torch.tensor([[-0.9, -1.2, -0.3], [0.9, 0.8, 0.3]], device=DEV)
    geo.prepare_bbox(bbox)
    geo.update_step(0, 1500)            # progressive mask: 4 + (1500-500)//125 = 12 levels active
    rng = np.random.default_rng(n)
    pts = (rng.random((n, 3)) * (N(bbox[1]) - N(bbox[0])) + N(bbox[0])).astype(np.float32)
    sdf, grad, feat = geo(T(pts), with_grad=True, with_feature=True)
    l0, l2 = geo.network.layers[0], geo.network.layers[2]
    mask = N(geo.prog.mask(1500, "cpu"))
    assert mask.sum() == 24
    sr, gr, fr = oracle.sdf_field(pts, N(geo.center), N(geo.scale), N(geo.grid_params), mask, N(l0.effective()),
                                  N(l0.bias), N(l2.effective()), N(l2.bias))
    np.testing.assert_allclose(N(sdf), sr, rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(N(feat), fr, rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(N(grad), gr, rtol=2e-3, atol=2e-3)           # |grad| ~ 1..50, hash J ~ 4096*0.05
    sdf2 = geo(T(pts), with_grad=False, with_feature=False)
    np.testing.assert_allclose(N(sdf2), sr, rtol=1e-4, atol=2e-5)


def test_tcnn_encoding_dropin_first_and_second_order(fields):
    """tinycudann.Encoding drop-in: forward, d/dparams, d/dx, and the DOUBLE backward the reference triggers with
    torch.autograd.grad(sdf, x, create_graph=True) (rf/geometry.py:165-172) -- vs float64 torch autograd."""
    from intrinsicavatar_amd import tinycudann as tcnn
    from tests import torch_ref as TR
    cfg = dict(otype="HashGrid", n_levels=16, n_features_per_level=2, log2_hashmap_size=19, base_resolution=16,
               per_level_scale=1.447269237440378, interpolation="Linear")
    enc = tcnn.Encoding(3, cfg, dtype=torch.float32).to(DEV)
    assert enc.n_output_dims == 32 and enc.params.numel() == fields.hash_n_entries() * 2
    with torch.no_grad():
        enc.params.mul_(300.0)                              # amplitude 3e-2 so that gradients are well above fp32 noise
    g = torch.Generator().manual_seed(0)
    n = 2000
    x = torch.rand((n, 3), generator=g).to(DEV).requires_grad_(True)
    W = torch.randn((32, 4), generator=g).to(DEV)
    y = enc(x)
    s = torch.tanh(y @ W).sum(-1)                           # a small "network" on top
    gx = torch.autograd.grad(s, x, torch.ones_like(s), create_graph=True)[0]
    loss = ((gx.norm(dim=-1) - 1.0) ** 2).mean() + y.pow(2).mean()      # eikonal-like term + a first-order term
    loss.backward()
    # reference
    x64 = x.detach().cpu().double().requires_grad_(True)
    p64 = enc.params.detach().cpu().double().requires_grad_(True)
    y64 = TR.hashgrid(x64, p64.reshape(-1, 2))
    s64 = torch.tanh(y64 @ W.cpu().double()).sum(-1)
    gx64 = torch.autograd.grad(s64, x64, torch.ones_like(s64), create_graph=True)[0]
    loss64 = ((gx64.norm(dim=-1) - 1.0) ** 2).mean() + y64.pow(2).mean()
    loss64.backward()
    # fp32 positions at scale 4096 carry ~2e-4 cell units of rounding vs the fp64 reference: |err| ~ amp * 2e-4
    np.testing.assert_allclose(N(y), y64.detach().numpy(), rtol=1e-3, atol=2e-5)
    gref = gx64.detach().numpy()
    # the input gradient is piecewise constant per cell: a point within fp32 rounding (~2e-4 cells at level 15) of a
    # cell face takes the neighbouring cell's slope in fp32 vs fp64 -> a few outlier points, the rest agrees tightly
    perr = np.abs(N(gx) - gref).max(-1) / np.abs(gref).max()
    assert (perr < 2e-3).mean() > 0.95, float((perr < 2e-3).mean())
    assert abs(float(loss) - float(loss64)) < 1e-3 * abs(float(loss64))
    a, b = enc.params.grad.cpu().double(), p64.grad
    assert float(b.norm()) > 0 and float((a - b).norm() / b.norm()) < 2e-2
    # SH drop-in incl. input gradient
    sh = tcnn.Encoding(3, dict(otype="SphericalHarmonics", degree=4))
    d = torch.rand((500, 3), generator=g).to(DEV).requires_grad_(True)
    (sh(d) * torch.arange(16, device=DEV)).sum().backward()
    d64 = d.detach().cpu().double().requires_grad_(True)
    (TR.sh4(d64) * torch.arange(16).double()).sum().backward()
    np.testing.assert_allclose(N(d.grad), d64.grad.numpy(), rtol=1e-4, atol=1e-4)
    assert tcnn.free_temporary_memory() is None


def test_sdf_only_query_path_equals_the_general_one(fields):
    """geometry.sdf_only (level-major hash result -> software-pipelined value head) and deformer.deform_sdf against the general
    forward / deform on the same points: the value head sums the output layer in another order (lane partials + LDS reduce instead
    of an MFMA tile) and evaluates Softplus without the threshold pass-through: 2e-6 absolute (SDF in metres); and bit-identical to
    ITSELF for every batch size / order (what ray-batch sharding relies on)."""
    from intrinsicavatar_amd import synthetic as S
    rs, rays, _ = S.build_frame(DEV, 96, 96, pose_seed=0, beta=0.01, num_samples_per_ray=64, grid_D=16, grid_H=64, grid_W=64,
                                smooth_iters=5, hash_amp=1e-2)
    geo = rs.geometry
    g = torch.Generator().manual_seed(0)
    for n in (fields.HASH_FWD_XCD_MIN + 37, 5, 33, 64 * 12 * 3 + 1):     # many tiles per wave / one ragged tile / two / exactly past a grid round
        x = (geo.center + (torch.rand((n, 3), generator=g).to(DEV) - 0.5) * geo.scale).contiguous()
        a = geo.sdf_only(x)
        b = geo.forward(x, with_grad=False, with_feature=False)
        assert float((a - b).abs().max()) < 2e-6, float((a - b).abs().max())
        k = n // 3
        assert torch.equal(geo.sdf_only(x[k:].contiguous()), a[k:])      # the value of a point does not depend on the batch
    r = rs.deformer.transform_rays_w2s(rays.float())
    t = torch.rand((rays.shape[0], 40), generator=g).to(DEV) * 2.0 + 4.0
    pts = (r[:, None, :3] + r[:, None, 3:6] * t[..., None]).reshape(-1, 3).contiguous()
    assert pts.shape[0] > fields.HASH_FWD_XCD_MIN
    d = rs.deformer.deform(pts, geo)
    s = rs.deformer.deform_sdf(pts, geo)
    assert float((s - d["sdf"]).abs().max()) < 2e-6 and int((s < 1e5).sum()) > 1000
    # spatial ordering of big batches does not change the values either
    old = rs.SORT_MIN_POINTS
    try:
        rs.SORT_MIN_POINTS = 1000
        s2 = rs._sdf_at(pts)
        rs.SORT_DROP_BITS = 6
        s3 = rs._sdf_at(pts)
    finally:
        rs.SORT_MIN_POINTS = old
        rs.SORT_DROP_BITS = 0
    assert torch.equal(s2, s) and torch.equal(s3, s)


def test_sdf_value_head_variants_agree(fields, monkeypatch):
    """ia_sdf_levels_fwd: the default kernel (operand-major LDS tiles, clamped loads, Softplus on 100 log2(e)-scaled weights) against
    its predecessors ("pipe12": bit-identical to "pipe8"; "tile": the one-tile-per-wave MFMA kernel) and the general forward, on
    ragged sizes: one point, around one and two tiles, around one round of the persistent grid, many tiles per wave."""
    from intrinsicavatar_amd import synthetic as S
    rs, _, _ = S.build_frame(DEV, 32, 32, pose_seed=0, beta=0.01, num_samples_per_ray=16, grid_D=16, grid_H=64, grid_W=64,
                             smooth_iters=5, hash_amp=1e-2)
    geo = rs.geometry
    g = torch.Generator().manual_seed(1)
    for n in (1, 31, 32, 33, 63, 65, 256 * 12 * 32 - 1, 256 * 12 * 32 + 1, 3_000_017):
        x = (geo.center + (torch.rand((n, 3), generator=g).to(DEV) - 0.5) * geo.scale).contiguous()
        out = {}
        for v in ("pipe2", "pipe2w12", "pipe12", "pipe8", "tile"):
            monkeypatch.setenv("IA_SDF_HEAD", v)
            out[v] = geo.sdf_only(x).clone()
        monkeypatch.delenv("IA_SDF_HEAD")
        assert torch.equal(geo.sdf_only(x), out["pipe2"])                           # the default
        assert torch.equal(out["pipe12"], out["pipe8"]) and torch.equal(out["pipe2"], out["pipe2w12"])
        ref = geo.forward(x, with_grad=False, with_feature=False)
        for v in out:
            assert float((out[v] - ref).abs().max()) < 2e-6, (n, v)
        assert float((out["pipe2"] - out["pipe12"]).abs().max()) < 1e-6


def test_sdf_head_with_gradient_on_level_major_results_equals_the_row_major_path(fields, monkeypatch):
    """big batches evaluate VolumeSDF.forward(with_grad=True) from the level-major gather results (ia_hashgrid_fwd_levels +
    ia_sdf_levels_fwd_grad: no [n,32] rows, no [n,32,3] Jacobian): sdf, 13 features and the analytic gradient equal the flat gather
    + row-major head (same arithmetic in the same order; ragged last tile)."""
    from intrinsicavatar_amd import synthetic as S
    rs, rays, _ = S.build_frame(DEV, 96, 96, pose_seed=0, beta=0.01, num_samples_per_ray=64, grid_D=16, grid_H=64, grid_W=64,
                                smooth_iters=5, hash_amp=1e-2)
    geo = rs.geometry
    g = torch.Generator().manual_seed(3)
    n = fields.HASH_FWD_XCD_MIN + 1237
    x = (geo.center + (torch.rand((n, 3), generator=g).to(DEV) - 0.5) * geo.scale).contiguous()
    a = geo.forward(x, with_grad=True, with_feature=True)
    monkeypatch.setenv("IA_SDF_GRAD_LEVELS", "0")
    b = geo.forward(x, with_grad=True, with_feature=True)
    for u, v, name in zip(a, b, ("sdf", "grad", "feature")):
        assert u.shape == v.shape
        assert torch.allclose(u, v, rtol=1e-6, atol=1e-7), (name, float((u - v).abs().max()))
    assert float(a[1].abs().max()) > 0.1


@pytest.mark.parametrize("drop_bits", [0, 3, 6])
def test_morton_order_is_a_stable_key_sort(drop_bits):
    """ia_morton_order: a permutation that sorts the bits [drop_bits, 30) of the Morton keys, stable (equal keys keep their
    input order) -- checked against torch.sort(stable=True) of ia_morton_keys' codes; int32 gather / scatter round trip."""
    import ctypes as C
    from intrinsicavatar_amd import _lib as L
    lib, st = L.lib(), L.stream()
    n = 3_000_017
    g = torch.Generator(device=DEV).manual_seed(drop_bits)
    pts = (torch.rand(n, 3, device=DEV, generator=g) * 3.0 - 1.5).contiguous()
    pts[::5] = pts[1::5][: pts[::5].shape[0]]                             # many equal keys
    origin = (C.c_float * 3)(-2.0, -2.0, -2.0)
    keys = torch.empty(n, dtype=torch.int32, device=DEV)
    L.check(lib.ia_morton_keys(L.i64(n), L.ptr(pts), origin, L.f32(100.0), L.ptr(keys), st), "keys")
    want = torch.sort(keys >> drop_bits, stable=True)[1]
    order = torch.empty(n, dtype=torch.int32, device=DEV)
    nb = int(lib.ia_morton_order_tmp_bytes(L.i64(n)))
    tmp = torch.empty(nb, dtype=torch.uint8, device=DEV)
    L.check(lib.ia_morton_order(L.i64(n), L.ptr(pts), origin, L.f32(100.0), L.i32(drop_bits), L.ptr(order), L.ptr(tmp), C.c_size_t(nb), st),
            "ia_morton_order")
    assert torch.equal(order.long(), want)
    ps = torch.empty_like(pts)
    L.check(lib.ia_gather_rows3_i32(L.i64(n), L.ptr(pts), L.ptr(order), L.ptr(ps), st), "gather")
    assert torch.equal(ps, pts[want])
    back = torch.empty(n, device=DEV)
    L.check(lib.ia_scatter_f32_i32(L.i64(n), L.ptr(ps[:, 0].contiguous()), L.ptr(order), L.ptr(back), st), "scatter")
    assert torch.equal(back, pts[:, 0])


def test_morton_order_ranks_by_lds_atomics_on_this_device_and_the_ballot_path_agrees():
    """csrc/sort.hip ranks the 64 elements of a wave instruction by one LDS atomic-with-return per lane, which is only stable if the
    LDS serves the lanes of one instruction in ascending order -- the library checks that on the device before its first sort
    (ia_sort_rank_mode() == 1 on gfx950) and otherwise ranks by ballots; IA_SORT_RANK=ballot forces that path (a fresh process:
    the mode is decided once), which must produce the same permutation."""
    import subprocess
    import sys
    code = r'''
import ctypes as C, torch, sys
from intrinsicavatar_amd import _lib as L
lib, st = L.lib(), L.stream()
n = 2_500_011
g = torch.Generator(device="cuda:0").manual_seed(5)
pts = (torch.rand(n, 3, device="cuda:0", generator=g) * 3.0 - 1.5).contiguous()
pts[::3] = pts[1::3][: pts[::3].shape[0]]
origin = (C.c_float * 3)(-2.0, -2.0, -2.0)
keys = torch.empty(n, dtype=torch.int32, device="cuda:0")
L.check(lib.ia_morton_keys(L.i64(n), L.ptr(pts), origin, L.f32(100.0), L.ptr(keys), st), "keys")
order = torch.empty(n, dtype=torch.int32, device="cuda:0")
nb = int(lib.ia_morton_order_tmp_bytes(L.i64(n)))
tmp = torch.empty(nb, dtype=torch.uint8, device="cuda:0")
L.check(lib.ia_morton_order(L.i64(n), L.ptr(pts), origin, L.f32(100.0), L.i32(0), L.ptr(order), L.ptr(tmp), C.c_size_t(nb), st), "order")
assert torch.equal(order.long(), torch.sort(keys, stable=True)[1])
print("MODE", int(lib.ia_sort_rank_mode()))
'''
    import os
    modes = {}
    for rank in ("", "ballot"):
        env = dict(os.environ, IA_SORT_RANK=rank)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600,
                           cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        assert r.returncode == 0, r.stderr[-2000:]
        modes[rank] = int(r.stdout.strip().split("MODE")[-1])
    assert modes == {"": 1, "ballot": 2}, modes
