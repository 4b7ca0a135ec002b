"""GPU (MI355X): the per-step operators of csrc/stepops.hip (and the single-launch scan of csrc/core.hip) against the torch expressions
they replace -- forward values and, for the differentiable ones, the gradients torch's autograd gives for the same expression.
These kernels exist to take launches off the host-bound 4096-ray training step; their arithmetic follows the torch
chain operation by operation, so the bars are a few ulp (reductions are summed in another order)."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module", autouse=True)
def _built():
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    from intrinsicavatar_amd import build
    build.build()


def _close(a, b, rtol, atol, what=""):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    err = (a - b).abs()
    assert bool((err <= atol + rtol * b.abs()).all()), (what, float(err.max()), float(b.abs().max()))


@pytest.mark.parametrize("dtype", [torch.int32, torch.int64])
@pytest.mark.parametrize("n", [0, 1, 5, 1024, 4097, 131072, 131073, 1_000_003])
def test_exclusive_scan_small_and_tiled_paths(n, dtype):
    """n <= 2^15: one workgroup, one launch (scan_small_kernel); above: the tiled protocol.  Values, the total, and in-place use."""
    from intrinsicavatar_amd import _lib as L
    g = torch.Generator().manual_seed(n)
    x = torch.randint(0, 7, (n,), generator=g).to(dtype).to(DEV)
    ref = torch.cumsum(x, 0) - x
    fn = L.lib().ia_exclusive_scan_i32 if dtype == torch.int32 else L.lib().ia_exclusive_scan_i64
    out, total = torch.empty_like(x), torch.full((1,), -7, dtype=dtype, device=DEV)
    L.check(fn(L.ptr(x), L.ptr(out), L.ptr(total), L.i64(n), L.ptr(L.scan_tmp(n, DEV)), L.stream()), "scan")
    assert torch.equal(out, ref) and int(total) == int(x.sum())
    y = x.clone()
    L.check(fn(L.ptr(y), L.ptr(y), L.ptr(None), L.i64(n), L.ptr(L.scan_tmp(n, DEV)), L.stream()), "scan in place")
    assert torch.equal(y, ref)


def test_normalize_points_is_the_torch_expression_bit_for_bit():
    from intrinsicavatar_amd import fields
    g = torch.Generator().manual_seed(0)
    x = (torch.rand((100_003, 3), generator=g) * 3 - 1.5).to(DEV)
    c, s = torch.tensor([0.1, -0.2, 0.05], device=DEV), torch.tensor([2.1, 2.6, 1.9], device=DEV)
    assert torch.equal(fields.normalize_points(x, c, s), (x - c) / s + 0.5)


def test_edge_min_sdf():
    from intrinsicavatar_amd import _lib as L
    g = torch.Generator().manual_seed(1)
    sdf = torch.randn(5001, generator=g).to(DEV)
    il = (torch.rand(5001, generator=g) > 0.3).to(DEV)
    out = torch.empty_like(sdf)
    L.check(L.lib().ia_edge_min_sdf(L.i64(5001), L.ptr(sdf), L.ptr(il), L.ptr(out), L.stream()), "ia_edge_min_sdf")
    nxt = torch.cat([sdf[1:], sdf[-1:]])
    assert torch.equal(out, torch.where(il, torch.minimum(sdf, nxt), torch.full_like(sdf, 1e10)))


def _torch_effective(mode, g, v, src, mul):
    if mode == 1:
        w = g * v / v.norm(dim=1, keepdim=True)
    elif mode == 2:
        w = v * torch.clamp(torch.nn.functional.softplus(g) / v.abs().sum(dim=1), max=1.0)[:, None]
    else:
        w = v
    if src is not None:
        w = w[:, src.long()]
    return w * mul[None] if mul is not None else w


@pytest.mark.parametrize("mode,M,N,perm", [(1, 64, 35, True), (1, 13, 64, False), (0, 64, 67, True), (2, 64, 48, True), (2, 64, 64, False), (2, 5, 64, False)])
def test_effective_weights_forward_and_backward_vs_torch(mode, M, N, perm):
    from intrinsicavatar_amd import fields
    gen = torch.Generator().manual_seed(M * 100 + N + mode)
    v = (torch.randn((M, N), generator=gen) * 0.3).to(DEV).requires_grad_(True)
    if mode == 1:
        g = (torch.rand((M, 1), generator=gen) + 0.5).to(DEV).requires_grad_(True)
    elif mode == 2:          # a bound that clamps about half of the rows
        g = torch.tensor([float(np.log(np.expm1(float(v.detach().abs().sum(1).median()))))], device=DEV, requires_grad=True)
    else:
        g = None
    src = torch.randperm(N, generator=gen).to(torch.int32).to(DEV) if perm else None
    mul = (torch.rand(N, generator=gen) > 0.2).float().to(DEV) * 0.5 + 0.25 if perm else None
    out = fields._EffW.apply(mode, g, v, src, mul)
    ref = _torch_effective(mode, g, v, src, mul)
    _close(out, ref, 3e-6, 1e-7, "forward")
    go = torch.randn((M, N), generator=gen).to(DEV)
    leaves = [t for t in (g, v) if t is not None]
    mine = torch.autograd.grad(out, leaves, go)
    theirs = torch.autograd.grad(ref, leaves, go)
    for a, b, nm in zip(mine, theirs, ("g", "v") if g is not None else ("v",)):
        assert a.shape == b.shape
        _close(a, b, 2e-5, 2e-6 * float(b.abs().max()), "d/d" + nm)
    if mode == 2:
        active = (torch.nn.functional.softplus(g) / v.abs().sum(1) <= 1.0)
        assert 0 < int(active.sum()) < M, "the test must exercise both sides of the clamp"


def test_module_effective_weights_match_the_torch_chain_and_are_cached_per_parameter_epoch():
    """VolumeSDF / VolumeRefDirRadiance / VolumeMaterial.effective_weights (kernel-backed) against the torch chain they replaced;
    no-grad requests are served from the cache until a parameter changes (through torch: _version; through optim.Adam: PARAM_EPOCH)."""
    from intrinsicavatar_amd import fields, optim
    geo, rad, mat = fields.VolumeSDF(seed=0).to(DEV), fields.VolumeRefDirRadiance(seed=1).to(DEV), fields.VolumeMaterial(seed=2).to(DEV)
    with torch.no_grad():
        geo.network.layers[0].weight_v[:, 3:] = torch.randn(64, 32, device=DEV) * 0.02
        for c in mat.network.lipshitz_bound_per_layer:
            c.mul_(0.3)
    for m in (geo, rad):
        m.update_step(0, 1100)          # 8 of 16 levels on
    l0, l2 = geo.network.layers[0], geo.network.layers[2]
    mask = geo.prog.mask(geo.global_step, DEV)
    assert 0 < float(mask.sum()) < 32
    W1 = fields._weight_norm(l0.weight_g, l0.weight_v)
    ref_geo = (torch.cat([W1[:, 3:] * mask[None], W1[:, :3]], 1), l0.bias, fields._weight_norm(l2.weight_g, l2.weight_v), l2.bias)
    for a, b in zip(geo.effective_weights(), ref_geo):
        _close(a, b, 3e-6, 1e-8)
    R1 = rad.network.layers[0].weight
    ref_rad = torch.cat([R1[:, 3:35] * mask[None], R1[:, :3], R1[:, 35:48], R1[:, 48:64] * rad.sh_mask, R1[:, 64:67]], 1)
    assert torch.equal(rad.effective_weights()[0], ref_rad)
    ws = mat.effective_weights(mask)
    for i in range(3):
        w = mat.network.weights_per_layer[i]
        sc = torch.clamp(torch.nn.functional.softplus(mat.network.lipshitz_bound_per_layer[i]) / w.abs().sum(1), max=1.0)
        w = w * sc[:, None]
        if i == 0:
            w = torch.cat([w[:, 3:35] * mask[None], w[:, :3], w[:, 35:48]], 1)
        _close(ws[2 * i], w, 3e-6, 1e-8)
    # gradients flow to the parameters
    loss = sum((t * t).sum() for t in geo.effective_weights()) + sum((t * t).sum() for t in ws)
    loss.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in (l0.weight_g, l0.weight_v, l2.weight_v, mat.network.lipshitz_bound_per_layer[0]))
    # cache: same objects until something changes
    with torch.no_grad():
        a1, a2 = geo.effective_weights(), geo.effective_weights()
        assert all(x is y for x, y in zip(a1, a2))
        l0.bias.add_(1.0)                                   # through torch: _version moves
        a3 = geo.effective_weights()
        assert a3[1] is not a1[1] and torch.equal(a3[1], l0.bias)
        opt = optim.Adam([l0.weight_g], lr=1e-2)
        l0.weight_g.grad = torch.ones_like(l0.weight_g)
        opt.step()                                          # behind torch's back: PARAM_EPOCH moves
        a4 = geo.effective_weights()
        assert a4[0] is not a3[0]
        _close(a4[0], torch.cat([fields._weight_norm(l0.weight_g, l0.weight_v)[:, 3:] * mask[None], fields._weight_norm(l0.weight_g, l0.weight_v)[:, :3]], 1), 3e-6, 1e-8)


def test_sg_image_forward_and_backward_vs_the_torch_expression():
    from intrinsicavatar_amd import pbr
    sg = pbr.EnvironmentLightSG(num_SGs=64, base_res=64, seed=4).to(DEV)
    img = sg.generate_image()
    ref = sg.generate_image_torch()
    assert img.shape == ref.shape == (64, 128, 3)
    _close(img, ref, 2e-5, 2e-6, "image")
    g = torch.Generator().manual_seed(3)
    go = torch.randn((64, 128, 3), generator=g).to(DEV)
    ps = [sg.axis, sg.log_lambda, sg.mu]
    mine = torch.autograd.grad(img, ps, go)
    theirs = torch.autograd.grad(ref, ps, go)
    for a, b, nm in zip(mine, theirs, ("axis", "log_lambda", "mu")):
        _close(a, b, 1e-4, 2e-5 * float(b.abs().max()), nm)


def test_envlight_pdf_tables_vs_the_torch_expression_and_sampling_still_follows_the_pmf():
    from intrinsicavatar_amd import pbr
    from tests.test_gpu_pbr import hdri
    base = torch.from_numpy(hdri(64, 128)).to(DEV)
    e = pbr.EnvironmentLightTensor(base)
    e.update_pdf()
    H, W, _ = base.shape
    sin_t = torch.sin((torch.arange(H, device=DEV) + 0.5) * math.pi / H)[:, None]
    lum = (0.2126 * base[..., 0] + 0.7152 * base[..., 1] + 0.0722 * base[..., 2]).clamp_min(0).double()
    w = lum * sin_t
    pmf = (w / w.sum()).float()
    cdf = torch.cumsum(pmf.reshape(-1).double(), 0)
    _close(e.pmf, pmf, 2e-6, 1e-12, "pmf")
    _close(e._cdf, cdf, 1e-12, 1e-12, "cdf")
    assert abs(float(e._cdf[-1]) - 1.0) < 1e-6 and bool((e._cdf[1:] >= e._cdf[:-1]).all())
    big = pbr.EnvironmentLightTensor(torch.rand((1024, 2048, 3), device=DEV))          # a 1024 x 2048 HDRI: 2048 tiles, same kernels
    big.update_pdf()
    b = big.base.detach()
    wb = (0.2126 * b[..., 0] + 0.7152 * b[..., 1] + 0.0722 * b[..., 2]).clamp_min(0).double() * torch.sin((torch.arange(1024, device=DEV) + 0.5) * math.pi / 1024)[:, None]
    pb = (wb / wb.sum()).float()
    _close(big.pmf, pb, 2e-6, 1e-13, "pmf 1024 x 2048")
    _close(big._cdf, torch.cumsum(pb.reshape(-1).double(), 0), 1e-11, 1e-11, "cdf 1024 x 2048")
    huge = pbr.EnvironmentLightTensor(torch.rand((2048, 4096, 3), device=DEV))         # > PDF_KERNEL_MAX_PIXELS: the torch route
    huge.update_pdf()
    assert abs(float(huge._cdf[-1]) - 1.0) < 1e-5


def test_uniform_sphere_stratified_vs_the_torch_expression():
    from intrinsicavatar_amd import pbr
    g = torch.Generator().manual_seed(5)
    u = torch.rand((512, 2), generator=g).to(DEV)
    dirs, inv_pdf = pbr.uniform_sphere_stratified(16, 32, u)
    i = torch.arange(16, device=DEV).repeat_interleave(32).float()
    j = torch.arange(32, device=DEV).repeat(16).float()
    z = 1.0 - 2.0 * (i + u[:, 0]) / 16
    phi = 2.0 * math.pi * (j + u[:, 1]) / 32
    r = torch.sqrt((1.0 - z * z).clamp_min(0.0))
    ref = torch.stack([r * torch.cos(phi), r * torch.sin(phi), z], -1)
    _close(dirs, ref, 0.0, 3e-7, "dirs")
    assert inv_pdf.shape == (512, 1) and torch.allclose(inv_pdf, torch.full_like(inv_pdf, 4 * math.pi))
    _close(dirs.norm(dim=-1), torch.ones(512), 0.0, 3e-7)


def test_material_affine_forward_and_backward():
    from intrinsicavatar_amd import fields, train_phys
    mat = fields.VolumeMaterial(seed=2)
    g = torch.Generator().manual_seed(6)
    m = torch.rand((10_001, 5), generator=g).to(DEV).requires_grad_(True)
    alb, rgh, mtl = train_phys._MaterialAffine.apply(m, mat)
    assert torch.equal(alb, m[:, :3] * mat.albedo_scale + mat.albedo_bias) and torch.equal(rgh, m[:, 3:4] * mat.roughness_scale + mat.roughness_bias)
    assert torch.equal(mtl, m[:, 4:5] * mat.metallic_scale + mat.metallic_bias)
    ga, gr = torch.randn((10_001, 3), generator=g).to(DEV), torch.randn((10_001, 1), generator=g).to(DEV)
    (gm,) = torch.autograd.grad([alb, rgh], [m], [ga, gr])          # metallic unused: its gradient is zero
    ref = torch.cat([ga * mat.albedo_scale, gr * mat.roughness_scale, torch.zeros_like(gr)], 1)
    assert torch.equal(gm, ref)


@pytest.mark.parametrize("n", [4096, 291_600])
@pytest.mark.parametrize("with_mask", [True, False])
def test_phys_loss_forward_and_backward_vs_the_torch_composition(with_mask, n):
    """train_phys.training_loss_phys: the fused kernel (IA_FUSED_LOSS, default) against the torch chain it replaces."""
    from intrinsicavatar_amd import train, train_phys
    g = torch.Generator().manual_seed(7)
    S = 50_000          # (n = 291 600: the full-frame path with workgroup partials)
    mk = lambda *s: torch.rand(s, generator=g).to(DEV)      # noqa: E731
    leaves = dict(comp_rgb=mk(n, 3), comp_rgb_phys=mk(n, 3) * 1.5, opacity=mk(n, 1) * 1.002 - 0.001, sdf_grad=(mk(S, 3) * 2 - 1) * 1.2)
    for t in leaves.values():
        t.requires_grad_(True)
    out = dict(leaves, valid=(mk(S) > 0.1))
    target, mask = mk(n, 3), ((mk(n) > 0.5).float() if with_mask else None)
    assert bool((leaves["opacity"] < 1e-3).any()) and bool((leaves["opacity"] > 1 - 1e-3).any())          # both sides of the clamp
    loss = train_phys.training_loss_phys(dict(out), target, mask)
    ref = train.training_loss(dict(out), target, mask) + 1.0 * (out["comp_rgb_phys"] - target).abs().mean()
    assert abs(float(loss) - float(ref)) <= 2e-6 * abs(float(ref)), (float(loss), float(ref))
    ks = ["comp_rgb", "comp_rgb_phys", "sdf_grad"] + (["opacity"] if with_mask else [])
    mine = torch.autograd.grad(loss, [leaves[k] for k in ks])
    theirs = torch.autograd.grad(ref, [leaves[k] for k in ks])
    for k, a, b in zip(ks, mine, theirs):
        _close(a, b, 2e-5, 1e-6 * float(b.abs().max()), k)


def test_laplace_alpha_intervals_and_dense_gather_of_secondary_results_are_bit_identical_to_the_two_step_forms():
    from intrinsicavatar_amd import render, pbr
    g = torch.Generator().manual_seed(8)
    n = 200_003
    sdf = (torch.randn(n, generator=g) * 0.05).to(DEV)
    ts = torch.rand(n, generator=g).to(DEV)
    te = ts + torch.rand(n, generator=g).to(DEV) * 0.03
    beta = torch.tensor([0.01], device=DEV)
    assert torch.equal(render.laplace_alpha_intervals(sdf, ts, te, beta), render.laplace_alpha(sdf, te - ts, beta))
    F_ = 50_001
    nrm = torch.nn.functional.normalize(torch.randn((F_, 3), generator=g), dim=-1).to(DEV)
    pos = torch.rand((F_, 3), generator=g).to(DEV)
    dirs = torch.nn.functional.normalize(torch.randn((F_, 3), generator=g), dim=-1).to(DEV)
    ro, rd, src, _ = pbr.secondary_rays(nrm, pos, dirs)
    M = ro.shape[0]
    assert 0 < M < F_
    tr, rgb = (torch.rand((M, 1), generator=g) * 1.2 - 0.1).to(DEV), torch.rand((M, 3), generator=g).to(DEV)
    a_tr, a_rgb = pbr.scatter_secondary(F_, src, tr, rgb)                 # (flag, slot) at hand: written point by point
    b_tr, b_rgb = pbr.scatter_secondary(F_, src.clone(), tr, rgb)         # an index list of unknown origin: zero fill + scatter
    assert torch.equal(a_tr, b_tr) and torch.equal(a_rgb, b_rgb)
    assert float(a_tr.max()) <= 1.0 and float(a_tr.min()) >= 0.0


def test_ray_transform_kernel_is_the_library_product_bit_for_bit():
    """SNARFDeformer.transform_rays_w2s as one launch: origins / directions equal the torch expression's [n,3] x [3,3] products (rocBLAS)
    and numpy's float32 products (the CPU oracle's) bit for bit; near / far equal numpy's."""
    from intrinsicavatar_amd import _lib as L
    g = torch.Generator().manual_seed(12)
    n = 300_007
    A = torch.linalg.qr(torch.randn((3, 3), generator=g))[0]
    w2s = torch.eye(4)
    w2s[:3, :3], w2s[:3, 3] = A, torch.randn(3, generator=g) * 2
    rays = torch.cat([torch.randn((n, 3), generator=g) * 3, torch.nn.functional.normalize(torch.randn((n, 3), generator=g), dim=-1), torch.zeros((n, 2))], 1)

    class D:            # the method only reads .w2s
        pass
    from intrinsicavatar_amd.deformer import SNARFDeformer
    d = D()
    d.w2s = w2s.to(DEV)
    out = SNARFDeformer.transform_rays_w2s(d, rays.to(DEV))
    rg, wg = rays.to(DEV), w2s.to(DEV)
    assert torch.equal(out[:, :3], rg[:, :3] @ wg[:3, :3].T + wg[None, :3, 3]) and torch.equal(out[:, 3:6], rg[:, 3:6] @ wg[:3, :3].T)
    rn, wn = rays.numpy(), w2s.numpy()
    on = rn[:, :3] @ wn[:3, :3].T + wn[None, :3, 3]
    nn = np.linalg.norm(on, axis=-1, keepdims=True)
    ref = np.concatenate([on, rn[:, 3:6] @ wn[:3, :3].T, nn - 1, nn + 1], 1).astype(np.float32)
    assert np.array_equal(out.cpu().numpy(), ref)
