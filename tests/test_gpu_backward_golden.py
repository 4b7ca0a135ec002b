"""GPU (MI355X): the BACKWARD of the HIP path against the reference's own autograd graph.

tests/golden/golden_backward.npz holds d loss / d parameter of the reference's `IntrinsicAvatarSystem.training_step`
(/root/reference/systems/intrinsic_avatar.py:160-301) on its own `IntrinsicAvatarModel.forward` in train() mode, run on CPU in
the build container by tests/golden/make_golden_backward.py (scene, rays and random tensors of golden_forward.npz's train run;
forward bit-identical to that fixture).  Here the same step runs on the HIP path -- `RenderStep.forward_train_` (differentiable:
every forward AND backward is a kernel behind the C ABI) -> the trainer's loss on the returned output dict -> `.backward()` --
and every parameter group's gradient is compared: both hash tables, the weight-normed SDF MLP (weight_g / weight_v / bias), the
radiance MLP, the Lipschitz material MLP (weights, biases, bounds), beta of the Laplace density, and the environment image.

Three loss compositions (make_golden_backward.py): `default` = configs/config.yaml as shipped; `allterms` = every term of
training_step switched on; `lipshitz` = material bounds scaled so that the Lipschitz normalisation is active.

Bars (tests/golden/grad_parity_bars.json): per group (max |diff| / max |ref|, cosine distance) at 3 x what the MI355X showed
(profiles/r05_grad_parity.json: worst group 7e-3 / 1e-5) under hard caps of 2e-2 / 2e-4 (tables: 5e-2 on the single worst entry of
the 87 k compared, 5e-2 on the projection); the loss itself to 2e-5 relative, every logged term to 3e-4.  The 50 MB table gradients are compared through per-level sums / L1 / L2
norms / a pseudo-random projection over ALL entries, the count of touched entries, and exact values on a 1-in-32 subset."""
import json
import os

import numpy as np
import pytest
import torch

from tests import forward_golden as FG

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
HERE = os.path.dirname(os.path.abspath(__file__))
VARIANTS = ("default", "allterms", "lipshitz", "uniform_default")
# the variants also driven through RenderStep.forward_backward_phys ITSELF (render.py: what bench.py and its `config4` object time) with
# the reference's loss composition (loss_config): the shipped training estimator and the `light` estimator
PHYS_VARIANTS = ("uniform_default", "default")


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def N(t):
    return t.detach().cpu().numpy()


@pytest.fixture(scope="module")
def G():
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    from intrinsicavatar_amd import build
    build.build()
    return FG.load()


@pytest.fixture(scope="module")
def B():
    return np.load(os.path.join(HERE, "golden", "golden_backward.npz"))


def lambdas(B, name):
    out = {}
    for s in B[f"{name}_lambdas"]:
        k, v = str(s).split("=", 1)
        out[k] = float(v)
    return out


def bce(x, t):            # systems/criterions.py:229-233
    return -(t * torch.log(x) + (1 - t) * torch.log(1 - x)).mean()


def trainer_loss(out, rgb, alpha, lam, material):
    """IntrinsicAvatarSystem.training_step (systems/intrinsic_avatar.py:160-301) on the output dict of the model: host code of the
    trainer, outside the operator boundary -- plain torch on whatever the model returned.  -> (loss, {term: value})."""
    from intrinsicavatar_amd import pbr
    F_ = torch.nn.functional
    v, vp = out["rays_valid_full"][..., 0], out["rays_valid_phys_full"][..., 0]
    t = {}
    t["rgb_mse"] = F_.mse_loss(out["comp_rgb_full"][v], rgb[v])                                   # :165-178
    t["rgb_l1"] = F_.l1_loss(out["comp_rgb_full"][v], rgb[v])
    t["rgb_phys_mse"] = F_.mse_loss(out["comp_rgb_phys_full"][vp], rgb[vp])                       # :181-212 (add_emitter False)
    t["rgb_phys_l1"] = F_.l1_loss(out["comp_rgb_phys_full"][vp], rgb[vp])
    t["rgb_demodulated"] = F_.l1_loss(pbr.luma(out["comp_demod_phys_full"][vp]), pbr.max_value(rgb[vp]))      # :217-224
    t["eikonal"] = ((torch.linalg.norm(out["sdf_grad_samples"], ord=2, dim=-1) - 1.0) ** 2).mean()             # :235-239
    op = torch.clamp(out["opacity"].squeeze(-1), 1.0e-3, 1.0 - 1.0e-3)                             # :242-252
    t["mask_mse"] = F_.mse_loss(op, alpha)
    t["mask_bce"] = bce(op, alpha)
    t["opaque"] = bce(op, op)                                                                      # :255-257
    t["sparsity"] = torch.exp(-lam["sparsity_scale"] * out["sdf_samples"].abs()).mean()           # :260-264
    reg = material.regularizations(out)                                                            # :285-290, models/pbr/material.py:53-87
    for k in ("normal_orientation", "albedo_smoothness", "roughness_smoothness", "metallic_smoothness", "albedo_entropy"):
        t[k] = reg[k]
    loss = 0.0
    for k, val in t.items():
        if lam.get("lambda_" + k, 0.0) != 0.0:
            loss = loss + lam["lambda_" + k] * val
    return loss, t


def run_step(G, B, name, through="forward_train_"):
    """one training step of variant `name` on the HIP path.  through = "forward_train_": RenderStep.forward_train_ + the test-side
    restatement of the trainer's loss (trainer_loss above) + backward; "forward_backward_phys": RenderStep.forward_backward_phys itself
    with loss_config = the variant's weights (the package's reference_training_loss), which runs its own backward."""
    tag = str(B[f"{name}_run"]) if f"{name}_run" in B.files else FG.TRAIN_RUN
    mode, spp = ("uniform_light", 512) if tag == FG.UNIFORM_TRAIN_RUN else ("light", 16)
    rs, mat, env, rays = FG.gpu_scene(G, tag)
    sc = float(B[f"{name}_lipshitz_scale"])
    if sc != 1.0:
        with torch.no_grad():
            for c in mat.network.lipshitz_bound_per_layer:
                c.mul_(sc)
    rnd = FG.explicit_randoms(G, tag)
    g = torch.Generator().manual_seed(0)
    mj = torch.cat([torch.from_numpy(rnd["material_jitter"]), torch.randn((4096, 3), generator=g)]).to(DEV)
    if mode == "light":          # training form of pbr_light_forward: an independent direction per foreground re-sample, no shuffle
        lu, su = torch.cat([torch.from_numpy(rnd["light_u"]), torch.rand((4096, 3), generator=g)]).to(DEV), None
    else:                        # uniform_light: one stratified set per step, shuffled per ray (:1392-1413)
        lu, su = T(rnd["stratified_u"]), T(rnd["shuffle_u"])
    params = {}
    for comp, mod in (("geometry", rs.geometry), ("radiance", rs.radiance), ("density", rs.density), ("material", mat), ("emitter", env)):
        for k, p in mod.named_parameters():
            if p.requires_grad and p.numel() > 0:
                params[f"{comp}.{k}"] = p
                p.grad = None
    if through == "forward_train_":
        d = rs.forward_train_(rays, mat, env, spp, lu, jitter=T(rnd["near_jitter"]), material_jitter=mj, background_color=T(G["background_color"]),
                              global_illumination=True, render_mode=mode, shuffle_u=su)
        loss, terms = trainer_loss(d, T(B["target_rgb"]), T(B["target_alpha"]), lambdas(B, name), mat)
        loss.backward()
    else:
        o = rs.forward_backward_phys(rays, T(B["target_rgb"]), mat, env, spp, lu, su, target_mask=T(B["target_alpha"]), jitter=T(rnd["near_jitter"]),
                                     render_mode=mode, background_color=T(G["background_color"]), global_illumination=True,
                                     light_sampling="per_point" if mode == "light" else "shared", material_jitter=mj,
                                     loss_config=lambdas(B, name))
        d, loss, terms = o["output_dict"], o["loss"], o["loss_terms"]
    torch.cuda.synchronize()
    return d, loss, terms, params


def group_stats(g, ref):
    g64, r64 = g.astype(np.float64).reshape(-1), ref.astype(np.float64).reshape(-1)
    scale = max(float(np.abs(r64).max()), 1e-30)
    cos = float((g64 * r64).sum() / max(np.linalg.norm(g64) * np.linalg.norm(r64), 1e-300))
    return dict(rel_max=float(np.abs(g64 - r64).max()) / scale, cos_dist=1.0 - cos, ref_max=scale)


def table_stats(g, B, key):
    """the table gradient through the summaries of make_golden_backward.table_gradient_summary."""
    mine = FG.table_gradient_summary(g.reshape(-1))
    ref = {k: B[f"{key}:{k}"] for k in mine}
    l1, l2 = np.maximum(ref["level_l1"], 1e-30), np.maximum(ref["level_l2"], 1e-30)
    st = dict(level_sum=float((np.abs(mine["level_sum"] - ref["level_sum"]) / l1).max()),
              level_l1=float((np.abs(mine["level_l1"] - ref["level_l1"]) / l1).max()),
              level_l2=float((np.abs(mine["level_l2"] - ref["level_l2"]) / l2).max()),
              level_probe=float((np.abs(mine["level_probe"] - ref["level_probe"]) / l2).max()),
              nnz_rel=float((np.abs(mine["level_nnz"] - ref["level_nnz"]) / np.maximum(ref["level_nnz"], 1)).max()))
    mine_at_ref = g.reshape(-1)[ref["sub_index"].astype(np.int64)]
    st.update({"sub_" + k: v for k, v in group_stats(mine_at_ref, ref["sub_value"]).items()})
    e = np.abs(mine_at_ref.astype(np.float64) - ref["sub_value"].astype(np.float64)) / max(float(np.abs(ref["sub_value"]).max()), 1e-30)
    st["sub_rel_p999"] = float(np.quantile(e, 0.999))
    st["sub_over_1e-2"] = int((e > 1e-2).sum())
    st["sub_n"] = int(ref["sub_index"].size)
    return st


def compare(G, B, name, through="forward_train_"):
    d, loss, terms, params = run_step(G, B, name, through)
    report = dict(loss=float(loss), loss_ref=float(B[f"{name}_loss"]), groups={}, tables={}, terms={})
    ref_terms = dict(str(s).split("=", 1) for s in B[f"{name}_loss_terms"])
    alias = dict(rgb_l1="train/loss_rgb", rgb_phys_l1="train/loss_rgb_phys")
    for k, v in terms.items():
        rk = alias.get(k, "train/loss_" + k)
        if rk in ref_terms:
            report["terms"][k] = (float(v), float(ref_terms[rk]))
    names = [str(s) for s in B[f"{name}_grad_names"]]
    assert sorted(names) == sorted(k for k, p in params.items() if p.grad is not None), (sorted(set(names) ^ set(params)))
    for pname in names:
        g = N(params[pname].grad)
        assert np.isfinite(g).all(), pname
        if pname.endswith("encoding.encoding.params"):
            report["tables"][pname] = table_stats(g, B, f"{name}_grad_{pname}")
        else:
            ref = B[f"{name}_grad_{pname}"]
            assert g.shape == ref.shape, (pname, g.shape, ref.shape)
            report["groups"][pname] = group_stats(g, ref)
    for k in ("comp_rgb_phys_full", "comp_albedo_full", "comp_roughness_full", "comp_metallic_full"):
        report.setdefault("forward", {})[k] = float(np.abs(N(d[k]) - B[f"{name}_out_{k}"]).max())
    return report


# (rel_max, cos_dist): 3 x the MI355X observation (profiles/r05_grad_parity.json), hard caps 2e-2 / 2e-4
CAP = (2e-2, 2e-4)
TABLE_CAP = dict(level_sum=2e-2, level_l1=2e-2, level_l2=2e-2, level_probe=5e-2, sub_rel_max=5e-2, sub_cos_dist=2e-4)
# uniform_light at 512 samples per pixel is a Monte-Carlo estimate with 401 408 visibility tests on this frame: a secondary ray whose
# transmittance lands on the other side of a threshold in the two implementations (a DISCRETE event; forward: comp_rgb_phys_full differs by
# up to 2.5e-3 here against 2e-4 under the `light` estimator at spp 16) moves the gradient of the few table entries its pixel's samples
# touch.  Observed (profiles/r06_grad_parity.json): 6 of 86 k compared entries above 1e-2 of the largest entry, the worst at 0.092, the
# 99.9th percentile at 5e-4.  So for this variant the single worst entry gets a wider cap, and the COUNT above 1e-2 and the 99.9th
# percentile are bounded instead.
MC_TABLE_CAP = dict(TABLE_CAP, sub_rel_max=0.25, sub_cos_dist=5e-4)
MC_VARIANTS = {"uniform_default": dict(max_entries_over_1e_2=20, p999=2e-3)}
BARS = {}
TABLE_BARS = {}


def _bars():
    p = os.path.join(HERE, "golden", "grad_parity_bars.json")
    if os.path.exists(p) and not BARS:
        j = json.load(open(p))
        BARS.update({k: tuple(v) for k, v in j["groups"].items()})
        TABLE_BARS.update(j["tables"])
    return BARS, TABLE_BARS


def test_forward_of_the_uniform_light_training_run_vs_the_reference(G):
    """forward_ in train() mode with the shipped estimator (uniform_light, spp 512): the reference's output dict of the
    `uniform_light_512_gi_train` run -- keys, sample count, composited maps, the Monte-Carlo image and the visibility map."""
    tag = FG.UNIFORM_TRAIN_RUN
    rs, mat, env, rays = FG.gpu_scene(G, tag)
    rnd = FG.explicit_randoms(G, tag)
    with torch.no_grad():
        d = rs.forward_train_(rays, mat, env, 512, T(rnd["stratified_u"]), jitter=T(rnd["near_jitter"]), material_jitter=T(rnd["material_jitter"]),
                              background_color=T(G["background_color"]), global_illumination=True, render_mode="uniform_light",
                              shuffle_u=T(rnd["shuffle_u"]))
    ref = {str(k): G[f"{tag}_out_{k}"] for k in G[tag + "_out_keys"]}
    assert sorted(k for k in d if k != "stats") == sorted(ref), sorted(set(d) ^ set(ref))
    n_ref = int(ref["num_samples"][0])
    assert abs(int(d["num_samples"][0]) - n_ref) <= 0.005 * n_ref
    for k, tol in (("comp_rgb", 2e-3), ("comp_normal", 4e-3), ("comp_albedo", 2e-3), ("comp_roughness", 2e-3), ("comp_metallic", 2e-3), ("opacity", 2e-3)):
        err = np.abs(N(d[k]) - ref[k]).reshape(ref[k].shape[0], -1).max(-1)
        assert (err <= tol).mean() >= 0.985 and err.mean() < 5e-4, (k, float((err > tol).mean()), float(err.max()))
    hit = ref["rays_valid"][:, 0]
    a, b = N(d["comp_rgb_phys"]), ref["comp_rgb_phys"]
    err, tol = np.abs(a - b).max(-1), 2e-2 * np.abs(b).max(-1) + 2e-2
    assert (err <= tol).mean() >= 0.97 and abs(a[hit].mean() - b[hit].mean()) <= 2e-2 * abs(b[hit].mean()), (float((err > tol).mean()), float(err.max()))
    ev = np.abs(N(d["visibility"]) - ref["visibility"]).max(-1)
    assert (ev <= 3e-2).mean() >= 0.97, float((ev > 3e-2).mean())
    if int(d["num_samples"][0]) == n_ref:          # same sample set: the jitter-pass maps and the per-sample training outputs line up
        for k in ("albedo_smoothness_loss_map", "roughness_smoothness_loss_map", "metallic_smoothness_loss_map", "normals_orientation_loss_map"):
            e = np.abs(N(d[k]) - ref[k]).max(-1)
            assert (e <= 2e-3 * max(float(np.abs(ref[k]).max()), 1e-6) + 1e-9).mean() >= 0.97, (k, float(e.max()), float(np.abs(ref[k]).max()))


def _check(rep, name, bars, tbars):
    assert abs(rep["loss"] - rep["loss_ref"]) <= 2e-5 * abs(rep["loss_ref"]), (rep["loss"], rep["loss_ref"])
    for k, (a, b) in rep["terms"].items():
        assert abs(a - b) <= 3e-4 * abs(b) + 1e-11, (k, a, b)
    for pname, st in rep["groups"].items():
        if st["ref_max"] <= 1e-30:               # a group the reference leaves at exactly zero (inactive Lipschitz clamp)
            assert st["rel_max"] * st["ref_max"] == 0.0, (pname, st)
            continue
        bar = bars.get(f"{name}/{pname}", CAP)
        assert st["rel_max"] <= min(bar[0], CAP[0]) and st["cos_dist"] <= min(bar[1], CAP[1]), (name, pname, st, bar)
    for pname, st in rep["tables"].items():
        bar = tbars.get(f"{name}/{pname}", {})
        for k, cap in (MC_TABLE_CAP if name in MC_VARIANTS else TABLE_CAP).items():
            assert st[k] <= min(bar.get(k, cap), cap), (name, pname, k, st, bar)
        assert st["nnz_rel"] <= 2e-3, (name, pname, st)                   # same samples -> same touched entries (exact zeros aside)
        if name in MC_VARIANTS:
            assert st["sub_over_1e-2"] <= MC_VARIANTS[name]["max_entries_over_1e_2"] and st["sub_rel_p999"] <= MC_VARIANTS[name]["p999"], (name, pname, st)
        else:
            assert st["sub_rel_p999"] <= 2e-3, (name, pname, st)


@pytest.mark.parametrize("name", PHYS_VARIANTS)
def test_forward_backward_phys_itself_vs_the_references_own_training_step(G, B, name):
    """RenderStep.forward_backward_phys -- the function bench.py's headline step and its `config4` object time -- with the reference's
    loss composition: loss value, every logged term and d loss / d parameter of all 25 groups against the reference's own
    training_step + backward (same bars as the forward_train_ route)."""
    rep = compare(G, B, name, through="forward_backward_phys")
    bars, tbars = _bars()
    _check(rep, name, bars, tbars)


@pytest.mark.parametrize("name", VARIANTS)
def test_gradients_vs_the_references_own_training_step(G, B, name):
    rep = compare(G, B, name)
    bars, tbars = _bars()
    _check(rep, name, bars, tbars)


if __name__ == "__main__":        # python -m tests.test_gpu_backward_golden  -> the observed table (profiles/r05_grad_parity.json)
    from intrinsicavatar_amd import build
    build.build()
    G_, B_ = FG.load(), np.load(os.path.join(HERE, "golden", "golden_backward.npz"))
    rep_ = {n: compare(G_, B_, n) for n in VARIANTS}
    rep_.update({n + "/forward_backward_phys": compare(G_, B_, n, through="forward_backward_phys") for n in PHYS_VARIANTS})
    print(json.dumps(rep_, indent=1))
