"""Helpers for tests/golden/golden_forward.npz (the reference's own IntrinsicAvatarModel.forward_, run on CPU by
tests/golden/make_golden_forward.py): rebuild the scene of the fixture for the CPU oracle (numpy) and for the HIP path."""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
RUNS = dict(light_16_nogi=("light", 16, False), light_64_gi=("light", 64, True), uniform_light_512_gi=("uniform_light", 512, True),
            mis_16_gi=("mis", 16, True), mats_16_gi=("mats", 16, True))
TRAIN_RUN = "light_16_gi_train"


def load():
    return np.load(os.path.join(HERE, "golden", "golden_forward.npz"))


def hash_tables():
    """the two tables of the golden scene: closed-form in the entry index (make_golden_forward.py init_params)."""
    from intrinsicavatar_amd import synthetic as S
    n = 12599920                  # floats: 6 299 960 entries x 2 features (tiny-cuda-nn's flat `params`)
    return S.hash_table_values(n, 11, 1e-2), S.hash_table_values(n, 12, 1e-2)


def rng_draws(G, tag):
    """[(kind, array)] in the order the reference drew them."""
    kinds = [str(k) for k in G[tag + "_rng_kinds"]]
    out = []
    for i, k in enumerate(kinds):
        a = G[f"{tag}_rng_{i}"]
        if k == "rand_formula":                                    # large uniform tensors are closed-form in the index (not stored)
            k, a = "rand", formula_uniforms(a[0], a[1])
        out.append((k, a))
    return out


def formula_uniforms(seed, n):
    from intrinsicavatar_amd import synthetic as S
    return ((S.hash_table_values(int(n), int(seed), 1.0).astype(np.float64) + 1.0) / 2.0).astype(np.float32)


def explicit_randoms(G, tag):
    """the random tensors of one run under the names the oracle / the package take them:
    occ_jitter [64^3, 3, 3]; light_u; shuffle_u; stratified_u [512, 2]; scatter_u [F, 6]; train: near_jitter, material_jitter."""
    d = rng_draws(G, tag)
    out = {}
    k0, a0 = d[0]
    assert k0 == "rand_like_formula"
    out["occ_jitter"] = formula_uniforms(a0[0], a0[1])            # flat: [64^3 * 3 * 3] (test grid, 3 points per voxel) or [64^3 * 3] (training grid)
    rest = d[1:]
    mode = tag.split("_")[0] if not tag.startswith("uniform") else "uniform_light"
    if tag.endswith("_train"):
        # stratified near-plane jitter, material jitter, the (unused) shuffle draw, emitter.sample(F)
        (ka, near), (kb, mj), (kc, _), (kd, lu) = rest
        assert (ka, kb, kc, kd) == ("rand_like", "randn_like", "rand", "emitter.sample")
        out.update(near_jitter=near, material_jitter=mj, light_u=lu)
        return out
    assert rest[0][0] == "emitter.sample"                      # prepare(): self.secondary_rays_d = emitter.sample(spp)
    out["light_u"] = rest[0][1]
    if mode in ("light", "uniform_light"):
        assert rest[1][0] == "rand"
        out["shuffle_u"] = rest[1][1].reshape(G["rays"].shape[0], -1)
        if mode == "uniform_light":
            assert rest[2][0] == "emitter.sample_uniform_sphere_stratified"
            out["stratified_u"] = rest[2][1]
    elif mode == "mats":
        assert rest[1][0] == "scatterer.sample"
        out["scatter_u"] = np.concatenate([rest[1][1], np.zeros_like(rest[1][1])], 1)
    elif mode == "mis":
        assert rest[1][0] == "scatterer.sample" and rest[2][0] == "emitter.sample"
        out["scatter_u"] = np.concatenate([rest[1][1], rest[2][1]], 1)
    return out


def oracle_scene(G, tag):
    """oracle/render_ref.Scene of the golden scene with run `tag`'s occupancy grid."""
    from oracle import render_ref as R
    st = lambda k: G["state_" + k]      # noqa: E731
    t_geo, t_rad = hash_tables()

    def wn(g, v):
        return (g * v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)
    bbox = G["rig_cano_bbox"]
    center, scale = ((bbox[0] + bbox[1]) / 2).astype(np.float32), (bbox[1] - bbox[0]).astype(np.float32)
    Ws, bs = [], []
    for i in range(3):
        w = st(f"material.network.weights_per_layer.{i}").astype(np.float64)
        c = np.log1p(np.exp(st(f"material.network.lipshitz_bound_per_layer.{i}").astype(np.float64)))
        Ws.append((w * np.minimum(c / np.abs(w).sum(1), 1.0)[:, None]).astype(np.float32))
        bs.append(st(f"material.network.biases_per_layer.{i}"))
    return R.Scene(
        w2s=G["rig_w2s"], tfs=G["rig_tfs"], voxel_J=G["rig_ref_voxel_J"], offset_kernel=G["rig_offset_kernel"], scale_kernel=G["rig_scale_kernel"],
        binaries=G[tag + "_occ_binaries"][0], aabb=G[tag + "_occ_aabb"][0], step=float(G["render_step_size"]), beta=float(abs(st("density.beta")) + 1e-4),
        geo_center=center, geo_scale=scale, geo_params=t_geo, geo_mask=np.ones(32, np.float32),
        geo_W1=wn(st("geometry.network.layers.0.weight_g"), st("geometry.network.layers.0.weight_v")), geo_b1=st("geometry.network.layers.0.bias"),
        geo_W2=wn(st("geometry.network.layers.2.weight_g"), st("geometry.network.layers.2.weight_v")), geo_b2=st("geometry.network.layers.2.bias"),
        rad_center=center, rad_scale=scale, rad_params=t_rad, rad_mask=np.ones(32, np.float32), rad_sh_mask=st("radiance.sh_mask")[0],
        rad_W=[st(f"radiance.network.layers.{i}.weight") for i in (0, 2, 4)], rad_b=[st(f"radiance.network.layers.{i}.bias") for i in (0, 2, 4)],
        mat_W=Ws, mat_b=bs, env_base=G["hdri"])


def gpu_scene(G, tag, dev="cuda:0"):
    """(RenderStep, material, emitter, rays) of the golden scene on the HIP path: the reference's state_dict loads into the
    package's modules under the reference's own keys (checkpoint.py)."""
    import torch
    from intrinsicavatar_amd import fields, pbr, render, checkpoint
    from intrinsicavatar_amd.deformer import SNARFDeformer
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)      # noqa: E731
    dfm = SNARFDeformer(T(G["rig_lbs_voxel_final"]), T(G["rig_offset_kernel"]), T(G["rig_scale_kernel"]), T(G["rig_cano_bbox"]))
    dfm.prepare(T(G["rig_tfs"]), T(G["rig_w2s"]))
    geo, rad = fields.VolumeSDF(seed=0).to(dev), fields.VolumeRefDirRadiance(seed=1).to(dev)
    dens, mat = fields.LaplaceDensity(beta_init=0.3).to(dev), fields.VolumeMaterial(seed=2).to(dev)
    t_geo, t_rad = hash_tables()
    sd = {"model." + str(k): torch.from_numpy(np.ascontiguousarray(G["state_" + str(k)])) for k in G["state_keys"] if "state_" + str(k) in G.files}
    sd["model.geometry.encoding.encoding.encoding.params"] = torch.from_numpy(t_geo)
    sd["model.radiance.xyz_encoding.encoding.encoding.params"] = torch.from_numpy(t_rad)
    rs = render.RenderStep(geo, rad, dens, dfm, T(G[tag + "_occ_binaries"]), T(G[tag + "_occ_aabb"]), float(G["render_step_size"]))
    checkpoint.load_reference_state_dict(rs, sd, material=mat, strict=False)
    for m in (geo, rad):
        m.update_step(250, 25000)                                  # systems/base.py:150
        m.prepare_bbox(dfm.bbox)
    env = pbr.EnvironmentLightTensor(T(G["hdri"]))
    env.update_pdf()
    return rs, mat, env, T(G["rays"])


# ----------------------------------------------------------------------------- golden_backward.npz: summaries of a table gradient
SUBSAMPLE = 32


def _S():
    from intrinsicavatar_amd import synthetic as S
    return S


def subsample_mask(idx):
    """1-in-SUBSAMPLE selection of table entries by a hash of the FLOAT index (numpy uint64 arithmetic, closed form)."""
    h = (idx.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(40)
    return (h % np.uint64(SUBSAMPLE)) == 0


def level_offsets():
    from tests import torch_ref as TR
    return np.asarray(TR.hash_cfg()[0], dtype=np.int64)


def table_gradient_summary(g):
    """g: flat float gradient of one table's `params` (entries x 2 features) -> dict of arrays (see the module docstring)."""
    off = level_offsets() * 2                                        # float offsets of the 16 levels (+ end)
    g64 = g.astype(np.float64)
    probe = _S().hash_table_values(g.size, 4242, 1.0).astype(np.float64)
    out = dict(level_sum=np.array([g64[off[i]:off[i + 1]].sum() for i in range(len(off) - 1)]),
               level_l1=np.array([np.abs(g64[off[i]:off[i + 1]]).sum() for i in range(len(off) - 1)]),
               level_l2=np.array([np.sqrt((g64[off[i]:off[i + 1]] ** 2).sum()) for i in range(len(off) - 1)]),
               level_probe=np.array([(g64[off[i]:off[i + 1]] * probe[off[i]:off[i + 1]]).sum() for i in range(len(off) - 1)]),
               level_nnz=np.array([int((g[off[i]:off[i + 1]] != 0).sum()) for i in range(len(off) - 1)]))
    nz = np.nonzero(g)[0]
    keep = nz[subsample_mask(nz)]
    out["sub_index"] = keep.astype(np.int32)
    out["sub_value"] = g[keep].astype(np.float32)
    return out
