"""Synthetic inputs for tests and bench: no dataset, checkpoint, SMPL .pkl or HDRI
is available, so rays, a 24-bone rig, skinning-weight grids and occupancy grids are generated here
(numpy, host side, seeded) with the shapes and magnitudes of the reference's PeopleSnapshot /
animation configs:

  * camera: K of load/animation/aist/cameras.npz (1080x1080) / downscale, identity extrinsic,
    subject translated to (0, 0.15, 5) (datasets/animation.py:19-27,129-130);
  * rays in SMPL space = transform_rays_w2s (models/deformers/snarf_deformer.py:128-147);
  * scene AABB = cube around the posed body x1.2 (snarf_deformer.py:24-35);
  * 64^3 occupancy grid of a union-of-capsules stick figure on the SMPL kinematic tree
    (models/pose/pose_encoder.py:30-56).
"""
import numpy as np

# rough SMPL rest-pose joints (metres, pelvis at origin, y up)
JOINTS = np.array([
    [0.00, 0.00, 0.00], [0.07, -0.09, 0.00], [-0.07, -0.09, 0.00], [0.00, 0.11, -0.02],
    [0.10, -0.47, 0.00], [-0.10, -0.47, 0.00], [0.00, 0.25, -0.01], [0.09, -0.87, -0.03],
    [-0.09, -0.87, -0.03], [0.00, 0.30, 0.00], [0.11, -0.93, 0.09], [-0.11, -0.93, 0.09],
    [0.00, 0.52, -0.03], [0.08, 0.43, -0.02], [-0.08, 0.43, -0.02], [0.00, 0.60, 0.01],
    [0.17, 0.45, -0.02], [-0.17, 0.45, -0.02], [0.43, 0.44, -0.04], [-0.43, 0.44, -0.04],
    [0.68, 0.44, -0.04], [-0.68, 0.44, -0.04], [0.77, 0.43, -0.05], [-0.77, 0.43, -0.05]], np.float32)
PARENTS = np.array([-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21])
RADII = np.array([0.12, 0.09, 0.09, 0.13, 0.07, 0.07, 0.14, 0.05, 0.05, 0.14, 0.04, 0.04, 0.06, 0.08, 0.08, 0.10,
                  0.06, 0.06, 0.045, 0.045, 0.04, 0.04, 0.035, 0.035], np.float32)
INIT_BONES = np.array([0, 1, 2, 4, 5, 10, 11, 12, 15, 16, 17, 18, 19], np.int32)   # deformer_torch.py:27

K_1080 = np.array([[2664.23, 0, 511.78], [0, 2664.69, 567.13], [0, 0, 1]], np.float64)


def rodrigues(rv):
    th = np.linalg.norm(rv)
    if th < 1e-8:
        return np.eye(3)
    k = rv / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def forward_kinematics(pose72, joints=JOINTS):
    """SMPL-style FK: pose [72] axis-angle -> A [24,4,4] (rest -> posed bone transforms)."""
    G = np.zeros((24, 4, 4))
    for j in range(24):
        R = rodrigues(pose72[3 * j:3 * j + 3])
        T = np.eye(4)
        T[:3, :3] = R
        T[:3, 3] = joints[j] - (joints[PARENTS[j]] if PARENTS[j] >= 0 else 0)
        G[j] = T if PARENTS[j] < 0 else G[PARENTS[j]] @ T
    A = G.copy()
    for j in range(24):
        A[j, :3, 3] = G[j, :3, 3] - G[j, :3, :3] @ joints[j]
    return A


def make_pose(seed=0, amplitude=0.25):
    rng = np.random.default_rng(seed)
    pose = rng.normal(0, amplitude, 72)
    pose[:3] = [np.pi, 0, 0]               # global orient: SMPL y-up -> camera y-down
    pose[3 * 16 + 2] -= 0.9                # arms down a bit
    pose[3 * 17 + 2] += 0.9
    return pose.astype(np.float64)


def reference_pose(name="male-3-casual", frame=0):
    """(pose72 [72] axis-angle = global_orient | body_pose, transl [3]) of one frame of the reference's own pose files:
    'male-3-casual' (load/peoplesnapshot/male-3-casual/poses/anim_nerf_train.npz: training frames 0 / 40 / 80 / 113) or 'aist'
    (load/animation/aist/poses.npz: out-of-distribution frames 0 / 100 / 200 / 319, translation re-based as
    datasets/animation.py:129-130).  The arrays travel as data in tests/golden/reference_poses.npz
    (tests/golden/make_reference_poses.py)."""
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "reference_poses.npz")
    z = np.load(path)
    frames = z[name + "_frames"].tolist()
    if frame not in frames:
        raise KeyError(f"{name}: frame {frame} is not in the committed set {frames}")
    i = frames.index(frame)
    pose = np.concatenate([z[name + "_global_orient"][i], z[name + "_body_pose"][i]]).astype(np.float64)
    return pose, z[name + "_transl"][i].astype(np.float64)


def parse_pose(spec):
    """'male-3-casual:0' / 'aist:100' -> reference_pose(...); 'synthetic:K' -> (make_pose(K), (0, 0.15, 5)); 'neutral' -> rest pose."""
    if spec in (None, "neutral"):
        return None, (0.0, 0.15, 5.0)
    name, _, k = spec.partition(":")
    if name == "synthetic":
        return make_pose(int(k or 0)), (0.0, 0.15, 5.0)
    return reference_pose(name, int(k or 0))


def make_rig(pose72=None, transl=(0.0, 0.15, 5.0)):
    """returns dict(tfs [1,24,4,4] (w2s @ A, root = identity), w2s [4,4], joints_posed [24,3] in SMPL space)."""
    if pose72 is None:
        pose72 = np.zeros(72)
        pose72[:3] = [np.pi, 0, 0]
    A = forward_kinematics(pose72)
    A[:, :3, 3] += np.asarray(transl)[None]
    s2w = A[0]
    w2s = np.linalg.inv(s2w)
    tfs = (w2s[None] @ A).astype(np.float32)[None]
    jp = (np.einsum("jab,jb->ja", tfs[0, :, :3, :3], JOINTS) + tfs[0, :, :3, 3]).astype(np.float32)
    return dict(tfs=tfs, w2s=w2s.astype(np.float32), joints_posed=jp)


def camera_rays(height, width, downscale_from_1080=None):
    """datasets/animation.py:14-27 with identity c2w. returns rays [H*W, 8] in WORLD space (o, d, near, far)."""
    ds = downscale_from_1080 if downscale_from_1080 is not None else 1080.0 / height
    K = K_1080.copy()
    K[:2] /= ds
    x, y = np.meshgrid(np.arange(width), np.arange(height), indexing="xy")
    xy = np.stack([x, y, np.ones_like(x)], -1).reshape(-1, 3).astype(np.float32)
    d = xy @ np.linalg.inv(K).T
    d = d / np.linalg.norm(d, axis=1, keepdims=True)
    o = np.zeros_like(d)
    near = np.zeros((d.shape[0], 1))
    far = np.zeros((d.shape[0], 1))
    return np.concatenate([o, d, near, far], -1).astype(np.float32)


def rays_world_to_smpl(rays, w2s):
    """snarf_deformer.py:128-147 transform_rays_w2s."""
    o = rays[:, :3] @ w2s[:3, :3].T + w2s[None, :3, 3]
    d = rays[:, 3:6] @ w2s[:3, :3].T
    dist = np.linalg.norm(o, axis=-1, keepdims=True)
    return np.concatenate([o, d, dist - 1, dist + 1], -1).astype(np.float32)


def capsule_sdf(p, joints_posed):
    """union-of-capsules SDF of the stick figure at points p [N,3] (SMPL space)."""
    best = np.full(p.shape[0], 1e9, np.float32)
    for j in range(1, 24):
        a, b = joints_posed[PARENTS[j]], joints_posed[j]
        ra, rb = RADII[PARENTS[j]], RADII[j]
        ab = b - a
        t = np.clip(((p - a) @ ab) / max(float(ab @ ab), 1e-12), 0, 1)
        c = a[None] + t[:, None] * ab[None]
        r = ra + (rb - ra) * t
        best = np.minimum(best, np.linalg.norm(p - c, axis=1) - r)
    return best


def body_aabb(joints_posed, factor=1.2):
    """get_bbox_from_smpl (snarf_deformer.py:24-35) on joints +- radii."""
    lo = (joints_posed - RADII[:, None]).min(0)
    hi = (joints_posed + RADII[:, None]).max(0)
    c = (hi + lo) / 2
    s = ((hi - lo) / 2).max() * factor
    return np.concatenate([c - s, c + s]).astype(np.float32)


def occupancy_grid(joints_posed, aabb, res=64, margin=0.02):
    """bool [res,res,res]; cell (x,y,z) occupied if the capsule body comes within half a cell diagonal."""
    cs = (aabb[3:] - aabb[:3]) / res
    g = (np.arange(res) + 0.5)
    X, Y, Z = np.meshgrid(g, g, g, indexing="ij")
    p = np.stack([X, Y, Z], -1).reshape(-1, 3).astype(np.float32) * cs[None] + aabb[None, :3]
    sd = capsule_sdf(p, joints_posed)
    return (sd < 0.5 * np.linalg.norm(cs) + margin).reshape(res, res, res)


def skinning_weight_grid(D=32, H=128, W=128, global_scale=1.2, smooth_iters=30, sigma=0.08):
    """[1,24,D,H,W] canonical skinning weights + (offset_kernel [1,1,3], scale_kernel [1,1,3]) following
    ForwardDeformer.switch_to_explicit (deformer_torch.py:139-197) on the rest-pose stick figure."""
    import torch
    verts_lo = (JOINTS - RADII[:, None]).min(0)
    verts_hi = (JOINTS + RADII[:, None]).max(0)
    offset = (verts_lo + verts_hi) * 0.5
    scale = (verts_hi - verts_lo).max() / 2 * global_scale
    ratio = H / D
    zz, yy, xx = np.meshgrid(np.linspace(-1, 1, D), np.linspace(-1, 1, H), np.linspace(-1, 1, W), indexing="ij")
    grid = np.stack([xx, yy, zz / ratio], -1).reshape(-1, 3).astype(np.float32) * scale + offset[None]
    d2 = np.zeros((24, grid.shape[0]), np.float32)
    for j in range(24):
        a = JOINTS[PARENTS[j]] if PARENTS[j] >= 0 else JOINTS[j]
        b = JOINTS[j]
        ab = b - a
        t = np.clip(((grid - a) @ ab) / max(float(ab @ ab), 1e-12), 0, 1)
        d2[j] = ((grid - (a[None] + t[:, None] * ab[None])) ** 2).sum(1)
    w = np.exp(-(d2 - d2.min(0, keepdims=True)) / (2 * sigma * sigma))
    w /= w.sum(0, keepdims=True)
    weights = torch.from_numpy(w.reshape(1, 24, D, H, W).astype(np.float32))
    for _ in range(smooth_iters):      # deformer_torch.py:246-252
        mean = (weights[:, :, 2:, 1:-1, 1:-1] + weights[:, :, :-2, 1:-1, 1:-1] + weights[:, :, 1:-1, 2:, 1:-1]
                + weights[:, :, 1:-1, :-2, 1:-1] + weights[:, :, 1:-1, 1:-1, 2:] + weights[:, :, 1:-1, 1:-1, :-2]) / 6.0
        weights[:, :, 1:-1, 1:-1, 1:-1] = (weights[:, :, 1:-1, 1:-1, 1:-1] - mean) * 0.7 + mean
        weights = weights / weights.sum(1, keepdim=True)
    offset_kernel = (-offset).reshape(1, 1, 3).astype(np.float32)
    scale_kernel = np.full((1, 1, 3), 1.0 / scale, np.float32)
    scale_kernel[..., 2] *= ratio
    bbox = np.stack([offset - scale * np.array([1, 1, 1 / ratio]), offset + scale * np.array([1, 1, 1 / ratio])])
    return weights.numpy(), offset_kernel, scale_kernel, bbox.astype(np.float32)


def make_scene(height=128, width=128, pose_seed=0, res=64):
    """everything traverse_grids needs for one frame."""
    rig = make_rig(None if pose_seed is None else make_pose(pose_seed))      # None: static neutral pose (tfs == identity)
    rays = rays_world_to_smpl(camera_rays(height, width), rig["w2s"])
    aabb = body_aabb(rig["joints_posed"])
    occ = occupancy_grid(rig["joints_posed"], aabb, res)
    return dict(rays=rays, aabb=aabb, binaries=occ, rig=rig)


# ----------------------------------------------------------------------------- full synthetic frame (torch / GPU side)
def build_frame(device="cuda:0", height=128, width=128, pose_seed=0, beta=0.01, hash_amp=1e-4, num_samples_per_ray=128,
                grid_D=32, grid_H=128, grid_W=128, smooth_iters=30, occ_res=64, seed=0, pose=None):
    """Random-init model of the reference's architecture + synthetic rig + per-frame occupancy grid.
    pose: None -> make_pose(pose_seed) (N(0, 0.25) joint angles; pose_seed None = neutral); or a spec for parse_pose():
    'male-3-casual:0', 'aist:100' ... = a frame of the reference's own pose files through plain forward kinematics.
    Returns (RenderStep, rays_world [H*W, 8] tensor, export dict of numpy arrays for the CPU oracle)."""
    import torch
    from . import fields, render
    from .deformer import SNARFDeformer

    w, offk, sck, bbox = skinning_weight_grid(grid_D, grid_H, grid_W, smooth_iters=smooth_iters)
    if pose is not None:
        rig = make_rig(*parse_pose(pose))
    else:
        rig = make_rig(None if pose_seed is None else make_pose(pose_seed))      # None: static neutral pose (tfs == identity)
    dev = torch.device(device)
    dfm = SNARFDeformer(torch.from_numpy(w).to(dev), torch.from_numpy(offk).to(dev), torch.from_numpy(sck).to(dev),
                        torch.from_numpy(bbox).to(dev))
    dfm.prepare(torch.from_numpy(rig["tfs"]).to(dev), torch.from_numpy(rig["w2s"]).to(dev))
    geo = fields.VolumeSDF(seed=seed).to(dev)
    rad = fields.VolumeRefDirRadiance(seed=seed + 1).to(dev)
    with torch.no_grad():
        # sphere init zeroes the hash-feature columns of the first SDF layer (network_utils.py:222-226); a trained
        # network has them populated, and the backward pass is only representative if they are non-zero
        gw = torch.Generator().manual_seed(seed + 3)
        geo.network.layers[0].weight_v[:, 3:] = (torch.randn((64, 32), generator=gw) * 0.02).to(dev)
    if hash_amp != 1e-4:
        with torch.no_grad():
            geo.grid_params.mul_(hash_amp / 1e-4)
            rad.grid_params.mul_(hash_amp / 1e-4)
    geo.prepare_bbox(dfm.bbox)
    rad.prepare_bbox(dfm.bbox)
    dens = fields.LaplaceDensity(beta_init=beta).to(dev)
    step = 4.330127018922194 / num_samples_per_ray     # |scene_aabb diag| / num_samples_per_ray, intrinsic_avatar.py:197-211
    aabb = body_aabb(rig["joints_posed"])
    # per-frame occupancy grid from the MODEL (prepare_test_occupancy_grid, intrinsic_avatar.py:307-381):
    # 3 jittered samples per voxel -> alpha -> max -> 3^3 dilation -> threshold (largest-CC filter: SURVEY 8(f).1)
    from . import occ_grid
    g = torch.Generator().manual_seed(seed + 7)
    rand = torch.rand((occ_res ** 3, 3, 3), generator=g).to(dev)
    beta_t = dens.get_beta().detach().reshape(1)

    def occ_eval_fn(x):
        return render.laplace_alpha(dfm.deform(x, geo)["sdf"], step, beta_t)

    _, binaries = occ_grid.compute_test_occupancy_grid(occ_eval_fn, torch.from_numpy(aabb).to(dev), occ_res, 3, 0.01, rand)
    binaries = binaries.contiguous()
    rs = render.RenderStep(geo, rad, dens, dfm, binaries, torch.from_numpy(aabb)[None].to(dev), step)
    rays = torch.from_numpy(camera_rays(height, width)).to(dev)
    N = lambda t: t.detach().cpu().numpy()      # noqa: E731
    l0, l2 = geo.network.layers[0], geo.network.layers[2]
    vJ = torch.empty((1, 12, grid_D, grid_H, grid_W), device=dev)
    from . import fast_snarf
    fast_snarf.precompute(dfm.lbs_voxel_final, dfm.tfs, None, vJ, dfm.offset_kernel, dfm.scale_kernel)
    rl = rad.network.layers
    export = dict(
        w2s=rig["w2s"], tfs=rig["tfs"], voxel_J=N(vJ), offset_kernel=offk.reshape(3), scale_kernel=sck.reshape(3),
        binaries=N(binaries[0]), aabb=aabb, step=step, beta=float(dens.get_beta().detach()),
        geo_center=N(geo.center), geo_scale=N(geo.scale), geo_params=N(geo.grid_params),
        geo_mask=N(geo.prog.mask(geo.global_step, "cpu")), geo_W1=N(l0.effective()), geo_b1=N(l0.bias),
        geo_W2=N(l2.effective()), geo_b2=N(l2.bias),
        rad_center=N(rad.center), rad_scale=N(rad.scale), rad_params=N(rad.grid_params),
        rad_mask=N(rad.prog.mask(rad.global_step, "cpu")), rad_sh_mask=N(rad.sh_mask[0]),
        rad_W=[N(rl[i].weight) for i in (0, 2, 4)], rad_b=[N(rl[i].bias) for i in (0, 2, 4)])
    return rs, rays, export


def repose(rs, pose, occ_res=64, seed=0):
    """the per-frame work of the reference's animation loop on an existing model (datasets/animation.py:129-130 -> the deformer's
    `prepare`, then prepare_test_occupancy_grid, intrinsic_avatar.py:307-381): new bone transforms -> skinning grids (K10) ->
    per-frame occupancy grid from the model.  `pose` as for build_frame(pose=...).  Returns the RenderStep (modified in place)."""
    import torch
    from . import render, occ_grid
    rig = make_rig(*parse_pose(pose))
    dfm, geo, dens = rs.deformer, rs.geometry, rs.density
    dev = dfm.device
    dfm.prepare(torch.from_numpy(rig["tfs"]).to(dev), torch.from_numpy(rig["w2s"]).to(dev))
    aabb = body_aabb(rig["joints_posed"])
    g = torch.Generator().manual_seed(seed + 7)
    rand = torch.rand((occ_res ** 3, 3, 3), generator=g).to(dev)
    beta_t = dens.get_beta().detach().reshape(1)
    step = rs.render_step_size

    def occ_eval_fn(x):
        return render.laplace_alpha(dfm.deform(x, geo)["sdf"], step, beta_t)

    _, binaries = occ_grid.compute_test_occupancy_grid(occ_eval_fn, torch.from_numpy(aabb).to(dev), occ_res, 3, 0.01, rand)
    rs.binaries, rs.aabbs = binaries.contiguous(), torch.from_numpy(aabb)[None].to(dev)
    return rs


def export_phys(material, env_base):
    """numpy arrays of the PBR branch for the CPU oracle (oracle/render_ref.py relight_step): the material head's
    Lipschitz-normalised weights in the REFERENCE's column order [xyz(3) | hash(32) | feat(13)] (network_utils.py:396-403)
    and the environment image."""
    import torch
    N = lambda t: t.detach().cpu().numpy()      # noqa: E731
    Ws, bs = [], []
    for i in range(3):
        w = material.network.weights_per_layer[i]
        c = torch.nn.functional.softplus(material.network.lipshitz_bound_per_layer[i])
        Ws.append(N(w * torch.clamp(c / w.abs().sum(dim=1), max=1.0)[:, None]))
        bs.append(N(material.network.biases_per_layer[i]))
    return dict(mat_W=Ws, mat_b=bs, env_base=N(env_base))


def hash_table_values(n, seed, amp):
    """n pseudo-random fp32 values in [-amp, amp) as a closed-form function of the entry index (64-bit integer mixing, 24
    mantissa bits): the 50 MB hash tables of a golden scene are regenerated bit-identically anywhere instead of being stored."""
    i = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        v = i * np.uint64(6364136223846793005) + np.uint64(seed) * np.uint64(1442695040888963407) + np.uint64(0x9E3779B97F4A7C15)
        v ^= v >> np.uint64(29)
        v *= np.uint64(0xBF58476D1CE4E5B9)
        v ^= v >> np.uint64(32)
    u = (v >> np.uint64(40)).astype(np.float64) / float(1 << 24)
    return ((u * 2.0 - 1.0) * amp).astype(np.float32)
