#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REFERENCE's own code.

Runs ONLY in the build container (needs /root/reference); never on the GPU box.
Nothing derived from the reference's sources is written into the repo: this script
copies the reference's CUDA kernel files to a scratch dir under /tmp, rewrites the
`<<<...>>>` launches into serial host loops there, compiles them as host C++ against
a macro header (defined below: `__global__` -> nothing, thread id -> loop counter),
executes the reference's own kernel BODIES serially on the CPU, and stores only
inputs + outputs as .npz fixtures.

  golden_resampling.npz   K1..K4  lib/nerfacc/cuda/csrc/cdf.cu
  golden_pack.npz         K5..K7  lib/nerfacc/cuda/csrc/pack.cu (+ pack.py:pack_data)
  golden_snarf.npz        K8..K10 models/deformers/fast_snarf/cuda/...
  golden_mlp.npz          VanillaMLP / LipshitzMLP  models/network_utils.py (imported, stubbed deps)

Usage:  python tests/golden/make_golden.py
"""
import importlib.util
import os
import re
import shutil
import subprocess
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.environ.get("IA_GOLDEN_OUT") or os.path.dirname(os.path.abspath(__file__))
# IA_GOLDEN_CONTRACT=fast: the SENSITIVITY run of tools/fma_contraction_sensitivity.py -- the same kernel bodies compiled WITH fused
# multiply-add contraction (-ffp-contract=fast -mfma), as nvcc's default --fmad=true does to the reference's build; the committed
# fixtures are the contraction-free build (source semantics).  Such a run must write somewhere else (IA_GOLDEN_OUT).
CONTRACT = os.environ.get("IA_GOLDEN_CONTRACT", "off")
assert CONTRACT == "off" or os.environ.get("IA_GOLDEN_OUT"), "a contraction run must not overwrite the committed fixtures"
FP_FLAGS = ["-ffp-contract=off"] if CONTRACT == "off" else ["-ffp-contract=fast", "-mfma"]
WORK = "/tmp/ia_golden_build" + ("" if CONTRACT == "off" else "_fma")

SHIM_H = r"""
#pragma once
#include <torch/extension.h>
#include <cmath>
#include <cstdint>
#include <climits>
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
static thread_local int g_tid = 0;
#define CUDA_GET_THREAD_ID(tid, Q) const int tid = g_tid; if (tid >= (int)(Q)) return
#define CHECK_CUDA(x)
#define CHECK_CONTIGUOUS(x) TORCH_CHECK(x.is_contiguous(), #x " must be contiguous")
#define CHECK_INPUT(x) CHECK_CONTIGUOUS(x)
#define CUDA_N_BLOCKS_NEEDED(Q, T) ((Q - 1) / T + 1)
#define DEVICE_GUARD(t)
template <class T> static inline T atomicAdd(T* p, T v) { T o = *p; *p += v; return o; }
// fast-snarf side
struct _dim1 { int x; };
static thread_local _dim1 blockIdx{0}, threadIdx{0}, blockDim{1};
#define C10_LAUNCH_BOUNDS_1(x)
#define C10_CUDA_KERNEL_LAUNCH_CHECK()
#define cudaDeviceSynchronize()
#define CUDA_KERNEL_LOOP_TYPE(i, n, T) for (T i = blockIdx.x, _once = 0; _once < 1 && i < (n); ++_once)
namespace at { namespace cuda { namespace detail {
template <typename T, typename I> struct TensorInfo { I sizes[8]; I strides[8]; };
template <typename T, typename I> TensorInfo<T, I> getTensorInfo(const at::Tensor& t) {
    TensorInfo<T, I> ti; for (int i = 0; i < t.dim(); i++) { ti.sizes[i] = (I)t.size(i); ti.strides[i] = (I)t.stride(i); } return ti; }
static inline int GET_BLOCKS(int64_t n, int t) { return (int)((n + t - 1) / t); }
}}}
"""


def _rewrite_nerfacc(src: str) -> str:
    src = src.replace('#include "include/helpers_cuda.h"', '#include "shim.h"')
    src = src.replace('#include "c10/core/TensorOptions.h"', "")
    src = re.sub(r"(\w+(?:<scalar_t>)?)<<<blocks, threads, 0, at::cuda::getCurrentCUDAStream\(\)>>>\(",
                 r"for (g_tid = 0; g_tid < (int)n_rays; ++g_tid) \1(", src)
    src = src.replace("AT_DISPATCH_FLOATING_TYPES_AND_HALF", "AT_DISPATCH_FLOATING_TYPES")
    return src


def _rewrite_snarf(src: str) -> str:
    src = re.sub(r'#include\s+[<"](c10/cuda|ATen/cuda)[^>"]*[>"]', "", src)
    src = '#include "shim.h"\n' + src
    src = re.sub(r"<<<\s*(?:at::cuda::detail::)?GET_BLOCKS\(count,\s*512\),\s*512,\s*0,\s*at::cuda::getCurrentCUDAStream\(\)\s*>>>\(",
                 "_LAUNCH_(", src, flags=re.S)
    # kernel<T>  _LAUNCH_(args)  ->  for (...) kernel<T>(args)
    src = re.sub(r"(\b(?:broyden_kernel|precompute_kernel|filter<scalar_t>)\s*)_LAUNCH_\(",
                 r"for (blockIdx.x = 0; blockIdx.x < (int)count; ++blockIdx.x) \1(", src, flags=re.S)
    src = src.replace(", torch::RestrictPtrTraits", "").replace(",torch::RestrictPtrTraits", "")
    src = src.replace("AT_DISPATCH_FLOATING_TYPES_AND_HALF", "AT_DISPATCH_FLOATING_TYPES")
    src = re.sub(r"PYBIND11_MODULE\(TORCH_EXTENSION_NAME, m\)\s*\{.*?\}\s*$", "", src, flags=re.S)
    return src


def build_reference_host_modules():
    from torch.utils.cpp_extension import load

    shutil.rmtree(WORK, ignore_errors=True)
    os.makedirs(WORK)
    open(f"{WORK}/shim.h", "w").write(SHIM_H)
    # --- lib/nerfacc
    for f in ("cdf.cu", "pack.cu"):
        s = open(f"{REF}/lib/nerfacc/cuda/csrc/{f}").read()
        open(f"{WORK}/{f[:-3]}.cpp", "w").write(_rewrite_nerfacc(s))
    decl = open(f"{REF}/lib/nerfacc/cuda/csrc/pybind.cu").read()
    decl = decl.replace('#include "include/helpers_cuda.h"', '#include "shim.h"').replace(
        '#include "include/helpers_math.h"', "").replace("(bool) CUB_SUPPORTS_SCAN_BY_KEY()", "false")
    open(f"{WORK}/bind_nerfacc.cpp", "w").write(decl)
    nerfacc = load(name="ia_ref_nerfacc_host" + ("" if CONTRACT == "off" else "_fma"), sources=[f"{WORK}/cdf.cpp", f"{WORK}/pack.cpp", f"{WORK}/bind_nerfacc.cpp"],
                   extra_cflags=["-O2", *FP_FLAGS, f"-I{WORK}"], build_directory=WORK, verbose=False)
    # --- fast-snarf
    os.makedirs(f"{WORK}/snarf", exist_ok=True)
    cu = f"{REF}/models/deformers/fast_snarf/cuda"
    srcs = []
    for sub, f in (("fuse_kernel", "fuse_cuda_kernel_fast.cu"), ("fuse_kernel", "fuse_cuda.cpp"),
                   ("filter", "filter.cu"), ("filter", "filter.cpp"),
                   ("precompute", "precompute.cu"), ("precompute", "precompute.cpp")):
        s = _rewrite_snarf(open(f"{cu}/{sub}/{f}").read())
        dst = f"{WORK}/snarf/{sub}_{f.replace('.cu', '_k.cpp')}"
        open(dst, "w").write(s)
        srcs.append(dst)
    open(f"{WORK}/snarf/bind.cpp", "w").write(
        '#include <torch/extension.h>\n'
        "void fuse_broyden(torch::Tensor &x, const torch::Tensor &xd_tgt, const torch::Tensor &grid, const torch::Tensor &grid_J_inv,"
        " const torch::Tensor &tfs, const torch::Tensor &bone_ids, bool align_corners, torch::Tensor &J_inv, torch::Tensor &is_valid,"
        " torch::Tensor &offset, torch::Tensor &scale, float cvg_threshold, float dvg_threshold);\n"
        "torch::Tensor filter(const torch::Tensor &x, const torch::Tensor &mask);\n"
        "void precompute(const torch::Tensor &voxel_w, const torch::Tensor &tfs, torch::Tensor &voxel_d, torch::Tensor &voxel_J,"
        " const torch::Tensor &offset, const torch::Tensor &scale);\n"
        'PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) { m.def("fuse_broyden", &fuse_broyden); m.def("filter", &filter); m.def("precompute", &precompute); }\n')
    srcs.append(f"{WORK}/snarf/bind.cpp")
    os.makedirs(f"{WORK}/snarf_build", exist_ok=True)
    snarf = load(name="ia_ref_snarf_host" + ("" if CONTRACT == "off" else "_fma"), sources=srcs, extra_cflags=["-O2", *FP_FLAGS, f"-I{WORK}"],
                 build_directory=f"{WORK}/snarf_build", verbose=False)
    return nerfacc, snarf


# ----------------------------------------------------------------------------- input generators
def gen_packed_samples(rng, n_rays, max_steps, p_empty=0.3):
    """random packed rays: sorted disjoint intervals, alphas, weights, sdfs with sign changes."""
    steps = rng.integers(1, max_steps + 1, n_rays)
    steps[rng.random(n_rays) < p_empty] = 0
    if n_rays > 3:
        steps[1] = 1          # single-sample ray
        steps[2] = 0
    cum = np.cumsum(steps)
    packed = np.stack([cum - steps, steps], -1).astype(np.int32)
    N = int(cum[-1])
    starts = np.zeros(N, np.float32)
    ends = np.zeros(N, np.float32)
    alphas = np.zeros(N, np.float32)
    sdfs = np.zeros(N, np.float32)
    for r in range(n_rays):
        b, s = packed[r]
        if s == 0:
            continue
        t0 = rng.uniform(0.5, 4.0)
        dts = rng.uniform(0.005, 0.06, s).astype(np.float32)
        gaps = (rng.random(s) < 0.2) * rng.uniform(0.01, 0.3, s)
        st = t0 + np.cumsum(dts + gaps) - dts
        starts[b:b + s] = st
        ends[b:b + s] = st + dts
        mode = r % 5
        if mode == 0:      # low density (sum w << 1) => bg samples
            alphas[b:b + s] = rng.uniform(0.0, 0.02, s)
        elif mode == 1:    # opaque
            alphas[b:b + s] = rng.uniform(0.3, 0.99, s)
        elif mode == 2:    # all zero weights
            alphas[b:b + s] = 0.0
        else:
            alphas[b:b + s] = rng.uniform(0.0, 0.5, s)
        sd = rng.normal(0.2, 0.3, s)
        if mode == 3:
            sd = np.sort(rng.normal(0.0, 0.3, s))[::-1]   # monotone decreasing: one crossing
        if mode == 4:
            sd = np.abs(sd) + 0.01                         # never crosses
        sdfs[b:b + s] = sd
    return packed, starts, ends, alphas, sdfs


def weights_from_alpha(packed, alphas, boost=1.0):
    w = np.zeros_like(alphas)
    for b, s in packed:
        T = np.float32(1.0)
        for j in range(s):
            w[b + j] = T * alphas[b + j]
            T = np.float32(T * (np.float32(1.0) - alphas[b + j]))
    return (w * np.float32(boost)).astype(np.float32)


def gen_edges(rng, n_rays, max_runs=3, max_run_len=12, p_empty=0.3):
    """edge lists like traverse_grids emits: runs of contiguous samples sharing edges."""
    vals, il, ir, steps = [], [], [], []
    for r in range(n_rays):
        if rng.random() < p_empty or r == 2:
            steps.append(0)
            continue
        t = rng.uniform(0.5, 4.0)
        n_e = 0
        for _ in range(int(rng.integers(1, max_runs + 1))):
            m = int(rng.integers(1, max_run_len + 1))
            t += rng.uniform(0.05, 0.4)
            dt = np.float32(0.0338)
            for k in range(m + 1):
                vals.append(np.float32(t + k * dt))
                il.append(k < m)
                ir.append(k > 0)
            t += m * dt
            n_e += m + 1
        steps.append(n_e)
    steps = np.asarray(steps)
    cum = np.cumsum(steps)
    packed = np.stack([cum - steps, steps], -1).astype(np.int32)
    return packed, np.asarray(vals, np.float32), np.asarray(il, bool), np.asarray(ir, bool)


def make_resampling(nerfacc, rng):
    out = {}
    T = torch.from_numpy
    case = 0
    for n_rays, max_steps, ns in ((64, 40, (4, 16, 256, 1024)), (257, 7, (2, 5, 16))):
        packed, st, en, al, sd = gen_packed_samples(rng, n_rays, max_steps)
        for boost in (1.0, 1.7):                       # boost>1: sum of weights may exceed 1
            w = weights_from_alpha(packed, al, boost)
            for n in ns:
                k = f"c{case}_"
                case += 1
                out[k + "packed_info"], out[k + "starts"], out[k + "ends"] = packed, st, en
                out[k + "weights"], out[k + "alphas"], out[k + "sdfs"], out[k + "n"] = w, al, sd, np.int32(n)
                r = nerfacc.ray_resampling(T(packed), T(st)[:, None], T(en)[:, None], T(w), T(sd), n)
                for nm, v in zip(("rpi", "ts", "offsets", "indices", "fg_counts", "bg_counts", "surface_idx"), r):
                    out[k + "k1_" + nm] = v.numpy()
                r = nerfacc.ray_resampling_fine(T(packed), T(st)[:, None], T(en)[:, None], T(w), n)
                for nm, v in zip(("rpi", "starts", "ends", "is_fg"), r):
                    out[k + "k3_" + nm] = v.numpy()
                r = nerfacc.ray_resampling_sdf_fine(T(packed), T(st)[:, None], T(en)[:, None], T(al), T(sd), n)
                for nm, v in zip(("rpi", "starts", "ends", "is_fg"), r):
                    out[k + "k4_" + nm] = v.numpy()
    out["n_cases"] = np.int32(case)
    # K2 on edge lists
    ecase = 0
    for n_rays, n in ((80, 16), (33, 4), (300, 16)):
        packed, vals, il, ir = gen_edges(rng, n_rays)
        for wmode in range(3):
            w = rng.uniform(0.0, [0.01, 0.2, 0.9][wmode], vals.shape[0]).astype(np.float32)
            w[~il] = 0.0
            k = f"e{ecase}_"
            ecase += 1
            out[k + "packed_info"], out[k + "vals"], out[k + "is_left"], out[k + "is_right"] = packed, vals, il, ir
            out[k + "weights"], out[k + "n"] = w, np.int32(n)
            r = nerfacc.ray_resampling_merge(T(packed), T(vals), T(il), T(ir), T(w), n)
            for nm, v in zip(("rpi", "vals", "dists", "is_left", "is_right", "is_resample", "is_fg"), r):
                out[k + "k2_" + nm] = v.numpy()
    out["n_edge_cases"] = np.int32(ecase)
    np.savez_compressed(f"{OUT}/golden_resampling.npz", **out)


def make_pack(nerfacc, rng):
    out = {}
    T = torch.from_numpy
    packed, st, en, al, sd = gen_packed_samples(rng, 200, 30)
    N = st.shape[0]
    data = rng.normal(size=(N, 3)).astype(np.float32)
    out.update(packed_info=packed, data=data)
    out["unpack_info"] = nerfacc.unpack_info(T(packed), N).numpy()
    out["unpack_mask"] = nerfacc.unpack_info_to_mask(T(packed), 32).numpy()
    out["unpack_data"] = nerfacc.unpack_data(T(packed), T(data), 32).numpy()
    # pack_data is pure torch (lib/nerfacc/pack.py:12-43) -- import by file path
    spec = importlib.util.spec_from_file_location("ref_pack", f"{REF}/lib/nerfacc/pack.py")
    sys.modules.setdefault("lib", types.ModuleType("lib"))
    sys.modules.setdefault("lib.nerfacc", types.ModuleType("lib.nerfacc"))
    sys.modules["lib.nerfacc.cuda"] = types.ModuleType("lib.nerfacc.cuda")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    d3 = rng.normal(size=(50, 16, 2)).astype(np.float32)
    mask = rng.random((50, 16)) < 0.4
    pd, pi = m.pack_data(T(d3), T(mask))
    out.update(pack_data_in=d3, pack_data_mask=mask, pack_data_out=pd.numpy(), pack_data_info=pi.numpy())
    np.savez_compressed(f"{OUT}/golden_pack.npz", **out)


def synthetic_rig(rng, D=8, H=32, W=32, n_pts=1500):
    """small synthetic 24-bone rig: smooth random skin-weight grid + random rigid tfs."""
    zz, yy, xx = np.meshgrid(np.linspace(-1, 1, D), np.linspace(-1, 1, H), np.linspace(-1, 1, W), indexing="ij")
    centres = rng.uniform(-0.8, 0.8, (24, 3))
    d2 = (xx[None] - centres[:, 0, None, None, None]) ** 2 + (yy[None] - centres[:, 1, None, None, None]) ** 2 \
        + (zz[None] - centres[:, 2, None, None, None]) ** 2
    wts = 1.0 / (d2 + 0.05) ** 2
    wts = (wts / wts.sum(0, keepdims=True)).astype(np.float16).astype(np.float32)[None]   # [1,24,D,H,W], fp16-exact so the fixture stays small
    tfs = np.tile(np.eye(4, dtype=np.float32), (1, 24, 1, 1))
    for j in range(24):
        ax = rng.normal(size=3)
        ax /= np.linalg.norm(ax)
        ang = rng.uniform(-0.5, 0.5)
        K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
        R = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K
        tfs[0, j, :3, :3] = R
        tfs[0, j, :3, 3] = rng.uniform(-0.1, 0.1, 3)
    offset = np.array([0.02, -0.3, 0.01], np.float32).reshape(1, 1, 3)       # offset_kernel (= -centre)
    scale = np.array([0.9, 0.9, 0.9 * 4], np.float32).reshape(1, 1, 3)        # scale_kernel (z * ratio)
    xd = rng.uniform(-1.0, 1.0, (1, n_pts, 3)).astype(np.float32)
    xd[..., 2] *= 0.25
    xd[..., 1] += 0.3
    return wts, tfs, offset, scale, xd


def make_snarf(snarf, rng):
    T = torch.from_numpy
    wts, tfs, offset, scale, xd = synthetic_rig(rng)
    _, _, D, H, W = wts.shape
    vd = torch.zeros(1, 3, D, H, W)
    vJ = torch.zeros(1, 12, D, H, W)
    snarf.precompute(T(wts), T(tfs), vd, vJ, T(offset), T(scale))
    bones = np.array([0, 1, 2, 4, 5, 10, 11, 12, 15, 16, 17, 18, 19], np.int32)
    n = xd.shape[1]
    x = torch.zeros(1, n, 13, 3)
    Ji = torch.zeros(1, n, 13, 3, 3)
    valid = torch.zeros(1, n, 13, dtype=torch.bool)
    snarf.fuse_broyden(x, T(xd), vd, vJ, T(tfs), T(bones), True, Ji, valid, T(offset), T(scale), 1e-5, 1e-1)
    mask = snarf.filter(x, valid)
    np.savez_compressed(f"{OUT}/golden_snarf.npz", voxel_w=wts.astype(np.float16), tfs=tfs, offset=offset, scale=scale,
                        xd=xd, bones=bones, voxel_d=vd.numpy(), voxel_J=vJ.numpy(), x=x.numpy(), J_inv=Ji.numpy(),
                        valid=valid.numpy(), filtered=mask.numpy())
    print("snarf: converged", int(valid.sum()), "after filter", int(mask.sum()), "of", valid.numel())


def make_mlp(rng):
    """reference VanillaMLP / LipshitzMLP (models/network_utils.py) imported with stubbed deps."""
    for name in ("tinycudann", "cv2", "pytorch_lightning", "pytorch_lightning.utilities",
                 "pytorch_lightning.utilities.rank_zero", "omegaconf", "systems", "systems.utils", "utils", "utils.misc"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["pytorch_lightning.utilities.rank_zero"].rank_zero_debug = lambda *a, **k: None
    sys.modules["pytorch_lightning.utilities.rank_zero"].rank_zero_info = lambda *a, **k: None
    sys.modules["systems.utils"].update_module_step = lambda *a, **k: None
    sys.modules["utils.misc"].config_to_primitive = lambda c: c
    sys.modules["utils.misc"].get_rank = lambda: 0
    sys.modules["omegaconf"].OmegaConf = object

    def load(name, path):
        spec = importlib.util.spec_from_file_location(name, path)
        m = importlib.util.module_from_spec(spec)
        sys.modules[name] = m
        spec.loader.exec_module(m)
        return m
    sys.modules.setdefault("models", types.ModuleType("models"))
    load("models.utils", f"{REF}/models/utils.py")
    nu = load("models.network_utils", f"{REF}/models/network_utils.py")
    torch.manual_seed(0)
    out = {}
    # SDF net: 35 -> 64 -> 13, sphere init, weight norm, Softplus(beta=100)
    sdf = nu.VanillaMLP(35, 13, dict(n_neurons=64, n_hidden_layers=1, sphere_init=True, sphere_init_radius=0.5,
                                     weight_norm=True, output_activation="none"))
    x = torch.cat([torch.rand(300, 3) * 2 - 1, torch.randn(300, 32) * 0.05], -1)
    # perturb so the test is not trivially the init
    with torch.no_grad():
        for p in sdf.parameters():
            p.add_(torch.randn_like(p) * 0.02)
    out["sdf_x"] = x.numpy()
    out["sdf_y"] = sdf(x).detach().numpy()
    for k, v in sdf.state_dict().items():
        out["sdf_sd_" + k] = v.numpy()
    # radiance net: 67 -> 64 -> 64 -> 3 ReLU (sigmoid applied outside, radiance.py:132-133)
    rad = nu.VanillaMLP(67, 3, dict(n_neurons=64, n_hidden_layers=2, output_activation="none"))
    x = torch.randn(300, 67) * 0.5
    out["rad_x"] = x.numpy()
    out["rad_y"] = rad(x).detach().numpy()
    for k, v in rad.state_dict().items():
        out["rad_sd_" + k] = v.numpy()
    # material net: Lipschitz 48 -> 64 -> 64 -> 5 (sigmoid applied outside, material.py:46-47)
    mat = nu.LipshitzMLP(48, 5, dict(n_neurons=64, n_hidden_layers=2, output_activation="none"))
    with torch.no_grad():
        for c in mat.lipshitz_bound_per_layer:
            c.mul_(0.3)        # make the Lipschitz clamp active
    x = torch.randn(300, 48) * 0.5
    out["mat_x"] = x.numpy()
    out["mat_y"] = mat(x).detach().numpy()
    for k, v in mat.state_dict().items():
        out["mat_sd_" + k] = v.numpy()
    np.savez_compressed(f"{OUT}/golden_mlp.npz", **out)


def main():
    assert os.path.isdir(REF), "needs /root/reference (build container only)"
    nerfacc, snarf = build_reference_host_modules()
    rng = np.random.default_rng(20260928)
    make_resampling(nerfacc, rng)
    make_pack(nerfacc, rng)
    make_snarf(snarf, rng)
    make_mlp(rng)
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(f"{OUT}/{f}") // 1024, "KiB")


if __name__ == "__main__":
    main()
