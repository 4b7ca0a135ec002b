#!/usr/bin/env python3
"""bench.py -- render_step throughput on MI355X (BASELINE.json metric: rays/sec (fwd+bwd) at 540x540, 1024 spp).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: either launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`, or started
   plainly -- WORLD_SIZE unset -- in which case bench.py re-executes itself under torch.distributed.run, one rank per GPU)

Default workload (`--workload headline`) = the configuration the metric is quoted on: ONE training-form pass
(forward + backward + optimiser step) of the render_step hot path over one 540x540 frame (291 600 primary rays) WITH the
physically based branch at samples_per_pixel = 1024: primary march + 2x importance resampling, fast-SNARF deformer,
SDF / radiance / material fields, volume-interaction re-sampling (1024 per ray), one light-importance-sampled secondary
ray per foreground re-sample (render_mode=light, training form of pbr_light_forward: models/intrinsic_avatar.py:755-861),
secondary march + zero-crossing resampling + shading of every secondary ray (compute_indirect_radiance, :396-545,
global_illumination on), Monte-Carlo estimator, composite, L1/eikonal/mask losses, backward to the hash grids, MLPs,
density, material head and the spherical-Gaussian environment light, fused Adam.  The frame is processed in ray chunks
of --ray-chunk rays with gradient accumulation (the reference trains on 4096-ray batches and tests on 4096-ray chunks;
288 GB of HBM take the whole 291 600-ray frame and 16 Mi secondary rays at a time: the default is one chunk).  `--workload config2` times BASELINE configs[1] (128 samples/ray, radiance + SDF geometry, no PBR branch) --
round 1's number, kept as a second key of the headline line (`config2_ms_per_step`).

Synthetic data: random-init networks of the reference's architecture, synthetic 24-bone rig, procedural light; all
inputs resident in HBM when the timed region starts; random numbers are drawn on the device inside the step (the
reference draws them per step too).  Multi-GPU: frames shard across ranks with replicated parameters (weak scaling: one
frame per rank per step), gradients all-reduced over RCCL.  Rank 0 prints ONE JSON line.

Extra objects in the line:
  roofline     -- the dominant C-ABI entry point of the step (by HIP-event time measured live on the launch stream).  For the
                  Broyden search ("cache-gather latency") it is priced against the bound that binds it: bytes
                  through the vector-memory (TCP/L1) path from the COUNTED trilinear fetches against 256 CUs x 64 B/clk;
                  `traffic` = PMC bytes per launch from the committed rocprofv3 passes of the same command, `hbm_side` what
                  that is against the HBM peak.  Any other dominant kernel is priced against HBM with SURVEY 8(d)'s bytes;
  mfma         -- the SDF head of the no-grad queries in-step: useful FLOP / live time against the fp32 MFMA peak;
  cpu_baseline -- the CPU oracle (a port; forward only) on a bounded ray sample of the same frame, rank 0 / N == 1.
"""
import argparse
import contextlib
import json
import os
import sys
import time

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")      # before the HIP runtime initialises (intrinsicavatar_amd/__init__.py: launch path)

# one process per GPU: the host side of a rank is one launch thread.  Library thread pools default to the number of
# visible CPUs (256 on the GPU boxes, 8 ranks per node) while a container's CPU quota is a fraction of that; an
# oversubscribed pool burns the cgroup quota and the launch thread gets throttled with it.
os.environ.setdefault("OMP_NUM_THREADS", "8")
for _v in ("OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "NUMEXPR_NUM_THREADS"):      # also keeps cpu_baseline's "cores": 1 honest
    os.environ.setdefault(_v, "1")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3     # fp32-in MFMA dense peak
L1_PEAK_GBPS = 256 * 64 * 2.4    # 256 CUs x 64 B/clk (vector L1 / TCP) x 2.4 GHz = 39.3 TB/s
L2_PEAK_GBPS = 34500.0           # MI355X_MICROARCH.md: L2 (per XCD 4 MiB, aggregate) ~34.5 TB/s
VOXEL_J_BYTES = 32 * 128 * 128 * 48
PROFILE_TAGS = ("r06", "r05", "r04", "r03", "r02")          # newest committed rocprofv3 PMC summary that knows the kernel wins


def algorithmic_bytes(name, calls, extra=None):
    """SURVEY.md 8(d) per-unit algorithmic bytes x the COUNTED units of every launch of one entry point.
    calls = [(ms, units, extras)].  Returns total bytes over the calls (None: no model for this entry point)."""
    n = sum(u for _, u, _ in calls)
    if name in ("ia_fuse_broyden", "ia_fuse_broyden_spec", "ia_fuse_broyden_spec_rows"):
        # SURVEY 8(d) row "Broyden (a7/K8)": per (point, init) 12 B target + 64 B bone row in, (1 + iters) x 8 corners x 48 B
        # gathered (SURVEY gives the <= 11-fetch upper bound; here the fetches are COUNTED by ia_broyden_stats: `extra`
        # = in-range corner loads of the step), 13 B out (x + valid; +36 B each for J_inv / fwd_J when requested).
        tot = 0
        for _, u, ex in calls:
            tot += u * ex["I"] * (12 + 64 + 13 + (36 if ex["J_inv"] else 0) + (36 if ex["fwd_J"] else 0))      # SURVEY 8(d) row (per item)
        return tot + (extra or 0) * 48
    if name in ("ia_hashgrid_fwd", "ia_hashgrid_fwd_xcd"):
        return n * (12 + 1024 + 128)             # 12 B in + 16 levels x 8 corners x 8 B gathered + 128 B out
    if name in ("ia_hashgrid_bwd", "ia_hashgrid_bwd_binned"):
        return n * (12 + 128 + 1024 + 1024)      # read-modify-write of the touched entries
    if name == "ia_sdf_fused":
        return n * (12 + 1024 + 4)               # encode + MLP fused: point in, gathers, one SDF out
    if name == "ia_accumulate_along_rays":
        return n * (4 + 8 + 12) + 0              # w, ray index, rgb per sample (ray outputs negligible)
    return None


# C-ABI entry point -> (mode, substrings of the kernels it launches in the rocprofv3 kernel names).  "alt": ONE of the kernels runs
# per call (per-launch traffic = launch-weighted mean); "seq": all of them run per call (sum of their per-call traffic)
PMC_KERNELS = {
    "ia_fuse_broyden": ("alt", ["broyden_persistent2_kernel", "broyden_persistent_kernel", "broyden_kernel"]),
    "ia_fuse_broyden_spec": ("alt", ["broyden_spec_kernel"]),
    "ia_fuse_broyden_spec_rows": ("alt", ["broyden_spec_kernel"]),
    "ia_hashgrid_fwd": ("alt", ["hash_fwd_kernel"]),
    "ia_hashgrid_fwd_xcd": ("seq", ["hash_fwd_xcd_kernel", "hash_transpose_kernel"]),
    "ia_hashgrid_bwd_binned": ("seq", ["hash_bin", "hash_reduce_kernel"]),
    "ia_mlp_fwd": ("alt", ["mlp_fwd_kernel<0", "mlp_fwd_kernel<1", "mlp_fwd_kernel<2"]),
    "ia_sdf_levels_fwd": ("alt", ["mlp_fwd_kernel<3"]),
}


def pmc_traffic(entry, calls_per_step=None):
    """(HBM-side bytes per launch of the entry point, source) from the committed rocprofv3 PMC passes of THIS command
    (profiles/<tag>_pmc_traffic.json: 2 x FETCH_SIZE + WRITE_SIZE per MI355X_MICROARCH.md), or (None, None)."""
    if entry not in PMC_KERNELS:
        return None, None
    mode, subs = PMC_KERNELS[entry]
    for tag in PROFILE_TAGS:
        path = os.path.join(ROOT, "profiles", f"{tag}_pmc_traffic.json")
        if not os.path.exists(path):
            continue
        ks = json.load(open(path))["kernels"]
        hits = [v for k, v in ks.items() if any(sub in k for sub in subs)]
        if not hits:
            continue
        tot = sum(v["hbm_side_bytes_per_launch"] * v["launches"] for v in hits)
        calls = sum(v["launches"] for v in hits) if mode == "alt" else max(v["launches"] for v in hits)
        return int(tot / max(calls, 1)), f"profiles/{tag}_pmc_traffic.json"
    return None, None


def pmc_kernel(substr):
    """(counters of the first kernel whose name contains `substr` in the newest committed PMC summary, its path) or (None, None)."""
    for tag in PROFILE_TAGS:
        path = os.path.join(ROOT, "profiles", f"{tag}_pmc_traffic.json")
        if not os.path.exists(path):
            continue
        for k, v in json.load(open(path))["kernels"].items():
            if substr in k:
                return v, f"profiles/{tag}_pmc_traffic.json"
    return None, None


def cgroup_throttle():
    """(nr_throttled, throttled_usec) of this container's CPU controller, or None (diagnostic: a throttled launch thread
    makes the step host-bound)."""
    try:
        kv = dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat").read().strip().splitlines())
        return int(kv.get("nr_throttled", 0)), int(kv.get("throttled_usec", 0))
    except (OSError, ValueError):
        return None


def build_headline(dev, hw, spp, rank, pose):
    """the synthetic frame + material head + training-time light of the headline workload."""
    from intrinsicavatar_amd import synthetic as S, fields, pbr
    rs, rays, export = S.build_frame(dev, hw, hw, pose=pose, beta=0.01, num_samples_per_ray=128)
    mat = fields.VolumeMaterial(seed=2).to(dev)
    sg = pbr.EnvironmentLightSG(num_SGs=64, base_res=256, seed=4).to(dev)       # configs/light/envlight_SG.yaml
    return rs, rays, export, mat, sg


def build_config4_step(rs, rays, mat, sg, dev, bg, n_batch=4096, spp=512, seed=4, sync=None, grad_scale=1.0, eik_denominator=None):
    """the reference's own training batch (BASELINE configs[3]; configs/sampler/edge.yaml:2, configs/config.yaml:46-48: uniform_light, spp 512):
    n_batch rays on the subject of the bench frame, RenderStep.forward_backward_phys + the fused Adam step.  -> (step callable, workload
    string) or (None, None) on a frame without the subject.  sync: parallel.OverlappedGradientAllReduce of a multi-rank run (the step
    then ends with sync.finish() before the optimiser step; grad_scale = 1 / world makes the summed all-reduce DDP's mean);
    eik_denominator: see RenderStep.forward_backward_phys (global sample count / world under ray-batch sharding)."""
    from intrinsicavatar_amd import optim, pbr
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        hit = torch.nonzero(rs.forward(rays)["opacity"][:, 0] > 0.5)[:, 0]          # pixels on the subject, like the trainer's fg sampler
    if hit.shape[0] == 0:
        return None, None                                           # a frame without the subject (tiny --hw): nothing to sample
    sel = hit[torch.randint(0, hit.shape[0], (n_batch,), generator=g).to(dev)]
    batch = rays[sel].contiguous()
    target = torch.rand((n_batch, 3), generator=g).to(dev)
    tmask = torch.ones(n_batch, device=dev)
    light_u = torch.rand((spp, 3), generator=g).to(dev)             # uniform_light: one stratified direction set per step (intrinsic_avatar.py:1392-1400)
    shuffle_u = torch.rand((n_batch, spp), generator=g).to(dev)     # the per-ray spp shuffle
    params = rs.parameters() + [p for p in mat.parameters() if p.requires_grad] + list(sg.parameters())
    opt, sched = optim.reference_optimizer(rs, grad_scale=grad_scale, material=mat, emitter=sg)

    def s4(local=False):
        """local=True: the same step without its exchange (no collective is issued: gradient hooks held, no finish())."""
        for p in params:
            p.grad = None
        img = sg.generate_image()
        leaf = img.detach().requires_grad_(True)
        emitter = pbr.EnvironmentLightTensor(leaf.detach())
        emitter.update_pdf()
        ctx = sync.no_sync() if (sync is not None and local) else contextlib.nullcontext()
        with ctx:
            o = rs.forward_backward_phys(batch, target, mat, emitter, spp, light_u, shuffle_u, target_mask=tmask, render_mode="uniform_light",
                                         env_base=leaf, background_color=bg, eik_denominator=(None if local else eik_denominator))
            if leaf.grad is not None:
                img.backward(leaf.grad)
        if sync is not None and not local:
            s4.reduced_bytes = sync.finish()
        opt.step()
        sched.step()
        return o
    s4.reduced_bytes = 0
    s4.params = params
    return s4, f"{n_batch} rays on the subject of the bench frame, PBR training step (uniform_light, spp {spp}), fwd+bwd+Adam"


def measure_config4(rs, rays, mat, sg, dev, bg, n_batch=4096, spp=512, steps=20, rank=0, world=1, rccl_at_one_rank=False):
    """The line's `config4` object -- the reference's 4096-ray training batch per GPU: ms per step, rays/s (whole job: world x n_batch rays per
    step, ray-batch sharding of ONE frame with replicated parameters, BASELINE configs[3]), the search launches of one step (points, ms) --
    the step's one large search batch is its secondary march (~2 100 march points per ray), not the 4096 rays' own samples -- and the
    host-orchestration figures of SURVEY 8(f) row 2 for one step: device launches, host read-backs, idle fraction of the device time line
    (tools/launch_audit.py: torch.profiler + sync debug mode).
    world > 1 (every rank calls this; rank 0 returns the object, the others None): each rank draws its own 4096 rays, the eikonal mean is
    normalised with the GLOBAL sample count (one scalar all-reduce per step, systems/intrinsic_avatar.py:235-239), gradients meet in the
    all-reduce (hook-launched for the two hash tables, one flat bucket for the rest), Adam averages them (grad_scale 1 / world).
    `gradient_allreduce`: bytes per step, ms_total (the exchange alone, back to back on idle GPUs), ms_exposed (step with the exchange -
    the same step without it, max over ranks); `sparse_exchange`: what an all-gather of the touched (index, value) pairs would move instead
    (SURVEY 8(e)), costed from the touched entries of this step's table gradients and, with a process group, timed."""
    from intrinsicavatar_amd import _lib as L, parallel
    import torch.distributed as dist
    group = dist.is_available() and dist.is_initialized()
    use_sync = group and (world > 1 or rccl_at_one_rank)
    probe, wl = build_config4_step(rs, rays, mat, sg, dev, bg, n_batch, spp, seed=4 + rank)          # (also tells whether the frame has the subject)
    ok = torch.tensor([1.0 if probe is not None else 0.0], device=dev)
    if group and world > 1:
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if float(ok) == 0.0:
        return None
    sync = parallel.OverlappedGradientAllReduce(probe.params, single_rank_too=rccl_at_one_rank) if use_sync else None
    den = (lambda n: parallel.allreduce_scalars([float(n)], dev)[0] / world) if (group and world > 1) else None
    s4, wl = build_config4_step(rs, rays, mat, sg, dev, bg, n_batch, spp, seed=4 + rank, sync=sync, grad_scale=1.0 / world, eik_denominator=den)
    del probe

    def timed(fn, k):
        torch.cuda.synchronize()
        if group and world > 1:
            dist.barrier()
        tc = time.perf_counter()
        for _ in range(k):
            o_ = fn()
        torch.cuda.synchronize()
        if group and world > 1:
            dist.barrier()
        dt_ = time.perf_counter() - tc
        if group and world > 1:
            t_ = torch.tensor([dt_], device=dev)
            dist.all_reduce(t_, op=dist.ReduceOp.MAX)
            dt_ = float(t_)
        return dt_ / k * 1e3, o_
    for _ in range(10):                 # (the first ~30 steps of a process drift down by ~5 %: allocator pools and clocks settle)
        o = s4()
    ms, o = timed(s4, steps)
    comm = sparse = None
    if sync is not None:
        for _ in range(2):
            s4(local=True)
        ms_local, _ = timed(lambda: s4(local=True), steps)
        s4()                                                        # leaves this rank's gradients of one step in .grad
        torch.cuda.synchronize()
        big = [p for p in s4.params if p.grad is not None and p.numel() >= parallel.BIG]
        nnz = [int((p.grad != 0).sum()) for p in big]
        ms_ar, _ = timed(lambda: parallel.allreduce_gradients(s4.params), 5)
        comm = dict(backend=dist.get_backend(), world=dist.get_world_size(), bytes_per_step=int(s4.reduced_bytes), ms_total=round(ms_ar, 3),
                    ms_exposed=round(ms - ms_local, 3), ms_per_step_without_the_exchange=round(ms_local, 3),
                    fraction_of_step_exposed=round((ms - ms_local) / ms, 4),
                    note="ms_total: the all-reduce alone (two table gradients + one flat bucket), 5 back-to-back repeats on otherwise idle GPUs; "
                         "ms_exposed: step with the exchange - the same step with the hooks held and no finish(), max over ranks")
        # the sparse alternative of SURVEY 8(e): all-gather of the touched (int32 index, fp32 value) pairs of the two tables, padded to the
        # largest rank, then a scatter-add of world x that many pairs on every rank
        cap = torch.tensor([float(sum(nnz))], device=dev)
        if world > 1:
            dist.all_reduce(cap, op=dist.ReduceOp.MAX)
        pairs = int(cap)
        payload = torch.zeros(pairs * 2, dtype=torch.int32, device=dev)
        gathered = torch.empty(pairs * 2 * dist.get_world_size(), dtype=torch.int32, device=dev)
        try:
            ms_ag, _ = timed(lambda: dist.all_gather_into_tensor(gathered, payload), 5)
        except Exception:                   # a backend without all_gather_into_tensor on device tensors (the gloo test hook)
            ms_ag = float("nan")
        dense = sum(p.numel() * 4 for p in big)
        sparse = dict(touched_entries_per_table=nnz, entries_per_table=[int(p.numel()) for p in big], pairs_per_rank_padded=pairs,
                      bytes_sent_per_rank=pairs * 8, bytes_received_per_rank=pairs * 8 * dist.get_world_size(), dense_allreduce_bytes=dense,
                      ms_all_gather=(round(ms_ag, 3) if ms_ag == ms_ag else None), ms_dense_all_reduce=round(ms_ar, 3),
                      note="all_gather_into_tensor of the padded pair lists timed like ms_total; the scatter-add of world x pairs that would "
                           "follow is not included (it is the binned hash backward's own final pass)")
        del payload, gathered
    if rank != 0:
        if sync is not None:
            sync.remove()
        return None
    lib = L.lib()
    lib.start()
    o = s4(local=True) if sync is not None else s4()
    det = lib.report(detail=True)
    search = [(round(c[0], 3), int(c[1])) for k in ("ia_fuse_broyden_spec_rows", "ia_fuse_broyden") for c in det.get(k, [])]
    host = None
    try:
        from tools import launch_audit as LA
        a = LA.audit((lambda: s4(local=True)) if sync is not None else s4, warm=0)
        host = dict(launches=a["device_launches"], aten_or_runtime_launches=a["aten_or_runtime_launches"], readbacks=a["readbacks"],
                    idle_frac=a["idle_frac"], span_ms_under_profiler=a["span_ms"], busy_ms_under_profiler=a["busy_ms"])
    except Exception as e:              # a diagnostic must not take the measurement down
        host = dict(error=f"{type(e).__name__}: {e}")
    if sync is not None:
        sync.remove()
    return dict(workload=wl + (f"; {world} ranks x {n_batch} rays of one frame (ray-batch sharding, replicated parameters)" if world > 1 else ""),
                n_gpus=world, ms_per_step=round(ms, 3), rays_per_s=round(world * n_batch / (ms * 1e-3), 1),
                secondary_rays_per_step=int(o["stats"]["n_secondary"]),
                kernel_ms_per_step=round(sum(c[0] for v in det.values() for c in v), 3),
                idle_frac_untraced_upper_bound=round(max(0.0, 1.0 - sum(c[0] for v in det.values() for c in v) / max(ms, 1e-9)), 4),
                idle_note="idle_frac is the device time line of one step UNDER torch.profiler, whose per-launch overhead widens every gap; "
                          "untraced, 1 - kernel_ms_per_step / ms_per_step bounds the idle share from above (kernel_ms_per_step = live HIP events "
                          "around the C-ABI entry points only: the ~0.35 ms per step of ATen kernels count as idle in it); the rocprofv3 kernel "
                          "trace of the same step (profiles/r06_config4_timeline.json) has the kernels busy 13.5 ms per step",
                search_launches_ms_points=search, gradient_allreduce=comm, sparse_exchange=sparse,
                host_figures_of="one step of this rank without the exchange" if sync is not None else "one step", **(host or {}))


def main():
    if os.environ.get("IA_SWITCH_INTERVAL"):          # experiment: the interpreter's thread switch interval (two host threads feed two streams)
        sys.setswitchinterval(float(os.environ["IA_SWITCH_INTERVAL"]))
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--hw", type=int, default=540)
    ap.add_argument("--spp", type=int, default=1024)
    ap.add_argument("--ray-chunk", type=int, default=int(os.environ.get("IA_RAY_CHUNK", str(1 << 19))))
    ap.add_argument("--workload", choices=["headline", "config2", "config4"], default="headline",
                    help="headline: the 540x540 frame at 1024 spp (BASELINE metric); config2: configs[1] (no PBR branch); config4: configs[3] -- the "
                         "reference's 4096-ray training batch per GPU, ray-batch sharded over the ranks (also a key of the headline line at every N)")
    ap.add_argument("--pose", default=os.environ.get("IA_BENCH_POSE", "male-3-casual:0"),
                    help="frame of the reference's pose files through plain FK: male-3-casual:{0,40,80,113} (peoplesnapshot training frames), "
                         "aist:{0,100,200,319} (animation, out of distribution), or synthetic:K (N(0, 0.25) joint angles, rounds 1-2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-config2", action="store_true", help="skip the secondary configs[1] measurement of the headline line")
    ap.add_argument("--no-config4", action="store_true", help="skip the 4096-ray training-batch measurement (configs[3] shape) of the headline line")
    ap.add_argument("--no-breakdown", action="store_true", help="skip the instrumented repeat (profiling runs)")
    ap.add_argument("--no-search-modes", action="store_true", help="skip the search-to-the-end timing / comparison of the deformer search")
    ap.add_argument("--no-optimizer", action="store_true", help="time fwd+bwd only (no Adam step)")
    ap.add_argument("--aten-profile", default=None, help="diagnostic: one extra step under torch.profiler; the ATen operators by device time "
                                                          "(with input shapes) go to this file")
    ap.add_argument("--pass", dest="mode", choices=["fwd+bwd", "fwd"], default="fwd+bwd",
                    help="fwd+bwd = training-step form of render_step (BASELINE metric); fwd = inference form (config2 only)")
    args = ap.parse_args()

    self_launch_ranks(args.gpus)                 # plain `python bench.py --gpus N`: start the N ranks ourselves
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (no CPU fallback in the product path)"
    # test hook for 1-GPU boxes: IA_BENCH_SHARE_GPU=1 puts every rank on cuda:0 and uses gloo (RCCL refuses two ranks on
    # one device) so the N>1 control flow can be exercised without an 8-GPU node; such a run is not a measurement
    share_gpu = os.environ.get("IA_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    # host threads spin while they wait for the GPU (HIP's default when CPUs outnumber GPUs): one rank burns ~1 CPU doing
    # so, eight ranks plus their RCCL proxies can exhaust a container's CPU quota and throttle each other.  With more than
    # one rank per host, waits block on an interrupt instead (costs ~10 us per size read-back, frees the CPUs).
    blocking = os.environ.get("IA_BLOCKING_SYNC", "1" if world > 1 else "0") == "1"
    if blocking:
        import ctypes
        try:
            _hip = ctypes.CDLL("libamdhip64.so")
            _hip.hipSetDevice(ctypes.c_int(local_rank))               # the flag belongs to THIS rank's device, not to device 0
            _rc = _hip.hipSetDeviceFlags(ctypes.c_uint(0x4))          # hipDeviceScheduleBlockingSync
            blocking = (_rc == 0)
        except OSError:
            blocking = False
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    # IA_BENCH_FORCE_RCCL=1 (test hook for 1-GPU boxes): a ONE-rank RCCL group and the all-reduce hooks even at N = 1, so that
    # init_process_group("nccl"), the hook-launched all-reduces on device tensors and finish() run through RCCL on this box
    force_rccl = world == 1 and os.environ.get("IA_BENCH_FORCE_RCCL") == "1"
    if world > 1 or force_rccl:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if force_rccl:
            import socket
            with socket.socket() as s_:
                s_.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(s_.getsockname()[1]))
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(dev))
        elif share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device(dev))

    from intrinsicavatar_amd import parallel as _par
    numa_node = _par.pin_to_gpu_numa_node(local_rank)          # host side of the step next to its GPU (2-socket hosts)
    # Buffer sizes follow the sample counts, which move a little every step once the optimiser updates the geometry; with
    # exact-size caching the allocator keeps growing and a multi-GiB hipMalloc on a freshly booted box costs ~100 ms.
    # Size classes (1/8 power-of-two steps) make the blocks of one step reusable by the next, and one up-front
    # reservation moves the remaining growth in front of the warm-up.  288 GB of HBM: the arena is under half of the device.
    torch.cuda.memory._set_allocator_settings("roundup_power2_divisions:8")
    if not share_gpu:
        # (with the secondary march on its own streams most of the working set lives in THEIR pools: a smaller reservation for the caller's)
        multi = int(os.environ.get("IA_SECONDARY_STREAMS", "2")) > 1
        _arena = torch.empty(int(os.environ.get("IA_BENCH_ARENA_GIB", ("16" if multi else "136") if args.workload == "headline" else "24")) << 30,
                             dtype=torch.uint8, device=dev)
        del _arena
    from intrinsicavatar_amd import build
    _build = build
    if rank == 0:
        build.build()
    if world > 1:
        dist.barrier()
    from intrinsicavatar_amd import _lib as L, parallel, optim, pbr

    headline = args.workload == "headline"
    rs, rays, export, mat, sg = build_headline(dev, args.hw, args.spp, rank, args.pose)
    n_rays = rays.shape[0]
    if args.workload == "config4":
        # BASELINE configs[3] on its own: K timed steps of the 4096-ray training batch per rank (ray-batch sharding of one frame)
        c4 = measure_config4(rs, rays, mat, sg, dev, torch.ones(3, device=dev), steps=max(args.steps, 1), rank=rank, world=world,
                             rccl_at_one_rank=force_rccl)
        if rank == 0:
            line = {"metric": "rays/sec (fwd+bwd), 4096-ray training batches per GPU (BASELINE configs[3]: uniform_light, spp 512)",
                    "value": c4 and c4["rays_per_s"], "unit": "rays/s", "n_gpus": world, "steps": max(args.steps, 1), "warmup": 10,
                    "ms_per_step": c4 and c4["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
                    "data": "synthetic", "config": {"workload": c4 and c4["workload"], "pose": args.pose, "parallelism": f"ray-batch sharding x{world}",
                                                    "library_sources_sha256_16": _build.source_fingerprint(), "library_built_from": _build.built_fingerprint()},
                    "roofline": None, "cpu_baseline": None, "config4": c4}
            _flush_c_stdout()
            print(json.dumps(line), flush=True)
        if world > 1 or force_rccl:
            dist.destroy_process_group()
        _flush_c_stdout()
        return
    # one frame per rank (frame-/ray-batch sharding, replicated parameters).  Weak scaling = the SAME per-GPU workload at
    # every N: each rank renders the same frame (pose 0) against its own target image and draws its own random numbers,
    # so the per-rank work at N=8 is the N=1 work and the gradients that meet in the all-reduce still differ per rank.
    g = torch.Generator().manual_seed(1234 + rank)
    target_rgb = torch.rand((n_rays, 3), generator=g).to(dev)
    target_mask = (torch.rand(n_rays, generator=g) > 0.5).float().to(dev)
    torch.manual_seed(99 + rank)
    bg = torch.ones(3, device=dev)
    frame_pipeline = int(os.environ.get("IA_FRAME_PIPELINE", "0"))       # experiment: n half-frames in flight (train_phys.forward_backward_phys_pipelined)
    if frame_pipeline > 1 and args.workload == "headline":
        args.ray_chunk = -(-n_rays // frame_pipeline)
    chunks = [(c0, min(c0 + args.ray_chunk, n_rays)) for c0 in range(0, n_rays, args.ray_chunk)]
    chunk_views = [(rays[a:b].contiguous(), target_rgb[a:b].contiguous(), target_mask[a:b].contiguous(), (b - a) / n_rays)
                   for a, b in chunks]

    params2 = rs.parameters()
    params = params2 + ([p for p in mat.parameters() if p.requires_grad] + list(sg.parameters()) if headline else [])

    def zero_grads():
        for p in params:
            p.grad = None

    def step_config2(sync=None):
        if args.mode == "fwd":
            return rs.forward(rays)
        for p in params2:
            p.grad = None
        out = rs.forward_backward(rays, target_rgb, target_mask)
        return out

    totals = {}

    def step_headline(sync=None):
        """one 540x540 frame: fwd + bwd with the PBR branch at `spp`, chunked with gradient accumulation."""
        zero_grads()
        # training-time light: SG lobes -> equirect image (differentiable); the chunks share one image and its gradient
        # is accumulated in a leaf and pushed through generate_image() once at the end
        img = sg.generate_image()
        leaf = img.detach().requires_grad_(True)
        emitter = pbr.EnvironmentLightTensor(leaf.detach())
        emitter.update_pdf()                                        # pbr_light_forward: `if self.training: update_pdf()`
        tot = dict(n_samples=0, n_edges0=0, n_samples0=0, n_resampled=0, n_fg=0, n_secondary=0)
        if frame_pipeline > 1 and sync is None and len(chunk_views) > 1:
            from intrinsicavatar_amd import train_phys as _tp
            got = _tp.forward_backward_phys_pipelined(rs, chunk_views, mat, emitter, args.spp, n_workers=frame_pipeline, render_mode="light",
                                                      env_base=leaf, background_color=bg, global_illumination=True, light_sampling="per_point")
            for k in tot:
                tot[k] = int(got.get(k, 0))
            if leaf.grad is not None:
                img.backward(leaf.grad)
            totals.update(tot)
            return dict(stats=tot)
        for ci, (r, t, m, frac) in enumerate(chunk_views):
            last = ci == len(chunk_views) - 1
            ctx = sync.no_sync() if (sync is not None and not last) else _null()
            with ctx:
                o = rs.forward_backward_phys(r, t, mat, emitter, args.spp, None, None, target_mask=m, render_mode="light",
                                             env_base=leaf, background_color=bg, global_illumination=True,
                                             light_sampling="per_point", loss_scale=frac)
            for k in tot:
                tot[k] += int(o["stats"].get(k, 0))
            del o
        if leaf.grad is not None:
            img.backward(leaf.grad)
        totals.update(tot)
        return dict(stats=tot)

    import contextlib
    _null = contextlib.nullcontext

    # one-time initialisation (not a step, and BEFORE the all-reduce hooks exist): the first launches load the code
    # objects and set kernel attributes; done on a 4096-ray slice so that the W warm-up steps see a warm library
    r0, t0_, m0 = rays[:4096].contiguous(), target_rgb[:4096].contiguous(), target_mask[:4096].contiguous()
    if headline:
        img0 = sg.generate_image().detach()
        e0 = pbr.EnvironmentLightTensor(img0)
        e0.update_pdf()
        rs.forward_backward_phys(r0, t0_, mat, e0, 64, None, None, target_mask=m0, render_mode="light",
                                 background_color=bg, global_illumination=True, light_sampling="per_point")
    elif args.mode == "fwd":
        rs.forward(r0)
    else:
        rs.forward_backward(r0, t0_, m0)
    zero_grads()
    torch.cuda.synchronize()

    # the one exchange step of the path: all-reduce(sum) of the gradients over RCCL/xGMI; the two 50 MB
    # hash-table gradients are launched from autograd hooks as soon as they are complete (overlap with the rest of backward)
    train = headline or args.mode != "fwd"
    sync = parallel.OverlappedGradientAllReduce(params, single_rank_too=force_rccl) if ((world > 1 or force_rccl) and train) else None
    reduced_bytes = [0]
    # the optimiser step of the iteration is inside the timed region: torch.optim.Adam semantics with the reference's
    # parameter groups and scheduler (configs/config.yaml:110-155), one fused launch; the summed all-reduce becomes DDP's
    # mean through grad_scale = 1/world
    opt = sched = None
    if train and not args.no_optimizer:
        opt, sched = optim.reference_optimizer(rs, grad_scale=1.0 / world, material=mat if headline else None,
                                               emitter=sg if headline else None)

    def step():
        out = step_headline(sync) if headline else step_config2(sync)
        if sync is not None:
            reduced_bytes[0] = sync.finish()
        if opt is not None:
            opt.step()
            sched.step()
        return out

    # CPU baseline + parity at the bench frame, BEFORE the first optimiser step: the oracle's scene bundle holds the parameters as
    # initialised, and the same rays go through the GPU path with the same random numbers right after the workers finish
    cpu = parity = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from intrinsicavatar_amd import synthetic as _S
        env0 = sg.generate_image().detach() if headline else None
        cpu, parity = cpu_baseline(rays, export, n_rays, headline, args.spp, _S.export_phys(mat, env0) if headline else None,
                                   gpu_model=(rs, mat if headline else None, env0))

    for _ in range(args.warmup):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    _flush_c_stdout()            # RCCL's start-up banner (C stdio, block-buffered on a pipe) leaves every rank NOW, not after the JSON line
    lib = L.lib()
    # ---- timed region: EXACTLY K steps, barrier + synchronize on both sides, nothing else running
    torch.cuda.reset_peak_memory_stats()          # the peaks reported are those of the timed steps (not of the up-front arena)
    thr0 = cgroup_throttle()
    cpu0 = time.process_time()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    streams_taken = getattr(rs, "last_secondary_streams", None)      # what the timed steps ran (RenderStep._secondary_streams_for: memory-gated)
    peak_alloc_timed, peak_reserved_timed = torch.cuda.max_memory_allocated(), torch.cuda.max_memory_reserved()
    cpu_busy = (time.process_time() - cpu0) / max(dt, 1e-9)          # CPUs this rank kept busy during the timed region
    thr1 = cgroup_throttle()
    # ---- ONE more step with a HIP-event pair around every C-ABI launch (on the launch stream): per-kernel durations
    # and counted units for the roofline / breakdown.  Kept out of the throughput region because the event records cost
    # host time that the un-instrumented step does not pay (ms_per_step_instrumented is reported next to it).
    if args.aten_profile and rank == 0:
        from torch.profiler import profile, ProfilerActivity
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
            step()
            torch.cuda.synchronize()
        with open(args.aten_profile, "w") as f:
            f.write(prof.key_averages(group_by_input_shape=True).table(sort_by="self_cuda_time_total", row_limit=60, max_name_column_width=60))
    detail, dt_instr, k_instr = {}, 0.0, 0
    # The instrumented step runs the secondary march on ONE stream: a roofline prices a kernel that has the device to itself, and the
    # timed region's streams (render.RenderStep.SECONDARY_STREAMS) let launches of two chunks share it.
    timed_streams = rs.SECONDARY_STREAMS
    rs.SECONDARY_STREAMS = 1
    if timed_streams > 1 and not args.no_breakdown:
        torch.cuda.empty_cache()          # the working set moves from the side streams' pools to the caller's: hand the cached blocks back first
        step()                            # ... and one untimed step to fill the caller's pool again (~130 GiB of hipMalloc)
        torch.cuda.synchronize()
    if not args.no_breakdown:
        k_instr = 1 if headline else args.steps
        lib.start()
        t1 = time.perf_counter()
        for _ in range(k_instr):
            out = step()
        torch.cuda.synchronize()
        dt_instr = time.perf_counter() - t1
        detail = lib.report(detail=True)
        if os.environ.get("IA_BENCH_DETAIL"):       # per-launch (ms, units) of the instrumented step(s), for tuning
            for name_, calls_ in sorted(detail.items(), key=lambda kv: -sum(c[0] for c in kv[1]))[:10]:
                print(name_, [(round(c[0], 2), c[1]) for c in calls_], file=sys.stderr)
    rs.SECONDARY_STREAMS = timed_streams
    if timed_streams > 1 and not args.no_breakdown:
        torch.cuda.empty_cache()
    if world > 1:
        t = torch.tensor([dt], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    value = world * n_rays * args.steps / dt

    # ---- the deformer search of the timed path against the reference's search-to-the-end (fuse_cuda_kernel_fast.cu:252-452 +
    # filter.cu:10-54), IN THIS RUN: (i) the same step with the early filter off (spec_eps = 0: every search runs to its end, K9 as a
    # pass) timed next to the timed figure; (ii) on march points of this frame, the candidate sets / min-SDF of the two searches compared
    search_modes = None
    if headline and rank == 0 and world == 1 and not args.no_search_modes and rs.deformer.spec_eps > 0:
        eps0 = rs.deformer.spec_eps
        try:
            rs.deformer.spec_eps = 0.0
            step()
            torch.cuda.synchronize()
            te = time.perf_counter()
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            exact_ms = (time.perf_counter() - te) / 2 * 1e3
        finally:
            rs.deformer.spec_eps = eps0
        cmp_ = None
        try:
            from tools import spec_search_probe as SP
            from intrinsicavatar_amd import fast_snarf as _fs
            pts = SP.march_points(rs, rays, 1 << 19)
            x0, v0 = SP.search(rs.deformer, pts, None)
            k0 = _fs.filter(x0, v0)
            rs.deformer.spec_eps = 0.0
            s0 = rs.deformer.deform_sdf(pts, rs.geometry)
            rs.deformer.spec_eps = eps0
            r_ = SP.compare(rs.deformer, rs.geometry, pts, eps0, (x0, v0, k0, s0))
            cmp_ = dict(points=int(pts.shape[0]), candidate_set_differs=int(round(r_["set_mismatch"] * pts.shape[0])),
                        distinct_root_lost=int(round(r_["lost_root"] * pts.shape[0])), min_sdf_bits_differ=int(round(r_["sdf_bits_differ"] * pts.shape[0])),
                        min_sdf_max_abs_diff=r_["sdf_max_abs"], completed_items_bit_identical=r_["completed_items_bit_identical"],
                        points_redone_with_the_filter_off=int(r_["redone_points"]), fetches_per_point=round(r_["fetches_per_point"], 2))
            del x0, v0, k0, s0, pts
        except Exception as e:
            cmp_ = dict(error=f"{type(e).__name__}: {e}")
        finally:
            rs.deformer.spec_eps = eps0
        # (iii) the product path's own canary (SNARFDeformer.spec_canary, IA_SPEC_CANARY): one more untimed step in which every
        # 1024th point of EVERY search batch of the step is searched again to the end + K9 and compared with the row the
        # early-filter search left (count, inits, bit-identical roots); must be 0
        canary = None
        try:
            rs.deformer.spec_canary = 1024
            rs.deformer.canary_totals(reset=True)
            step()
            cc = rs.deformer.canary_totals()
            canary = dict(every=1024, points_checked=int(cc[0]), candidate_rows_differ=int(cc[1]), overflow_points_count_only=int(cc[2]))
        except Exception as e:
            canary = dict(error=f"{type(e).__name__}: {e}")
        finally:
            rs.deformer.spec_canary = 0
        search_modes = dict(
            canary_on_one_step=canary,
            timed="K9-consistent early filter (csrc/snarf.hip: retire inside the eps-box of a tight later root, same voxel cell, only where the TRUE "
                  "skinning Jacobian is tight all over that cell (ia_cell_tightness); redo a point with the filter off when a completed root is "
                  "1e-4 .. 2e-4 from a recorded one)",
            cell_tightness_table=rs.deformer.cell_tight is not None,
            eps=eps0, ms_per_step=round(ms_per_step, 3),
            search_to_the_end=dict(ms_per_step=round(exact_ms, 3), rays_per_s=round(n_rays / (exact_ms * 1e-3), 1),
                                   note="IA_BROYDEN_SPEC_EPS=0: all 13 searches of every point to their end + K9 pass; 2 steps after 1 warm-up, same process"),
            vs_search_to_the_end_on_this_frame=cmp_)

    # ---- secondary key of the headline line: round 1's configs[1] step (radiance + SDF, no PBR branch), same frame
    config2 = None
    if headline and rank == 0 and world == 1 and not args.no_config2:
        opt2, sched2 = optim.reference_optimizer(rs)

        def s2():
            for p in params2:
                p.grad = None
            o = rs.forward_backward(rays, target_rgb, target_mask)
            opt2.step()
            sched2.step()
            return o
        for _ in range(2):
            s2()
        torch.cuda.synchronize()
        tc = time.perf_counter()
        for _ in range(10):
            s2()
        torch.cuda.synchronize()
        config2 = (time.perf_counter() - tc) / 10 * 1e3

    # ---- third key: BASELINE configs[3] shape = the reference's OWN training batch (configs/sampler/edge.yaml:2: 4096 rays per step
    # and GPU; PBR branch, render_mode=uniform_light, spp 512, fwd + bwd + Adam), rays drawn on the subject of the same frame
    config4 = None
    if headline and not args.no_config4:
        # EVERY rank: at N > 1 the batches are sharded over the ranks and the gradients meet in the all-reduce (the regime where the
        # 101 MB exchange is a visible share of a step, unlike the 300 ms headline step)
        if sync is not None:
            sync.remove()                   # the headline step's gradient hooks: config 4 registers its own on the same parameters
        config4 = measure_config4(rs, rays, mat, sg, dev, bg, rank=rank, world=world, rccl_at_one_rank=force_rccl)

    if rank == 0:
        stats = dict(out["stats"])
        stats["n_rays"] = n_rays
        per_call = {k: (len(v), sum(c[0] for c in v)) for k, v in detail.items()}
        total_ms = sum(v[1] for v in per_call.values())
        roofline = l1 = mfma = hash_gather = None
        breakdown = {}
        if per_call:
            dname, (dcalls, dms) = max(per_call.items(), key=lambda kv: kv[1][1])
            bro = None
            if dname in ("ia_fuse_broyden", "ia_fuse_broyden_spec", "ia_fuse_broyden_spec_rows"):
                # one extra untimed step; with N > 1 a LOCAL one on rank 0 (all chunks under no_sync, no finish(), no optimiser step:
                # no collective is issued, the other ranks are not involved), so the roofline is the same object at every N
                def local_step():
                    if sync is None:
                        return step()
                    with sync.no_sync():
                        return step_headline(None) if headline else step_config2(None)
                bro = count_broyden_fetches(local_step, dev, rs.deformer)[dname]
                zero_grads()
            ab = algorithmic_bytes(dname, detail[dname], extra=(bro[1] * k_instr if bro else None))
            stats["deform_points"] = sum(u for k in ("ia_fuse_broyden", "ia_fuse_broyden_spec", "ia_fuse_broyden_spec_rows") for _, u, _ in detail.get(k, [])) // k_instr      # counted, not estimated
            stats["hash_points"] = sum(u for k in ("ia_hashgrid_fwd", "ia_hashgrid_fwd_xcd") for _, u, _ in detail.get(k, [])) // k_instr
            if ab:
                avg_us = dms / dcalls * 1e3
                achieved = ab / (dms * 1e-3) / 1e9
                traffic, tsrc = pmc_traffic(dname)
                roofline = dict(kernel=dname, bound="hbm", achieved=round(achieved, 1), peak=HBM_PEAK_GBPS, unit="GB/s",
                                frac=round(achieved / HBM_PEAK_GBPS, 4), traffic=traffic, traffic_source=tsrc,
                                avg_launch_us=round(avg_us, 1), launches_per_step=dcalls / k_instr,
                                units_per_step=sum(u for _, u, _ in detail[dname]) // k_instr,
                                algorithmic_bytes_per_launch=int(ab / dcalls),
                                share_of_kernel_time=round(dms / max(total_ms, 1e-9), 3))
            if bro:
                # SURVEY 8(d) classes this stage "cache-gather latency": voxel_J is 25 MB, the gathers are served by L1 / L2 /
                # Infinity Cache, so the counted gather bytes can exceed what HBM could ever deliver -- an HBM ratio of them is
                # not a fraction of anything.  The bound that binds is the CU's vector-memory (L1 / texture-addresser) path:
                # `frac` = counted corner loads x 48 B / time against 256 CUs x 64 B/clk.  What HBM itself sees sits next to it:
                # `hbm_side` (PMC bytes of the committed rocprofv3 passes / the live launch time) and the compulsory bytes
                # (targets in, x / valid out, the grid once per launch).
                c = bro
                sec = dms / k_instr * 1e-3
                items = max(c[2] + c[3] + c[4] + c[5] + c[6], 1)
                comp = sum(u * (12 + (48 if ex.get("rows") else ex["I"] * 13) + ex["I"] * ((36 if ex["J_inv"] else 0) + (36 if ex["fwd_J"] else 0)))
                           + VOXEL_J_BYTES for _, u, ex in detail[dname]) / k_instr
                l1_bytes = c[1] * 48.0
                l1_rate = l1_bytes / sec / 1e9
                hbm_side = None
                if roofline["traffic"]:
                    hs = roofline["traffic"] * (dcalls / k_instr) / sec / 1e9
                    hbm_side = dict(GBps=round(hs, 1), frac_of_hbm_peak=round(hs / HBM_PEAK_GBPS, 4), source=roofline["traffic_source"])
                roofline.update(
                    bound="l1 (cache-gather, SURVEY 8(d)): vector-memory path, 256 CUs x 64 B/clk x 2.4 GHz",
                    achieved=round(l1_rate, 1), peak=round(L1_PEAK_GBPS, 1), frac=round(l1_rate / L1_PEAK_GBPS, 4),
                    launch_durations=("live HIP events of one extra step with the secondary march on ONE stream (the kernel has the device to itself); "
                                      f"the timed region runs it on {timed_streams}"),
                    algorithmic_model="COUNTED trilinear fetches (the search kernel's own counters / ia_broyden_stats, one extra untimed step) x in-range corners x 48 B "
                                      "through the L1 path / live HIP-event time of the entry point",
                    fetches_per_step=int(c[0]), corner_loads_per_step=int(c[1]), bytes_per_corner=48,
                    fetches_per_item=round(c[0] / items, 3), Gfetch_per_s=round(c[0] / sec / 1e9, 2),
                    items=(dict(converged=int(c[2]), diverged=int(c[3]), exhausted=int(c[4])) if dname == "ia_fuse_broyden" else
                           dict(retired_by_the_early_filter=int(c[5]), completed_valid=int(c[6]), other=int(c[4]))),
                    speculative_early_filter=(None if dname == "ia_fuse_broyden" else dict(
                        eps=rs.deformer.spec_eps, exact_search_fetches_per_step=int(c[7]),
                        fetches_removed=round(1.0 - c[0] / max(c[7], 1), 4))),
                    hbm_side=hbm_side,
                    survey_8d_gather_bytes_GBps=round(achieved, 1),
                    compulsory_hbm_GBps=round(comp / sec / 1e9, 1), compulsory_hbm_frac=round(comp / sec / 1e9 / HBM_PEAK_GBPS, 4),
                    note="a 16-byte wave gather occupies the vector-memory path for >= 16 clk (64 lanes x 16 B at 64 B/clk); frac = 16 clk / the "
                         "average clk per issued gather, idle lanes included (DESIGN 4.5)")
                l1 = None
            # north star's second target (MFMA utilisation of the batched MLP evaluation), in-step: the SDF head of the no-grad
            # queries (35 -> 64 -> 1: 2 x (35 x 64 + 64) useful FLOP per point) over its live HIP-event time
            for hname, flop_pt in (("ia_sdf_levels_fwd", 2.0 * (35 * 64 + 64)),):
                if hname in detail:
                    hms = sum(c_[0] for c_ in detail[hname])
                    hpts = sum(c_[1] for c_ in detail[hname])
                    tf = hpts * flop_pt / (hms * 1e-3) / 1e12
                    mfma = dict(kernel=hname, bound="mfma", achieved=round(tf, 2), peak=MFMA_F32_PEAK_TFLOPS, unit="TFLOP/s",
                                frac=round(tf / MFMA_F32_PEAK_TFLOPS, 4), points_per_step=hpts // k_instr, ms_per_step=round(hms / k_instr, 3),
                                useful_flop_per_point=flop_pt, note="fp32 MFMA (v_mfma_f32_32x32x2_f32) dense peak; in-step, live HIP events")
            # north star: "rocprof HBM GB/s (traversal, hash lookup)".  The XCD-partitioned gather (one table at a time, each XCD's L2 holds
            # the 4 MB table it gathers from) is priced where it is bound: the L2 -> L1 line traffic of its gathers against the L2 peak,
            # with what HBM itself sees next to it (live HIP-event time of the entry point; counters from the committed PMC passes)
            hname = "ia_hashgrid_fwd_xcd"
            if hname in detail:
                hms = sum(c_[0] for c_ in detail[hname]) / k_instr
                hpts = sum(c_[1] for c_ in detail[hname]) // k_instr
                pk, psrc = pmc_kernel("hash_fwd_xcd_kernel<false>")
                hash_gather = dict(kernel=hname, points_per_step=hpts, ms_per_step=round(hms, 3), launches_per_call=12,
                                   gathered_bytes_per_point=1024, survey_8d_bytes_per_point=12 + 1024 + 128,
                                   gathered_GBps=round(hpts * 1024 / (hms * 1e-3) / 1e9, 1), compulsory_hbm_bytes_per_point=12 + 128,
                                   compulsory_hbm_GBps=round(hpts * 140 / (hms * 1e-3) / 1e9, 1))
                if pk:
                    lpc = sum(1 for _ in detail[hname]) / k_instr * 12          # kernel launches per step (12 per call: 11 hashed levels + the dense set)
                    hbm = pk["hbm_side_bytes_per_launch"] * lpc
                    hash_gather.update(counters_source=psrc, l2_hit_rate=pk.get("l2_hit_rate"), l1_hit_rate=pk.get("l1_hit_rate"),
                                       hbm_side_GBps=round(hbm / (hms * 1e-3) / 1e9, 1), hbm_side_frac_of_peak=round(hbm / (hms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                                       hbm_side_over_compulsory=round(hbm / max(hpts * 140.0, 1.0), 2))
                    if pk.get("l2_to_l1_bytes_per_launch"):
                        l2l1 = pk["l2_to_l1_bytes_per_launch"] * lpc / (hms * 1e-3) / 1e9
                        hash_gather.update(bound="l2 -> l1 line traffic of the gathers (128 B per request) against the L2 peak, MI355X_MICROARCH.md",
                                           achieved=round(l2l1, 1), peak=L2_PEAK_GBPS, unit="GB/s", frac=round(l2l1 / L2_PEAK_GBPS, 4))
            if mfma is not None:
                pk, psrc = pmc_kernel("sdf_head_pipelined2_kernel")
                if pk and pk.get("clock_GHz"):
                    clk = pk["clock_GHz"]
                    mfma.update(clock_GHz_under_kernel=clk, clock_source=psrc + " (GRBM_GUI_ACTIVE / dispatch duration)",
                                peak_at_measured_clock=round(MFMA_F32_PEAK_TFLOPS * clk / 2.4, 1),
                                frac_at_measured_clock=round(mfma["achieved"] / (MFMA_F32_PEAK_TFLOPS * clk / 2.4), 4))
            breakdown = {k: dict(calls_per_step=v[0] / k_instr, ms_per_step=round(v[1] / k_instr, 3))
                         for k, v in sorted(per_call.items(), key=lambda kv: -kv[1][1])[:12]}
        wl = (f"{args.hw}x{args.hw} frame ({n_rays} rays), fwd+bwd+Adam WITH the PBR branch: 128 samples/ray primary march, 2x importance "
              f"resampling, fast-SNARF deformer (13 inits), SDF/radiance/material fields, samples_per_pixel={args.spp} volume-interaction "
              f"re-samples per ray, render_mode=light (one light-importance-sampled secondary ray per foreground re-sample, training form), "
              f"secondary march 64 steps + zero-crossing resampling + shading (global_illumination on), SG environment light; "
              f"{len(chunks)} ray chunk(s) of <= {args.ray_chunk} rays (gradient accumulation across chunks), secondary rays in chunks of 16 Mi; "
              f"random-init fields, synthetic 24-bone rig posed by frame '{args.pose}' of the reference's pose files (plain FK)") if headline else \
             (f"{args.hw}x{args.hw} frame ({n_rays} rays), 128 samples/ray, radiance + SDF geometry, fast-SNARF deformer (13 inits), "
              f"2x importance resampling, random-init hash-grid/MLP fields, synthetic 24-bone rig posed by frame '{args.pose}' (plain FK)")
        base_metric = None
        try:
            base_metric = json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
        except (OSError, KeyError, ValueError):
            pass
        # BASELINE.json's own metric string when the run IS that configuration (540x540, 1024 spp, fwd+bwd with the PBR branch)
        metric = ((base_metric if (base_metric and args.hw == 540 and args.spp == 1024) else
                   f"rays/sec (fwd+bwd) at {args.hw}x{args.hw}, {args.spp} spp") if headline else
                  (f"rays/sec (fwd+bwd) at {args.hw}x{args.hw}, no PBR branch (configs[1])" if args.mode == "fwd+bwd" else f"rays/sec (fwd) at {args.hw}x{args.hw}"))
        line = {
            "metric": metric, "value": round(value, 1), "unit": "rays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl, "pass": "fwd+bwd" if headline else args.mode, "spp": args.spp if headline else 0,
                       "render_mode": "light" if headline else None, "global_illumination": bool(headline), "pose": args.pose,
                       "ray_chunk": args.ray_chunk if headline else n_rays,
                       "host_numa_node": numa_node, "host_cpus_busy": round(cpu_busy, 2), "blocking_sync": bool(blocking),
                       "host_cpu_throttled_ms_in_timed_region": (None if not (thr0 and thr1) else round((thr1[1] - thr0[1]) / 1e3, 1)),
                       "optimizer_step_in_timed_region": bool(opt is not None), "frames_per_step_per_gpu": 1,
                       "peak_device_memory_GiB": round(peak_alloc_timed / 2 ** 30, 1),
                       "peak_reserved_memory_GiB": round(peak_reserved_timed / 2 ** 30, 1),
                       "parallelism": f"frame/ray-batch sharding x{world}", "samples": stats,
                       "secondary_march_streams": timed_streams, "secondary_march_streams_taken": streams_taken,
                       "gradient_allreduce": (None if sync is None else dict(backend=dist.get_backend(), world=dist.get_world_size(),
                                                                            bytes_per_step=int(reduced_bytes[0]))),
                       "library_sources_sha256_16": _build.source_fingerprint(), "library_built_from": _build.built_fingerprint()},
            "roofline": roofline, "mfma": mfma, "hash_gather": hash_gather, "cpu_baseline": cpu, "parity_on_bench_frame": parity, "deformer_search": search_modes,
            "kernel_breakdown_ms_per_step": breakdown,
            "ms_per_step_instrumented": round(dt_instr / max(k_instr, 1) * 1e3, 3),
            "abi_kernel_ms_per_step": round(total_ms / max(k_instr, 1), 3),
            "secondary_rays_per_s": (round(world * stats.get("n_secondary", 0) * args.steps / dt, 1) if headline else None),
            "config4": config4,
            "config2_ms_per_step": (round(config2, 3) if config2 else None),
            "config2_rays_per_s": (round(n_rays / (config2 * 1e-3), 1) if config2 else None),
        }
        _flush_c_stdout()
        print(json.dumps(line), flush=True)          # the ONE line of the contract, and the last thing on stdout
    if world > 1 or force_rccl:
        dist.destroy_process_group()
    _flush_c_stdout()


def _flush_c_stdout():
    """libraries that write to C stdout (RCCL prints a version banner when its first communicator is made) sit in a block buffer when
    stdout is a pipe and would come out at exit, AFTER the JSON line a driver reads last: flush them where they happen."""
    import ctypes
    try:
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
    except Exception:       # noqa: BLE001 -- best effort
        pass


def self_launch_ranks(n_gpus, script=None):
    """`python bench.py --gpus N` without a launcher (WORLD_SIZE unset, N > 1): re-exec the same command line under
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N` on 127.0.0.1 (one rank per GPU, RCCL) and exit with its
    status.  Refuses only when fewer than N devices are visible (IA_BENCH_SHARE_GPU=1, the 1-GPU test hook, lifts that).
    The reference's analogue is launch.py:83-98 (one process, Lightning spawns the DDP ranks)."""
    if n_gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < n_gpus and os.environ.get("IA_BENCH_SHARE_GPU") != "1":
        sys.exit(f"--gpus {n_gpus}: only {have} GPU(s) visible")
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), script or os.path.abspath(sys.argv[0])] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    sys.exit(subprocess.call(cmd, env=env))


def count_broyden_fetches(step, dev, dfm):
    """one extra (untimed) step that counts what the searches cost.  Exact searches (ia_fuse_broyden): every call is followed
    by ia_broyden_stats on the same inputs.  Speculative searches (ia_fuse_broyden_spec): the kernel's own counters, plus
    ia_broyden_stats on the same inputs for what the exact search WOULD have fetched.
    -> {entry point: [fetches, in-range corner loads, converged, diverged, exhausted | other, retired, completed valid, exact fetches]}"""
    from intrinsicavatar_amd import fast_snarf, _lib as L
    cnt = torch.zeros(17, dtype=torch.int64, device=dev)        # exact entry point
    cnt_x = torch.zeros(17, dtype=torch.int64, device=dev)      # what the exact search would cost on the speculative calls' inputs
    spec = torch.zeros(5, dtype=torch.int64, device=dev)
    n_spec_items = [0]
    orig, orig_spec, orig_rows = fast_snarf.fuse_broyden, fast_snarf.fuse_broyden_spec, fast_snarf.fuse_broyden_spec_rows

    def stats(xd_tgt, voxel_J, tfs, bone_ids, offset, scale, cvg, dvg, out):
        cl = isinstance(voxel_J, fast_snarf.ChannelLastVoxelJ)
        vj = voxel_J.data if cl else voxel_J.contiguous()
        D, H, W = (vj.shape[1:4] if cl else vj.shape[2:5])
        B, N, _ = xd_tgt.shape
        L.check(L.lib().ia_broyden_stats(L.i32(B), L.i64(N), L.i32(bone_ids.shape[0]), L.ptr(xd_tgt.contiguous().float()), L.ptr(vj),
                                         L.i32(1 if cl else 0), L.i32(D), L.i32(H), L.i32(W), L.ptr(tfs.contiguous().float()),
                                         L.ptr(bone_ids.contiguous().to(torch.int32)), L.ptr(offset.reshape(3).contiguous().float()),
                                         L.ptr(scale.reshape(3).contiguous().float()), L.f32(cvg), L.f32(dvg), L.ptr(out), L.stream()),
                "ia_broyden_stats")

    def wrapped(x, xd_tgt, voxel, voxel_J, tfs, bone_ids, align_corners, J_inv, is_valid, offset, scale, cvg, dvg, fwd_J=None):
        orig(x, xd_tgt, voxel, voxel_J, tfs, bone_ids, align_corners, J_inv, is_valid, offset, scale, cvg, dvg, fwd_J=fwd_J)
        stats(xd_tgt, voxel_J, tfs, bone_ids, offset, scale, cvg, dvg, cnt)

    def wrapped_spec(x, xd_tgt, voxel_J, tfs, bone_ids, J_inv, is_valid, offset, scale, cvg, dvg, eps, fwd_J=None, counters=None, cell_tight=None):
        orig_spec(x, xd_tgt, voxel_J, tfs, bone_ids, J_inv, is_valid, offset, scale, cvg, dvg, eps, fwd_J=fwd_J, counters=spec, cell_tight=cell_tight)
        stats(xd_tgt, voxel_J, tfs, bone_ids, offset, scale, cvg, dvg, cnt_x)
        n_spec_items[0] += xd_tgt.shape[1] * bone_ids.shape[0]
    def wrapped_rows(x, xd_tgt, voxel_J, tfs, bone_ids, J_inv, cnt_, meta, start, oh, osc, tot, offset, scale, cvg, dvg, eps, fwd_J=None, counters=None,
                     order=None, n_points=None, cell_tight=None):
        orig_rows(x, xd_tgt, voxel_J, tfs, bone_ids, J_inv, cnt_, meta, start, oh, osc, tot, offset, scale, cvg, dvg, eps, fwd_J=fwd_J, counters=spec,
                  order=order, cell_tight=cell_tight)
        stats(xd_tgt, voxel_J, tfs, bone_ids, offset, scale, cvg, dvg, cnt_x)
        n_spec_items[0] += xd_tgt.shape[1] * bone_ids.shape[0]
    fast_snarf.fuse_broyden, fast_snarf.fuse_broyden_spec, fast_snarf.fuse_broyden_spec_rows = wrapped, wrapped_spec, wrapped_rows
    try:
        step()
        torch.cuda.synchronize()
    finally:
        fast_snarf.fuse_broyden, fast_snarf.fuse_broyden_spec, fast_snarf.fuse_broyden_spec_rows = orig, orig_spec, orig_rows
    c, cx, sp = cnt.cpu().tolist(), cnt_x.cpu().tolist(), spec.cpu().tolist()
    other = max(n_spec_items[0] - sp[1] - sp[2], 0)            # searches that ended by themselves without a valid root
    spec_row = [sp[0], sp[4], 0, 0, other, sp[1], sp[2], cx[0]]
    return {"ia_fuse_broyden": c[:5] + [0, 0, c[0]], "ia_fuse_broyden_spec": spec_row, "ia_fuse_broyden_spec_rows": spec_row}


def usable_cores():
    """host cores this process may actually use: the affinity mask, capped by the cgroup's CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def parity_on_bench_frame(gpu_model, sample, edges, refs, spp, dev):
    """the CPU oracle's renderings of the cpu_baseline sample (one .npz per worker) against the GPU path on the SAME rays with
    the SAME random numbers (a worker draws light_u [spp,3] then shuffle_u [n,spp] from default_rng(first ray index),
    oracle/render_ref.py relight_step): per map max / p99 / mean absolute difference, and the discrete flips counted."""
    from intrinsicavatar_amd import pbr
    rs, mat, env_img = gpu_model
    keys = ("comp_rgb", "comp_normal", "opacity", "depth", "albedo", "roughness", "metallic", "comp_rgb_phys")
    errs = {k: [] for k in keys}
    flips = dict(rays=0, sample_count=0, has_samples=0, resampled_layout=0, n_fg_ref=0, n_fg_gpu=0)
    emitter = None
    if spp:
        emitter = pbr.EnvironmentLightTensor(env_img)
        emitter.update_pdf()
    white = torch.ones(3, device=dev)
    with torch.no_grad():
        for k, ref in enumerate(refs):
            a, b = edges[k], edges[k + 1]
            if b <= a:
                continue
            r = torch.from_numpy(sample[a:b]).to(dev)
            if spp:
                rng = np.random.default_rng(a)
                light_u = torch.from_numpy(rng.random((spp, 3), dtype=np.float32)).to(dev)
                shuffle_u = torch.from_numpy(rng.random((b - a, spp), dtype=np.float32)).to(dev)
                out = rs.relight(r, mat, emitter, spp, light_u, shuffle_u, background_color=white, global_illumination=True)
            else:
                out = rs.forward(r)
            for key in keys:
                if key in ref and key in out:
                    errs[key].append(np.abs(out[key].detach().cpu().numpy().reshape(b - a, -1) - ref[key].reshape(b - a, -1)).max(-1))
            cg, cr = out["packed_info"][:, 1].cpu().numpy(), ref["packed_info"][:, 1]
            flips["rays"] += b - a
            flips["sample_count"] += int((cg != cr).sum())
            flips["has_samples"] += int(((cg > 0) != (cr > 0)).sum())
            if "resampled_packed_info" in ref and "resampled_packed_info" in out:
                flips["resampled_layout"] += int((out["resampled_packed_info"].cpu().numpy() != ref["resampled_packed_info"]).any(-1).sum())
                flips["n_fg_ref"] += int(ref["n_fg"])
                flips["n_fg_gpu"] += int(out["stats"]["n_fg"])
    maps = {}
    for key, v in errs.items():
        if v:
            e = np.concatenate(v)
            maps[key] = dict(max=float(e.max()), p99=float(np.quantile(e, 0.99)), mean=float(e.mean()))
    return dict(rays=flips["rays"], maps_abs_err=maps,
                flips=dict(rays_with_other_sample_count=flips["sample_count"], rays_hit_vs_miss=flips["has_samples"],
                           rays_with_other_resample_layout=flips["resampled_layout"],
                           foreground_resamples=(flips["n_fg_gpu"], flips["n_fg_ref"])),
                against="oracle/render_ref.py (relight_step eval form / render_step) on the cpu_baseline sample of the timed frame, same rays and "
                        "random numbers, parameters as initialised (before the first optimiser step); comp_rgb_phys is a Monte-Carlo estimate: a "
                        "visibility sample that falls on the other side of a threshold moves a pixel by Lo / spp")


def cpu_baseline(rays, export, n_rays, headline, spp, phys=None, gpu_model=None):
    """the CPU oracle (oracle/: a port of the reference's algorithm, test infrastructure) timed on this box's host cores on a
    bounded sample of the same frame: one single-threaded worker process per core (oracle/cpu_worker.py), each on its share
    of the sample; rate = sample rays / slowest worker's compute time.  Forward only: the oracle has no backward."""
    import subprocess
    import tempfile
    from oracle import oracle as O, cpu_worker as CW
    O.build()
    cores = min(usable_cores(), int(os.environ.get("IA_CPU_BASELINE_CORES", "32")))
    per = 150 if headline else 6000                       # rays per worker: ~4 s of work each
    n_sample = min(per * cores, n_rays)
    stride = max(1, n_rays // n_sample)
    sample = rays[::stride][:n_sample].cpu().numpy()
    with tempfile.TemporaryDirectory() as td:
        CW.save_scene(f"{td}/scene.npz", dict(export, **(phys or {})))
        np.save(f"{td}/rays.npy", sample)
        edges = [len(sample) * k // cores for k in range(cores + 1)]
        env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), OMP_NUM_THREADS="1")
        tc = time.perf_counter()
        procs = [subprocess.Popen([sys.executable, "-m", "oracle.cpu_worker", f"{td}/scene.npz", f"{td}/rays.npy", str(edges[k]),
                                   str(edges[k + 1]), str(spp if headline else 0), f"{td}/out{k}.npz"], cwd=ROOT, env=env,
                                  stdout=subprocess.PIPE, stderr=subprocess.DEVNULL) for k in range(cores)]
        times = []
        for p_ in procs:
            out, _ = p_.communicate(timeout=600)
            if p_.returncode != 0:
                return dict(value=None, unit="rays/s", cores=cores, kind="port", sample="worker failed"), None
            times.append(float(out.decode().strip().splitlines()[-1]))
        wall = time.perf_counter() - tc
        # the rays the oracle just rendered, rendered by the GPU path with the same random numbers: parity AT the bench frame
        parity = None
        if gpu_model is not None:
            try:
                parity = parity_on_bench_frame(gpu_model, sample, edges, [dict(np.load(f"{td}/out{k}.npz")) for k in range(cores)],
                                               spp if headline else 0, rays.device)
            except Exception as e:              # a diagnostic must not take the measurement down
                parity = dict(error=f"{type(e).__name__}: {e}")
    tcpu = max(times)
    form = (f"relight_step (render_step FORWARD with the PBR branch at {spp} spp: render_mode=light in its eval form -- shared light "
            f"directions shuffled per ray --, secondary rays + indirect shading on)") if headline else \
           "render_step forward WITHOUT the PBR branch (configs[1] form)"
    return dict(value=round(len(sample) / tcpu, 1), unit="rays/s", cores=cores, kind="port", passes="forward only (the oracle has no backward)",
                sample=f"every {stride}th ray of the same 540x540 frame ({len(sample)} rays), oracle/render_ref.py {form}; {cores} "
                       f"single-threaded worker processes, slowest {tcpu:.1f} s, mean {sum(times) / len(times):.1f} s, {wall:.1f} s incl. start-up",
                single_core_rays_per_s=round(len(sample) / sum(times), 2)), parity


if __name__ == "__main__":
    main()
