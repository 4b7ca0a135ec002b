#!/usr/bin/env python3
"""bench.py -- render_step throughput on MI355X (BASELINE.json metric: rays/sec at 540x540).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`)

A "step" is one pass of the render_step hot path over one 540x540 frame (291 600 primary rays) of
synthetic input: BASELINE config 2 (128 samples/ray, radiance + SDF geometry, fast-SNARF deformer,
random-init network of the reference's architecture, synthetic 24-bone rig; inputs resident in HBM).
Multi-GPU: frames / ray batches shard across ranks with replicated parameters (weak scaling: one
frame per rank per step).  Rank 0 prints ONE JSON line.

Extra objects in the line:
  roofline     -- the dominant kernel of the step (by HIP-event time measured live in the timed region)
                  priced against its roofline with SURVEY.md 8(d)'s algorithmic bytes;
  cpu_baseline -- the CPU oracle (oracle/render_ref.py, a port) on a bounded ray sample of the same frame,
                  rank 0 / N == 1 only.
"""
import argparse
import json
import os
import sys
import time

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


def algorithmic_bytes(name, stats):
    """SURVEY.md 8(d) per-unit algorithmic bytes x units of ONE step, per C-ABI entry point."""
    n, E0, S0 = stats["n_rays"], stats["n_edges0"], stats["n_samples0"]
    trav = 48 * n + 16 * S0 + 14 * E0
    P = stats["deform_points"]          # total points pushed through the deformer in one step
    Q = stats["sdf_points"]             # total candidates through the SDF network
    return {
        "ia_traverse_grids_count": trav, "ia_traverse_grids_fill": trav,
        # Broyden: compulsory HBM traffic only -- 12 B point in, 13 inits x 49 B out (x 12, J_inv 36, valid 1) and the
        # 25.2 MB voxel_J grid once per launch; the <= 11 x 8-corner x 48 B gathers per (point, init) of SURVEY 8(d) are
        # served by L2 / Infinity Cache (the grid is resident), they are reported as gather traffic in DESIGN.md
        "ia_fuse_broyden": P * (12 + 13 * 49) + 3 * 25_165_824,
        # hash grid fwd: 12 B in + 16 levels x 8 corners x 8 B gathered + 128 B out (+384 B Jacobian when asked)
        "ia_hashgrid_fwd": (Q + stats["n_samples"]) * (12 + 1024 + 128),
        "ia_hashgrid_bwd": 2 * stats["n_samples"] * (12 + 128 + 1024 + 1024),     # read-modify-write atomics
        "ia_hashgrid_bwd_binned": 2 * stats["n_samples"] * (12 + 128 + 1024 + 1024),
        "ia_mlp_fwd": Q * (35 + 13) * 4 + stats["n_samples"] * (67 + 3) * 4,
    }.get(name)


# C-ABI entry point -> kernels it launches (substring match on the rocprofv3 kernel names); "alt": one of them runs per
# call, "seq": all of them run once per call
PMC_KERNELS = {
    "ia_fuse_broyden": ("alt", ["broyden_persistent_kernel", "broyden_kernel"]),
    "ia_hashgrid_fwd": ("alt", ["hash_fwd_kernel<false>", "hash_fwd_kernel<true>"]),
    "ia_hashgrid_bwd_binned": ("seq", ["hash_bin_kernel", "hash_reduce_kernel"]),
    "ia_mlp_fwd": ("alt", ["mlp_fwd_kernel"]),
    "ia_mlp_bwd_fused": ("alt", ["mlp2_train_kernel"]),
    "ia_sdf_mlp_bwd_fused": ("alt", ["sdf_train_kernel"]),
}


def pmc_traffic(entry):
    """HBM-side bytes per launch of `entry` from the committed rocprofv3 PMC passes (profiles/r01_pmc_traffic.json:
    2 x FETCH_SIZE + WRITE_SIZE, per MI355X_MICROARCH.md), or None when the entry point has not been profiled."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if entry not in PMC_KERNELS or not os.path.exists(path):
        return None
    mode, subs = PMC_KERNELS[entry]
    ks = json.load(open(path))["kernels"]
    hits = [v for k, v in ks.items() if any(sub in k for sub in subs)]
    if not hits:
        return None
    tot = sum(v["hbm_side_bytes_per_launch"] * v["launches"] for v in hits)
    calls = sum(v["launches"] for v in hits) if mode == "alt" else max(v["launches"] for v in hits)
    return int(tot / max(calls, 1))


def cgroup_throttle():
    """(nr_throttled, throttled_usec) of this container's CPU controller, or None (diagnostic: a throttled launch thread
    makes the step host-bound)."""
    try:
        kv = dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat").read().strip().splitlines())
        return int(kv.get("nr_throttled", 0)), int(kv.get("throttled_usec", 0))
    except (OSError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--hw", type=int, default=540)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-optimizer", action="store_true", help="time fwd+bwd only (no Adam step)")
    ap.add_argument("--pass", dest="mode", choices=["fwd+bwd", "fwd"], default="fwd+bwd",
                    help="fwd+bwd = training-step form of render_step (BASELINE metric); fwd = inference form")
    args = ap.parse_args()

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
            _rc = _hip.hipSetDeviceFlags(ctypes.c_uint(0x4))          # hipDeviceScheduleBlockingSync
            blocking = (_rc == 0)
        except OSError:
            blocking = False
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device(dev))

    from intrinsicavatar_amd import parallel as _par
    numa_node = _par.pin_to_gpu_numa_node(local_rank)          # host side of the step next to its GPU (2-socket hosts)
    # Buffer sizes follow the sample counts, which move a little every step once the optimiser updates the geometry; with
    # exact-size caching the allocator keeps growing (13 -> 34 GiB over 14 steps) and a 5 GiB hipMalloc on a freshly booted
    # box costs ~100 ms -- inside the timed region that showed up as 75-110 ms "steps".  Size classes (1/8 power-of-two
    # steps) make the blocks of one step reusable by the next, and one up-front reservation moves the remaining growth in
    # front of the warm-up.  288 GB of HBM: the 24 GiB arena is 8 % of the device.
    torch.cuda.memory._set_allocator_settings("roundup_power2_divisions:8")
    _arena = torch.empty(24 << 30, dtype=torch.uint8, device=dev)
    del _arena
    from intrinsicavatar_amd import build
    if rank == 0:
        build.build()
    if world > 1:
        dist.barrier()
    from intrinsicavatar_amd import synthetic as S, _lib as L, parallel

    # one frame per rank (frame-/ray-batch sharding, replicated parameters).  Weak scaling = the SAME per-GPU workload at
    # every N: each rank renders the configs[1] frame (pose 0) against its own target image, so the per-rank work at N=8
    # is exactly the N=1 work and the gradients that meet in the all-reduce still differ per rank.
    rs, rays, export = S.build_frame(dev, args.hw, args.hw, pose_seed=0, beta=0.01, num_samples_per_ray=128)
    n_rays = rays.shape[0]

    params = rs.parameters()
    g = torch.Generator().manual_seed(1234 + rank)
    target_rgb = torch.rand((n_rays, 3), generator=g).to(dev)
    target_mask = (torch.rand(n_rays, generator=g) > 0.5).float().to(dev)

    # one-time initialisation (not a step, and BEFORE the all-reduce hooks exist): the first launches load the code
    # objects and set kernel attributes; done on a 4096-ray slice so that the W warm-up steps see a warm library
    if args.mode == "fwd":
        rs.forward(rays[:4096].contiguous())
    else:
        rs.forward_backward(rays[:4096].contiguous(), target_rgb[:4096].contiguous(), target_mask[:4096].contiguous())
        for p in params:
            p.grad = None
    torch.cuda.synchronize()

    # the one exchange step of the path: all-reduce(sum) of the gradients over RCCL/xGMI; the two 50 MB
    # hash-table gradients are launched from autograd hooks as soon as they are complete (overlap with the rest of backward)
    sync = parallel.OverlappedGradientAllReduce(params) if (world > 1 and args.mode != "fwd") else None
    # the optimiser step of the iteration is inside the timed region: torch.optim.Adam semantics with the reference's
    # parameter groups and linear warm-up (configs/config.yaml:110-148), one fused launch; the summed all-reduce becomes
    # DDP's mean through grad_scale = 1/world
    opt = sched = None
    if args.mode != "fwd" and not args.no_optimizer:
        from intrinsicavatar_amd import optim
        opt, sched = optim.reference_optimizer(rs, grad_scale=1.0 / world)

    def step():
        if args.mode == "fwd":
            return rs.forward(rays)
        for p in params:
            p.grad = None
        out = rs.forward_backward(rays, target_rgb, target_mask)
        if sync is not None:
            sync.finish()
        if opt is not None:
            opt.step()
            sched.step()
        return out

    for _ in range(args.warmup):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    lib = L.lib()
    # ---- timed region: EXACTLY K steps, barrier + synchronize on both sides, nothing else running
    thr0 = cgroup_throttle()
    cpu0 = time.process_time()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    cpu_busy = (time.process_time() - cpu0) / max(dt, 1e-9)          # CPUs this rank kept busy during the timed region
    thr1 = cgroup_throttle()
    # ---- the same K steps again with a HIP-event pair around every C-ABI launch (on the launch stream): per-kernel
    # durations for the roofline / breakdown.  Kept out of the throughput region because the ~600 event records per step
    # cost host time that the un-instrumented step does not pay (ms_per_step_instrumented is reported next to it).
    lib.start()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    dt_instr = time.perf_counter() - t1
    per_call = lib.report()
    if world > 1:
        t = torch.tensor([dt], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    value = world * n_rays * args.steps / dt

    if rank == 0:
        stats = dict(out["stats"])
        stats["n_rays"] = n_rays
        # points through the deformer per step: edges (it0) + intervals (it1) + final samples
        calls_b, ms_b = per_call.get("ia_fuse_broyden", (0, 0.0))
        stats["deform_points"] = stats["n_edges0"] + 2 * stats["n_samples"]       # ~ (it1 has fewer intervals; upper est.)
        stats["sdf_points"] = stats["deform_points"]                               # ~1 surviving candidate / point
        total_ms = sum(v[1] for v in per_call.values())
        dom = max(per_call.items(), key=lambda kv: kv[1][1])
        dname, (dcalls, dms) = dom
        launches_per_step = dcalls / args.steps
        ab = algorithmic_bytes(dname, stats)
        roofline = None
        if ab:
            per_launch_bytes = ab / launches_per_step
            avg_us = dms / dcalls * 1e3
            achieved = per_launch_bytes / (avg_us * 1e-6) / 1e9
            roofline = dict(kernel=dname, bound="hbm", achieved=round(achieved, 1), peak=HBM_PEAK_GBPS, unit="GB/s",
                            frac=round(achieved / HBM_PEAK_GBPS, 4), traffic=pmc_traffic(dname),
                            traffic_source="profiles/r01_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE per "
                                           "launch, separate passes, tools/pmc_traffic.py)",
                            avg_launch_us=round(avg_us, 1), launches_per_step=launches_per_step,
                            algorithmic_bytes_per_launch=int(per_launch_bytes),
                            share_of_kernel_time=round(dms / max(total_ms, 1e-9), 3))
        breakdown = {k: dict(calls_per_step=v[0] / args.steps, ms_per_step=round(v[1] / args.steps, 3))
                     for k, v in sorted(per_call.items(), key=lambda kv: -kv[1][1])[:10]}
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            from oracle import render_ref as R, oracle as O
            O.build()
            stride = max(1, n_rays // 24000)
            sample = rays[::stride].cpu().numpy()
            sc = R.Scene(**export)
            tc = time.perf_counter()
            R.render_step(sc, sample)
            tcpu = time.perf_counter() - tc
            cpu = dict(value=round(sample.shape[0] / tcpu, 1), unit="rays/s", cores=1, kind="port",
                       sample=f"every {stride}th ray of the same 540x540 frame ({sample.shape[0]} rays), "
                              f"oracle/render_ref.py render_step forward, {tcpu:.1f} s single-threaded")
        line = {
            "metric": "rays/sec (fwd+bwd) at 540x540" if args.mode == "fwd+bwd" else "rays/sec (fwd) at 540x540", "value": round(value, 1), "unit": "rays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.hw}x{args.hw} frame ({n_rays} rays), 128 samples/ray, radiance + SDF geometry, "
                                   "fast-SNARF deformer (13 inits), 2x importance resampling, random-init hash-grid/MLP "
                                   "fields, synthetic 24-bone rig",
                       "pass": args.mode, "host_numa_node": numa_node, "host_cpus_busy": round(cpu_busy, 2), "blocking_sync": bool(blocking),
                       "host_cpu_throttled_ms_in_timed_region": (None if not (thr0 and thr1) else round((thr1[1] - thr0[1]) / 1e3, 1)), "optimizer_step_in_timed_region": bool(opt is not None), "frames_per_step_per_gpu": 1, "parallelism": f"frame/ray-batch sharding x{world}",
                       "samples": stats},
            "roofline": roofline, "cpu_baseline": cpu, "kernel_breakdown_ms_per_step": breakdown,
            "ms_per_step_instrumented": round(dt_instr / args.steps * 1e3, 3),
            "abi_kernel_ms_per_step": round(total_ms / args.steps, 3),
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
