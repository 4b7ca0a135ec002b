"""GPU (MI355X): the N > 1 training path with two REAL processes (one box has one GPU, so both ranks share cuda:0 and talk
over gloo -- RCCL refuses two ranks on one device; the collective semantics are the same): every rank renders its ray shard
of one frame with RenderStep.forward_backward, gradients meet in OverlappedGradientAllReduce (hash-table all-reduces launched
from autograd hooks), the fused Adam folds DDP's 1/world into the step.  The parameters after the step must equal those of a
single process that took the whole batch."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build():
    from intrinsicavatar_amd import build
    build.build()
    from intrinsicavatar_amd import synthetic as S
    rs, rays, _ = S.build_frame("cuda:0", 40, 40, pose_seed=0, beta=0.05, num_samples_per_ray=64, grid_D=16, grid_H=64, grid_W=64,
                                smooth_iters=5, hash_amp=1e-2)
    g = torch.Generator().manual_seed(5)
    n = rays.shape[0]
    target = torch.rand((n, 3), generator=g).to("cuda:0")
    mask = (torch.rand(n, generator=g) > 0.5).float().to("cuda:0")
    return rs, rays, target, mask


def _step(rs, rays, target, mask, world, sync, chunks=1):
    from intrinsicavatar_amd import optim
    params = rs.parameters()
    opt, _ = optim.reference_optimizer(rs, grad_scale=1.0 / world, warmup_steps=None, milestones=None)
    for p in params:
        p.grad = None
    n = rays.shape[0]
    edges = [n * k // chunks for k in range(chunks + 1)]
    for k in range(chunks):
        a, b = edges[k], edges[k + 1]
        ctx = sync.no_sync() if (sync is not None and k < chunks - 1) else _null()
        with ctx:
            # lambda_eik = 0: the eikonal term is a mean over SAMPLES, whose count differs per shard (the trainer passes the
            # global count, eik_denominator); the ray-mean terms shard exactly when the shards are equal-sized
            rs.forward_backward(rays[a:b].contiguous(), target[a:b].contiguous(), mask[a:b].contiguous(), lambda_eik=0.0,
                                loss_scale=(b - a) / n)
    if sync is not None:
        sync.finish()
    opt.step()
    return [p.detach().clone().cpu() for p in params]


def _null():
    import contextlib
    return contextlib.nullcontext()


def _worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from intrinsicavatar_amd import parallel
    rs, rays, target, mask = _build()
    a, b = parallel.shard_range(rays.shape[0], rank, world)
    sync = parallel.OverlappedGradientAllReduce(rs.parameters())
    after = _step(rs, rays[a:b], target[a:b], mask[a:b], world, sync, chunks=2 if rank == 0 else 1)     # rank 0 also accumulates over 2 chunks
    torch.save(after, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_training_step_equals_single_process(tmp_path):
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(tmp_path / f"r{k}.pt") for k in range(world)]
    rs, rays, target, mask = _build()
    before = [p.detach().clone().cpu() for p in rs.parameters()]
    single = _step(rs, rays, target, mask, 1, None)
    moved = 0
    for i, (p0, p1, ps, pb) in enumerate(zip(r[0], r[1], single, before)):
        assert torch.equal(p0, p1), f"ranks disagree on parameter {i}"                       # replicas stay replicas
        # one Adam step moves every touched entry by ~lr; the two-rank result must be the single-process one (fp32 sums of
        # the same gradients in a different order: hash-table entries agree to a fraction of the step)
        step = (ps - pb).abs().max().item()
        if step > 0:
            moved += 1
            assert (p0 - ps).abs().max().item() <= 0.02 * step + 1e-7, (i, (p0 - ps).abs().max().item(), step)
    assert moved >= 4


def _run_line(cmd, timeout=900):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, IA_BENCH_SHARE_GPU="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable] + cmd, cwd=root, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_starts_its_own_ranks():
    """the driver's command form is plain `python3 bench.py --gpus N --steps K --warmup W` (no launcher, WORLD_SIZE unset):
    bench.py must re-execute itself under torch.distributed.run.  Two ranks share the box's one GPU over gloo
    (IA_BENCH_SHARE_GPU=1) -- the control flow of N > 1, not a measurement."""
    line = _run_line(["bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0", "--hw", "64", "--spp", "16"])
    assert line["n_gpus"] == 2 and line["steps"] == 1 and line["value"] > 0 and line["scaling"] == "weak"
    assert line["config"]["samples"]["n_fg"] > 0 and line["config"]["samples"]["n_secondary"] > 0


def test_bench_line_of_one_rank_carries_the_objects_the_contract_names():
    """`python bench.py` on one GPU (small frame): ONE JSON line with the contract's keys, `roofline` and `cpu_baseline`, and the
    secondary workloads `config4` (the reference's 4096-ray training batch) and `config2_ms_per_step`."""
    line = _run_line(["bench.py", "--steps", "1", "--warmup", "1", "--hw", "96", "--spp", "16", "--no-search-modes"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["vs_baseline"] is None and line["dtype"] == "f32"
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(line["roofline"]) and 0 < line["roofline"]["frac"] <= 1
    assert {"value", "unit", "cores", "kind", "sample"} <= set(line["cpu_baseline"])
    c4 = line["config4"]
    assert c4 is not None and c4["ms_per_step"] > 0 and c4["secondary_rays_per_step"] > 0 and len(c4["search_launches_ms_points"]) >= 3
    assert line["config2_ms_per_step"] > 0


def test_relight_bench_starts_its_own_ranks():
    line = _run_line(["tools/relight_bench.py", "--gpus", "2", "--frames", "2", "--hw", "48", "--spp", "16"])
    assert line["n_gpus"] == 2 and line["frames"] == 2 and line["secondary_rays"] > 0


def _rccl_worker(out_path, port):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    assert dist.get_backend() == "nccl"
    from intrinsicavatar_amd import parallel
    rs, rays, target, mask = _build()
    sync = parallel.OverlappedGradientAllReduce(rs.parameters(), single_rank_too=True)
    assert sync.active and len(sync.order) >= 2                       # the two hash tables go out from the autograd hooks
    launched = []
    orig = dist.all_reduce

    def counting(t, *a, **k):
        launched.append((tuple(t.shape), t.device.type, bool(k.get("async_op", False))))
        return orig(t, *a, **k)
    dist.all_reduce = counting
    try:
        after = _step(rs, rays, target, mask, 1, sync, chunks=2)    # two chunks: the first under no_sync
    finally:
        dist.all_reduce = orig
    assert parallel.allreduce_scalars([3.0, 4.0], "cuda:0") == [3.0, 4.0]
    torch.save(dict(after=after, launched=launched), out_path)
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_allreduce_through_rccl_on_one_rank(tmp_path):
    """RCCL itself (torch.distributed backend "nccl"): init_process_group on the GPU, OverlappedGradientAllReduce's hook-launched
    asynchronous all-reduces of the two 50 MB-class table gradients on DEVICE tensors, no_sync() over ray chunks, finish() with the
    flat bucket of the small tensors -- in a one-rank group (a box has one GPU), where a sum over ranks returns its input: the
    parameters after the step must equal those of the step without any collective."""
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    import torch.multiprocessing as mp
    out = str(tmp_path / "rccl.pt")
    ctx = mp.get_context("spawn")
    p = ctx.Process(target=_rccl_worker, args=(out, _free_port()))
    p.start()
    p.join(600)
    assert p.exitcode == 0, p.exitcode
    r = torch.load(out)
    big = [l for l in r["launched"] if l[2]]
    assert len(big) >= 2 and all(l[1] == "cuda" for l in r["launched"]), r["launched"]
    assert any(not l[2] for l in r["launched"])                         # the flat bucket of the small tensors
    rs, rays, target, mask = _build()
    before = [p_.detach().clone().cpu() for p_ in rs.parameters()]
    single = _step(rs, rays, target, mask, 1, None, chunks=2)
    moved = 0
    for i, (a, b, pb) in enumerate(zip(r["after"], single, before)):
        # same bar as the two-rank test: the hash-table gradient of the dense coarse levels meets in float atomics (last-bit run-to-run
        # differences), so two runs of the SAME step agree to a fraction of the Adam step, not bit for bit
        step = (b - pb).abs().max().item()
        if step > 0:
            moved += 1
            assert (a - b).abs().max().item() <= 0.02 * step + 1e-7, (i, (a - b).abs().max().item(), step)
        else:
            assert torch.equal(a, b), i
    assert moved >= 4


def test_bench_step_through_rccl_on_one_rank():
    """bench.py's own step (headline workload, small frame) with a one-rank RCCL group: IA_BENCH_FORCE_RCCL=1."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, IA_BENCH_FORCE_RCCL="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--steps", "2", "--warmup", "1", "--hw", "64", "--spp", "16", "--ray-chunk", "2048",
                        "--no-cpu-baseline", "--no-config2", "--no-config4", "--no-search-modes"], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert lines[-1].startswith("{"), lines[-3:]          # the JSON line is the LAST thing on stdout (RCCL's banner is flushed before it)
    assert sum(l.startswith("{") for l in lines) == 1
    line = json.loads(lines[-1])
    ar = line["config"]["gradient_allreduce"]
    assert ar["backend"] == "nccl" and ar["world"] == 1 and ar["bytes_per_step"] > 90e6, ar      # two 50.4 MB tables + the small bucket
    assert line["roofline"] is not None and line["value"] > 0


# ----------------------------------------------------------------------------- BASELINE configs[3]: ray-batch sharding of the PBR training step
def _phys_scene():
    from intrinsicavatar_amd import build
    build.build()
    from intrinsicavatar_amd import synthetic as S, fields, pbr
    rs, rays, _ = S.build_frame("cuda:0", 40, 40, pose_seed=0, beta=0.05, num_samples_per_ray=64, grid_D=16, grid_H=64, grid_W=64,
                                smooth_iters=5, hash_amp=1e-2)
    mat = fields.VolumeMaterial(seed=2).to("cuda:0")
    sg = pbr.EnvironmentLightSG(num_SGs=16, base_res=32, seed=4).to("cuda:0")
    g = torch.Generator().manual_seed(9)
    with torch.no_grad():
        hit = torch.nonzero(rs.forward(rays)["opacity"][:, 0] > 0.5)[:, 0]
    sel = hit[torch.randint(0, hit.shape[0], (512,), generator=g).to("cuda:0")]          # 2 x 256 rays on the subject
    batch = rays[sel].contiguous()
    target = torch.rand((512, 3), generator=g).to("cuda:0")
    mask = (torch.rand(512, generator=g) > 0.3).float().to("cuda:0")
    light_u = torch.rand((512, 3), generator=g).to("cuda:0")                              # ONE stratified direction set per step, every rank
    shuffle_u = torch.rand((512, 512), generator=g).to("cuda:0")
    return rs, mat, sg, batch, target, mask, light_u, shuffle_u


def _phys_step(scene, a, b, world, sync, eik_denominator):
    """bench.py's config-4 step (build_config4_step) on rays [a, b) of the batch."""
    from intrinsicavatar_amd import optim, pbr
    rs, mat, sg, batch, target, mask, light_u, shuffle_u = scene
    params = rs.parameters() + [p for p in mat.parameters() if p.requires_grad] + list(sg.parameters())
    opt, _ = optim.reference_optimizer(rs, grad_scale=1.0 / world, material=mat, emitter=sg, warmup_steps=None, milestones=None)
    for p in params:
        p.grad = None
    img = sg.generate_image()
    leaf = img.detach().requires_grad_(True)
    emitter = pbr.EnvironmentLightTensor(leaf.detach())
    emitter.update_pdf()
    o = rs.forward_backward_phys(batch[a:b].contiguous(), target[a:b].contiguous(), mat, emitter, 512, light_u, shuffle_u[a:b].contiguous(),
                                 target_mask=mask[a:b].contiguous(), render_mode="uniform_light", env_base=leaf,
                                 background_color=torch.ones(3, device="cuda:0"), eik_denominator=eik_denominator)
    img.backward(leaf.grad)
    if sync is not None:
        sync.finish()
    grads = [(p.grad.detach() / world).clone().cpu() for p in params]          # what Adam sees (grad_scale 1 / world)
    opt.step()
    return [p.detach().clone().cpu() for p in params], int(o["n_samples"]), grads


def _phys_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from intrinsicavatar_amd import parallel
    scene = _phys_scene()
    rs, mat, sg = scene[0], scene[1], scene[2]
    params = rs.parameters() + [p for p in mat.parameters() if p.requires_grad] + list(sg.parameters())
    sync = parallel.OverlappedGradientAllReduce(params)
    a, b = parallel.shard_range(512, rank, world)
    # the eikonal term is a mean over ALL samples of the global batch (systems/intrinsic_avatar.py:235-239): global count / world here,
    # so that the average of the ranks' losses (grad_scale 1 / world) is the global-batch loss
    den = lambda n: parallel.allreduce_scalars([float(n)], "cuda:0")[0] / world      # noqa: E731
    after, n_s, grads = _phys_step(scene, a, b, world, sync, den)
    torch.save(dict(after=after, n_samples=n_s, grads=grads), os.path.join(out_dir, f"p{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_config4_step_sharded_by_ray_batch_equals_the_single_process_step(tmp_path):
    """BASELINE configs[3] / bench.py's `config4` at N > 1: the PBR training step (uniform_light, spp 512, material head, SG light) with the
    batch sharded by rays over two REAL ranks (gloo, both on the box's one GPU), the eikonal mean normalised with the GLOBAL sample count,
    the gradients summed by OverlappedGradientAllReduce and averaged inside Adam -- against one process that takes the whole batch."""
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    mp.spawn(_phys_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(tmp_path / f"p{k}.pt") for k in range(world)]
    scene = _phys_scene()
    params = scene[0].parameters() + [p for p in scene[1].parameters() if p.requires_grad] + list(scene[2].parameters())
    before = [p.detach().clone().cpu() for p in params]
    single, n_all, g_single = _phys_step(scene, 0, 512, 1, None, None)
    assert r[0]["n_samples"] + r[1]["n_samples"] == n_all                          # sampling is per ray: sharding invariant
    moved = 0
    for i, (p0, p1, g0, gs, ps, pb) in enumerate(zip(r[0]["after"], r[1]["after"], r[0]["grads"], g_single, single, before)):
        assert torch.equal(p0, p1), f"ranks disagree on parameter {i}"             # replicas stay replicas
        # the averaged gradient of the two shards IS the gradient of the whole batch (fp32 sums in another order; the table gradients meet
        # in float atomics): 1e-3 of the tensor's largest entry.  (The parameters after the step are not compared entry by entry: the
        # first Adam step moves an entry by lr * sign(g), so an entry whose gradient is at the noise floor may go either way.)
        scale = float(gs.abs().max())
        if scale > 0:
            moved += 1
            assert float((g0 - gs).abs().max()) <= 1e-3 * scale + 1e-12, (i, float((g0 - gs).abs().max()), scale)
            big = gs.abs() > 1e-2 * scale                                          # entries with a clear gradient took the same step
            assert float(((p0 - ps).abs() * big).max()) <= 0.05 * float((ps - pb).abs().max()) + 1e-7, i
    assert moved >= 10


def test_config4_workload_of_the_bench_with_two_ranks_and_through_rccl_on_one():
    """`bench.py --workload config4`: two ranks sharing the GPU over gloo (the N > 1 control flow: sharded batches, global-count loss
    normalisation, all-reduce timing) and a one-rank RCCL group (the same path through librccl)."""
    import json
    import subprocess
    import sys
    line = _run_line(["bench.py", "--workload", "config4", "--gpus", "2", "--steps", "2", "--hw", "96"])
    c4 = line["config4"]
    assert line["n_gpus"] == 2 and c4["n_gpus"] == 2 and line["value"] == c4["rays_per_s"] > 0
    ar = c4["gradient_allreduce"]
    assert ar["world"] == 2 and ar["bytes_per_step"] > 90e6 and ar["ms_total"] > 0 and "ms_exposed" in ar
    assert c4["sparse_exchange"]["touched_entries_per_table"][0] > 0
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, IA_BENCH_FORCE_RCCL="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, "bench.py", "--workload", "config4", "--gpus", "1", "--steps", "2", "--hw", "96"], cwd=root, env=env,
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    ar = line["config4"]["gradient_allreduce"]
    assert ar["backend"] == "nccl" and ar["world"] == 1 and ar["bytes_per_step"] > 90e6
    assert line["config4"]["sparse_exchange"]["ms_all_gather"] is not None
