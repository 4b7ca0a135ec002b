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


def test_relight_bench_starts_its_own_ranks():
    line = _run_line(["tools/relight_bench.py", "--gpus", "2", "--frames", "2", "--hw", "48", "--spp", "16"])
    assert line["n_gpus"] == 2 and line["frames"] == 2 and line["secondary_rays"] > 0
