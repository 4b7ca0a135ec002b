"""CPU, world_size 2, gloo: the N > 1 host logic (ray sharding + gradient all-reduce) of the multi-GPU path."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from intrinsicavatar_amd import parallel
    torch.manual_seed(0)                                  # replicated parameters
    params = [torch.nn.Parameter(torch.randn(parallel.BIG + 5)), torch.nn.Parameter(torch.randn(64, 35)),
              torch.nn.Parameter(torch.randn(13)), torch.nn.Parameter(torch.randn(3)), torch.nn.Parameter(torch.tensor(0.3))]
    g = torch.Generator().manual_seed(100 + rank)         # rank-specific "gradients"
    local = []
    for i, p in enumerate(params):
        if i == 3 and rank == 1:
            local.append(torch.zeros_like(p))             # unused parameter on this rank -> contributes zeros
            continue
        p.grad = torch.randn(p.shape, generator=g)
        local.append(p.grad.clone())
    nbytes = parallel.allreduce_gradients(params)
    torch.save(dict(local=local, reduced=[p.grad.clone() for p in params], nbytes=nbytes,
                    shard=parallel.shard_range(291600, rank, world),
                    scal=parallel.allreduce_scalars([float(rank + 1), 10.0], "cpu")), os.path.join(out_dir, f"r{rank}.pt"))
    dist.destroy_process_group()


def test_allreduce_gradients_and_sharding_world2(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(tmp_path / f"r{k}.pt") for k in range(world)]
    for i in range(5):
        expect = r[0]["local"][i] + r[1]["local"][i]
        for k in range(world):
            torch.testing.assert_close(r[k]["reduced"][i], expect)
    assert r[0]["nbytes"] == r[1]["nbytes"] == sum(t.numel() * 4 for t in r[0]["local"])
    (a0, a1), (b0, b1) = r[0]["shard"], r[1]["shard"]
    assert a0 == 0 and a1 == b0 and b1 == 291600 and abs((a1 - a0) - (b1 - b0)) <= 1
    assert r[0]["scal"] == r[1]["scal"] == [3.0, 20.0]


def _worker_overlap(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from intrinsicavatar_amd import parallel
    torch.manual_seed(0)
    big1, big2 = torch.nn.Parameter(torch.randn(parallel.BIG + 3)), torch.nn.Parameter(torch.randn(parallel.BIG))
    small = torch.nn.Parameter(torch.randn(64, 35))
    unused = torch.nn.Parameter(torch.randn(parallel.BIG))       # big parameter that gets no gradient on any rank
    sync = parallel.OverlappedGradientAllReduce([big1, big2, small, unused])
    g = torch.Generator().manual_seed(7 + rank)
    w1, w2, w3 = torch.randn(big1.shape, generator=g), torch.randn(big2.shape, generator=g), torch.randn(small.shape, generator=g)
    for _ in range(2):                                            # two steps: hooks must survive zeroed grads
        for p in (big1, big2, small, unused):
            p.grad = None
        loss = (big1 * w1).sum() + (big2 * w2).sum() + (small * w3).sum()
        loss.backward()                                           # hooks fire here
        nbytes = sync.finish()
    torch.save(dict(local=[w1, w2, w3], reduced=[big1.grad.clone(), big2.grad.clone(), small.grad.clone()],
                    unused=float(unused.grad.abs().sum()), nbytes=nbytes), os.path.join(out_dir, f"o{rank}.pt"))
    dist.destroy_process_group()


def test_overlapped_gradient_allreduce_world2(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker_overlap, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(tmp_path / f"o{k}.pt") for k in range(world)]
    for i in range(3):
        expect = r[0]["local"][i] + r[1]["local"][i]
        for k in range(world):
            torch.testing.assert_close(r[k]["reduced"][i], expect)
    assert r[0]["unused"] == r[1]["unused"] == 0.0


def test_shard_range_covers_everything():
    from intrinsicavatar_amd.parallel import shard_range
    for n in (0, 1, 7, 4096, 291600):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_numa_pinning_helper_degrades_gracefully():
    from intrinsicavatar_amd import parallel
    assert parallel._parse_cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11}
    assert parallel._parse_cpulist("") == set()
    import os
    before = os.sched_getaffinity(0)
    assert parallel.pin_to_gpu_numa_node(0) is None          # no GPU here: nothing changes
    assert os.sched_getaffinity(0) == before
