"""per-step wall time / device malloc+free counts of the training step with the optimiser in the loop (sample counts
change from step to step, so buffer sizes do too).  usage: python tools/alloc_probe.py [allocator settings string]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

if len(sys.argv) > 1 and sys.argv[1]:
    torch.cuda.memory._set_allocator_settings(sys.argv[1])
from intrinsicavatar_amd import build
build.build()
from intrinsicavatar_amd import synthetic as S, optim

dev = "cuda:0"
rs, rays, _ = S.build_frame(dev, 540, 540, pose_seed=0, beta=0.01, num_samples_per_ray=128)
n = rays.shape[0]
g = torch.Generator().manual_seed(1234)
target = torch.rand((n, 3), generator=g).to(dev)
mask = (torch.rand(n, generator=g) > 0.5).float().to(dev)
params = rs.parameters()
opt, sched = optim.reference_optimizer(rs)
use_opt = os.environ.get("NO_OPT") != "1"
for i in range(14):
    st0 = torch.cuda.memory_stats()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for p in params:
        p.grad = None
    out = rs.forward_backward(rays, target, mask)
    t1 = time.perf_counter()
    if use_opt:
        opt.step()
        sched.step()
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    st1 = torch.cuda.memory_stats()
    print(f"step {i}: {1e3 * (t3 - t0):7.2f} ms (fb host {1e3 * (t1 - t0):6.2f}, opt host {1e3 * (t2 - t1):5.2f})  n_samples {out['n_samples']}  "
          f"dev_alloc +{st1['num_device_alloc'] - st0['num_device_alloc']} dev_free +{st1['num_device_free'] - st0['num_device_free']}  "
          f"reserved {st1['reserved_bytes.all.current'] / 2**30:.2f} GiB  loss {float(out['loss']):.5f}", flush=True)
