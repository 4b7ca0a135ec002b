#!/usr/bin/env python3
"""Do the deformer search (texture-path bound) and the SDF network's hash gathers (L2-request bound) overlap when two
independent point batches run on two HIP streams (one host thread each)?  -> ms for two batches in sequence on one stream
against the same two batches on two streams."""
import json, os, sys, threading, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from intrinsicavatar_amd import build; build.build()
from intrinsicavatar_amd import synthetic as S
dev = "cuda:0"
n = int(os.environ.get("IA_N", 24_000_000))
rs, rays, _ = S.build_frame(dev, 540, 540, pose_seed=0, beta=0.01, num_samples_per_ray=128)
lo, hi = rs.aabbs[0, :3], rs.aabbs[0, 3:]
g = torch.Generator(device=dev).manual_seed(0)
from intrinsicavatar_amd.render import ray_points
ro, rd, far, ts, te, ri, pinfo, st = rs.sample(rays)
base = ray_points(ro, rd, ri, ts, te)                      # the primary samples: points near the body, as the secondary march sees
rep = (n + base.shape[0] - 1) // base.shape[0]
batches = [(base.repeat(rep, 1)[:n] + 0.03 * torch.randn(n, 3, device=dev, generator=g)).contiguous() for _ in range(2)]
def work(b):
    return rs._sdf_at(b)
for b in batches: work(b)
torch.cuda.synchronize()
t0 = time.perf_counter(); ref = [work(b) for b in batches]; torch.cuda.synchronize(); seq = time.perf_counter() - t0
def masked_stream(words):
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    st = C.c_void_p()
    arr = (C.c_uint32 * len(words))(*words)
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), C.c_uint32(len(words)), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value)
mode = os.environ.get("IA_MASK", "none")
if mode == "none":
    streams = [torch.cuda.Stream() for _ in batches]
else:
    # groups of 8 consecutive mask bits alternate between the two streams: half of every XCD's CUs each, whichever way the
    # driver interleaves the bits over XCDs / shader engines
    a = [0x00FF00FF] * 8 if mode == "half" else [0x3F3F3F3F] * 8          # "half": 128 / 128 CUs, otherwise 192 / 64
    b = [w ^ 0xFFFFFFFF for w in a]
    streams = [masked_stream(a), masked_stream(b)]
out = [None, None]
def run(i):
    with torch.cuda.stream(streams[i]):
        out[i] = work(batches[i])
    streams[i].synchronize()
def both():
    th = [threading.Thread(target=run, args=(i,)) for i in range(2)]
    for t in th: t.start()
    for t in th: t.join()
both(); torch.cuda.synchronize()
t0 = time.perf_counter(); both(); torch.cuda.synchronize(); par = time.perf_counter() - t0
same = all(torch.equal(a, b) for a, b in zip(ref, out))
from intrinsicavatar_amd import _lib as L
lib = L.lib(); lib.start(); work(batches[0]); per = lib.report()
print(json.dumps(dict(breakdown_ms={k: round(v[1], 2) for k, v in sorted(per.items(), key=lambda kv: -kv[1][1])[:6]}, points_per_batch=n, mask=mode, sequential_ms=round(seq * 1e3, 1), two_streams_ms=round(par * 1e3, 1), identical=same)))
