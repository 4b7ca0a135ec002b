#!/usr/bin/env python3
"""Asymmetric two-stream pipeline: stage 1 (search, texture-path bound) of batch k+1 on stream A while stage 2 (candidate packing,
hash gathers + SDF head, select; L2-request / MFMA bound) of batch k runs on stream B, one host thread per stage.
-> ms for K batches in sequence on one stream against the pipelined schedule; outputs must be identical."""
import json, os, queue, sys, threading, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from intrinsicavatar_amd import build; build.build()
from intrinsicavatar_amd import synthetic as S, _lib as L
from intrinsicavatar_amd.render import ray_points
dev = "cuda:0"
n, K = int(os.environ.get("IA_N", 24_000_000)), int(os.environ.get("IA_K", 4))
rs, rays, _ = S.build_frame(dev, 540, 540, pose_seed=0, beta=0.01, num_samples_per_ray=128)
g = torch.Generator(device=dev).manual_seed(0)
ro, rd, far, ts, te, ri, pinfo, st = rs.sample(rays)
base = ray_points(ro, rd, ri, ts, te)
rep = (n + base.shape[0] - 1) // base.shape[0]
batches = [(base.repeat(rep, 1)[:n] + 0.03 * torch.randn(n, 3, device=dev, generator=g)).contiguous() for _ in range(K)]
dfm, geo = rs.deformer, rs.geometry
lib = L.lib()

def stage1(pts):
    order = rs._spatial_order(pts)
    ps = torch.empty_like(pts)
    L.check(lib.ia_gather_rows3_i32(L.i64(n), L.ptr(pts), L.ptr(order), L.ptr(ps), L.stream()), "g")
    x, valid, _ = dfm.search(ps)
    return order, x, valid

def stage2(order, x, valid):
    cand_x, _, cnt, start, Q = dfm._pack_candidates(x, valid, with_src=False)
    csdf = geo.sdf_only(cand_x)
    sdf_s = torch.empty(n, device=dev)
    L.check(lib.ia_deform_select_min(L.i64(n), L.ptr(start), L.ptr(cnt), L.ptr(csdf), L.ptr(sdf_s), L.stream()), "s")
    sdf = torch.empty_like(sdf_s)
    L.check(lib.ia_scatter_f32_i32(L.i64(n), L.ptr(sdf_s), L.ptr(order), L.ptr(sdf), L.stream()), "sc")
    return sdf

def sequential():
    return [stage2(*stage1(b)) for b in batches]

for _ in range(2): ref = sequential()
torch.cuda.synchronize()
t0 = time.perf_counter(); ref = sequential(); torch.cuda.synchronize(); seq = time.perf_counter() - t0

sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
def pipelined():
    q = queue.Queue(maxsize=2)
    out = [None] * K
    def a():
        with torch.cuda.stream(sA):
            for k, b in enumerate(batches):
                r = stage1(b)
                ev = torch.cuda.Event(); ev.record(sA)
                q.put((k, r, ev))
        q.put(None)
    def bfun():
        with torch.cuda.stream(sB):
            while True:
                it = q.get()
                if it is None:
                    break
                k, r, ev = it
                sB.wait_event(ev)
                for t in r: t.record_stream(sB)
                out[k] = stage2(*r)
    th = [threading.Thread(target=a), threading.Thread(target=bfun)]
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    return out
pipelined()
t0 = time.perf_counter(); got = pipelined(); par = time.perf_counter() - t0
same = all(torch.equal(a, b) for a, b in zip(ref, got))
print(json.dumps(dict(points_per_batch=n, batches=K, sequential_ms=round(seq * 1e3, 1), pipelined_ms=round(par * 1e3, 1), identical=same,
                      broyden_waves=os.environ.get("IA_BR2_WAVES_NOTE", "5"))))
