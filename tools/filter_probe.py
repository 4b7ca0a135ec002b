#!/usr/bin/env python3
"""ia_deform_filter_compact on a synthetic batch shaped like one launch of the headline step's secondary march (49 M points x 13
candidates, 58 % valid, clustered roots): ms per launch and effective HBM rate."""
import ctypes as C, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from intrinsicavatar_amd import build; build.build()
from intrinsicavatar_amd import _lib as L
dev = "cuda:0"
P, I = int(os.environ.get("IA_P", 49_000_000)), 13
g = torch.Generator(device=dev).manual_seed(0)
x = torch.rand(P, I, 3, device=dev, generator=g)
dup = torch.rand(P, I, device=dev, generator=g) < 0.5
x = torch.where(dup[..., None], x[:, :1] + (torch.rand(P, I, 3, device=dev, generator=g) - 0.5) * 1.6e-4, x).contiguous()
valid = (torch.rand(P, I, device=dev, generator=g) < 0.58).contiguous()
del dup
lib, st = L.lib(), L.stream()
cnt = torch.empty(P, dtype=torch.int32, device=dev); start = torch.empty_like(cnt); tot = torch.empty(1, dtype=torch.int32, device=dev)
out = torch.empty(P * I, 3, device=dev)
nb = int(lib.ia_deform_filter_compact_tmp_bytes(L.i64(P))); tmp = torch.empty((nb + 7) // 8, dtype=torch.int64, device=dev)
res = {}
nb2 = int(lib.ia_deform_filter_tiles_tmp_bytes(L.i64(P))); tmp2 = torch.empty((nb2 + 7) // 8, dtype=torch.int64, device=dev)
xw = x.clone()
for dbg in ("single_pass", "tiles"):
    def run():
        if dbg == "single_pass":
            L.check(lib.ia_deform_filter_compact(L.i64(P), L.i32(I), L.ptr(x), L.ptr(valid), L.ptr(cnt), L.ptr(start), L.ptr(out), L.ptr(None),
                                                 L.ptr(None), L.ptr(tot), L.ptr(tmp), C.c_size_t(tmp.numel() * 8), st), "fcc")
        else:        # (x is consumed: later repetitions filter the packed leftovers -- same traffic, the timing is what is measured)
            L.check(lib.ia_deform_filter_tiles(L.i64(P), L.i32(I), L.ptr(xw), L.ptr(valid), L.ptr(cnt), L.ptr(start), L.ptr(None), L.ptr(None),
                                               L.ptr(tot), L.ptr(tmp2), C.c_size_t(tmp2.numel() * 8), st), "tiles")
            L.check(lib.ia_deform_pack_tiles(L.i64(P), L.i32(I), L.ptr(xw), L.ptr(None), L.ptr(start), L.ptr(out), L.ptr(None), L.ptr(tmp2), st),
                    "pack")
    for _ in range(2): run()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    Q = int(tot.item())
    byt = P * (I * 13 + 8) + Q * 12
    res[dbg] = dict(ms=round(ms, 3), Q_per_point=round(Q / P, 3), GBps=round(byt / ms / 1e6, 1))
print(json.dumps(res))
