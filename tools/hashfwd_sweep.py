"""flat vs XCD-partitioned hash forward over batch sizes (threshold HASH_FWD_XCD_MIN in fields.py)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from intrinsicavatar_amd import build; build.build()
from intrinsicavatar_amd import fields
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
table = (torch.rand(fields.hash_n_entries() * 2, generator=g) * 2e-4 - 1e-4).to(dev)
def t_(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for n in (1 << 15, 1 << 16, 1 << 17, 1 << 18, 1 << 19, 1 << 20, 1 << 21):
    base = torch.rand((n // 16, 1, 3), generator=g) * 0.5 + 0.25
    dirs = torch.nn.functional.normalize(torch.randn((n // 16, 1, 3), generator=g), dim=-1)
    x = (base + dirs * (torch.arange(16).float()[None, :, None] * 0.0135)).reshape(-1, 3).clamp(0, 1).to(dev).contiguous()
    res = []
    for m in ("flat", "xcd"):
        os.environ["IA_HASH_FWD"] = m
        res.append(t_(lambda: fields.hashgrid_forward(x, table)))
    print(f"n = {n:8d}: flat {res[0] * 1e3:8.1f} us   xcd {res[1] * 1e3:8.1f} us")
