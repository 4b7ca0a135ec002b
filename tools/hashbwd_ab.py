"""A/B of the two fill schedules of ia_hashgrid_bwd_binned (IA_HASHBWD_FILL=direct|staged, read once per process)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from intrinsicavatar_amd import build; build.build()
from intrinsicavatar_amd import fields
dev = "cuda:0"
n = 4_400_000
g = torch.Generator().manual_seed(0)
base = torch.rand((n // 16, 1, 3), generator=g) * 0.5 + 0.25
dirs = torch.nn.functional.normalize(torch.randn((n // 16, 1, 3), generator=g), dim=-1)
t = torch.arange(16).float()[None, :, None] * 0.0135
x = (base + dirs * t).reshape(-1, 3).clamp(0, 1).to(dev).contiguous()
n = x.shape[0]
table = torch.zeros(fields.hash_n_entries() * 2, device=dev)
gE = torch.randn((n, 32), generator=g).to(dev)
gG = torch.randn((n, 32), generator=g).to(dev)
q = torch.randn((n, 3), generator=g).to(dev)
def t_(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
mask = 0x0FFF
print(os.environ.get("IA_HASHBWD_FILL", "staged"),
      "first-order %.2f ms" % t_(lambda: fields.hashgrid_backward(x, gE, table, level_mask=mask)),
      " with second-order %.2f ms" % t_(lambda: fields.hashgrid_backward(x, gE, table, g_jac=gG, q=q, level_mask=mask)))
table.zero_(); fields.hashgrid_backward(x, gE, table, g_jac=gG, q=q, level_mask=mask)
print("checksum %.6e  absmax %.6e" % (float(table.double().sum()), float(table.abs().max())))
