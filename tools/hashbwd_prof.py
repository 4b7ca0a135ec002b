import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from intrinsicavatar_amd import build; build.build()
from intrinsicavatar_amd import fields
dev = "cuda:0"
n = 3_800_000
g = torch.Generator().manual_seed(0)
base = torch.rand((n // 64, 1, 3), generator=g) * 0.5 + 0.25
dirs = torch.nn.functional.normalize(torch.randn((n // 64, 1, 3), generator=g), dim=-1)
t = torch.arange(64).float()[None, :, None] * 0.002
x = (base + dirs * t).reshape(-1, 3).clamp(0, 1).to(dev).contiguous()
n = x.shape[0]
table = torch.zeros(fields.hash_n_entries() * 2, device=dev)
gE = torch.randn((n, 32), device=dev)
gG = torch.randn((n, 32), device=dev)
q = torch.randn((n, 3), device=dev)
for _ in range(3):
    fields.hashgrid_backward(x, gE, table, method="binned")
    fields.hashgrid_backward(x, gE, table, g_jac=gG, q=q, method="binned")
torch.cuda.synchronize()
