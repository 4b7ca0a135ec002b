#!/usr/bin/env python3
"""List every host<->device synchronisation point of one training step (torch sync debug mode), by source line."""
import collections, os, sys, warnings, traceback
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from intrinsicavatar_amd import build; build.build()
from intrinsicavatar_amd import synthetic as S
dev = "cuda:0"
rs, rays, _ = S.build_frame(dev, 540, 540, pose_seed=0, beta=0.01, num_samples_per_ray=128)
n = rays.shape[0]
g = torch.Generator().manual_seed(1)
trgb = torch.rand((n, 3), generator=g).to(dev); tmask = (torch.rand(n, generator=g) > 0.5).float().to(dev)
params = rs.parameters()
for _ in range(2):
    rs.forward_backward(rays, trgb, tmask)
torch.cuda.synchronize()
cnt = collections.Counter()
def hook(message, category, filename, lineno, file=None, line=None):
    st = traceback.extract_stack()
    for fr in reversed(st):
        if "intrinsicavatar_amd" in fr.filename and "sync_audit" not in fr.filename:
            cnt[f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.line.strip()[:90]}"] += 1
            break
    else:
        cnt["other"] += 1
warnings.showwarning = hook
warnings.simplefilter("always")
torch.cuda.set_sync_debug_mode("warn")
rs.forward_backward(rays, trgb, tmask)
torch.cuda.set_sync_debug_mode("default")
print("total syncs:", sum(cnt.values()))
for k, v in cnt.most_common():
    print(v, k)
