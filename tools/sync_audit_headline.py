#!/usr/bin/env python3
"""every host<->device synchronisation point of ONE headline step (540 x 540, spp 1024, light, GI; fwd + bwd), by source line
(torch sync debug mode; the secondary march on one stream so that the warnings come from one thread)."""
import collections, os, sys, traceback, warnings
os.environ["IA_SECONDARY_STREAMS"] = "1"
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from intrinsicavatar_amd import pbr
dev = "cuda:0"
hw = int(os.environ.get("IA_HW", "540"))
rs, rays, export, mat, sg = bench.build_headline(dev, hw, 1024, 0, "male-3-casual:0")
n = rays.shape[0]
g = torch.Generator().manual_seed(1)
trgb = torch.rand((n, 3), generator=g).to(dev); tmask = (torch.rand(n, generator=g) > 0.5).float().to(dev)
bg = torch.ones(3, device=dev)
params = rs.parameters() + [p for p in mat.parameters() if p.requires_grad] + list(sg.parameters())

def step():
    for p in params:
        p.grad = None
    img = sg.generate_image()
    leaf = img.detach().requires_grad_(True)
    em = pbr.EnvironmentLightTensor(leaf.detach()); em.update_pdf()
    o = rs.forward_backward_phys(rays, trgb, mat, em, 1024, None, None, target_mask=tmask, render_mode="light", env_base=leaf,
                                 background_color=bg, global_illumination=True, light_sampling="per_point")
    img.backward(leaf.grad)
    return o
for _ in range(2):
    step()
torch.cuda.synchronize()
cnt = collections.Counter()
def hook(message, category, filename, lineno, file=None, line=None):
    for fr in reversed(traceback.extract_stack()):
        if "intrinsicavatar_amd" in fr.filename:
            cnt[f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.line.strip()[:100]}"] += 1
            break
    else:
        cnt["other"] += 1
warnings.showwarning = hook
warnings.simplefilter("always")
torch.cuda.set_sync_debug_mode("warn")
step()
torch.cuda.set_sync_debug_mode("default")
print("total syncs:", sum(cnt.values()))
for k, v in cnt.most_common():
    print(v, k)
