#!/usr/bin/env python3
"""Where do the launches, the read-backs and the idle time of ONE config-4 step come from?  (f2 of SURVEY 8(f): host orchestration.)

    python tools/launch_audit.py [--out gpurun_out/launch_audit.json] [--headline]

One step of bench.py's `config4` workload (the reference's 4096-ray training batch: uniform_light, spp 512, fwd + bwd + Adam;
--headline: the 540 x 540 headline step instead) under torch.profiler (with_stack) and torch's sync debug mode:
  launches      every device kernel / memcpy / memset of the step, counted; ATen / runtime launches attributed to the innermost
                source line under intrinsicavatar_amd/ (the C-ABI kernels are launched through ctypes and carry no ATen op: they are
                the rest, listed by kernel name)
  readbacks     host <-> device synchronisations by source line (torch.cuda.set_sync_debug_mode("warn"))
  idle          device time line of the step: span, busy, idle, the kernels the device waited in front of
"""
import argparse
import collections
import json
import os
import sys
import traceback
import warnings

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def site_of(stack):
    """innermost frame of a profiler stack (list of 'file(line): fn' strings, innermost first) that lies in the package."""
    for fr in stack:
        if "intrinsicavatar_amd/" in fr and "/_lib.py" not in fr:
            f = fr.split("intrinsicavatar_amd/")[-1]
            return f.strip()
    for fr in stack:
        if "bench.py" in fr or "tools/" in fr:
            return os.path.basename(fr.split("(")[0]) + "(" + fr.split("(", 1)[1] if "(" in fr else fr
    return "other"


def aten_sites(step):
    """every ATen operator of one step that touches a device tensor, attributed to the innermost source line under
    intrinsicavatar_amd/ (forward: the Python stack at dispatch; operators run by the autograd engine have no Python frame of the
    package and are attributed to the backward node's name).  -> [(site, ops)] most common first."""
    from torch.utils._python_dispatch import TorchDispatchMode
    sites = collections.Counter()
    elems = collections.Counter()

    class Mode(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            out = func(*args, **(kwargs or {}))
            name = str(func.name()) if hasattr(func, "name") else str(func)
            if any(n in name for n in ("aten::view", "aten::_unsafe_view", "aten::reshape", "aten::slice", "aten::select", "aten::expand", "aten::t",
                                       "aten::transpose", "aten::detach", "aten::alias", "aten::unsqueeze", "aten::squeeze", "aten::as_strided",
                                       "aten::empty", "aten::permute", "aten::_local_scalar_dense", "aten::is_", "aten::unbind", "aten::split",
                                       "aten::lift_fresh", "aten::_to_copy")) and "aten::_to_copy" not in name:
                return out
            site = None
            for fr in reversed(traceback.extract_stack()):
                if "intrinsicavatar_amd/" in fr.filename and not fr.filename.endswith("_lib.py"):
                    site = f"{os.path.basename(fr.filename)}:{fr.lineno} {(fr.line or '').strip()[:90]}"
                    break
                if fr.filename.endswith("bench.py"):
                    site = f"bench.py:{fr.lineno} {(fr.line or '').strip()[:90]}"
                    break
            if site is None:
                node = getattr(torch._C, "_current_autograd_node", lambda: None)()
                site = f"backward of {node.name()}" if node is not None else "autograd engine / other"
            sites[site + " | " + name.replace("aten::", "")] += 1
            o0 = out[0] if isinstance(out, (tuple, list)) and out else out
            if isinstance(o0, torch.Tensor) and o0.is_cuda:
                elems[site] += o0.numel()
            return out
    with Mode():
        step()
    torch.cuda.synchronize()
    agg = collections.Counter()
    for k, v in sites.items():
        agg[k.split(" | ")[0]] += v
    aten_sites.elements = elems.most_common(40)          # output elements written per source line: where the LARGE element-wise launches are
    return [(k, v, sorted((kk.split(" | ")[1], vv) for kk, vv in sites.items() if kk.startswith(k + " | "))) for k, v in agg.most_common(120)]


def audit(step, warm=2):
    from torch.profiler import profile, ProfilerActivity
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    # ---- read-backs
    syncs = collections.Counter()
    sync_order = []

    def hook(message, category, filename, lineno, file=None, line=None):
        if "prototype feature" in str(message):            # the debug mode's own one-time notice, not a synchronisation
            return
        chain = [f"{os.path.basename(fr.filename)}:{fr.lineno}" for fr in traceback.extract_stack()
                 if "intrinsicavatar_amd" in fr.filename and "launch_audit" not in fr.filename and not fr.filename.endswith("_lib.py")]
        sync_order.append(" > ".join(chain[-4:]))
        for fr in reversed(traceback.extract_stack()):
            if "intrinsicavatar_amd" in fr.filename and "launch_audit" not in fr.filename:
                syncs[f"{os.path.basename(fr.filename)}:{fr.lineno} {(fr.line or '').strip()[:100]}"] += 1
                return
        syncs["other: " + str(message)[:80]] += 1
    old = warnings.showwarning
    warnings.showwarning = hook
    warnings.simplefilter("always")
    torch.cuda.set_sync_debug_mode("warn")
    try:
        step()
    finally:
        torch.cuda.set_sync_debug_mode("default")
        warnings.showwarning = old
    torch.cuda.synchronize()
    # ---- launches + time line
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        step()
        torch.cuda.synchronize()
    evs = prof.events()
    dev_evs = [e for e in evs if getattr(e, "device_type", None) is not None and str(e.device_type).endswith("CUDA")]
    by_site = collections.Counter()
    by_site_us = collections.Counter()
    by_op = collections.Counter()
    aten_launches = 0
    seen_kernels = set()
    for e in evs:
        ks = getattr(e, "kernels", None) or []
        if not ks or str(getattr(e, "device_type", "")).endswith("CUDA"):
            continue
        # count every kernel once, at the innermost CPU op that owns it
        fresh = [k for k in ks if id(k) not in seen_kernels]
        if not fresh:
            continue
        if e.cpu_children and any(getattr(c, "kernels", None) for c in e.cpu_children):
            continue
        for k in fresh:
            seen_kernels.add(id(k))
        site = site_of(list(e.stack or []))
        by_site[site] += len(fresh)
        by_site_us[site] += sum(getattr(k, "duration", 0) for k in fresh)
        by_op[e.name] += len(fresh)
        aten_launches += len(fresh)
    names = collections.Counter()
    name_us = collections.Counter()
    spans = []
    for e in dev_evs:
        names[e.name[:90]] += 1
        tr = e.time_range
        name_us[e.name[:90]] += tr.end - tr.start
        spans.append((tr.start, tr.end, e.name[:70]))
    spans.sort()
    idle = busy = 0.0
    gaps = []
    if spans:
        t0, busy_end = spans[0][0], spans[0][0]
        for s, e_, nm in spans:
            if s > busy_end:
                gaps.append((s - busy_end, nm))
                busy += e_ - s
            else:
                busy += max(0.0, e_ - busy_end)
            busy_end = max(busy_end, e_)
        idle = sum(g for g, _ in gaps)
        span = busy_end - t0
    else:
        span = 0.0
    gap_by = collections.Counter()
    for g_, nm in gaps:
        gap_by[nm] += g_
    # what the HOST did while the device waited: for the longest gaps, the host-side operators / runtime calls that started inside
    # the gap (a read-back shows as hipMemcpyWithStream / hipStreamSynchronize, a slow Python section as a long list of small operators)
    cpu_evs = sorted(((e.time_range.start, e.time_range.end, e.name, site_of(list(e.stack or []))) for e in evs
                      if not str(getattr(e, "device_type", "")).endswith("CUDA")), key=lambda t: t[0])
    gap_context = []
    prev_end = None
    timeline = []
    busy_end = spans[0][0] if spans else 0
    last = None
    for s, e_, nm in spans:
        if s > busy_end and last is not None:
            timeline.append((s - busy_end, busy_end, s, last, nm))
        if e_ >= busy_end:
            last = nm
        busy_end = max(busy_end, e_)
    for g_, a, b, before, after in sorted(timeline, reverse=True)[:14]:
        inside = [(n, st) for (cs, ce, n, st) in cpu_evs if a - 30.0 <= cs <= b]
        cnt = collections.Counter(n for n, _ in inside)
        sites = collections.Counter(st for _, st in inside if st != "other")
        gap_context.append(dict(gap_us=round(g_, 1), at_ms=round((a - spans[0][0]) / 1e3, 3), after_kernel=before, before_kernel=after,
                                host_events=cnt.most_common(12), host_sites=sites.most_common(6)))
    return dict(
        device_launches=len(dev_evs), aten_or_runtime_launches=aten_launches, abi_or_unattributed_launches=len(dev_evs) - aten_launches,
        readbacks=sum(syncs.values()), readbacks_by_line=syncs.most_common(), readbacks_in_order=sync_order,
        span_ms=round(span / 1e3, 3), busy_ms=round(busy / 1e3, 3), idle_ms=round(idle / 1e3, 3), idle_frac=round(idle / max(span, 1e-9), 4),
        launches_by_source_line=[(s, n, round(by_site_us[s] / 1e3, 3)) for s, n in by_site.most_common(80)],
        launches_by_aten_op=by_op.most_common(40),
        kernels_by_name=names.most_common(70),
        kernel_ms_by_name=[(k, round(v / 1e3, 3), names[k]) for k, v in name_us.most_common(40)],
        idle_us_in_front_of=[(round(v, 1), k) for k, v in gap_by.most_common(25)], longest_gaps=gap_context)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--headline", action="store_true")
    ap.add_argument("--hw", type=int, default=540)
    args = ap.parse_args()
    import bench as B
    from intrinsicavatar_amd import build
    build.build()
    dev = "cuda:0"
    torch.cuda.memory._set_allocator_settings("roundup_power2_divisions:8")
    rs, rays, export, mat, sg = B.build_headline(dev, args.hw, 1024, 0, "male-3-casual:0")
    bg = torch.ones(3, device=dev)
    if args.headline:
        # the 540 x 540 step: ATen operators by source line and by output elements (no profiler trace: the step is too large for one)
        import contextlib
        from intrinsicavatar_amd import pbr, optim
        n = rays.shape[0]
        g = torch.Generator().manual_seed(1234)
        target, tmask = torch.rand((n, 3), generator=g).to(dev), (torch.rand(n, generator=g) > 0.5).float().to(dev)
        params = rs.parameters() + [p for p in mat.parameters() if p.requires_grad] + list(sg.parameters())
        opt, sched = optim.reference_optimizer(rs, material=mat, emitter=sg)

        def hstep():
            for p in params:
                p.grad = None
            img = sg.generate_image()
            leaf = img.detach().requires_grad_(True)
            emitter = pbr.EnvironmentLightTensor(leaf.detach())
            emitter.update_pdf()
            rs.forward_backward_phys(rays, target, mat, emitter, 1024, None, None, target_mask=tmask, render_mode="light", env_base=leaf,
                                     background_color=bg, global_illumination=True, light_sampling="per_point")
            img.backward(leaf.grad)
            opt.step()
        rs.SECONDARY_STREAMS = 1            # the dispatch mode is per thread: keep the march on the caller's
        for _ in range(2):
            hstep()
        torch.cuda.synchronize()
        res = dict(workload="headline step (540 x 540, spp 1024), one march stream", aten_ops_by_source_line=aten_sites(hstep),
                   output_elements_by_source_line=aten_sites.elements)
        txt = json.dumps(res, indent=1)
        if args.out:
            open(args.out, "w").write(txt)
        print(txt)
        return
    step, info = B.build_config4_step(rs, rays, mat, sg, dev, bg)
    res = audit(step)
    res["workload"] = info
    res["aten_ops_by_source_line"] = aten_sites(step)
    txt = json.dumps(res, indent=1)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        open(args.out, "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main()
