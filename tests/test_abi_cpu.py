"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/ia_amd.h declares.
(No compute calls here -- there is no GPU in the build container.)"""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def so_path():
    from intrinsicavatar_amd import build
    return build.build()


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "ia_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(ia_[a-z0-9_]+)\s*\(", hdr)))


def test_header_symbols_exported(so_path):
    lib = ctypes.CDLL(so_path)
    syms = _declared_symbols()
    assert len(syms) >= 20
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, f"declared in include/ia_amd.h but not exported: {missing}"


def test_version_and_error_string(so_path):
    lib = ctypes.CDLL(so_path)
    lib.ia_last_error.restype = ctypes.c_char_p
    assert lib.ia_version() >= 100
    assert isinstance(lib.ia_last_error(), bytes)
    lib.ia_scan_tmp_bytes.restype = ctypes.c_int64
    assert lib.ia_scan_tmp_bytes(ctypes.c_int64(10_000_000)) > 8 * 10_000_000 // 1024


def test_no_cpu_fallback():
    """the product path must fail loudly on CPU tensors instead of silently falling back."""
    import torch
    from intrinsicavatar_amd import lib_nerfacc, _lib
    with pytest.raises((NotImplementedError, _lib.IaError)):
        lib_nerfacc.unpack_info(torch.zeros((4, 2), dtype=torch.int32), 0)
    with pytest.raises((NotImplementedError, _lib.IaError)):
        lib_nerfacc.pack_info(torch.zeros(4, dtype=torch.int64), 2)


def test_package_never_imports_oracle():
    pkg = os.path.join(ROOT, "intrinsicavatar_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "libia_oracle" not in src, f


def test_install_aliases_resolves_the_reference_imports():
    """the import statements of the reference (models/intrinsic_avatar.py:20-46, models/volrend.py:10-14,
    models/network_utils.py:7) land in this package after install_aliases(); no GPU needed to import."""
    import subprocess, sys
    code = ("import intrinsicavatar_amd as ia; ia.install_aliases();"
            "import nerfacc; from nerfacc import (RayIntervals, OccGridEstimator, traverse_grids, render_visibility_from_alpha,"
            " render_visibility_from_density, render_weight_from_alpha, accumulate_along_rays);"
            "from nerfacc.volrend import render_weight_from_density, render_weight_from_alpha as r2, accumulate_along_rays as a2;"
            "from lib.torch_pbr import rgb_to_srgb, luminance, luma, max_value; import lib.torch_pbr;"
            "from lib.nerfacc import (ray_resampling, ray_resampling_merge, ray_resampling_fine, ray_resampling_sdf_fine, pack_info,"
            " pack_data, unpack_info, unpack_data);"
            "import tinycudann as tcnn; assert hasattr(tcnn, 'Encoding') and hasattr(tcnn, 'free_temporary_memory');"
            "assert nerfacc.__name__ == 'intrinsicavatar_amd.nerfacc'; print('ok')")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-1500:]


def _call_sites():
    """(file, line, entry point, [arg kind or None], has_star) of every `<lib>.ia_*(...)` call in the package, bench and tools.
    arg kind = the ctypes class the argument expression evidently has (L.i64 / L.i32 / L.f32 / L.ptr / L.stream / C.c_*)."""
    import ast
    import ctypes as C
    kinds = {"i64": C.c_int64, "i32": C.c_int, "f32": C.c_float, "ptr": C.c_void_p, "stream": C.c_void_p,
             "c_size_t": C.c_size_t, "c_int64": C.c_int64, "c_int": C.c_int, "c_float": C.c_float, "c_void_p": C.c_void_p,
             "c_uint32": C.c_uint32, "c_uint64": C.c_uint64}
    files = [os.path.join(ROOT, "bench.py")]
    for d in ("intrinsicavatar_amd", "tools"):
        files += [os.path.join(ROOT, d, f) for f in sorted(os.listdir(os.path.join(ROOT, d))) if f.endswith(".py")]
    out = []
    for path in files:
        tree = ast.parse(open(path).read())
        for node in ast.walk(tree):
            if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr.startswith("ia_"):
                args, star = [], False
                for a in node.args:
                    if isinstance(a, ast.Starred):
                        star = True
                        continue
                    k = None
                    if isinstance(a, ast.Call) and isinstance(a.func, ast.Attribute):
                        k = kinds.get(a.func.attr)
                    args.append(k)
                out.append((os.path.relpath(path, ROOT), node.lineno, node.func.attr, args, star))
    return out


def test_every_call_site_matches_the_header():
    """host logic: the loader takes argtypes / restype of all entry points from include/ia_amd.h; every call site in the package
    must pass the declared NUMBER of arguments and, where the expression shows its ctypes class, the declared TYPE."""
    from intrinsicavatar_amd import _lib
    protos = _lib.header_prototypes()
    assert len(protos) == len(_declared_symbols())
    sites = _call_sites()
    assert len(sites) > 100
    bad = []
    for f, line, name, args, star in sites:
        if name not in protos:
            bad.append((f, line, name, "not declared in include/ia_amd.h"))
            continue
        want = protos[name][1]
        if (not star and len(args) != len(want)) or (star and len(args) > len(want)):
            bad.append((f, line, name, f"{len(args)} arguments, header declares {len(want)}"))
            continue
        if not star:
            for i, (k, w) in enumerate(zip(args, want)):
                if k is not None and k is not w:
                    bad.append((f, line, name, f"argument {i}: {k.__name__} passed, header declares {w.__name__}"))
    assert not bad, "\n".join(map(str, bad))


def test_loader_declares_argtypes_for_every_entry_point(so_path):
    from intrinsicavatar_amd import _lib
    l = _lib.lib()
    for name, (restype, argtypes) in _lib.header_prototypes().items():
        fn = getattr(l._cdll, name)
        assert fn.argtypes is not None and list(fn.argtypes) == argtypes and fn.restype is restype, name
    with pytest.raises((TypeError, ctypes.ArgumentError)):
        l.ia_scan_tmp_bytes()                            # wrong arity is an error in Python, not a garbage read
