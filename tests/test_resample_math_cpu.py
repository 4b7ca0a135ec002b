"""CPU: the PRODUCT's re-sampling arithmetic (intrinsicavatar_amd/csrc/resample_math.h -- the functions resample.hip's kernels
are made of: per-ray CDF tables, per-element inversion) replayed on the host by tests/resample_harness.c and held, bit for bit,

  * to the golden vectors of the reference's own K1..K4 kernel bodies (tests/golden/golden_resampling.npz), and
  * to the oracle on seeded ragged batches (rays without samples, one-sample rays, zero weights, weights summing above 1,
    sign-change patterns, gaps in the edge lists, n from 2 to 1024).

The GPU tests (tests/test_gpu_parity.py) run the same cases through the kernels themselves."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("rs") / "libresample_harness.so")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-fvisibility=hidden", "-o", so,
                           os.path.join(HERE, "resample_harness.c"), "-lm"])
    return C.CDLL(so)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _rpi(oracle, pi, n, add):
    from oracle.oracle import _resample_info
    return _resample_info(np.ascontiguousarray(pi, np.int32), n, add)


def k1(h, oracle, pi, st, en, w, sd, n):
    pi = np.ascontiguousarray(pi, np.int32)
    rpi, T = _rpi(oracle, pi, n, 0)
    ts, offs, idx = np.full(T, -7, np.float32), np.full(T, -7, np.float32), np.full(T, -7, np.int64)
    fg, bg, surf = np.full(w.shape[0], -7, np.int32), np.full(pi.shape[0], -7, np.int32), np.full(pi.shape[0], -7, np.int64)
    h.rs_h_k1(C.c_int64(pi.shape[0]), _p(pi), _p(st), _p(en), _p(w), _p(sd), C.c_int(n), _p(rpi), _p(ts), _p(offs), _p(surf), _p(idx),
              _p(fg), _p(bg), C.c_int64(w.shape[0]))
    return rpi, ts[:, None], offs[:, None], idx, fg, bg, surf


def k2(h, oracle, pi, vals, il, ir, w, n):
    pi = np.ascontiguousarray(pi, np.int32)
    rpi, T = _rpi(oracle, pi, n, 1)
    ov, od = np.full(T, -7, np.float32), np.full(T, -7, np.float32)
    fl = [np.full(T, 9, np.uint8) for _ in range(4)]
    il8, ir8 = np.ascontiguousarray(il, np.uint8), np.ascontiguousarray(ir, np.uint8)
    h.rs_h_k2(C.c_int64(pi.shape[0]), _p(pi), _p(vals), _p(il8), _p(ir8), _p(w), C.c_int(n), _p(rpi), _p(ov), _p(od), _p(fl[0]), _p(fl[1]),
              _p(fl[2]), _p(fl[3]), C.c_int64(vals.shape[0]))
    return rpi, ov, od, fl[0].astype(bool), fl[1].astype(bool), fl[2].astype(bool), fl[3].astype(bool)


def k34(h, oracle, sdf_mode, pi, st, en, wa, sd, n):
    pi = np.ascontiguousarray(pi, np.int32)
    rpi, T = _rpi(oracle, pi, n, 0)
    os_, oe, fg = np.full(T, -7, np.float32), np.full(T, -7, np.float32), np.full(T, 9, np.uint8)
    h.rs_h_k34(C.c_int(sdf_mode), C.c_int64(pi.shape[0]), _p(pi), _p(st), _p(en), _p(wa), _p(sd), C.c_int(n), _p(rpi), _p(os_), _p(oe),
               _p(fg), C.c_int64(wa.shape[0]))
    return rpi, os_[:, None], oe[:, None], fg.astype(bool)


def k34_small(h, oracle, sdf_mode, pi, st, en, wa, sd, n):
    pi = np.ascontiguousarray(pi, np.int32)
    rpi, T = _rpi(oracle, pi, n, 0)
    os_, oe, fg = np.full(T, -7, np.float32), np.full(T, -7, np.float32), np.full(T, 9, np.uint8)
    rc = h.rs_h_k34_small(C.c_int(sdf_mode), C.c_int64(pi.shape[0]), _p(pi), _p(st), _p(en), _p(wa), _p(sd), C.c_int(n), _p(rpi), _p(os_),
                          _p(oe), _p(fg))
    assert rc == 0
    return rpi, os_[:, None], oe[:, None], fg.astype(bool)


def _eq(got, ref, what):
    for i, (a, b) in enumerate(zip(got, ref)):
        assert a.shape == b.shape, (what, i, a.shape, b.shape)
        np.testing.assert_array_equal(a, b, err_msg=f"{what} output {i}")


def test_golden_vectors_of_the_reference_kernels(harness, oracle, golden_dir):
    g = np.load(os.path.join(golden_dir, "golden_resampling.npz"))
    for c in range(int(g["n_cases"])):
        k = f"c{c}_"
        pi, st, en = g[k + "packed_info"], g[k + "starts"], g[k + "ends"]
        w, al, sd, n = g[k + "weights"], g[k + "alphas"], g[k + "sdfs"], int(g[k + "n"])
        _eq(k1(harness, oracle, pi, st, en, w, sd, n), [g[k + "k1_" + nm] for nm in ("rpi", "ts", "offsets", "indices", "fg_counts", "bg_counts", "surface_idx")], f"K1 case {c}")
        _eq(k34(harness, oracle, 0, pi, st, en, w, sd, n), [g[k + "k3_" + nm] for nm in ("rpi", "starts", "ends", "is_fg")], f"K3 case {c}")
        _eq(k34(harness, oracle, 1, pi, st, en, al, sd, n), [g[k + "k4_" + nm] for nm in ("rpi", "starts", "ends", "is_fg")], f"K4 case {c}")
    for c in range(int(g["n_edge_cases"])):
        k = f"e{c}_"
        _eq(k2(harness, oracle, g[k + "packed_info"], g[k + "vals"], g[k + "is_left"], g[k + "is_right"], g[k + "weights"], int(g[k + "n"])),
            [g[k + "k2_" + nm] for nm in ("rpi", "vals", "dists", "is_left", "is_right", "is_resample", "is_fg")], f"K2 case {c}")


def _ragged(rng, n_rays, max_steps, p_empty=0.2):
    steps = rng.integers(1, max_steps + 1, n_rays)
    steps[rng.random(n_rays) < p_empty] = 0
    steps[rng.random(n_rays) < 0.05] = 1
    base = np.cumsum(steps) - steps
    return np.stack([base, steps], -1).astype(np.int32), int(steps.sum())


@pytest.mark.parametrize("seed,n", [(0, 2), (1, 4), (6, 7), (7, 8), (2, 16), (3, 64), (4, 256), (5, 1024)])
def test_sample_lists_against_the_oracle(harness, oracle, seed, n):
    rng = np.random.default_rng(seed)
    pi, S = _ragged(rng, 700, 48)
    st, en = np.zeros(S, np.float32), np.zeros(S, np.float32)
    for b, s in pi:
        if s:
            dts = rng.uniform(0.005, 0.05, s).astype(np.float32)
            t = rng.uniform(3, 5) + np.cumsum(dts) - dts
            st[b:b + s], en[b:b + s] = t, t + dts
    al = rng.uniform(0, 0.5, S).astype(np.float32)
    al[rng.random(S) < 0.1] = 0.0
    al[rng.random(S) < 0.02] = 1.0
    sd = rng.normal(0.05, 0.2, S).astype(np.float32)
    w, _ = oracle.render_weight_from_alpha(al, pi)
    for weights, tag in ((w, "T*alpha"), ((w * 3.0).astype(np.float32), "sum>1"), (np.zeros_like(w), "zero"), ((w * 1e-6).astype(np.float32), "tiny")):
        _eq(k1(harness, oracle, pi, st, en, weights, sd, n), oracle.ray_resampling(pi, st, en, weights, sd, n), f"K1 {tag} n={n}")
        _eq(k34(harness, oracle, 0, pi, st, en, weights, sd, n), oracle.ray_resampling_fine(pi, st, en, weights, n), f"K3 {tag} n={n}")
    _eq(k34(harness, oracle, 1, pi, st, en, al, sd, n), oracle.ray_resampling_sdf_fine(pi, st, en, al, sd, n), f"K4 n={n}")
    if n <= 8:                  # the register-resident form the kernels use for few points per ray
        _eq(k34_small(harness, oracle, 1, pi, st, en, al, sd, n), oracle.ray_resampling_sdf_fine(pi, st, en, al, sd, n), f"K4 small n={n}")
        for weights in (w, (w * 3.0).astype(np.float32), np.zeros_like(w)):
            _eq(k34_small(harness, oracle, 0, pi, st, en, weights, sd, n), oracle.ray_resampling_fine(pi, st, en, weights, n), f"K3 small n={n}")


@pytest.mark.parametrize("seed,n", [(0, 1), (1, 2), (2, 16), (3, 64), (4, 512)])
def test_merged_edge_lists_against_the_oracle(harness, oracle, seed, n):
    rng = np.random.default_rng(100 + seed)
    pi, E = _ragged(rng, 600, 40)
    vals = np.zeros(E, np.float32)
    il, ir = np.zeros(E, bool), np.zeros(E, bool)
    w = np.zeros(E, np.float32)
    for b, s in pi:
        if s == 0:
            continue
        t = rng.uniform(2, 4) + np.cumsum(rng.uniform(0.01, 0.06, s))
        vals[b:b + s] = t.astype(np.float32)
        # runs of contiguous intervals separated by gaps: an edge is a left edge when an interval starts at it
        gap = rng.random(s) < 0.25                     # gap[k]: no interval between edge k and k + 1
        gap[s - 1] = True
        for k in range(s):
            il[b + k] = not gap[k]
            ir[b + k] = k > 0 and not gap[k - 1]
        if rng.random() < 0.2:
            il[b] = True                                # a first edge that claims an interval the next edge does not close
        w[b:b + s] = np.where(gap, rng.uniform(0, 0.3, s) * (rng.random(s) < 0.3), rng.uniform(0, 0.4, s)).astype(np.float32)
    for weights, tag in ((w, "plain"), ((w * 4).astype(np.float32), "sum>1"), (np.zeros_like(w), "zero")):
        _eq(k2(harness, oracle, pi, vals, il, ir, weights, n), oracle.ray_resampling_merge(pi, vals, il, ir, weights, n), f"K2 {tag} n={n}")
