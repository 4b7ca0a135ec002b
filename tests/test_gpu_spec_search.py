"""GPU (MI355X): the K9-consistent early filter of the Broyden search (ia_fuse_broyden_spec[_rows], csrc/snarf.hip), the product
path of every query batch.  The reference runs all 13 searches of a point to their end (fuse_cuda_kernel_fast.cu:252-452) and
filter.cu:10-54 drops every root that has a later init's root within 1e-4; the product search walks a point's inits in reverse and
retires a search once it enters the retirement box of a TIGHT root a later init has found (same voxel cell, within eps, own
J_inv estimate sane: K9 would drop it wherever exactly it ends), and redoes a point with the filter off when a completed root is
neither surely a duplicate nor surely distinct (snarf.hip).

Bars: eps = 0 is the exact search, bit for bit; with the default eps every COMPLETED item is the exact search's item bit for
bit, at least a quarter of the fetches are gone, and -- with the per-cell tightness table of the true skinning Jacobian
(ia_cell_tightness) that the product path passes -- the candidate SET after K9 EQUALS the exact search's on every point, so the
min-over-candidates SDF is bit-identical everywhere (the same quantities tools/spec_search_probe.py reports for the eight
reference poses in profiles/r04_spec_search_probe_poses.jsonl: 0 of 145.8 M points differ; without the table 1 ... 12 per pose)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def march():
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    from tools import spec_search_probe as SP
    from intrinsicavatar_amd import synthetic as S, fast_snarf
    rs, rays, _ = S.build_frame(DEV, 128, 128, pose_seed=0, beta=0.01)
    pts = SP.march_points(rs, rays, 1 << 17)
    assert pts.shape[0] > 500_000
    x0, v0 = SP.search(rs.deformer, pts, None)
    k0 = fast_snarf.filter(x0, v0)
    old = rs.deformer.spec_eps
    rs.deformer.spec_eps = 0.0
    s0 = rs.deformer.deform_sdf(pts, rs.geometry)
    rs.deformer.spec_eps = old
    return SP, rs, pts, (x0, v0, k0, s0)


def test_eps_zero_is_the_exact_search(march):
    SP, rs, pts, (x0, v0, _, _) = march
    cnt = torch.zeros(5, dtype=torch.int64, device=DEV)
    x1, v1 = SP.search(rs.deformer, pts, 0.0, counters=cnt)
    assert torch.equal(v0, v1)
    assert torch.equal(torch.where(v0[..., None], x0, torch.zeros_like(x0)), torch.where(v1[..., None], x1, torch.zeros_like(x1)))
    x2, v2 = SP.search(rs.deformer, pts, 0.0)                       # the variant without counters (the one the product path runs)
    assert torch.equal(v0, v2) and torch.equal(torch.where(v0[..., None], x0, torch.zeros_like(x0)), torch.where(v2[..., None], x2, torch.zeros_like(x2)))
    c = cnt.cpu().tolist()
    # nothing retired; completed valid items = valid items, plus the 13 searches again of every point that was redone (a completed
    # root between 1e-4 and 2e-4 of a recorded one sends its point through the exact redo whatever eps is)
    assert c[1] == 0 and int(v0.sum()) <= c[2] <= int(v0.sum()) + 13 * c[3] and c[3] < 1e-3 * pts.shape[0]


def test_outputs_with_jinv_and_fwd_J_are_the_exact_ones(march):
    """the optional outputs of the training path (J_inv, fwd_J) of every completed item."""
    SP, rs, pts, _ = march
    from intrinsicavatar_amd import fast_snarf
    dfm = rs.deformer
    sub = pts[:300_000].contiguous()
    P, I = sub.shape[0], 13
    vj = fast_snarf.ChannelLastVoxelJ(dfm.voxel_J_cl)

    def run(spec):
        x = torch.zeros((1, P, I, 3), device=DEV); Ji = torch.zeros((1, P, I, 3, 3), device=DEV); Fw = torch.zeros((1, P, I, 3, 3), device=DEV)
        v = torch.zeros((1, P, I), dtype=torch.bool, device=DEV)
        if spec:
            fast_snarf.fuse_broyden_spec(x, sub[None], vj, dfm.tfs, dfm.init_bones, Ji, v, dfm.offset_kernel, dfm.scale_kernel, 1e-5, 1e-1, 1e-3, fwd_J=Fw)
        else:
            fast_snarf.fuse_broyden(x, sub[None], None, vj, dfm.tfs, dfm.init_bones, True, Ji, v, dfm.offset_kernel, dfm.scale_kernel, 1e-5, 1e-1, fwd_J=Fw)
        return x, Ji, Fw, v
    x0, J0, F0, v0 = run(False)
    x1, J1, F1, v1 = run(True)
    assert int((v1 & ~v0).sum()) == 0 and int(v1.sum()) < int(v0.sum())
    m = v1[..., None]
    assert torch.equal(torch.where(m, x1, torch.zeros_like(x1)), torch.where(m, x0, torch.zeros_like(x0)))
    m = v1[..., None, None]
    assert torch.equal(torch.where(m, J1, torch.zeros_like(J1)), torch.where(m, J0, torch.zeros_like(J0)))
    assert torch.equal(torch.where(m, F1, torch.zeros_like(F1)), torch.where(m, F0, torch.zeros_like(F0)))


@pytest.mark.parametrize("eps", [1e-3])
def test_speculation_removes_fetches_and_changes_almost_nothing(march, eps):
    SP, rs, pts, ref = march
    r = SP.compare(rs.deformer, rs.geometry, pts, eps, ref)
    print(r)
    assert r["completed_items_bit_identical"]
    exact_fetches = SP.compare(rs.deformer, rs.geometry, pts, 0.0, ref)["fetches"]
    assert r["fetches"] <= 0.75 * exact_fetches, (r["fetches"], exact_fetches)
    P = pts.shape[0]
    assert rs.deformer.cell_tight is not None
    assert r["set_mismatch"] == 0.0            # the candidate set after K9 IS the exact search's, on every point
    assert r["lost_root"] == 0.0
    assert r["sdf_bits_differ"] == 0.0 and r["sdf_max_abs"] == 0.0
    assert 0 < r["redone_points"] < 1e-3 * P   # points the kernel searched again with the filter off


def test_candidate_sets_equal_on_the_hardest_reference_pose():
    """aist frame 319 (out-of-distribution animation pose): the pose on which the filter WITHOUT the cell table differed from
    search-to-the-end + K9 most often (12 of 18.1 M march points, two distinct roots lost; tools/k9_mismatch_dump.py: pairs of roots next
    to a fold of the skinning map).  4.5 M march points: candidate sets, min-SDF bits identical; without the table the same batch differs."""
    from tools import spec_search_probe as SP
    from intrinsicavatar_amd import synthetic as S, fast_snarf
    rs, rays, _ = S.build_frame(DEV, 540, 540, pose_seed=0, beta=0.01, pose="aist:319")
    pts = SP.march_points(rs, rays, 1 << 19)
    assert pts.shape[0] > 4_000_000
    dfm = rs.deformer
    x0, v0 = SP.search(dfm, pts, None)
    k0 = fast_snarf.filter(x0, v0)
    old = dfm.spec_eps
    dfm.spec_eps = 0.0
    s0 = dfm.deform_sdf(pts, rs.geometry)
    dfm.spec_eps = old
    r = SP.compare(dfm, rs.geometry, pts, 1e-3, (x0, v0, k0, s0))
    assert r["completed_items_bit_identical"] and r["set_mismatch"] == 0.0 and r["lost_root"] == 0.0 and r["sdf_max_abs"] == 0.0, r
    assert r["fetches"] < 0.75 * int(torch.zeros(1).item() + 49.7 * pts.shape[0])          # ~35 of 49.7 fetches per point
    frac_vetoed = 1.0 - float((dfm.cell_tight[:-1, :-1, :-1] & 1).float().mean())
    assert 0.02 < frac_vetoed < 0.2, frac_vetoed


def test_cell_tightness_table_against_a_float64_evaluation(march):
    """ia_cell_tightness: per voxel cell, bit 0 set iff the TRUE Jacobian of g(x) = A(x) x + b(x) - xd (weight-gradient term included) keeps the sign
    of its determinant and |J^-1|_F <= 2.5 at the cell's 27 sample points.  Against the same quantity in float64 torch on 40 000 random
    cells + the cells the march's roots fall into; cells within 1e-3 (relative) of the threshold or of det = 0 may go either way."""
    from intrinsicavatar_amd import fast_snarf
    SP, rs, pts, (x0, v0, k0, _) = march
    dfm = rs.deformer
    tab = dfm.cell_tight
    _, D, H, W, _ = dfm.voxel_J_cl.shape
    assert tab.shape == (D, H, W) and tab.dtype == torch.uint8
    assert int(tab[-1].sum()) == 0 and int(tab[:, -1].sum()) == 0 and int(tab[:, :, -1].sum()) == 0       # the last index of an axis is no cell
    frac = float((tab[:-1, :-1, :-1] & 1).float().mean())
    assert 0.5 < frac < 0.995, frac
    g = torch.Generator().manual_seed(0)
    cz, cy, cx = (torch.randint(0, n - 1, (40_000,), generator=g).to(DEV) for n in (D, H, W))
    # + the cells of the march's surviving roots (where the table is actually read)
    roots = x0[0][k0[0]][:200_000]
    gq = (roots + dfm.offset_kernel) * dfm.scale_kernel
    rx = ((gq[:, 0] + 1) / 2 * (W - 1)).floor().long().clamp(0, W - 2)
    ry = ((gq[:, 1] + 1) / 2 * (H - 1)).floor().long().clamp(0, H - 2)
    rz = ((gq[:, 2] + 1) / 2 * (D - 1)).floor().long().clamp(0, D - 2)
    cz, cy, cx = torch.cat([cz, rz]), torch.cat([cy, ry]), torch.cat([cx, rx])
    vJ = dfm.voxel_J_cl[0].double()
    off, sc = dfm.offset_kernel.double(), dfm.scale_kernel.double()
    dims = torch.tensor([W, H, D], device=DEV, dtype=torch.float64)
    dco = sc * (dims - 1) / 2
    n = cz.shape[0]
    worst = torch.zeros(n, dtype=torch.float64, device=DEV)
    dmin, dmax = torch.full((n,), 1e300, dtype=torch.float64, device=DEV), torch.full((n,), -1e300, dtype=torch.float64, device=DEV)
    corners = [vJ[cz + ((c >> 2) & 1), cy + ((c >> 1) & 1), cx + (c & 1)].reshape(n, 3, 4) for c in range(8)]
    idx = torch.stack([cx, cy, cz], -1).double()
    for tz in (0.02, 0.5, 0.98):
        for ty in (0.02, 0.5, 0.98):
            for tx in (0.02, 0.5, 0.98):
                t = torch.tensor([tx, ty, tz], dtype=torch.float64, device=DEV)
                x = ((idx + t) / (dims - 1) * 2 - 1) / sc - off
                J = torch.zeros((n, 3, 3), dtype=torch.float64, device=DEV)
                for c in range(8):
                    w = [(t[a] if (c >> a) & 1 else 1 - t[a]) for a in range(3)]
                    sg = [(1.0 if (c >> a) & 1 else -1.0) for a in range(3)]
                    p_c = torch.einsum("nij,nj->ni", corners[c][:, :, :3], x) + corners[c][:, :, 3]
                    dw = torch.stack([sg[0] * w[1] * w[2] * dco[0], w[0] * sg[1] * w[2] * dco[1], w[0] * w[1] * sg[2] * dco[2]])
                    J += corners[c][:, :, :3] * (w[0] * w[1] * w[2]) + p_c[:, :, None] * dw[None, None, :]
                det = torch.linalg.det(J)
                inv = torch.linalg.inv(torch.where((det.abs() > 1e-30)[:, None, None], J, torch.eye(3, dtype=torch.float64, device=DEV).expand(n, 3, 3)))
                nrm = torch.where(det.abs() > 1e-30, inv.reshape(n, 9).norm(dim=-1), torch.full_like(det, 1e300))
                worst = torch.maximum(worst, nrm)
                dmin, dmax = torch.minimum(dmin, det), torch.maximum(dmax, det)
    one_sign = (dmin > 0) | (dmax < 0)
    want = one_sign & (worst <= fast_snarf.CELL_TAU)
    clear = ((worst - fast_snarf.CELL_TAU).abs() > 1e-3 * fast_snarf.CELL_TAU) & (torch.minimum(dmin.abs(), dmax.abs()) > 1e-6)
    got = (tab[cz, cy, cx] & 1).bool()
    assert int(clear.sum()) > 0.99 * n
    assert torch.equal(got[clear], want[clear]), int((got[clear] != want[clear]).sum())
    assert 0.02 < float((~want).float().mean()) < 0.6            # both kinds of cells are sampled
    # bit 2: the sign of det in a tight cell; bit 1: the cell and its 26 neighbours are tight with ONE sign (no cell cut of the retirement box)
    sign_pos = (tab[cz, cy, cx] & 4).bool()
    assert torch.equal(sign_pos[clear & want], (dmin > 0)[clear & want])
    assert int(((tab & 1) == 0).logical_and((tab & 6) != 0).sum()) == 0          # bits 1 / 2 only on tight cells
    import torch.nn.functional as F
    pos = ((tab & 5) == 5).float()[None, None]
    neg = ((tab & 5) == 1).float()[None, None]
    allpos = -F.max_pool3d(-F.pad(pos, (1, 1, 1, 1, 1, 1), value=0.0), 3, 1) > 0.5
    allneg = -F.max_pool3d(-F.pad(neg, (1, 1, 1, 1, 1, 1), value=0.0), 3, 1) > 0.5
    free3 = (allpos | allneg)[0, 0]
    free3[:, :, -2:] = False; free3[:, -2:, :] = False; free3[-2:, :, :] = False   # a neighbour that is no cell (last index of an axis): keep the cut
    assert torch.equal((tab & 2).bool(), free3)
    assert 0.4 < float(free3[:-1, :-1, :-1].float().mean()) < float((tab & 1)[:-1, :-1, :-1].float().mean())


def test_without_the_cell_table_the_rule_is_round_4a(march):
    """cell_tight = None (C-ABI callers that do not build the table): the search still only completes the exact search's items and K9's
    candidate set differs on at most a few points of a batch -- the figures of profiles/r04_spec_search_probe.json."""
    SP, rs, pts, ref = march
    dfm = rs.deformer
    saved = dfm.cell_tight
    try:
        dfm.cell_tight = None
        r = SP.compare(dfm, rs.geometry, pts, 1e-3, ref)
    finally:
        dfm.cell_tight = saved
    assert r["completed_items_bit_identical"] and r["set_mismatch"] * pts.shape[0] <= 2.5 and r["lost_root"] == 0.0


def test_product_path_is_independent_of_the_batch(march):
    """the product path speculates for every batch size: a point's result does not depend on which other points share its launch
    (ray-batch sharding invariance); spec_eps = 0 gives the reference's exact search."""
    SP, rs, pts, (_, _, _, s0) = march
    dfm = rs.deformer
    assert dfm.spec_eps > 0
    cnt = torch.zeros(5, dtype=torch.int64, device=DEV)
    dfm.spec_counters = cnt
    try:
        s1 = dfm.deform_sdf(pts, rs.geometry)
        assert int(cnt[0]) > 0 and int(cnt[1]) > 0
        assert torch.equal(s1, s0)
        for a, b in ((0, 50_000), (123_457, 131_000), (400_000, 400_001)):
            assert torch.equal(dfm.deform_sdf(pts[a:b].contiguous(), rs.geometry), s1[a:b])
    finally:
        dfm.spec_counters = None
    old = dfm.spec_eps
    try:
        dfm.spec_eps = 0.0
        assert torch.equal(dfm.deform_sdf(pts[:50_000].contiguous(), rs.geometry), s0[:50_000])
    finally:
        dfm.spec_eps = old


def test_candidate_rows_equal_search_plus_k9_pack(march):
    """the search with the bookkeeping in the kernel (ia_fuse_broyden_spec_rows + ia_deform_rows_count / _pack) against the same
    speculative search followed by K9 + count + pack (ia_deform_filter_tiles / ia_deform_pack_tiles): identical packed lists."""
    SP, rs, pts, _ = march
    dfm = rs.deformer
    assert dfm.SPEC_ROWS and pts.shape[0] >= dfm.SPEC_MIN_POINTS
    a = dfm._candidates(pts, with_src=True, want_fwd=True)
    try:
        type(dfm).SPEC_ROWS = False
        b = dfm._candidates(pts, with_src=True, want_fwd=True)
    finally:
        type(dfm).SPEC_ROWS = True
    assert a[4] == b[4] and a[4] > pts.shape[0] // 2                      # Q
    for k in (0, 1, 2, 3):                                               # cand_x, cand_src, cnt, start
        assert torch.equal(a[k], b[k]), k
    src = a[1].long()
    assert torch.equal(a[5].reshape(-1, 9)[src], b[5].reshape(-1, 9)[src])    # fwd_J of the candidates
    s_rows = dfm.deform_sdf(pts, rs.geometry)
    try:
        type(dfm).SPEC_ROWS = False
        s_k9 = dfm.deform_sdf(pts, rs.geometry)
    finally:
        type(dfm).SPEC_ROWS = True
    assert torch.equal(s_rows, s_k9)


def test_candidates_leave_the_packing_kernel_in_hash_grid_coordinates(march):
    """ia_deform_rows_pack with norm_center / norm_scale: (x - center) / scale + 0.5 on the way out (models/rf/geometry.py:155), the bits
    of the three elementwise passes it replaces."""
    SP, rs, pts, _ = march
    dfm, geo = rs.deformer, rs.geometry
    a = dfm._candidates(pts, with_src=True)
    b = dfm._candidates(pts, with_src=True, normalize=(geo.center, geo.scale))
    assert a[4] == b[4] and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    assert torch.equal(b[0], (a[0] - geo.center) / geo.scale + 0.5)
    assert torch.equal(geo.sdf_only(b[0], normalized=True), geo.sdf_only(a[0]))


def test_points_that_run_out_of_row_slots_are_redone_exactly(march, monkeypatch):
    """a completed root that finds the point's row full (a 4th distinct root: ~1e-8 of the march distribution), or that lies between
    1e-4 and 2e-4 of a recorded one, sends its point through the exact redo: all 13 searches with the filter off, K9 applied to the
    13 results as filter.cu:10-54 writes it (rows_flagged_kernel), survivors beyond the row's three slots as overflow records.  The
    kernel's test hook IA_SPEC_TEST_SLOTS=1 leaves ONE slot, so every point with a second root takes that path (tens of thousands
    here, some with four and more survivors): the packed candidate lists must equal exact search + K9 + pack for EVERY point."""
    SP, rs, pts, _ = march
    dfm = rs.deformer
    sub = pts[:250_000].contiguous()                                 # a fifth of the points has a second root: stays inside the list of redone points
    old = dfm.spec_eps
    try:
        dfm.spec_eps = 0.0
        type(dfm).SPEC_ROWS = False
        want = dfm._candidates(sub, with_src=True)                   # exact search + K9 + pack
    finally:
        dfm.spec_eps = old
        type(dfm).SPEC_ROWS = True
    monkeypatch.setenv("IA_SPEC_TEST_SLOTS", "1")
    monkeypatch.setenv("IA_BR_SMALL_MAX", "0")                        # the early-filter kernel, whatever the batch size
    got = dfm._candidates(sub, with_src=True)
    monkeypatch.delenv("IA_SPEC_TEST_SLOTS")
    monkeypatch.delenv("IA_BR_SMALL_MAX")
    assert 10_000 < dfm.last_overflow_records <= dfm._ovf_cap, (dfm.last_overflow_records, dfm._ovf_cap)      # redone in the kernel, not by the fallback
    assert int(got[2].max()) >= 3
    diff = int((got[2] != want[2]).sum())
    assert diff <= 2, diff                                            # counts per point (the filter itself may differ on ~1e-7 of the points)
    if diff == 0:
        assert got[4] == want[4]
        for k in (0, 1, 3):                                           # cand_x, cand_src, start
            assert torch.equal(got[k], want[k]), k


def test_a_full_overflow_list_falls_back_to_k9(march):
    """more points to redo than the flagged list holds (never seen; forced here): the batch is redone through is_valid + K9."""
    from intrinsicavatar_amd import fast_snarf
    SP, rs, pts, _ = march
    dfm = rs.deformer
    want = dfm._candidates(pts, with_src=True)
    orig = fast_snarf.fuse_broyden_spec_rows
    calls = []

    def forced(x_rows, xd, vj, tfs, bones, J_inv, cnt, meta, start, oh, osc, tot, *a, **k):
        orig(x_rows, xd, vj, tfs, bones, J_inv, cnt, meta, start, oh, osc, tot, *a, **k)
        tot[1] = dfm._ovf_cap + 1
        calls.append(1)
    fast_snarf.fuse_broyden_spec_rows = forced
    try:
        got = dfm._candidates(pts, with_src=True)
    finally:
        fast_snarf.fuse_broyden_spec_rows = orig
    assert calls and got[4] == want[4]
    for k in (0, 1, 2, 3):
        assert torch.equal(got[k], want[k]), k


def test_canary_counts_nothing_on_the_product_path_and_catches_a_loosened_filter(march):
    """SNARFDeformer.spec_canary (IA_SPEC_CANARY): every k-th point of a batch searched again to the end + K9 and compared with the row
    the early-filter search left.  Product settings: 0 differ.  With the cell-tightness veto off and eps = 5e-3 (~2e-4 of
    the candidate sets differ) the canary reports it -- it is a detector, not a constant."""
    SP, rs, pts, _ = march
    dfm = rs.deformer
    old = (dfm.spec_canary, dfm.spec_eps, dfm.cell_tight)
    try:
        dfm.spec_canary = 4
        dfm.canary_totals(reset=True)
        order = torch.randperm(pts.shape[0], device=DEV).to(torch.int32)
        dfm._candidates(pts, with_src=False)
        dfm._candidates(pts, with_src=True, order=order)                     # through the permutation as well
        n, bad, ovf = dfm.canary_totals(reset=True)
        assert n >= pts.shape[0] // 2 - 2 and bad == 0 and ovf <= 1e-4 * n, (n, bad, ovf)     # ovf: points with a 4th survivor (count compared only)
        # the loosened filter changes the candidate set of only a few of these 1.1 M points (Q differs by one to three on this frame): EVERY
        # point is checked here (k = 1) -- with every 4th the test depended on which residue class the handful of bad points fell in
        dfm.spec_eps, dfm.cell_tight = 2e-2, None
        dfm.spec_canary = 1
        dfm._candidates(pts, with_src=False)
        n2, bad2, _ = dfm.canary_totals(reset=True)
        assert n2 == pts.shape[0] and 0 < bad2 < 2e-3 * n2, (n2, bad2)
    finally:
        dfm.spec_canary, dfm.spec_eps, dfm.cell_tight = old


def test_overflow_scratch_of_the_wrong_size_is_refused(march):
    """fuse_broyden_spec_rows checks the work area against ia_spec_rows_overflow_bytes(N) (it grows with N)."""
    from intrinsicavatar_amd import fast_snarf, _lib as L
    SP, rs, pts, _ = march
    dfm = rs.deformer
    P = 100_000
    sub = pts[:P].contiguous()
    i32 = lambda: torch.empty(P, dtype=torch.int32, device=DEV)      # noqa: E731
    small = torch.empty(int(L.lib().ia_spec_rows_overflow_bytes(L.i64(P))) - 1, dtype=torch.uint8, device=DEV)
    with pytest.raises(RuntimeError, match="ovf_scratch"):
        fast_snarf.fuse_broyden_spec_rows(torch.empty((P, 3, 3), device=DEV), sub[None], fast_snarf.ChannelLastVoxelJ(dfm.voxel_J_cl), dfm.tfs,
                                          dfm.init_bones, None, i32(), i32(), i32(), i32(), small, torch.empty(2, dtype=torch.int32, device=DEV),
                                          dfm.offset_kernel, dfm.scale_kernel, 1e-5, 1e-1, 1e-3)


def test_split_candidate_layout_gives_the_same_sdf_and_the_same_candidates(march):
    """SDF-only queries gather their candidates in the split layout (ia_deform_rows_pack_split: first candidates in point order, the others
    after them) because a point's second candidate lies on another body part (-6 % in the hash gather).  Same candidates, same min-SDF, bit
    for bit, with and without the spatial permutation; points with a 4th survivor (overflow records) included."""
    SP, rs, pts, _ = march
    dfm, geo = rs.deformer, rs.geometry
    order = torch.randperm(pts.shape[0], device=DEV).to(torch.int32)
    cls = type(dfm)
    try:
        cls.SPLIT_CANDIDATES = False
        want = dfm.deform_sdf(pts, geo)
        want_o = dfm.deform_sdf(pts, geo, order=order)
        a = dfm._candidates(pts, with_src=False)
        cls.SPLIT_CANDIDATES = True
        got = dfm.deform_sdf(pts, geo)
        got_o = dfm.deform_sdf(pts, geo, order=order)
        b = dfm._candidates(pts, with_src=False, split=True)
    finally:
        cls.SPLIT_CANDIDATES = True
    assert torch.equal(got, want) and torch.equal(got_o, want_o) and torch.equal(want, want_o)
    cand_a, cnt, start, Q = a[0], a[2], a[3], a[4]
    cand_b, n_first = b[0], int(b[8])
    first_pos = b[7][0] + b[7][1][torch.arange(pts.shape[0], device=DEV) // 1024]
    assert b[4] == Q and n_first == int((cnt > 0).sum()) and int(cnt.max()) >= 3
    has = cnt > 0
    assert torch.equal(cand_b[first_pos[has].long()], cand_a[start[has].long()])                      # first candidates
    more = torch.nonzero(cnt > 1)[:, 0]
    tail = n_first + (start[more] - first_pos[more]).long()
    assert torch.equal(cand_b[tail], cand_a[start[more].long() + 1])                                   # second candidates
    assert torch.equal(torch.sort(cand_b.reshape(-1))[0], torch.sort(cand_a.reshape(-1))[0])           # the same multiset


def test_small_batches_search_all_inits_side_by_side_with_the_same_candidates(march, monkeypatch):
    """batches of up to IA_BR_SMALL_MAX points (default 2^18: the reference's 4096-ray training batches) run one lane per (point, init)
    search to its end and K9 literally (broyden_items_rows_kernel + rows_flagged_kernel) instead of the early-filter kernel's 13 serial
    searches per lane: the packed candidates, their source inits, the per-point counts and the Jacobians of every candidate are the
    early-filter path's, bit for bit -- through a permutation as well."""
    SP, rs, pts, _ = march
    dfm = rs.deformer
    for n, with_order in ((1, False), (777, False), (60_000, True), (262_144, False)):
        sub = pts[1000:1000 + n].contiguous()
        order = torch.randperm(n, device=DEV).to(torch.int32) if with_order else None
        monkeypatch.setenv("IA_BR_SMALL_MAX", "0")
        want = dfm._candidates(sub, with_src=True, want_fwd=True, want_jinv=True, order=order)
        assert dfm._tls.n_over >= 0
        monkeypatch.delenv("IA_BR_SMALL_MAX")
        got = dfm._candidates(sub, with_src=True, want_fwd=True, want_jinv=True, order=order)
        assert dfm._tls.n_over == 0                                     # nothing "redone": every point went through the list by design
        assert got[4] == want[4], (n, got[4], want[4])
        for k in (0, 1, 2, 3):                                          # cand_x, cand_src, cnt, start
            assert torch.equal(got[k], want[k]), (n, k)
        src = got[1].long()
        for k in (5, 6):                                                # fwd_J, J_inv [P, I, 3, 3]: rows of the candidates
            a, b = got[k].reshape(-1, 9)[src], want[k].reshape(-1, 9)[src]
            assert torch.equal(a.view(torch.int32), b.view(torch.int32)), (n, k)
    # the SDF-only product call on a small batch = the rows of the full batch
    s_full = dfm.deform_sdf(pts[:300_000].contiguous(), rs.geometry)
    assert torch.equal(dfm.deform_sdf(pts[:40_000].contiguous(), rs.geometry), s_full[:40_000])
