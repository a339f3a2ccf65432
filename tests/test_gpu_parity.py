"""GPU parity tests proper: the HIP path, called through the C-ABI, against the CPU oracle on the same
seeded inputs and against the committed golden fixtures (tests/golden/).

Bars (BASELINE.json north_star): neighbour indices bit-exact -- tied candidate distances included: the oracle runs the
literal std::priority_queue and the kernels replay libstdc++'s heap whenever they detect a tie --, residuals / normal
equations / ESIKF state within 1e-5 RELATIVE.
"""
import os
import threading

import numpy as np
import pytest

import sr_livo_amd as srl
from sr_livo_amd import capi, synth

pytestmark = pytest.mark.gpu

RTOL = 1e-5          # the tolerance north_star states for floating point
TIGHT = 1e-9         # what we actually expect (same algorithm, FP64, different summation order only)
INT_MAX = 2**31 - 1


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    if a.size == 0:
        return 0.0
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def state16(sw):
    return np.concatenate([sw["q_pred"], sw["t_pred"], sw["vel"], np.zeros(6)])


@pytest.fixture(scope="module")
def ctx_small(golden):
    ctx = srl.Context(0)
    ctx.map_upload(golden["map_keys"], golden["map_counts"], golden["map_xyz"])
    yield ctx
    ctx.close()


def gpu_pass(ctx, raw, q, t, t_last, frame_id=100, **optkw):
    opts = srl.default_opts(**optkw)
    ctx.sweep_upload(raw)
    ctx.set_taps(1)
    neq, rc = ctx.build_residuals(capi.make_frame(q, t, t_last, frame_id=frame_id), opts)
    ids, status, ncand = ctx.fetch_neighbors(K=opts.max_number_neighbors)
    res = ctx.fetch_residuals()
    ctx.set_taps(0)
    return dict(neq=neq, rc=rc, ids=ids, status=status, ncand=ncand, **res)


def eigen_gap(ids, map_xyz):
    """(lambda_1 - lambda_0) / lambda_2 of every keypoint's neighbourhood (NaN without a full id row): where it
    vanishes the normal is decided by rounding inside the eigen-solver and no two solvers agree on it."""
    gap = np.full(len(ids), np.nan)
    flat = np.asarray(map_xyz, np.float64).reshape(-1, 3)
    for k, row in enumerate(ids):
        if row.min() < 0:
            continue
        P = flat[row]
        E = P - P.sum(0) / len(P)
        w = np.linalg.eigvalsh(E.T @ E)
        gap[k] = (w[1] - w[0]) / max(w[2], 1e-300)
    return gap


def check_pass_against(g, ref, prefix, tol=TIGHT, well_posed=None):
    """ref: golden dict with keys prefix_one_*.  well_posed (bool per keypoint, optional): keypoints whose normal is
    determined by the data; the others are compared on ids / status / a2D only (and the normal equations are skipped if
    any of them was accepted)."""
    st_ref = ref[f"{prefix}_one_status"]
    assert np.array_equal(g["status"], st_ref), "status (accepted set / cut-off) differs"
    visited = st_ref != 3
    # neighbour indices: bit-exact, tied distances included (the kernels replay libstdc++'s heap on ties)
    assert np.array_equal(g["ids"][visited], ref[f"{prefix}_one_ids"][visited]), "neighbour ids differ"
    has_plane = (st_ref == 1) | (st_ref == 2)
    acc = st_ref == 2
    assert rel(g["a2D"][has_plane], ref[f"{prefix}_one_a2D"][has_plane]) < tol, "a2D"
    ok = np.ones(len(st_ref), bool) if well_posed is None else well_posed
    for key in ("normal", "weight", "norm_offset", "distance"):
        assert rel(g[key][has_plane & ok], ref[f"{prefix}_one_{key}"][has_plane & ok]) < tol, key
    assert rel(g["jacobian"][acc & ok], ref[f"{prefix}_one_jacobian"][acc & ok]) < tol
    assert g["neq"].num_residuals == int(ref[f"{prefix}_one_num_residuals"])
    assert g["neq"].success == int(ref[f"{prefix}_one_success"])
    if not np.any(acc & ~ok):
        assert rel(np.array(g["neq"].HtH).reshape(6, 6), ref[f"{prefix}_one_HtH"]) < tol
        assert rel(np.array(g["neq"].Hth), ref[f"{prefix}_one_Hth"]) < tol
        assert rel(g["neq"].loss_sum, ref[f"{prefix}_one_loss"]) < tol


# ----------------------------------------------------------------------------- vs the reference's own translation units
@pytest.fixture(scope="module")
def gref():
    """tests/golden/golden_ref_tu.npz: outputs of /root/reference/src/optimize.cpp & co. compiled in place
    (oracle/_ref/libref_path.so, tests/golden/make_golden_ref.py) on the scenes of golden_small.npz."""
    return dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_ref_tu.npz"), allow_pickle=False))


def check_pass_against_reference_tu(g, gref, prefix, raw, tol=TIGHT, ok=None):
    """The reference returns the accepted residuals as a list in push order (plane_residuals, optimize.cpp:100-103): the
    device's accepted keypoints, in keypoint order, must be that list."""
    acc = g["status"] == 2
    if ok is None:
        ok = np.ones(int(acc.sum()), bool)
    assert int(acc.sum()) == len(gref[f"{prefix}_ref_distance"]) == g["neq"].num_residuals == int(gref[f"{prefix}_ref_num_residuals"])
    assert g["neq"].success == int(gref[f"{prefix}_ref_success"])
    assert np.array_equal(raw[acc], gref[f"{prefix}_ref_location"])            # the same keypoints, not just as many
    for key in ("normal", "weight", "norm_offset", "distance", "jacobian"):
        assert rel(g[key][acc][ok], gref[f"{prefix}_ref_{key}"][ok]) < tol, key
    if ok.all():
        assert rel(g["neq"].loss_sum, float(gref[f"{prefix}_ref_loss"])) < tol


@pytest.mark.parametrize("prefix,frame_id,max_res", [("full", 100, INT_MAX), ("cut600", 100, 600), ("init", 5, INT_MAX), ("neg1", 100, -1)])
def test_one_pass_matches_reference_tu_golden(ctx_small, golden, gref, prefix, frame_id, max_res):
    g = gpu_pass(ctx_small, golden["raw"], golden["q_pred"], golden["t_pred"], golden["t_last"], frame_id=frame_id, max_num_residuals=max_res)
    check_pass_against_reference_tu(g, gref, prefix, golden["raw"])


@pytest.mark.parametrize("prefix,frame_id,max_res", [("full", 100, INT_MAX), ("cut600", 100, 600), ("init", 5, INT_MAX), ("neg1", 100, -1)])
def test_full_solve_matches_reference_tu_golden(golden, gref, prefix, frame_id, max_res):
    """updateIEKF of the reference's src/optimize.cpp:133-314 (compiled in place) vs the device path + host algebra."""
    lio = srl.Lio(0)
    try:
        lio.ctx.map_upload(golden["map_keys"], golden["map_counts"], golden["map_xyz"])
        lio.eskf_set_state(golden[f"{prefix}_eskf_state0"])
        lio.eskf_set_cov(golden[f"{prefix}_eskf_cov0"])
        opts = srl.default_opts(max_num_residuals=max_res)
        r = lio.update_iekf(opts, golden["raw"], golden[f"{prefix}_state0"], golden["t_last"], frame_id=frame_id, log_iters=20)
        assert (r["rc"] == 0) == (int(gref[f"{prefix}_ref_solve_rc"]) == 1)
        assert r["num_residuals"] == int(gref[f"{prefix}_ref_solve_num_residuals"])
        assert rel(r["state"], gref[f"{prefix}_ref_solve_state"]) < TIGHT
        assert rel(lio.eskf_get_state(), gref[f"{prefix}_ref_solve_eskf_state"]) < TIGHT
        assert rel(lio.eskf_get_cov(), gref[f"{prefix}_ref_solve_eskf_cov"]) < 1e-8
    finally:
        lio.close()


@pytest.mark.parametrize("prefix,kw,frame_id", [("tie", {}, 100), ("tie5", dict(max_number_neighbors=5, min_number_neighbors=5), 100), ("tieinit", {}, 5)])
def test_tie_scene_neighbours_match_reference_tu_golden(golden, gref, prefix, kw, frame_id):
    """Neighbour lists on the tie scene = what the real std::priority_queue inside the reference's searchNeighbors left
    (src/optimize.cpp:394-422, compiled in place): coordinates of the device's ids, in order, bit for bit."""
    ctx = srl.Context(0)
    try:
        ctx.map_upload(golden["tie_map_keys"], golden["tie_map_counts"], golden["tie_map_xyz"])
        g = gpu_pass(ctx, golden["tie_raw"], golden["tie_q"], golden["tie_t"], golden["tie_t_last"], frame_id=frame_id, max_num_residuals=INT_MAX, **kw)
    finally:
        ctx.close()
    flat = golden["tie_map_xyz"].reshape(-1, 3)
    cnt = gref[f"{prefix}_ref_num_neighbors"]
    assert np.array_equal((g["ids"] >= 0).sum(axis=1), cnt)
    for i in range(len(cnt)):
        assert np.array_equal(flat[g["ids"][i, : cnt[i]]], gref[f"{prefix}_ref_neighbors"][i, : cnt[i]]), i
    # exact-plane lattice neighbourhoods: the normal is well determined only where the eigen-gap is (see eigen_gap)
    acc = g["status"] == 2
    gap = eigen_gap(g["ids"], golden["tie_map_xyz"])
    check_pass_against_reference_tu(g, gref, prefix, golden["tie_raw"], tol=1e-6, ok=(gap[acc] > 1e-6))


# ----------------------------------------------------------------------------- one pass vs golden
@pytest.mark.parametrize("prefix,frame_id,max_res", [("full", 100, INT_MAX), ("cut600", 100, 600), ("init", 5, INT_MAX), ("neg1", 100, -1)])
def test_one_pass_matches_golden(ctx_small, golden, prefix, frame_id, max_res):
    g = gpu_pass(ctx_small, golden["raw"], golden["q_pred"], golden["t_pred"], golden["t_last"], frame_id=frame_id, max_num_residuals=max_res)
    check_pass_against(g, golden, prefix)
    assert TIGHT < RTOL
    if prefix in ("full", "init"):
        assert g["neq"].sum_candidates == int(golden[f"{prefix}_one_sum_candidates"])
        assert g["neq"].last_visited == len(golden["raw"]) - 1
    if prefix == "cut600":
        assert g["neq"].last_visited == int(golden["cut600_one_num_visited"]) - 1


@pytest.mark.parametrize("mode", [1, 2])
def test_general_selection_paths_match(ctx_small, golden, mode):
    """select_mode=1 forces the streaming-extraction selection (overflow fallback), select_mode=2 the general
    two-pass threshold + tie-aware rank path, for every keypoint; both must equal the fast path's result."""
    g = gpu_pass(ctx_small, golden["raw"], golden["q_pred"], golden["t_pred"], golden["t_last"], max_num_residuals=INT_MAX, select_mode=mode)
    check_pass_against(g, golden, "full")
    assert g["neq"].num_fallback == len(golden["raw"])
    g0 = gpu_pass(ctx_small, golden["raw"], golden["q_pred"], golden["t_pred"], golden["t_last"], max_num_residuals=INT_MAX)
    assert g0["neq"].num_fallback == 0
    assert np.array_equal(np.array(g0["neq"].HtH), np.array(g["neq"].HtH))


def test_fused_final_reduction_equals_the_reduce_kernel(ctx_small, golden, oracle_lib, oracle_backend):
    """No ordered cut possible + no taps + one rank: the last workgroup of the association kernel finishes the sum itself.
    Same result as the separate reduce kernel up to FP64 summation order, bitwise reproducible, at 2k and 64k keypoints."""
    f = capi.make_frame(golden["q_pred"], golden["t_pred"], golden["t_last"])
    opts = srl.default_opts(max_num_residuals=INT_MAX)
    ctx_small.sweep_upload(golden["raw"])
    outs = {}
    for fused in (1, 0, 1):
        ctx_small.set_fused_reduce(fused)
        runs = [ctx_small.build_residuals(f, opts)[0] for _ in range(3)]
        for r in runs[1:]:
            assert np.array_equal(np.array(r.HtH), np.array(runs[0].HtH)) and np.array_equal(np.array(r.Hth), np.array(runs[0].Hth))
        outs.setdefault(fused, runs[0])
    ctx_small.set_fused_reduce(1)
    a, b = outs[1], outs[0]
    assert rel(np.array(a.HtH), np.array(b.HtH)) < 1e-13 and rel(np.array(a.Hth), np.array(b.Hth)) < 1e-12
    assert rel(a.loss_sum, b.loss_sum) < 1e-13
    for k in ("num_residuals", "success", "sum_candidates", "last_visited", "nan_error", "num_fallback"):
        assert getattr(a, k) == getattr(b, k), k
    assert rel(np.array(a.HtH).reshape(6, 6), golden["full_one_HtH"]) < TIGHT
    # 64k keypoints: 256 sixteen-wave workgroups, every XCD involved in the hand-off; many launches back to back
    pts, L = synth.map_candidates(31, 200_000)
    sw = synth.make_sweep(32, 65536, L)
    ctx = srl.Context(0)
    try:
        ctx.map_insert(pts)
        ctx.sweep_upload(sw["raw"])
        f2 = capi.make_frame(sw["q_pred"], sw["t_pred"], sw["t_last"])
        ctx.set_fused_reduce(0)
        ref = ctx.build_residuals(f2, opts)[0]
        ctx.set_fused_reduce(1)
        first = ctx.build_residuals(f2, opts)[0]
        for _ in range(200):
            r = ctx.build_residuals(f2, opts)[0]
            assert np.array_equal(np.array(r.HtH), np.array(first.HtH)) and r.num_residuals == ref.num_residuals
        assert rel(np.array(first.HtH), np.array(ref.HtH)) < 1e-13 and first.sum_candidates == ref.sum_candidates
    finally:
        ctx.close()


def test_prefetched_sweep_is_the_sweep_after_the_swap(ctx_small, golden):
    """srl_sweep_prefetch uploads the NEXT sweep on the copy stream while the current one is solved; after srl_sweep_swap
    every result is that of the prefetched sweep, bit for bit -- from pageable and from page-locked sources, interleaved."""
    f = capi.make_frame(golden["q_pred"], golden["t_pred"], golden["t_last"])
    opts = srl.default_opts(max_num_residuals=INT_MAX)
    A = golden["raw"]
    B = golden["raw"][::-1].copy()[:1500]                      # a different sweep of a different size
    ref = {}
    for name, sw in (("A", A), ("B", B)):
        ctx_small.sweep_upload(sw)
        ref[name] = np.array(ctx_small.build_residuals(f, opts)[0].HtH)
    assert not np.array_equal(ref["A"][:1], ref["B"][:1])
    pin = srl.PinnedArray(B.shape)
    pin.array[:] = B
    ctx_small.sweep_upload(A)
    cur = "A"
    for it in range(12):
        nxt = "B" if cur == "A" else "A"
        src = A if nxt == "A" else (pin.array if it % 2 else B)
        ctx_small.sweep_prefetch(src)                          # next sweep in flight ...
        for _ in range(3):                                     # ... while the current one is solved
            assert np.array_equal(np.array(ctx_small.build_residuals(f, opts)[0].HtH), ref[cur])
        ctx_small.sweep_swap()
        cur = nxt
        assert np.array_equal(np.array(ctx_small.build_residuals(f, opts)[0].HtH), ref[cur])
        assert ctx_small.sweep_shard()[2] == (len(A) if cur == "A" else len(B))
    with pytest.raises(srl.SrlError):
        ctx_small.sweep_swap()                                 # nothing prefetched
    pin.close()


def test_idempotent_bitwise(ctx_small, golden):
    a = gpu_pass(ctx_small, golden["raw"], golden["q_pred"], golden["t_pred"], golden["t_last"], max_num_residuals=INT_MAX)
    b = gpu_pass(ctx_small, golden["raw"], golden["q_pred"], golden["t_pred"], golden["t_last"], max_num_residuals=INT_MAX)
    assert np.array_equal(np.array(a["neq"].HtH), np.array(b["neq"].HtH))
    assert np.array_equal(np.array(a["neq"].Hth), np.array(b["neq"].Hth))
    assert np.array_equal(a["ids"], b["ids"])


# ----------------------------------------------------------------------------- full solve vs golden
@pytest.mark.parametrize("prefix,frame_id,max_res", [("full", 100, INT_MAX), ("cut600", 100, 600), ("init", 5, INT_MAX)])
def test_full_solve_matches_golden(golden, prefix, frame_id, max_res):
    lio = srl.Lio(0)
    try:
        lio.ctx.map_upload(golden["map_keys"], golden["map_counts"], golden["map_xyz"])
        lio.eskf_set_state(golden[f"{prefix}_eskf_state0"])
        lio.eskf_set_cov(golden[f"{prefix}_eskf_cov0"])
        opts = srl.default_opts(max_num_residuals=max_res)
        r = lio.update_iekf(opts, golden["raw"], golden[f"{prefix}_state0"], golden["t_last"], frame_id=frame_id, log_iters=20)
        assert r["rc"] == 0
        assert r["iters"] == int(golden[f"{prefix}_solve_rc"])            # iteration count equal
        assert r["num_residuals"] == int(golden[f"{prefix}_solve_num_residuals"])
        log_ref = golden[f"{prefix}_solve_log"]
        assert rel(r["log"][:, :42], log_ref[:, :42]) < TIGHT             # HtH, Hth per iteration
        assert rel(r["log"][:, 42:59], log_ref[:, 42:59]) < 1e-8           # d_x per iteration
        assert np.array_equal(r["log"][:, 59], log_ref[:, 59])
        assert rel(r["state"], golden[f"{prefix}_solve_state"]) < 1e-9     # p_state (q, t, v, ba, bg)
        assert rel(lio.eskf_get_state(), golden[f"{prefix}_solve_eskf_state"]) < 1e-9
        assert rel(lio.eskf_get_cov(), golden[f"{prefix}_solve_eskf_cov"]) < 1e-8
    finally:
        lio.close()


def test_not_enough_residuals_is_reported(golden):
    """max_num_residuals = -1 (class default): one keypoint visited, solve fails."""
    lio = srl.Lio(0)
    try:
        lio.ctx.map_upload(golden["map_keys"], golden["map_counts"], golden["map_xyz"])
        lio.eskf_set_state(golden["neg1_eskf_state0"]); lio.eskf_set_cov(golden["neg1_eskf_cov0"])
        r = lio.update_iekf(srl.default_opts(max_num_residuals=-1), golden["raw"], golden["neg1_state0"], golden["t_last"])
        assert r["rc"] == capi.SRL_ERR_NOT_ENOUGH_RESIDUALS
        assert r["num_residuals"] == int(golden["neg1_solve_num_residuals"])
        assert int(golden["neg1_solve_rc"]) == -1
        assert np.array_equal(r["state"], golden["neg1_state0"])          # pose untouched
    finally:
        lio.close()


# ----------------------------------------------------------------------------- live oracle comparison
@pytest.fixture(scope="module")
def scene100k(oracle_lib, oracle_backend):
    pts, L = synth.map_candidates(20250304 + 1, 100_000)
    m = oracle_lib.Map(oracle_backend)
    m.add_points(pts)
    return dict(map=m, L=L, candidates=pts)


@pytest.mark.parametrize("pattern,n,seed", [("livox", 4096, 11), ("ouster16", 3000, 12)])
def test_config1_like_pass_vs_oracle(oracle_lib, scene100k, pattern, n, seed):
    """C1: N = 4096 (and a ragged ring-indexed 3000), P = 100k: every output of one pass vs the live oracle."""
    m = scene100k["map"]
    sw = synth.make_sweep(seed, n, scene100k["L"], pattern=pattern)
    ctx = srl.Context(0)
    try:
        ctx.map_upload(*m.export())
        g = gpu_pass(ctx, sw["raw"], sw["q_pred"], sw["t_pred"], sw["t_last"], max_num_residuals=INT_MAX)
        o = m.build_plane_residuals(oracle_lib.default_opts(max_num_residuals=INT_MAX), sw["raw"], sw["q_pred"], sw["t_pred"], sw["t_last"])
        ref = {f"x_one_{k}": v for k, v in o.items() if isinstance(v, np.ndarray)}
        ref.update(x_one_num_ties=o["neq"].num_ties, x_one_num_residuals=o["neq"].num_residuals, x_one_success=o["neq"].success,
                   x_one_loss=o["neq"].loss_sum)
        check_pass_against(g, ref, "x")
        assert g["neq"].sum_candidates == o["neq"].sum_candidates
        assert np.array_equal(g["ncand"].sum(), o["neq"].sum_candidates)
    finally:
        ctx.close()


def test_edge_cases_empty_far_and_ragged(oracle_lib, scene100k):
    m = scene100k["map"]
    ctx = srl.Context(0)
    try:
        ctx.map_upload(*m.export())
        L = scene100k["L"]
        sw = synth.make_sweep(5, 70, L)          # ragged: 70 keypoints = 1 block + 6
        raw = sw["raw"].copy()
        raw[3] = [500.0, 500.0, 30.0]            # far outside the map: zero candidates
        raw[10] = [1e-9, -1e-9, 0.0]             # at the sensor origin
        g = gpu_pass(ctx, raw, sw["q_pred"], sw["t_pred"], sw["t_last"], max_num_residuals=INT_MAX)
        o = m.build_plane_residuals(oracle_lib.default_opts(max_num_residuals=INT_MAX), raw, sw["q_pred"], sw["t_pred"], sw["t_last"])
        assert np.array_equal(g["status"], o["status"])
        assert g["status"][3] == 0 and g["ncand"][3] == 0
        assert np.array_equal(g["ids"], o["ids"])
        assert rel(np.array(g["neq"].HtH).reshape(6, 6), o["HtH"]) < TIGHT
        # single keypoint
        g1 = gpu_pass(ctx, raw[:1], sw["q_pred"], sw["t_pred"], sw["t_last"], max_num_residuals=INT_MAX)
        assert g1["neq"].num_residuals in (0, 1) and g1["neq"].success == 0
    finally:
        ctx.close()


def test_empty_map_and_missing_inputs_fail_loudly():
    ctx = srl.Context(0)
    try:
        opts = srl.default_opts()
        f = capi.make_frame([1, 0, 0, 0], [0, 0, 0], [0, 0, 0])
        with pytest.raises(srl.SrlError):
            ctx.build_residuals(f, opts)                       # no map
        ctx.map_upload(np.zeros((0, 3), np.int16), np.zeros(0, np.int32), np.zeros((0, 20, 3), np.float32))
        with pytest.raises(srl.SrlError):
            ctx.build_residuals(f, opts)                       # no sweep
        ctx.sweep_upload(np.random.default_rng(0).normal(size=(100, 3)))
        neq, _ = ctx.build_residuals(f, opts)
        assert neq.num_residuals == 0 and neq.success == 0 and neq.sum_candidates == 0
        with pytest.raises(srl.SrlError):
            ctx.build_residuals(f, srl.default_opts(max_number_neighbors=40))   # unsupported K
    finally:
        ctx.close()


def test_search_neighbors_api_vs_oracle(oracle_lib, scene100k):
    m = scene100k["map"]
    ctx = srl.Context(0)
    try:
        ctx.map_upload(*m.export())
        rng = np.random.default_rng(3)
        q = np.column_stack([rng.uniform(-20, 20, 300), rng.uniform(-20, 20, 300), rng.uniform(-2.0, 3.0, 300)])
        for nb, K in ((1, 20), (2, 20), (1, 5), (2, 32)):
            ids, xyz, nf = ctx.search_neighbors(q, nb=nb, K=K)
            for i in range(len(q)):
                r = m.search_neighbors(q[i], nb=nb, K=K)
                assert nf[i] == r["n"]
                assert np.array_equal(ids[i, : r["n"]], r["ids"])
                assert np.array_equal(xyz[i, : r["n"]].astype(np.float64), r["xyz"])
    finally:
        ctx.close()


# ----------------------------------------------------------------------------- ties: the reference's heap order
@pytest.fixture(scope="module")
def ctx_tie(golden):
    ctx = srl.Context(0)
    ctx.map_upload(golden["tie_map_keys"], golden["tie_map_counts"], golden["tie_map_xyz"])
    yield ctx
    ctx.close()


PLANE_TOL = 1e-6     # exact-lattice neighbourhoods are exact planes: sigma_3 = sqrt(|lambda_0|) amplifies eps ||A|| to ~1e-8


@pytest.mark.parametrize("mode", [0, 1, 2, 5])
@pytest.mark.parametrize("prefix,K", [("tie", 20), ("tie5", 5), ("tieinit", 20)])
def test_tied_distances_follow_the_reference_heap(ctx_tie, golden, prefix, K, mode):
    """synth.lattice_scene: ~86 % of the keypoints have exactly tied candidate distances, inside the K nearest and across
    the cut.  Which tied points survive, and in which order, is libstdc++'s heap order (optimize.cpp:394-404,411-422); ids
    must equal the oracle's (literal std::priority_queue) for EVERY keypoint, through every selection path:
    0 fast path + replay on detection, 1 extraction + replay, 2 general two-pass + replay, 5 replay for all.
    "tieinit" = the same scene in init mode (frame_id < 20: r = 2, 125 voxels -- the looped fast path)."""
    assert int(golden[f"{prefix}_one_num_ties"]) > 1000
    g = gpu_pass(ctx_tie, golden["tie_raw"], golden["tie_q"], golden["tie_t"], golden["tie_t_last"], max_num_residuals=INT_MAX,
                 max_number_neighbors=K, min_number_neighbors=K, select_mode=mode, frame_id=5 if prefix == "tieinit" else 100)
    # 5 lattice points are often rotationally symmetric (lambda_0 = lambda_1): their normal is not defined by the data
    gap = eigen_gap(golden[f"{prefix}_one_ids"], golden["tie_map_xyz"])
    well = ~(gap < 1e-6)
    assert well.mean() > (0.9 if K == 20 else 0.3)
    check_pass_against(g, golden, prefix, tol=PLANE_TOL, well_posed=well)
    assert g["neq"].sum_candidates == int(golden[f"{prefix}_one_sum_candidates"])
    if mode == 0:
        # the fast path must have handed (at least) the tied keypoints to the replay, and only a minority of the rest
        assert g["neq"].num_fallback >= int(golden[f"{prefix}_one_num_ties"])


def test_forced_heap_replay_equals_fast_path_on_tie_free_data(ctx_small, golden):
    g5 = gpu_pass(ctx_small, golden["raw"], golden["q_pred"], golden["t_pred"], golden["t_last"], max_num_residuals=INT_MAX, select_mode=5)
    check_pass_against(g5, golden, "full")
    assert g5["neq"].num_fallback == len(golden["raw"])
    g5i = gpu_pass(ctx_small, golden["raw"], golden["q_pred"], golden["t_pred"], golden["t_last"], frame_id=5, max_num_residuals=INT_MAX, select_mode=5)
    check_pass_against(g5i, golden, "init")


def test_search_neighbors_api_on_tied_points(oracle_lib, oracle_backend, golden, ctx_tie):
    m = oracle_lib.Map(oracle_backend)
    m.import_(golden["tie_map_keys"], golden["tie_map_counts"], golden["tie_map_xyz"])
    q = golden["tie_raw"][:200] + golden["tie_t"]
    for mode in (0, 1, 5):
        ctx_tie.set_search_select_mode(mode)
        for nb, K in ((1, 20), (2, 20), (1, 4), (2, 32)):
            ids, xyz, nf = ctx_tie.search_neighbors(q, nb=nb, K=K)
            ties = 0
            for i in range(len(q)):
                r = m.search_neighbors(q[i], nb=nb, K=K)
                ties += int(r["tie"])
                assert nf[i] == r["n"]
                assert np.array_equal(ids[i, : r["n"]], r["ids"]), (mode, nb, K, i)
            assert ties > 100
    ctx_tie.set_search_select_mode(0)


def test_device_sqrt_is_correctly_rounded(ctx_small):
    """The tie replay compares sqrt(d2) like the reference (norm(), optimize.cpp:395): the device's sqrt must round like
    the host's (IEEE correctly rounded)."""
    rng = np.random.default_rng(8)
    x = np.concatenate([rng.uniform(0, 4, 200_000), rng.uniform(0, 1e-6, 50_000), 10.0 ** rng.uniform(-300, 300, 50_000),
                        np.nextafter(np.arange(1, 2000, dtype=np.float64) ** 2, 0), np.arange(0, 2000, dtype=np.float64) ** 2,
                        np.nextafter(np.arange(1, 2000, dtype=np.float64) ** 2, np.inf), [0.0, 5e-324, 2.2250738585072014e-308]])
    assert np.array_equal(ctx_small.device_sqrt(x), np.sqrt(x))


# ----------------------------------------------------------------------------- NaN planarity: visited keypoints only
def _nan_scene(golden):
    """the small golden map + one voxel far away holding 20 IDENTICAL points (addPointToMap could never build it; a
    legal srl_map_upload): a keypoint next to it gets a zero scatter matrix, a2D = 0/0 = NaN (optimize.cpp:343-350)."""
    keys = np.concatenate([golden["map_keys"], np.array([[300, 300, 30]], np.int16)])
    counts = np.concatenate([golden["map_counts"], np.array([20], np.int32)])
    xyz = np.concatenate([golden["map_xyz"], np.full((1, 20, 3), [300.5, 300.5, 30.5], np.float32)])
    return keys, counts, xyz


@pytest.mark.parametrize("pos,max_res,expect_nan", [(100, INT_MAX, True), (100, 600, True), (1500, 600, False), (1500, INT_MAX, True),
                                                    (2047, 2040, False), (0, -1, True), (5, -1, False)])
def test_nan_planarity_only_counts_for_visited_keypoints(oracle_lib, oracle_backend, golden, pos, max_res, expect_nan):
    """The reference throws at optimize.cpp:348-350 only for keypoints its sequential loop reaches before the break at
    :107; a degenerate neighbourhood behind the cut is never looked at."""
    keys, counts, xyz = _nan_scene(golden)
    m = oracle_lib.Map(oracle_backend)
    m.import_(keys, counts, xyz)
    raw = golden["raw"].copy()
    R = synth.quat_to_rot(golden["q_pred"] / np.linalg.norm(golden["q_pred"]))
    raw[pos] = R.T @ (np.array([300.5, 300.5, 30.45]) - golden["t_pred"])       # lands next to the degenerate voxel
    o = m.build_plane_residuals(oracle_lib.default_opts(max_num_residuals=max_res), raw, golden["q_pred"], golden["t_pred"], golden["t_last"])
    assert bool(o["neq"].nan_error) == expect_nan
    ctx = srl.Context(0)
    try:
        ctx.map_upload(keys, counts, xyz)
        for taps in (1, 0):            # taps off: the single-rank prefix pass for finite max_num_residuals
            ctx.sweep_upload(raw)
            ctx.set_taps(taps)
            neq, rc = ctx.build_residuals(capi.make_frame(golden["q_pred"], golden["t_pred"], golden["t_last"]), srl.default_opts(max_num_residuals=max_res))
            assert (rc == capi.SRL_ERR_NAN_PLANARITY) == expect_nan and bool(neq.nan_error) == expect_nan
            if not expect_nan:
                assert neq.num_residuals == o["neq"].num_residuals and neq.last_visited == o["neq"].num_visited - 1
                assert rel(np.array(neq.HtH).reshape(6, 6), o["HtH"]) < TIGHT
        ctx.set_taps(0)
    finally:
        ctx.close()


def test_collinear_neighbourhood_is_handled_like_the_reference(oracle_lib, oracle_backend, golden):
    """20 collinear neighbours (lambda_0 = lambda_1 = 0): the planarity weight vanishes, the keypoint still contributes with
    weight 0.1 exp(..) (optimize.cpp:87-88) along a normal that only rounding decides.  What IS defined must agree with the
    oracle: neighbour ids, status, a2D ~ 0, the weight, and a unit normal orthogonal to the line; nothing blows up."""
    line = np.column_stack([400.05 + 0.045 * np.arange(20), np.full(20, 400.5), np.full(20, 40.5)]).astype(np.float32)
    keys = np.concatenate([golden["map_keys"], np.array([[400, 400, 40]], np.int16)])
    counts = np.concatenate([golden["map_counts"], np.array([20], np.int32)])
    xyz = np.concatenate([golden["map_xyz"], line[None, :, :]])
    m = oracle_lib.Map(oracle_backend)
    m.import_(keys, counts, xyz)
    raw = golden["raw"].copy()
    R = synth.quat_to_rot(golden["q_pred"] / np.linalg.norm(golden["q_pred"]))
    pos = [3, 700, 1999]
    for i, p in enumerate(pos):
        raw[p] = R.T @ (np.array([400.3 + 0.2 * i, 400.5, 40.45]) - golden["t_pred"])
    o = m.build_plane_residuals(oracle_lib.default_opts(max_num_residuals=INT_MAX), raw, golden["q_pred"], golden["t_pred"], golden["t_last"])
    ctx = srl.Context(0)
    try:
        ctx.map_upload(keys, counts, xyz)
        g = gpu_pass(ctx, raw, golden["q_pred"], golden["t_pred"], golden["t_last"], max_num_residuals=INT_MAX)
        assert g["rc"] == 0 and not o["neq"].nan_error
        assert np.array_equal(g["ids"], o["ids"]) and np.array_equal(g["status"], o["status"])
        for p in pos:
            assert o["status"][p] in (1, 2)
            assert abs(g["a2D"][p]) < 1e-6 and abs(o["a2D"][p]) < 1e-6
            assert abs(g["weight"][p] - o["weight"][p]) < 1e-9 * abs(o["weight"][p]) + 1e-12
            for n in (g["normal"][p], o["normal"][p]):
                assert abs(np.linalg.norm(n) - 1.0) < 1e-9 and abs(n[0]) < 1e-5          # the line runs along x
        # every other keypoint is untouched by the extra voxel
        others = np.ones(len(raw), bool); others[pos] = False
        has = ((o["status"] == 1) | (o["status"] == 2)) & others
        assert rel(g["normal"][has], o["normal"][has]) < TIGHT and rel(g["distance"][has], o["distance"][has]) < TIGHT
    finally:
        ctx.close()


def test_nan_planarity_raises_through_the_class_surface(golden):
    keys, counts, xyz = _nan_scene(golden)
    raw = golden["raw"].copy()
    R = synth.quat_to_rot(golden["q_pred"] / np.linalg.norm(golden["q_pred"]))
    raw[7] = R.T @ (np.array([300.5, 300.5, 30.45]) - golden["t_pred"])
    lio = srl.Lio(0)
    try:
        lio.ctx.map_upload(keys, counts, xyz)
        lio.eskf_set_state(golden["full_eskf_state0"]); lio.eskf_set_cov(golden["full_eskf_cov0"])
        # the mirror rethrows std::runtime_error("error") (optimize.cpp:348-350); the C handle reports it as a status
        with pytest.raises(srl.SrlError) as ei:
            lio.update_iekf(srl.default_opts(max_num_residuals=INT_MAX), raw, golden["full_state0"], golden["t_last"])
        assert ei.value.status == capi.SRL_ERR_NAN_PLANARITY
        # behind the cut-off of the shipped max_num_residuals the same keypoint is never reached: the solve goes through
        raw2 = golden["raw"].copy()
        raw2[1900] = raw[7]
        lio.eskf_set_state(golden["cut600_eskf_state0"]); lio.eskf_set_cov(golden["cut600_eskf_cov0"])
        r = lio.update_iekf(srl.default_opts(max_num_residuals=600), raw2, golden["cut600_state0"], golden["t_last"])
        assert r["rc"] == 0 and r["num_residuals"] == 600
    finally:
        lio.close()


@pytest.mark.parametrize("n,kpw", [(4096, 0), (7000, 0), (4096, 4)])
def test_fused_ordered_cut_equals_the_two_kernel_path_and_the_oracle(oracle_lib, scene100k, n, kpw):
    """Finite max_num_residuals on sweeps of >= 2 048 keypoints: the last workgroup of the association kernel applies the
    sequential loop's cut itself (optimize.cpp:107) from published rows, acceptance masks and record granules.  Against
    the separate reduce kernel on the same launch (srl_debug_set_fused_reduce(0)) and against the oracle, for budgets that
    stop in the first workgroup, in the middle, on the very last accepted keypoint, and not at all."""
    m = scene100k["map"]
    sw = synth.make_sweep(77, n, scene100k["L"], pattern="livox")
    o_all = m.build_plane_residuals(oracle_lib.default_opts(max_num_residuals=INT_MAX), sw["raw"], sw["q_pred"], sw["t_pred"], sw["t_last"])
    total = int(o_all["neq"].num_residuals)
    ctx = srl.Context(0)
    try:
        ctx.map_upload(*m.export())
        ctx.sweep_upload(sw["raw"])
        ctx.set_launch_shape(kpw, 16 if kpw else 0)
        f = capi.make_frame(sw["q_pred"], sw["t_pred"], sw["t_last"])
        for max_res in (1, 2, 31, 32, 33, 600, 1000, total - 1, total, min(total + 1, n)):
            opts = srl.default_opts(max_num_residuals=max_res)
            ctx.set_fused_reduce(1)
            a, rca = ctx.build_residuals(f, opts)
            a2, _ = ctx.build_residuals(f, opts)
            ctx.set_fused_reduce(0)
            b, rcb = ctx.build_residuals(f, opts)
            ctx.set_fused_reduce(1)
            o = m.build_plane_residuals(oracle_lib.default_opts(max_num_residuals=max_res), sw["raw"], sw["q_pred"], sw["t_pred"], sw["t_last"])
            assert rca == rcb == 0
            for x in (a, b):
                assert x.num_residuals == o["neq"].num_residuals == min(max_res, total), max_res
                assert x.last_visited == o["neq"].num_visited - 1, max_res
                assert rel(np.array(x.HtH).reshape(6, 6), o["HtH"]) < TIGHT and rel(np.array(x.Hth), o["Hth"]) < TIGHT, max_res
            assert np.array_equal(np.array(a.HtH), np.array(a2.HtH)) and np.array_equal(np.array(a.Hth), np.array(a2.Hth))   # deterministic
            assert rel(np.array(a.HtH), np.array(b.HtH)) < 1e-13 and abs(a.loss_sum - b.loss_sum) <= 1e-13 * abs(b.loss_sum)
    finally:
        ctx.close()


# ----------------------------------------------------------------------------- max_num_residuals <= 0, empty sweeps
def test_class_default_max_num_residuals_stops_at_the_first_keypoint_with_a_plane(oracle_lib, oracle_backend, golden):
    """optimize.cpp:107 sits behind the `continue` of :78-79: with max_num_residuals = -1 the loop passes over keypoints
    that have too few neighbours and stops at the first one that has a plane."""
    m = oracle_lib.Map(oracle_backend)
    m.import_(golden["map_keys"], golden["map_counts"], golden["map_xyz"])
    raw = golden["raw"].copy()
    raw[:37] = raw[:37] + np.array([0.0, 0.0, 900.0])             # the first 37 keypoints see no map at all
    o = m.build_plane_residuals(oracle_lib.default_opts(max_num_residuals=-1), raw, golden["q_pred"], golden["t_pred"], golden["t_last"])
    assert o["neq"].num_visited == 38 and np.all(o["status"][:37] == 0)
    ctx = srl.Context(0)
    try:
        ctx.map_upload(golden["map_keys"], golden["map_counts"], golden["map_xyz"])
        g = gpu_pass(ctx, raw, golden["q_pred"], golden["t_pred"], golden["t_last"], max_num_residuals=-1)
        assert np.array_equal(g["status"], o["status"])
        assert g["neq"].last_visited == 37 and g["neq"].num_residuals == o["neq"].num_residuals == 1
        assert rel(np.array(g["neq"].HtH).reshape(6, 6), o["HtH"]) < TIGHT
        # the same across 4 logical shards: the stop keypoint lives in shard 0 here, then in shard 1
        for shift in (0, 600):
            raw2 = golden["raw"].copy()
            raw2[: 37 + shift] = raw2[: 37 + shift] + np.array([0.0, 0.0, 900.0])
            o2 = m.build_plane_residuals(oracle_lib.default_opts(max_num_residuals=-1), raw2, golden["q_pred"], golden["t_pred"], golden["t_last"])
            neq = _logical_shards_pass(golden, raw2, 4, -1)
            assert neq.last_visited == o2["neq"].num_visited - 1 == 37 + shift
            assert neq.num_residuals == o2["neq"].num_residuals
            assert rel(np.array(neq.HtH).reshape(6, 6), o2["HtH"]) < TIGHT
    finally:
        ctx.close()


def test_empty_keypoint_set_is_a_failed_solve_not_an_error(golden):
    """0 keypoints: the loop of optimize.cpp:68 does not run, :110 reports failure, process() carries on."""
    ctx = srl.Context(0)
    try:
        ctx.map_upload(golden["map_keys"], golden["map_counts"], golden["map_xyz"])
        ctx.sweep_upload(np.zeros((0, 3)))
        neq, rc = ctx.build_residuals(capi.make_frame(golden["q_pred"], golden["t_pred"], golden["t_last"]), srl.default_opts())
        assert rc == 0 and neq.num_residuals == 0 and neq.success == 0 and neq.last_visited == -1
        assert not np.any(np.array(neq.HtH)) and not np.any(np.array(neq.Hth))
    finally:
        ctx.close()
    lio = srl.Lio(0)
    try:
        lio.ctx.map_upload(golden["map_keys"], golden["map_counts"], golden["map_xyz"])
        lio.eskf_set_state(golden["full_eskf_state0"]); lio.eskf_set_cov(golden["full_eskf_cov0"])
        r = lio.update_iekf(srl.default_opts(), np.zeros((0, 3)), golden["full_state0"], golden["t_last"])
        assert r["rc"] == capi.SRL_ERR_NOT_ENOUGH_RESIDUALS and r["num_residuals"] == 0
        assert np.array_equal(r["state"], golden["full_state0"])
        # and the next sweep solves normally on the same handle
        r2 = lio.update_iekf(srl.default_opts(max_num_residuals=INT_MAX), golden["raw"], golden["full_state0"], golden["t_last"])
        assert r2["rc"] == 0 and r2["iters"] == int(golden["full_solve_rc"])
    finally:
        lio.close()


# ----------------------------------------------------------------------------- large coordinates, the truncation seam
@pytest.mark.parametrize("shift", [(20000.0, -20000.0, 150.0), (-19990.25, 19000.5, -40.0), (0.0, 0.0, 0.0)])
def test_far_from_the_origin_and_across_the_truncation_seam(oracle_lib, oracle_backend, shift):
    """+-20 km: an FP32 ulp is 2 mm there, which stresses the FP32 prefilter's margin (thr = c0 + c1 tau) -- the result is
    still the exact FP64 top-K.  shift = 0: the scene straddles the coordinate planes, where truncation toward zero makes
    voxel 0 two metres wide and insert keys (FP32 position) can differ from query keys (FP64 position)."""
    shift = np.array(shift)
    pts, L = synth.map_candidates(4242, 60_000)
    sw = synth.make_sweep(4243, 6000, L)
    m = oracle_lib.Map(oracle_backend)
    m.add_points(pts + shift)
    t_pred = sw["t_pred"] + shift
    t_last = sw["t_last"] + shift
    raw = sw["raw"].copy()
    if not shift.any():
        # keypoints ON the seam: world coordinates within +-1e-7 of the planes x = 0 / y = 0 / z = -1, and exactly on them
        R = synth.quat_to_rot(sw["q_pred"] / np.linalg.norm(sw["q_pred"]))
        rng = np.random.default_rng(1)
        for i in range(0, 600):
            pw = R @ raw[i] + t_pred
            ax = i % 3
            pw[ax] = [0.0, 0.0, -1.0][ax] + (0.0 if i % 2 else rng.uniform(-1e-7, 1e-7))
            raw[i] = R.T @ (pw - t_pred)
    ctx = srl.Context(0)
    try:
        ctx.map_upload(*m.export())
        g = gpu_pass(ctx, raw, sw["q_pred"], t_pred, t_last, max_num_residuals=INT_MAX)
        o = m.build_plane_residuals(oracle_lib.default_opts(max_num_residuals=INT_MAX), raw, sw["q_pred"], t_pred, t_last)
        ref = {f"x_one_{k}": v for k, v in o.items() if isinstance(v, np.ndarray)}
        ref.update(x_one_num_residuals=o["neq"].num_residuals, x_one_success=o["neq"].success, x_one_loss=o["neq"].loss_sum)
        # norm_offset = -n . nn0 is ~2e4 at +-20 km: compare absolutely scaled fields with the scene's scale
        check_pass_against(g, ref, "x", tol=1e-8)
        assert o["neq"].num_residuals > 4000
        assert g["neq"].sum_candidates == o["neq"].sum_candidates
        # device-side insertion builds the same map there (FP32 keys / positions at 2 mm resolution)
        ctx2 = srl.Context(0)
        try:
            ctx2.map_insert(pts + shift)
            k2, c2, x2 = ctx2.map_download()
            k1, c1, x1 = m.export()
            assert np.array_equal(k1, k2) and np.array_equal(c1, c2) and np.array_equal(x1, x2)
        finally:
            ctx2.close()
    finally:
        ctx.close()


def test_transform_points_matches_reference_formula(golden):
    ctx = srl.Context(0)
    try:
        rng = np.random.default_rng(4)
        raw = rng.normal(size=(1000, 3)) * 20
        q = np.array([0.9, 0.1, -0.3, 0.2])      # deliberately NOT normalised (utility.cpp:317 uses q as is)
        t = np.array([1.0, -2.0, 0.5])
        R_il = synth.quat_to_rot(synth.quat_from_rotvec([0.1, 0.2, -0.1])); t_il = np.array([0.05, 0.02, -0.03])
        out = ctx.transform_points(raw, q, t, R_il, t_il)
        w, x, y, z = q
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                      [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                      [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
        ref = (raw @ R_il.T + t_il) @ R.T + t
        assert rel(out, ref) < 1e-14
    finally:
        ctx.close()


# ----------------------------------------------------------------------------- map insertion (next row f1)
def test_map_insert_reproduces_the_sequential_map(oracle_lib, oracle_backend):
    """addPointsToMap on the device must give the bit-identical map (keys, counts, slot order, FP32 xyz)
    and therefore identical point ids -- in several batches, from empty, growing the table."""
    pts, L = synth.map_candidates(99, 60_000)
    m = oracle_lib.Map(oracle_backend)
    ctx = srl.Context(0)
    try:
        cuts = [0, 7, 5000, 40000, len(pts)]
        for a, b in zip(cuts[:-1], cuts[1:]):
            added_o = m.add_points(pts[a:b])
            added_g = ctx.map_insert(pts[a:b])
            assert added_g == added_o
            assert ctx.map_size() == (m.size(), m.num_voxels())
        ko, co, xo = m.export()
        kg, cg, xg = ctx.map_download()
        assert np.array_equal(kg, ko) and np.array_equal(cg, co) and np.array_equal(xg, xo)
        # re-inserting the same points adds nothing (idempotence of the min-distance rule)
        assert ctx.map_insert(pts[:20000]) == m.add_points(pts[:20000]) == 0
        # min_num_points > 0 never creates voxels (lioOptimization.cpp:437)
        far = pts[:100] + np.array([1000.0, 0, 0])
        assert ctx.map_insert(far, min_num_points=3) == m.add_points(far, min_num_points=3)
        assert ctx.map_size() == (m.size(), m.num_voxels())
        # a pass on the inserted map gives the oracle's neighbour ids
        sw = synth.make_sweep(100, 1024, L)
        g = gpu_pass(ctx, sw["raw"], sw["q_pred"], sw["t_pred"], sw["t_last"], max_num_residuals=INT_MAX)
        o = m.build_plane_residuals(oracle_lib.default_opts(max_num_residuals=INT_MAX), sw["raw"], sw["q_pred"], sw["t_pred"], sw["t_last"])
        assert np.array_equal(g["ids"], o["ids"])
    finally:
        ctx.close()


# ----------------------------------------------------------------------------- sharded path on one device
def _run_logical_shards(golden, raw, G, max_res):
    """G logical shards on one device (threads, one context each, host callbacks standing in for RCCL): same partition /
    ordered cut-off / reduction code as the multi-GPU path.  Returns the per-rank gpu_pass results."""
    barrier = threading.Barrier(G)
    lock = threading.Lock()
    box = {"sum": None, "gather": [0] * G, "n": 0}
    results = [None] * G
    errors = []

    def worker(rank):
        try:
            ctx = srl.Context(0)
            ctx.map_upload(golden["map_keys"], golden["map_counts"], golden["map_xyz"])

            def allreduce(buf):
                with lock:
                    box["sum"] = buf.copy() if box["n"] == 0 else box["sum"] + buf
                    box["n"] += 1
                barrier.wait()
                buf[:] = box["sum"]
                barrier.wait()
                if rank == 0:
                    box["n"] = 0
                barrier.wait()

            def allgather(v):
                box["gather"][rank] = v
                barrier.wait()
                out = list(box["gather"])
                barrier.wait()
                return out

            ctx.comm_set_host_callbacks(G, rank, allreduce, allgather)
            results[rank] = gpu_pass(ctx, raw, golden["q_pred"], golden["t_pred"], golden["t_last"], max_num_residuals=max_res)
            results[rank]["shard"] = ctx.sweep_shard()
            ctx.close()
        except Exception as e:  # noqa: BLE001
            errors.append(e)
            try:
                barrier.abort()
            except Exception:
                pass

    ts = [threading.Thread(target=worker, args=(r,)) for r in range(G)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errors, errors
    return results


def _logical_shards_pass(golden, raw, G, max_res):
    return _run_logical_shards(golden, raw, G, max_res)[0]["neq"]


@pytest.mark.parametrize("max_res", [INT_MAX, 600, 37, -1])
def test_logical_shards_reproduce_single_rank(golden, max_res):
    """G = 4 logical shards: the result must equal the single context result (bit-exact counts, 1e-12 on the sums:
    only the summation order differs)."""
    G = 4
    ref_ctx = srl.Context(0)
    ref_ctx.map_upload(golden["map_keys"], golden["map_counts"], golden["map_xyz"])
    ref = gpu_pass(ref_ctx, golden["raw"], golden["q_pred"], golden["t_pred"], golden["t_last"], max_num_residuals=max_res)
    ref_ctx.close()
    results = _run_logical_shards(golden, golden["raw"], G, max_res)
    status = np.concatenate([results[r]["status"] for r in range(G)])
    ids = np.concatenate([results[r]["ids"] for r in range(G)])
    assert np.array_equal(status, ref["status"])
    assert np.array_equal(ids[status != 3], ref["ids"][status != 3])
    for r in range(G):
        n = results[r]["neq"]
        assert n.num_residuals == ref["neq"].num_residuals and n.success == ref["neq"].success
        assert n.last_visited == ref["neq"].last_visited
        assert rel(np.array(n.HtH), np.array(ref["neq"].HtH)) < 1e-12
        assert rel(np.array(n.Hth), np.array(ref["neq"].Hth)) < 1e-12
        assert np.array_equal(np.array(n.HtH), np.array(results[0]["neq"].HtH))   # identical on every rank


_TWO_RANK_SCRIPT = r"""
import os, sys, time
sys.path.insert(0, sys.argv[1])
import sr_livo_amd as srl
rank, path = int(sys.argv[2]), sys.argv[3]
ctx = srl.Context(0)
if rank == 0:
    uid = srl.Context.comm_unique_id()
    open(path + ".tmp", "wb").write(bytes(uid)); os.replace(path + ".tmp", path)
else:
    t0 = time.time()
    while not os.path.exists(path):
        if time.time() - t0 > 60: raise SystemExit(3)
        time.sleep(0.05)
    uid = open(path, "rb").read()
print("backend", srl.comm_backend_info(), flush=True)
try:
    ctx.comm_init_rank(2, rank, uid)
except srl.SrlError as e:
    print("REFUSED", e.status, str(e)[:200], flush=True)
    raise SystemExit(7)
print("CONNECTED", flush=True)
"""


def test_two_processes_on_one_gpu_fail_loudly_not_hang(tmp_path):
    """Two ranks of one communicator on the SAME device (all a 1-GPU box can offer): RCCL must refuse the duplicate GPU and
    srl_comm_init_rank must hand that back as SRL_ERR_COMM in both processes within the time limit -- a mis-launched job
    (two ranks mapped to one GPU) has to die with a message, not sit in a bootstrap loop."""
    import subprocess
    import sys as _sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = str(tmp_path / "uid.bin")
    env = dict(os.environ, NCCL_DEBUG="WARN")
    procs = [subprocess.Popen([_sys.executable, "-c", _TWO_RANK_SCRIPT, root, str(r), path], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env)
             for r in range(2)]
    outs = []
    try:
        for p in procs:
            out, _ = p.communicate(timeout=150)
            outs.append((p.returncode, out.decode(errors="replace")))
    except subprocess.TimeoutExpired:
        for p in procs:
            p.kill()
        pytest.fail("two ranks on one GPU hung instead of failing: " + repr(outs))
    for rc, out in outs:
        assert rc == 7 and f"REFUSED {capi.SRL_ERR_COMM}" in out, (rc, out[-600:])
        assert "backend" in out


def test_rccl_single_rank_communicator(golden, monkeypatch):
    """The RCCL path itself -- ncclCommInitRank, ncclAllGather of the accepted counts, ncclAllReduce of the 48
    doubles on the context's stream -- with a 1-rank communicator (SRL_FORCE_COLLECTIVES runs the collectives
    even though one rank needs none).  Results must equal the communicator-free path bit for bit."""
    ctx = srl.Context(0)
    try:
        ctx.map_upload(golden["map_keys"], golden["map_counts"], golden["map_xyz"])
        for max_res in (600, INT_MAX, -1):   # -1: the count of keypoints with a plane is what gets gathered
            ctx.comm_set_host_callbacks(1, 0, lambda b: None, lambda v: [v])      # detach any communicator
            base = gpu_pass(ctx, golden["raw"], golden["q_pred"], golden["t_pred"], golden["t_last"], max_num_residuals=max_res)
            monkeypatch.setenv("SRL_FORCE_COLLECTIVES", "1")
            ctx.comm_init_rank(1, 0, srl.Context.comm_unique_id())
            g = gpu_pass(ctx, golden["raw"], golden["q_pred"], golden["t_pred"], golden["t_last"], max_num_residuals=max_res)
            monkeypatch.delenv("SRL_FORCE_COLLECTIVES")
            assert np.array_equal(np.array(g["neq"].HtH), np.array(base["neq"].HtH))
            assert np.array_equal(np.array(g["neq"].Hth), np.array(base["neq"].Hth))
            assert g["neq"].num_residuals == base["neq"].num_residuals and g["neq"].last_visited == base["neq"].last_visited
            assert np.array_equal(g["status"], base["status"])
            # park the communicator (bench.py's "replicas" leg), run unsharded, bring it back
            ctx.comm_suspend(True)
            assert ctx.sweep_shard()[1] == len(golden["raw"])
            p = gpu_pass(ctx, golden["raw"], golden["q_pred"], golden["t_pred"], golden["t_last"], max_num_residuals=max_res)
            assert np.array_equal(np.array(p["neq"].HtH), np.array(base["neq"].HtH))
            ctx.comm_suspend(False)
    finally:
        ctx.close()


def test_rccl_single_rank_fused_pass(golden, monkeypatch):
    """Sharded sweep, no ordered cut possible: every rank's pass stays FUSED (its finishing workgroup leaves the rank's result in
    a device-side mailbox), then one ncclAllReduce of 50 doubles and the publish kernel -- no reduce kernel.  1-rank
    communicator: the result must equal the communicator-free fused pass bit for bit, pass after pass."""
    ctx = srl.Context(0)
    try:
        ctx.map_upload(golden["map_keys"], golden["map_counts"], golden["map_xyz"])
        ctx.sweep_upload(golden["raw"])
        f = capi.make_frame(golden["q_pred"], golden["t_pred"], golden["t_last"])
        opts = srl.default_opts(max_num_residuals=INT_MAX)
        base, _ = ctx.build_residuals(f, opts)
        monkeypatch.setenv("SRL_FORCE_COLLECTIVES", "1")
        ctx.comm_init_rank(1, 0, srl.Context.comm_unique_id())
        monkeypatch.delenv("SRL_FORCE_COLLECTIVES")
        ctx.sweep_upload(golden["raw"])
        for fused in (1, 0, 1):
            ctx.set_fused_reduce(fused)
            for _ in range(3):
                g, _ = ctx.build_residuals(f, opts)
                if fused:
                    assert np.array_equal(np.array(g.HtH), np.array(base.HtH)) and np.array_equal(np.array(g.Hth), np.array(base.Hth))
                    assert g.loss_sum == base.loss_sum
                else:       # the reduce kernel adds the workgroup partials in another order
                    assert rel(np.array(g.HtH), np.array(base.HtH)) < 1e-12 and rel(np.array(g.Hth), np.array(base.Hth)) < 1e-12
                assert g.num_residuals == base.num_residuals and g.last_visited == base.last_visited
                assert g.sum_candidates == base.sum_candidates
    finally:
        ctx.close()


def test_profiling_modes_report_consistent_kernel_times(golden):
    """srl_set_profiling: mode 1 (four events + sync per call) and mode 2 (one lazily read event pair per association
    launch) must count the same launches, the same algorithmic bytes, and kernel times of the same size."""
    ctx = srl.Context(0)
    try:
        ctx.map_upload(golden["map_keys"], golden["map_counts"], golden["map_xyz"])
        ctx.sweep_upload(golden["raw"])
        f = capi.make_frame(golden["q_pred"], golden["t_pred"], golden["t_last"])
        opts = srl.default_opts(max_num_residuals=INT_MAX)
        out = {}
        for mode in (1, 2):
            for _ in range(3):
                ctx.build_residuals(f, opts)
            ctx.set_profiling(mode)
            for _ in range(600 if mode == 2 else 20):          # mode 2: more launches than the event ring holds
                ctx.build_residuals(f, opts)
            t = ctx.timing(); ctx.set_profiling(0)
            out[mode] = (t.calls, t.sum_assoc_ms / t.calls, t.sum_algorithmic_bytes / t.calls)
        assert out[1][0] == 20 and out[2][0] == 600
        assert out[1][2] == out[2][2] > 0
        assert 0.5 < out[1][1] / out[2][1] < 2.0 and out[2][1] > 0
    finally:
        ctx.close()


def test_prefix_pass_for_finite_max_num_residuals(oracle_lib, scene100k):
    """max_num_residuals = 600 (the shipped yaml value) on a 20k-keypoint sweep: without taps the context runs only a
    prefix of the sweep (4 x 600 + 2048 keypoints) -- the result must be the one of the full pass (same accepted set,
    same cut; sums equal up to FP64 summation order, the prefix runs with fewer keypoints per workgroup) and the oracle's;
    when the prefix cannot hold 600 accepted keypoints (its keypoints see no map) the full pass is taken."""
    m = scene100k["map"]
    sw = synth.make_sweep(77, 20_000, scene100k["L"])
    far = sw["raw"].copy(); far[:6000] += np.array([0.0, 0.0, 500.0])      # the first 6000 keypoints find no neighbours
    ctx = srl.Context(0)
    try:
        ctx.map_insert(scene100k["candidates"])
        for raw in (sw["raw"], far):
            for max_res in (600, 37):
                full = gpu_pass(ctx, raw, sw["q_pred"], sw["t_pred"], sw["t_last"], max_num_residuals=max_res)     # taps on: full pass
                opts = srl.default_opts(max_num_residuals=max_res)
                ctx.sweep_upload(raw)
                ctx.set_profiling(1)
                neq, rc = ctx.build_residuals(capi.make_frame(sw["q_pred"], sw["t_pred"], sw["t_last"]), opts)
                t = ctx.timing(); ctx.set_profiling(0)
                assert rc == 0 and neq.num_residuals == full["neq"].num_residuals == max_res
                assert neq.last_visited == full["neq"].last_visited
                assert rel(np.array(neq.HtH), np.array(full["neq"].HtH)) < 1e-13
                assert rel(np.array(neq.Hth), np.array(full["neq"].Hth)) < 1e-12 and rel(neq.loss_sum, full["neq"].loss_sum) < 1e-13
                # one prefix pass when it suffices; prefix + full pass when it does not
                pre = -(-(4 * max_res + 2048) // 64) * 64
                assert (t.calls, t.sum_keypoints) == ((1, pre) if raw is sw["raw"] else (2, pre + len(raw)))
                o = m.build_plane_residuals(oracle_lib.default_opts(max_num_residuals=max_res), raw, sw["q_pred"], sw["t_pred"], sw["t_last"], full=False)
                assert o["neq"].num_residuals == neq.num_residuals and rel(np.array(neq.HtH).reshape(6, 6), o["HtH"]) < TIGHT
    finally:
        ctx.close()


def test_bulk_map_insert_long_and_odd_segments(oracle_lib, oracle_backend):
    """Bulk batches (> 250 k points) must reproduce the sequential addPointsToMap (lioOptimization.cpp:409-445) bit for bit
    whatever the segments look like: dense voxels that fill to 20 early, thin slivers that never fill (thousands of points,
    nearly all closer than min_distance to the few stored ones), segment lengths around the replay kernel's batch size and its
    multiples (7 / 8 / 9 / 16 / 17, 48 ... 129), duplicated points, negative coordinates, a second bulk batch onto the partly
    filled map, and min_num_points > 0 on existing voxels."""
    rng = np.random.default_rng(4242)
    parts = []
    # dense room surfaces (the usual case)
    pts, _ = synth.map_candidates(4243, 150_000)
    parts.append(pts)
    # slivers: 3000 points inside a 0.12 m ball per voxel -> one stored point, everything else rejected, never full
    for c in range(40):
        centre = np.array([100.5 + c, 0.5, 0.5])
        parts.append(centre + rng.uniform(-0.06, 0.06, size=(3000, 3)))
    # exact segment lengths around the thresholds, on a coarse lattice so that most points are accepted
    for n_seg, x0 in ((48, 200), (49, 201), (64, 202), (65, 203), (128, 204), (129, 205), (7, 206), (8, 207), (9, 208), (16, 209), (17, 210), (1, 211)):
        parts.append(np.array([x0, 0, 0]) + rng.uniform(0.02, 0.98, size=(n_seg, 3)))
    # duplicated points and a voxel on negative coordinates
    dup = np.array([-300.25, -7.5, 3.25]) + np.zeros((500, 3))
    parts.append(dup)
    parts.append(np.array([-300.0, -8.0, 3.0]) + rng.uniform(0.0, 0.99, size=(5000, 3)))
    allp = np.concatenate(parts)
    allp = allp[rng.permutation(len(allp))]
    assert len(allp) > 250_000
    m = oracle_lib.Map(oracle_backend)
    ctx = srl.Context(0)
    try:
        half = len(allp) // 2
        for a, b, mnp in ((0, half, 0), (half, len(allp), 0), (0, 120_000, 3), (half, half + 130_000, 25)):
            batch = allp[a:b] + (0.07 if mnp else 0.0)                  # the min_num_points batches: shifted copies onto existing voxels
            added_o = m.add_points(batch, min_num_points=mnp)
            added_g = ctx.map_insert(batch, min_num_points=mnp)
            assert added_g == added_o, (a, b, mnp)
            assert ctx.map_size() == (m.size(), m.num_voxels())
        ko, co, xo = m.export()
        kg, cg, xg = ctx.map_download()
        assert np.array_equal(kg, ko) and np.array_equal(cg, co) and np.array_equal(xg, xo)
        assert (co == 20).sum() > 100 and (co < 5).sum() > 40
    finally:
        ctx.close()


def test_volumetric_map_all_27_voxels_occupied(oracle_lib, oracle_backend):
    """A map that fills space (not surfaces): every one of the 27 probed voxels is occupied and full, so the fast path
    runs its 9-round instance, the survivor set is large, and min-distance pruning shapes the slabs.  Ids, status and
    normal equations against the oracle, for the default K = 20 and for K = 32 (more than 32 survivors: long rank loop)."""
    rng = np.random.default_rng(4)
    pts = rng.uniform(-6.0, 6.0, (400_000, 3))                     # ~230 candidates per 1 m voxel: all voxels saturate at 20
    raw = rng.uniform(-4.0, 4.0, (6000, 3))
    q = synth.quat_from_rotvec([0.01, -0.02, 0.03]); t = np.array([0.1, -0.05, 0.02]); t_last = t - 0.01
    m = oracle_lib.Map(oracle_backend)
    m.add_points(pts)
    ctx = srl.Context(0)
    try:
        assert ctx.map_insert(pts) == m.size()
        for K in (20, 32):
            kw = dict(max_number_neighbors=K, min_number_neighbors=min(K, 20), max_num_residuals=INT_MAX)
            g = gpu_pass(ctx, raw, q, t, t_last, **kw)
            o = m.build_plane_residuals(oracle_lib.default_opts(**kw), raw, q, t, t_last)
            assert o["neq"].num_ties == 0
            assert int(g["ncand"].min()) == 27 * 20                 # every probed voxel present and full
            assert np.array_equal(g["status"], o["status"]) and np.array_equal(g["ids"], o["ids"])
            assert g["neq"].num_residuals == o["neq"].num_residuals
            assert rel(np.array(g["neq"].HtH).reshape(6, 6), o["HtH"]) < TIGHT and rel(np.array(g["neq"].Hth), o["Hth"]) < TIGHT
            assert g["neq"].num_fallback == 0 or K == 32
    finally:
        ctx.close()


# ----------------------------------------------------------------------------- class-surface forms
def test_signature_compatible_build_plane_residuals(golden):
    lio = srl.Lio(0)
    try:
        lio.ctx.map_upload(golden["map_keys"], golden["map_counts"], golden["map_xyz"])
        r = lio.build_plane_residuals(srl.default_opts(max_num_residuals=INT_MAX), golden["raw"], golden["full_state0"], golden["t_last"])
        acc = golden["full_one_status"] == 2
        assert r["success"] and len(r["rows"]) == acc.sum()
        assert rel(r["rows"][:, 3:6], golden["full_one_normal"][acc]) < TIGHT
        assert rel(r["rows"][:, 6:12], golden["full_one_jacobian"][acc]) < TIGHT
        assert rel(r["rows"][:, 13], golden["full_one_distance"][acc]) < TIGHT
        assert rel(r["loss_sum"], golden["full_one_loss"]) < TIGHT
        assert rel(r["keypoint_world"], golden["full_one_point_world"]) < 1e-15
    finally:
        lio.close()


def test_optimize_end_to_end_vs_oracle(oracle_lib, golden, oracle_backend):
    """optimize(): gridSampling (host, tr1 order) -> updateIEKF -> re-transform, against the oracle fed
    with the same keypoint selection."""
    lio = srl.Lio(0)
    try:
        lio.ctx.map_upload(golden["map_keys"], golden["map_counts"], golden["map_xyz"])
        lio.eskf_set_state(golden["full_eskf_state0"]); lio.eskf_set_cov(golden["full_eskf_cov0"])
        raw = golden["raw"]
        world = golden["full_one_point_world"]
        opts = srl.default_opts(max_num_residuals=600)
        r = lio.optimize(opts, 1.0, raw, world, golden["full_state0"], golden["t_last"])
        assert r["rc"] == 0
        kidx = r["keypoint_index"]
        assert len(kidx) >= 50 and len(np.unique(kidx)) == len(kidx)
        # oracle on the same keypoints
        m2 = oracle_lib.Map(oracle_backend)
        pts, _ = synth.map_candidates(int(golden["map_seed"]), int(golden["map_target"]))
        m2.add_points(pts)
        assert np.array_equal(m2.export()[2], golden["map_xyz"])
        e = oracle_lib.Eskf(oracle_backend)
        e.set_state(golden["full_eskf_state0"]); e.set_cov(golden["full_eskf_cov0"])
        u = oracle_lib.update_iekf(m2, e, oracle_lib.default_opts(max_num_residuals=600), raw[kidx], golden["full_state0"], golden["t_last"])
        assert u["rc"] == r["iters"]
        assert rel(r["state"], u["state"]) < 1e-9
        # re-transformed frame points (optimize.cpp:441-445)
        q = r["state"][0:4]; t = r["state"][4:7]
        assert rel(r["world"], lio.ctx.transform_points(raw, q, t)) < 1e-15
    finally:
        lio.close()


# ----------------------------------------------------------------------------- BASELINE.json configs at full size
def _full_size_case(oracle_lib, oracle_backend, n_kp, map_pts, pattern, seed, check_oracle_pass=True):
    """Build the map twice (oracle sequential insert, device insert), compare them bit for bit, then one pass
    and one full solve against the oracle, plus size-independent properties."""
    pts, L = synth.map_candidates(seed, map_pts)
    sw = synth.make_sweep(seed + 1000, n_kp, L, pattern=pattern)
    m = oracle_lib.Map(oracle_backend)
    m.add_points(pts)
    lio = srl.Lio(0)
    try:
        lio.add_points_to_map(pts)
        assert lio.map_size() == m.size()
        kg, cg, xg = lio.ctx.map_download()
        ko, co, xo = m.export()
        assert np.array_equal(kg, ko) and np.array_equal(cg, co) and np.array_equal(xg, xo)

        g = gpu_pass(lio.ctx, sw["raw"], sw["q_pred"], sw["t_pred"], sw["t_last"], max_num_residuals=INT_MAX)
        # properties that need no oracle: idempotence (bitwise), every id points at a stored point,
        # neighbours sorted by distance, checksum of per-keypoint candidates = kernel's own total
        g2 = gpu_pass(lio.ctx, sw["raw"], sw["q_pred"], sw["t_pred"], sw["t_last"], max_num_residuals=INT_MAX)
        assert np.array_equal(np.array(g["neq"].HtH), np.array(g2["neq"].HtH)) and np.array_equal(g["ids"], g2["ids"])
        assert int(g["ncand"].sum()) == g["neq"].sum_candidates
        ids = g["ids"]; ok = ids >= 0
        assert np.all(cg[ids[ok] // 20] > ids[ok] % 20)
        R = synth.quat_to_rot(sw["q_pred"] / np.linalg.norm(sw["q_pred"]))
        pw = sw["raw"] @ R.T + sw["t_pred"]
        full = ok.all(1)
        nbp = xg.reshape(-1, 3)[ids[full]].astype(np.float64)
        d = np.linalg.norm(nbp - pw[full][:, None, :], axis=2)
        assert np.all(np.diff(d, axis=1) >= 0)
        if check_oracle_pass:
            o = m.build_plane_residuals(oracle_lib.default_opts(max_num_residuals=INT_MAX), sw["raw"], sw["q_pred"], sw["t_pred"], sw["t_last"])
            ref = {f"x_one_{k}": v for k, v in o.items() if isinstance(v, np.ndarray)}
            ref.update(x_one_num_ties=o["neq"].num_ties, x_one_num_residuals=o["neq"].num_residuals, x_one_success=o["neq"].success,
                       x_one_loss=o["neq"].loss_sum)
            check_pass_against(g, ref, "x")
            assert g["neq"].sum_candidates == o["neq"].sum_candidates
        # full solve: both the throughput setting and the shipped max_num_residuals = 600
        for max_res in (INT_MAX, 600):
            e = oracle_lib.Eskf(oracle_backend)
            synth.eskf_prior(e, sw["q_pred"], sw["t_pred"], sw["vel"])
            lio.eskf_set_state(e.get_state()); lio.eskf_set_cov(e.get_cov())
            st = state16(sw)
            opts = srl.default_opts(max_num_residuals=max_res)
            r = lio.update_iekf(opts, sw["raw"], st, sw["t_last"])
            u = oracle_lib.update_iekf(m, e, oracle_lib.opts_from_product(opts), sw["raw"], st, sw["t_last"])
            assert r["rc"] == 0 and r["iters"] == u["rc"] and r["num_residuals"] == u["num_residuals"]
            assert rel(r["state"], u["state"]) < 1e-9
            assert rel(lio.eskf_get_cov(), e.get_cov()) < 1e-8
    finally:
        lio.close()


def test_config2_r3live_24k_on_1M(oracle_lib, oracle_backend):
    """BASELINE configs[1]: Livox-like sweep (24 576 keypoints), 1 M-point map, full ESIKF solve."""
    _full_size_case(oracle_lib, oracle_backend, *synth.CONFIGS["C2"])


def test_config3_ouster16_16k_on_2M(oracle_lib, oracle_backend):
    """BASELINE configs[2]: ring-indexed Ouster-16-like sweep (16 384 keypoints), 2 M-point map."""
    _full_size_case(oracle_lib, oracle_backend, *synth.CONFIGS["C3"])


def test_headline_64k_on_1M(oracle_lib, oracle_backend):
    """The metric's own configuration: 65 536-keypoint Livox-like sweep, 1 M-point map."""
    _full_size_case(oracle_lib, oracle_backend, *synth.CONFIGS["HEADLINE"])


def test_spread_sweep_off_cache_parity(oracle_lib, oracle_backend):
    """the workload whose working set is NOT cache resident -- keypoints area-uniform over the whole scene in random
    order (synth.make_spread_sweep: about one keypoint per voxel, consecutive keypoints far apart) instead of a lidar cone -- at a size the
    oracle finishes in seconds: 32 768 keypoints over the 2 M-pt map of config 3.  One pass against the oracle (neighbour ids bit-exact,
    records 1e-9), the candidate census, and a full solve whose later passes start from the neighbourhood bounds."""
    n_kp, map_pts, _pattern, seed = 32_768, 2_000_000, "livox", synth.CONFIGS["C3"][3]
    pts, L = synth.map_candidates(seed, map_pts)
    sw = synth.make_spread_sweep(seed + 77, n_kp, pts, L)
    # spread indeed: tens of thousands of distinct keypoint voxels (a 70-degree cone of this size touches a few hundred)
    R = synth.quat_to_rot(sw["q_pred"] / np.linalg.norm(sw["q_pred"]))
    kv = np.trunc(sw["raw"] @ R.T + sw["t_pred"]).astype(np.int64)
    assert len(np.unique(kv, axis=0)) > 20_000
    m = oracle_lib.Map(oracle_backend)
    m.add_points(pts)
    lio = srl.Lio(0)
    try:
        lio.add_points_to_map(pts)
        assert lio.map_size() == m.size()
        g = gpu_pass(lio.ctx, sw["raw"], sw["q_pred"], sw["t_pred"], sw["t_last"], max_num_residuals=INT_MAX)
        with oracle_lib.threads(min(os.cpu_count() or 1, 32)):
            o = m.build_plane_residuals(oracle_lib.default_opts(max_num_residuals=INT_MAX), sw["raw"], sw["q_pred"], sw["t_pred"], sw["t_last"])
        ref = {f"x_one_{k}": v for k, v in o.items() if isinstance(v, np.ndarray)}
        ref.update(x_one_num_ties=o["neq"].num_ties, x_one_num_residuals=o["neq"].num_residuals, x_one_success=o["neq"].success, x_one_loss=o["neq"].loss_sum)
        check_pass_against(g, ref, "x")
        assert g["neq"].sum_candidates == o["neq"].sum_candidates and int(g["ncand"].sum()) == g["neq"].sum_candidates
        e = oracle_lib.Eskf(oracle_backend)
        synth.eskf_prior(e, sw["q_pred"], sw["t_pred"], sw["vel"])
        lio.eskf_set_state(e.get_state()); lio.eskf_set_cov(e.get_cov())
        st = state16(sw)
        opts = srl.default_opts(max_num_residuals=INT_MAX)
        r = lio.update_iekf(opts, sw["raw"], st, sw["t_last"])
        with oracle_lib.threads(min(os.cpu_count() or 1, 32)):
            u = oracle_lib.update_iekf(m, e, oracle_lib.opts_from_product(opts), sw["raw"], st, sw["t_last"])
        assert r["rc"] == 0 and r["iters"] == u["rc"] >= 2 and r["num_residuals"] == u["num_residuals"]
        assert rel(r["state"], u["state"]) < 1e-9 and rel(lio.eskf_get_cov(), e.get_cov()) < 1e-8
    finally:
        lio.close()


def test_config4_dense_256k_on_10M_logical_shards(oracle_lib, oracle_backend):
    """BASELINE configs[3] on one device: 262 144 keypoints, 10 M-point / ~500 k-voxel map.  Map equality
    with the sequential insert, kernel properties, the oracle over the WHOLE sweep (ids / status bit-exact, records and normal
    equations to tolerance), and additivity over 8 point-range shards (the sum the exchange step performs)."""
    n_kp, map_pts, pattern, seed = synth.CONFIGS["C4"]
    pts, L = synth.map_candidates(seed, map_pts)
    sw = synth.make_sweep(seed + 1000, n_kp, L, pattern=pattern)
    m = oracle_lib.Map(oracle_backend)
    m.add_points(pts)
    ctx = srl.Context(0)
    try:
        assert ctx.map_insert(pts) == m.size()
        kg, cg, xg = ctx.map_download()
        ko, co, xo = m.export()
        assert np.array_equal(kg, ko) and np.array_equal(cg, co) and np.array_equal(xg, xo)
        assert len(cg) > 400_000 and m.size() > 9_000_000
        g = gpu_pass(ctx, sw["raw"], sw["q_pred"], sw["t_pred"], sw["t_last"], max_num_residuals=INT_MAX)
        assert int(g["ncand"].sum()) == g["neq"].sum_candidates and g["neq"].num_fallback == 0
        # the oracle over ALL 262 144 keypoints (its keypoint loop visited by up to 64 OpenMP threads, committed in keypoint
        # order: bit-identical to its single-thread run): ids and status bit-exact, records and normal equations to tolerance
        with oracle_lib.threads(min(os.cpu_count() or 1, 64)):
            o = m.build_plane_residuals(oracle_lib.default_opts(max_num_residuals=INT_MAX), sw["raw"], sw["q_pred"], sw["t_pred"], sw["t_last"])
        assert o["neq"].num_ties == 0
        assert np.array_equal(g["status"], o["status"]) and np.array_equal(g["ids"], o["ids"])
        acc = o["status"] == 2
        assert acc.sum() == g["neq"].num_residuals == o["neq"].num_residuals and g["neq"].sum_candidates == o["neq"].sum_candidates
        assert rel(g["jacobian"][acc], o["jacobian"][acc]) < TIGHT and rel(g["distance"][acc], o["distance"][acc]) < TIGHT
        assert rel(g["weight"][acc], o["weight"][acc]) < TIGHT
        assert rel(np.array(g["neq"].HtH), o["HtH"].ravel()) < 1e-10 and rel(np.array(g["neq"].Hth), o["Hth"]) < 1e-9
        # additivity over 8 contiguous shards
        HtH = np.zeros(36); Hth = np.zeros(6); nres = 0
        for r in range(8):
            b, c = srl.shard_range(n_kp, 8, r)
            gs = gpu_pass(ctx, sw["raw"][b:b + c], sw["q_pred"], sw["t_pred"], sw["t_last"], max_num_residuals=INT_MAX)
            HtH += np.array(gs["neq"].HtH); Hth += np.array(gs["neq"].Hth); nres += gs["neq"].num_residuals
            assert np.array_equal(gs["ids"], g["ids"][b:b + c])
        assert nres == g["neq"].num_residuals
        assert rel(HtH, np.array(g["neq"].HtH)) < 1e-12 and rel(Hth, np.array(g["neq"].Hth)) < 1e-11
    finally:
        ctx.close()


# ----------------------------------------------------------------------------- multi-sweep replay (path + map insertion chained)
def test_multi_sweep_replay_matches_oracle(oracle_lib, oracle_backend):
    """Six consecutive sweeps along a trajectory: IMU predict -> optimize (gridSampling + updateIEKF +
    re-transform) -> addPointsToMap, on the device-resident map, against the oracle doing the same sequence.
    Poses, covariances and the final maps must agree (the maps bit for bit)."""
    rng = np.random.default_rng(2024)
    pts, L = synth.map_candidates(555, 120_000)
    # initial map: only part of the scene, the sweeps fill in the rest
    init = pts[: len(pts) // 3]
    m = oracle_lib.Map(oracle_backend)
    m.add_points(init)
    lio = srl.Lio(0)
    try:
        lio.add_points_to_map(init)
        e = oracle_lib.Eskf(oracle_backend)
        sw0 = synth.make_sweep(600, 6000, L)
        synth.eskf_prior(e, sw0["q_gt"], sw0["t_gt"], np.zeros(3))
        lio.eskf_set_noise(0.1, 0.1, 0.0001, 0.0001)          # same IMU noise model / last IMU sample as the oracle filter
        lio.eskf_init_imu(np.array([0.0, 0.0, 9.81]), np.zeros(3))
        lio.eskf_set_state(e.get_state()); lio.eskf_set_cov(e.get_cov())
        opts_p = srl.default_opts(max_num_residuals=600)
        opts_o = oracle_lib.opts_from_product(opts_p)
        t_last = sw0["t_gt"].copy()
        acc = np.array([0.0, 0.0, 9.81]); gyr = np.zeros(3)
        for k in range(6):
            sw = synth.make_sweep(600 + k, 6000, L)          # new viewpoint each sweep
            # "IMU": a few predict steps, then teleport the prior near the sweep's true pose (both filters alike)
            for _ in range(3):
                e.predict(0.01, acc, gyr); lio.eskf_predict(0.01, acc, gyr)
            s = e.get_state(); s[0:3] = sw["t_pred"]; s[3:7] = sw["q_pred"]; s[7:10] = 0.0
            e.set_state(s); lio.eskf_set_state(s)
            st = np.concatenate([sw["q_pred"], sw["t_pred"], np.zeros(9)])
            R0 = synth.quat_to_rot(sw["q_pred"])
            world0 = sw["raw"] @ R0.T + sw["t_pred"]       # buildFrame's point field (pose prior)
            frame_id = 100 + k
            g = lio.optimize(opts_p, 1.5, sw["raw"], world0, st, t_last, frame_id=frame_id)
            assert g["rc"] == 0
            kidx = g["keypoint_index"]
            u = oracle_lib.update_iekf(m, e, opts_o, sw["raw"][kidx], st, t_last, frame_id=frame_id)
            assert u["rc"] == g["iters"] and u["num_residuals"] == g["num_residuals"]
            assert rel(g["state"], u["state"]) < 1e-9
            assert rel(lio.eskf_get_cov(), e.get_cov()) < 1e-8
            # re-transformed frame -> map (optimize.cpp:441-445, lioOptimization.cpp:1027)
            q, t = u["state"][0:4], u["state"][4:7]
            w, x, y, z = q
            Rq = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                           [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                           [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
            assert rel(g["world"], sw["raw"] @ Rq.T + t) < 1e-12
            # both maps receive the SAME world points (the product's), so the comparison stays bit-exact
            m.add_points(g["world"])
            lio.add_points_to_map(g["world"])
            assert lio.map_size() == m.size()
            t_last = u["state"][4:7].copy()
        kg, cg, xg = lio.ctx.map_download()
        ko, co, xo = m.export()
        assert np.array_equal(kg, ko) and np.array_equal(cg, co) and np.array_equal(xg, xo)
        assert m.size() > len(init) * 0.5
    finally:
        lio.close()


# ----------------------------------------------------------------------------- non-default frames and options
def test_extrinsics_unnormalised_quaternion_and_voxel_size(oracle_lib, oracle_backend):
    """Non-identity R_imu_lidar / t_imu_lidar, an un-normalised state quaternion (optimize.cpp:35 normalises for the
    association but :95/:101 do not -- SURVEY Appendix B.10), and a 0.5 m voxel map (exercises the FP64 division
    in the voxel key and denser candidate sets), K = 12 / min 10, power_planarity 3, other weights."""
    pts, L = synth.map_candidates(901, 60_000)
    sw = synth.make_sweep(902, 3000, L)
    R_il = synth.quat_to_rot(synth.quat_from_rotvec([0.03, -0.02, 0.05])); t_il = np.array([0.08, -0.03, 0.02])
    raw = (sw["raw"] - t_il) @ R_il            # so that R_il raw + t_il reproduces the sweep in the IMU frame
    q = sw["q_pred"] * 1.0007
    for voxel, cap_min_dist in ((0.5, 0.1), (1.0, 0.15)):
        m = oracle_lib.Map(oracle_backend)
        m.add_points(pts, voxel_size=voxel, min_dist=cap_min_dist)
        ctx = srl.Context(0)
        try:
            assert ctx.map_insert(pts, voxel_size=voxel, min_dist=cap_min_dist) == m.size()
            kg, cg, xg = ctx.map_download(); ko, co, xo = m.export()
            assert np.array_equal(kg, ko) and np.array_equal(cg, co) and np.array_equal(xg, xo)
            kw = dict(size_voxel_map=voxel, max_number_neighbors=12, min_number_neighbors=10, power_planarity=3.0,
                      weight_alpha=0.7, weight_neighborhood=0.3, max_dist_to_plane_icp=0.2, max_num_residuals=INT_MAX)
            opts = srl.default_opts(**kw)
            ctx.sweep_upload(raw); ctx.set_taps(1)
            f = capi.make_frame(q, sw["t_pred"], sw["t_last"], R_il=R_il, t_il=t_il)
            neq, rc = ctx.build_residuals(f, opts)
            ids, status, ncand = ctx.fetch_neighbors(K=12); res = ctx.fetch_residuals(); ctx.set_taps(0)
            o = m.build_plane_residuals(oracle_lib.default_opts(**kw), raw, q, sw["t_pred"], sw["t_last"], R_il=R_il, t_il=t_il)
            assert o["neq"].num_ties == 0
            assert np.array_equal(status, o["status"]) and np.array_equal(ids, o["ids"])
            hp = o["status"] >= 1; acc = o["status"] == 2
            assert acc.sum() > 1000
            for k in ("normal", "a2D", "weight", "norm_offset", "distance"):
                assert rel(res[k][hp], o[k][hp]) < TIGHT, k
            assert rel(res["jacobian"][acc], o["jacobian"][acc]) < TIGHT
            assert rel(np.array(neq.HtH).reshape(6, 6), o["HtH"]) < TIGHT and rel(np.array(neq.Hth), o["Hth"]) < TIGHT
        finally:
            ctx.close()


def test_scratch_table_epoch_wrap_changes_nothing():
    """The frame pipeline's scratch hash tables are never cleared per frame: entries carry a 16-bit epoch and the tables are cleared when
    it wraps (every 65 535 frames).  A context whose epoch wraps in the middle of a 7-frame stream selects the same keypoints and builds
    the same map as one far from the wrap -- through the frames before the wrap (stale entries of higher epochs), the wrapping frame
    (tables cleared, epoch 1) and the frames after it."""
    pts, L = synth.map_candidates(1401, 40_000)
    a = srl.Context(0); b = srl.Context(0)
    try:
        for c in (a, b):
            c.map_insert(pts[:15_000])
        for k, n in enumerate((8_000, 8_000, 5_000, 12_000, 8_000, 300, 9_000)):
            if k == 1:
                a.set_frame_epoch(2)                      # frames 1, 2 use the last two epochs, frame 3 wraps
            sw = synth.make_sweep(1410 + k, n, L)
            q, t = sw["q_pred"], sw["t_pred"]
            got = []
            for c in (a, b):
                c.frame_upload(sw["raw"])
                kp = c.frame_select_keypoints(q, t, 0.9)
                world, _ = c.frame_commit(sw["q_gt"], sw["t_gt"], want_added=(k % 2 == 0))
                got.append((kp, world))
            assert np.array_equal(got[0][0], got[1][0]) and np.array_equal(got[0][1], got[1][1]), f"frame {k}"
            assert a.map_size() == b.map_size(), f"frame {k}"
        ka, ca, xa = a.map_download(); kb, cb, xb = b.map_download()
        assert np.array_equal(ka, kb) and np.array_equal(ca, cb) and np.array_equal(xa, xb)
    finally:
        a.close(); b.close()


# ----------------------------------------------------------------------------- frame-resident pipeline (rows f1/f2)
def test_deferred_commit_equals_the_synchronous_one_over_a_stream_of_frames():
    """srl_frame_commit with num_added = NULL only enqueues the insertion (the map's totals are folded in by the next reader); the
    passes of the next frame are ordered behind it on the stream.  Eight frames of different sizes (so that the host exchange block and
    the scratch tables change size and the pool blocks are recycled while the previous insertion may still run): selection, normal
    equations of two passes, world points, map totals and the final map are bit-identical to the synchronous form."""
    pts, L = synth.map_candidates(1301, 60_000)
    a = srl.Context(0); b = srl.Context(0)
    pinned = []
    try:
        for c in (a, b):
            c.map_insert(pts[:20_000])
        opts = srl.default_opts(max_num_residuals=INT_MAX)
        rng = np.random.default_rng(5)
        total_added = 0
        for k, n in enumerate((9_000, 3_000, 24_000, 700, 24_000, 12_345, 1, 6_000)):
            sw = synth.make_sweep(1310 + k, n, L)
            q, t = sw["q_pred"], sw["t_pred"]
            f = capi.make_frame(q, t, sw["t_last"])
            pin = srl.PinnedArray((n, 3)); pinned.append(pin)
            out = []
            for c, deferred in ((a, False), (b, True)):
                c.frame_upload(sw["raw"])
                kp = c.frame_select_keypoints(q, t, 1.2)
                n1, _ = c.build_residuals(f, opts)
                n2, _ = c.build_residuals(f, opts)
                if deferred:
                    world, added = c.frame_commit(sw["q_gt"], sw["t_gt"], want_added=False, world_out=pin.array)
                    assert added is None
                    world = world.copy()
                else:
                    world, added = c.frame_commit(sw["q_gt"], sw["t_gt"])
                    total_added += added
                out.append((kp, np.array(n1.HtH), np.array(n2.Hth), n1.num_residuals, world))
            for x, y in zip(out[0], out[1]):
                assert np.array_equal(x, y), f"frame {k}: deferred and synchronous commits diverge"
            if k == 4:
                assert a.map_size() == b.map_size()            # a reader in the middle of the stream settles the totals
        assert a.map_size() == b.map_size() and a.map_size()[0] > 20_000 and total_added > 0
        ka, ca, xa = a.map_download(); kb, cb, xb = b.map_download()
        assert np.array_equal(ka, kb) and np.array_equal(ca, cb) and np.array_equal(xa, xb)
    finally:
        for pin in pinned:
            pin.close()
        a.close(); b.close()


def test_frame_pipeline_selects_reference_keypoints_and_commits(oracle_lib, oracle_backend):
    """srl_frame_upload -> srl_frame_select_keypoints -> srl_build_residuals -> srl_frame_commit against
    transformPoint + gridSampling + buildPlaneResiduals + addPointsToMap of the oracle: same keypoints in the
    same (tr1 iteration) order, bit-identical normal equations to the upload-the-keypoints path, bit-identical map."""
    pts, L = synth.map_candidates(1201, 80_000)
    sw = synth.make_sweep(1202, 30_000, L)
    R_il = synth.quat_to_rot(synth.quat_from_rotvec([0.02, 0.01, -0.04])); t_il = np.array([0.05, 0.02, -0.03])
    raw = (sw["raw"] - t_il) @ R_il
    q = sw["q_pred"] * 0.9996           # un-normalised: transformPoint uses q as is
    t = sw["t_pred"]
    m = oracle_lib.Map(oracle_backend)
    m.add_points(pts[: len(pts) // 2])
    ctx = srl.Context(0); ctx2 = srl.Context(0)
    try:
        for c in (ctx, ctx2):
            assert c.map_insert(pts[: len(pts) // 2]) == m.size()
        for size in (1.5, 0.4):
            world = oracle_lib.transform_points(raw, q, t, R_il, t_il, backend=oracle_backend)
            want = oracle_lib.grid_sampling(world, size, backend=oracle_backend)
            ctx.frame_upload(raw)
            got = ctx.frame_select_keypoints(q, t, size, R_il, t_il)
            assert np.array_equal(got, want), "keypoint set/order differs from gridSampling"
            assert ctx.sweep_shard() == (0, len(want), len(want))
        # the selection is the resident sweep: same normal equations as uploading those keypoints
        opts = srl.default_opts(max_num_residuals=INT_MAX)
        f = capi.make_frame(q, t, sw["t_last"], R_il=R_il, t_il=t_il)
        n1, _ = ctx.build_residuals(f, opts)
        ctx2.sweep_upload(raw[want])
        n2, _ = ctx2.build_residuals(f, opts)
        assert n1.num_residuals == n2.num_residuals > 500
        assert np.array_equal(np.array(n1.HtH), np.array(n2.HtH)) and np.array_equal(np.array(n1.Hth), np.array(n2.Hth))
        o = m.build_plane_residuals(oracle_lib.default_opts(max_num_residuals=INT_MAX), raw[want], q, t, sw["t_last"], R_il=R_il, t_il=t_il)
        assert rel(np.array(n1.HtH).reshape(6, 6), o["HtH"]) < TIGHT and n1.num_residuals == o["neq"].num_residuals
        # commit with another pose: transform + insert on the device
        q2 = sw["q_gt"]; t2 = sw["t_gt"]
        world2, added = ctx.frame_commit(q2, t2, R_il=R_il, t_il=t_il)
        want2 = oracle_lib.transform_points(raw, q2, t2, R_il, t_il, backend=oracle_backend)
        assert np.array_equal(world2, want2), "device transformPoint is not bit-identical"
        before = m.size()
        m.add_points(want2)
        assert added == m.size() - before and added > 300
        kg, cg, xg = ctx.map_download(); ko, co, xo = m.export()
        assert np.array_equal(kg, ko) and np.array_equal(cg, co) and np.array_equal(xg, xo)
        # empty frame
        ctx.frame_upload(np.zeros((0, 3)))
        assert len(ctx.frame_select_keypoints(q, t, 1.0)) == 0
        assert ctx.frame_commit(q2, t2)[1] == 0
    finally:
        ctx.close(); ctx2.close()


def test_resident_replay_matches_oracle(oracle_lib, oracle_backend):
    """The multi-sweep replay with the frame kept in HBM: optimize_resident (device keypoint selection + ESIKF)
    and commit_frame (device re-transform + addPointsToMap) against the oracle running
    transformPoint -> gridSampling -> updateIEKF -> transformPoint -> addPointsToMap."""
    pts, L = synth.map_candidates(555, 120_000)
    init = pts[: len(pts) // 3]
    m = oracle_lib.Map(oracle_backend)
    m.add_points(init)
    lio = srl.Lio(0)
    try:
        lio.add_points_to_map(init)
        e = oracle_lib.Eskf(oracle_backend)
        sw0 = synth.make_sweep(600, 6000, L)
        synth.eskf_prior(e, sw0["q_gt"], sw0["t_gt"], np.zeros(3))
        lio.eskf_set_noise(0.1, 0.1, 0.0001, 0.0001)
        lio.eskf_init_imu(np.array([0.0, 0.0, 9.81]), np.zeros(3))
        lio.eskf_set_state(e.get_state()); lio.eskf_set_cov(e.get_cov())
        opts_p = srl.default_opts(max_num_residuals=600)
        opts_o = oracle_lib.opts_from_product(opts_p)
        t_last = sw0["t_gt"].copy()
        acc = np.array([0.0, 0.0, 9.81]); gyr = np.zeros(3)
        for k in range(5):
            sw = synth.make_sweep(600 + k, 8000, L)
            for _ in range(2):
                e.predict(0.01, acc, gyr); lio.eskf_predict(0.01, acc, gyr)
            s = e.get_state(); s[0:3] = sw["t_pred"]; s[3:7] = sw["q_pred"]; s[7:10] = 0.0
            e.set_state(s); lio.eskf_set_state(s)
            st = np.concatenate([sw["q_pred"], sw["t_pred"], np.zeros(9)])
            frame_id = 100 + k
            g = lio.optimize_resident(opts_p, 1.5, sw["raw"], st, t_last, frame_id=frame_id)
            assert g["rc"] == 0
            world0 = oracle_lib.transform_points(sw["raw"], sw["q_pred"], sw["t_pred"], backend=oracle_backend)
            kidx = oracle_lib.grid_sampling(world0, 1.5, backend=oracle_backend)
            assert np.array_equal(g["keypoint_index"], kidx)
            u = oracle_lib.update_iekf(m, e, opts_o, sw["raw"][kidx], st, t_last, frame_id=frame_id)
            assert u["rc"] == g["iters"] and u["num_residuals"] == g["num_residuals"]
            assert rel(g["state"], u["state"]) < 1e-9
            assert rel(lio.eskf_get_cov(), e.get_cov()) < 1e-8
            world, added = lio.commit_frame(g["state"])
            assert np.array_equal(world, oracle_lib.transform_points(sw["raw"], g["state"][0:4], g["state"][4:7], backend=oracle_backend))
            assert rel(world, oracle_lib.transform_points(sw["raw"], u["state"][0:4], u["state"][4:7], backend=oracle_backend)) < 1e-9
            before = m.size()
            m.add_points(world)                 # same world points into both maps: the comparison stays bit-exact
            assert added == m.size() - before and lio.map_size() == m.size()
            t_last = u["state"][4:7].copy()
        kg, cg, xg = lio.ctx.map_download(); ko, co, xo = m.export()
        assert np.array_equal(kg, ko) and np.array_equal(cg, co) and np.array_equal(xg, xo)
    finally:
        lio.close()


# ----------------------------------------------------------------------------- sweep reconstruction (row f4)
def _imu_track(rng, S, t0, dt):
    st = np.zeros((S, 17))
    q = synth.quat_from_rotvec([0.1, -0.05, 0.3]); p = np.array([1.0, 2.0, 0.3]); v = np.array([1.5, -0.4, 0.1])
    for k in range(S):
        st[k, 0] = t0 + k * dt
        st[k, 1:4] = rng.normal(0, 0.5, 3); st[k, 4:7] = rng.normal(0, 0.3, 3)
        st[k, 7:10] = p; st[k, 10:14] = q; st[k, 14:17] = v
        q = synth.quat_mul(q, synth.quat_from_rotvec(st[k, 4:7] * dt)); p = p + v * dt; v = v + st[k, 1:4] * dt
    return st


UNDISTORT_TOL = 1e-12     # device sin / cos / acos vs glibc: a few ulp of FP64 on O(10 m) coordinates


@pytest.mark.parametrize("mode", [capi.MC_CONSTANT_VELOCITY, capi.MC_IMU, capi.MC_NONE])
def test_frame_undistort_matches_oracle(oracle_lib, oracle_backend, mode):
    """distortFrameByConstant / distortFrameByImu + transformAllImuPoint on the device vs the oracle, including
    points on interval boundaries, a time reversal that stops the IMU interval walk, and identical quaternions
    (slerp's linear branch)."""
    rng = np.random.default_rng(21)
    st = _imu_track(rng, 11, 200.0, 0.01)
    n = 50_000
    raw = rng.uniform(-30, 30, (n, 3)); rt = np.sort(rng.uniform(0, 100.0, n))
    rt[0] = 0.0; rt[-1] = 100.0; rt[100] = 10.0; rt[101] = 10.0 + 4e-4
    rt = np.sort(rt)
    R_il = synth.quat_to_rot(synth.quat_from_rotvec([0.02, 0.01, -0.04])); t_il = np.array([0.05, 0.02, -0.03])
    sentinel = rng.normal(0, 1, (n, 3))
    ctx = srl.Context(0)
    try:
        cases = [(rt, st)]
        rt_back = rt.copy(); rt_back[30_000] = rt_back[5]            # goes back in time: the walk stops there
        cases.append((rt_back, st))
        st_same = st.copy(); st_same[:, 10:14] = st[0, 10:14]             # q_begin == q_end
        cases.append((rt, st_same))
        for rl, s in cases:
            imu_g, raw_g = ctx.frame_undistort(raw, rl, s, 200.0, mode, R_il, t_il, imu_point_in=sentinel)
            imu_o, k = oracle_lib.distort_frame(raw, rl, s, 200.0, mode, R_il, t_il, imu_point_in=sentinel, backend=oracle_backend)
            raw_o = oracle_lib.transform_all_imu_point(imu_o, s, R_il, t_il, backend=oracle_backend)
            assert rel(imu_g, imu_o) < UNDISTORT_TOL and rel(raw_g, raw_o) < UNDISTORT_TOL
            if mode == capi.MC_IMU and rl is rt_back:
                assert k == 30_000 and np.array_equal(imu_g[k:], sentinel[k:])      # untouched behind the stop
            if mode == capi.MC_NONE:
                assert np.array_equal(imu_g, sentinel)
        # the corrected sweep is resident: take a subset as the frame, select keypoints from it
        imu_g, raw_g = ctx.frame_undistort(raw, rt, st, 200.0, mode, R_il, t_il)
        order = oracle_lib.build_frame_order(raw, 0.5, backend=oracle_backend)
        ctx.frame_take(order)
        q = synth.quat_from_rotvec([0.0, 0.0, 0.2]); t = np.array([0.3, -0.1, 0.0])
        got = ctx.frame_select_keypoints(q, t, 1.5, R_il, t_il)
        want = oracle_lib.grid_sampling(oracle_lib.transform_points(raw_g[order], q, t, R_il, t_il, backend=oracle_backend), 1.5, backend=oracle_backend)
        assert np.array_equal(got, want)
        with pytest.raises(srl.SrlError):
            ctx.frame_take(np.array([n], dtype=np.int32))
    finally:
        ctx.close()


@pytest.mark.parametrize("mc", [capi.MC_CONSTANT_VELOCITY, capi.MC_IMU])
def test_replay_driver_matches_the_reference_node_golden(gref, mc):
    """golden `run{mc}_*` (tests/golden/make_golden_ref.py): the reference's OWN node -- src/lioOptimization.cpp compiled in
    place: constructor, readParameters, imuHandler, getMeasurements, run, process, buildFrame, stateEstimation, optimize,
    addPointsToMap -- fed 40 sweeps of sensor streams.  The product's replay driver gets the same streams cut into
    measurements the way getMeasurements cuts them: frame sizes, residual-driven poses, filter and the final map."""
    from replay_reference import REPLAY_OO, REPLAY_SEQ, replay_inputs
    st, parts, gt = replay_inputs()
    oo = dict(REPLAY_OO, motion_compensation=mc)
    icp_p = srl.default_opts(max_num_residuals=REPLAY_SEQ["max_num_residuals"])
    pre = f"run{mc}"
    lio = srl.Lio(0)
    try:
        lio.set_initial_flag(False)
        lio.set_odometry_options(icp=icp_p, **oo)
        row = 0
        for i, ms in enumerate(parts):
            got = lio.run_measurement(ms["time_frame"], ms["imu_t"], ms["imu_acc"], ms["imu_gyr"], ms["pts_raw"], ms["pts_timestamp"],
                                      ms["time_sweep_begin"], ms["time_sweep_offset"])
            assert got["rc"] == 0
            if not got["processed"]:
                continue
            assert i == int(gref[f"{pre}_measurement"][row]) and got["success"]
            assert got["frame_points"] == int(gref[f"{pre}_frame_points"][row])
            assert rel(got["state"], gref[f"{pre}_state"][row]) < TIGHT
            assert rel(lio.eskf_get_state(), gref[f"{pre}_eskf_state"][row]) < TIGHT
            assert rel(lio.eskf_get_cov(), gref[f"{pre}_eskf_cov"][row]) < 1e-8
            assert lio.map_size() == int(gref[f"{pre}_map_points"][row])
            f = lio.last_frame()
            assert rel(f["raw_point"].sum(0), gref[f"{pre}_raw_sum"][row]) < 1e-11 and rel(f["point"].sum(0), gref[f"{pre}_point_sum"][row]) < TIGHT
            row += 1
        assert row == len(gref[f"{pre}_measurement"]) == 9
        k, c, x = lio.ctx.map_download()
        order = np.lexsort((k[:, 2], k[:, 1], k[:, 0]))
        assert np.array_equal(k[order], gref[f"{pre}_map_keys"]) and np.array_equal(c[order], gref[f"{pre}_map_counts"])
        assert np.array_equal(x[order], gref[f"{pre}_map_xyz"])
    finally:
        lio.set_initial_flag(False)
        lio.close()


@pytest.mark.parametrize("mc", [capi.MC_CONSTANT_VELOCITY, capi.MC_IMU])
def test_replay_driver_matches_reference_loop(oracle_lib, oracle_backend, mc):
    """The ROS-free replay driver (tryInit -> IMU propagation -> stateInitialization -> buildFrame on the device ->
    device keypoints + ESIKF -> device map insert -> sliding window) against the same loop restated on the oracle's
    pieces (tests/replay_reference.py): per-frame poses, counts, frame contents and the final maps."""
    from replay_reference import OracleReplay
    pts, L = synth.map_candidates(555, 60_000)
    meas, gt, _ = synth.make_sequence(31, 7, 24_000, L)
    oo = dict(init_voxel_size=0.2, init_sample_voxel_size=1.0, init_num_frames=6, num_for_initialization=10, voxel_size=0.2,
              sample_voxel_size=1.5, max_num_points_in_voxel=20, min_distance_points=0.1, motion_compensation=mc, initialization=0,
              point_time_enable=1, acc_cov=0.1, gyr_cov=0.1, b_acc_cov=1e-4, b_gyr_cov=1e-4)
    icp_p = srl.default_opts(max_num_residuals=600)
    ref = OracleReplay(oracle_lib, oracle_backend, oo, oracle_lib.opts_from_product(icp_p))
    lio = srl.Lio(0)
    try:
        lio.set_initial_flag(False)
        lio.set_odometry_options(icp=icp_p, **oo)
        processed = 0
        margins = []
        for i, ms in enumerate(meas):
            want = ref.run_measurement(ms)
            got = lio.run_measurement(ms["time_frame"], ms["imu_t"], ms["imu_acc"], ms["imu_gyr"], ms["pts_raw"], ms["pts_timestamp"],
                                      ms["time_sweep_begin"], ms["time_sweep_offset"])
            assert got["rc"] == 0
            assert got["processed"] == (want is not None) and got["initialized"] == ref.initial_flag
            assert got["index_frame"] == ref.index_frame
            if want is None:
                continue
            processed += 1
            assert got["success"] and want["success"]
            assert got["frame_points"] == want["frame_points"] and got["keypoints"] == want["keypoints"]
            assert got["iters"] == want["iters"] and got["num_residuals"] == want["num_residuals"]
            assert got["points_added"] == want["points_added"]
            assert rel(got["state"], want["state"]) < 1e-9
            f = lio.last_frame(); fo = ref.frames[-1]
            assert rel(f["raw_point"], fo["raw"]) < 1e-11 and rel(f["imu_point"], fo["imu_point"]) < 1e-11
            assert rel(f["point"], fo["point"]) < 1e-9
            assert rel(lio.eskf_get_cov(), ref.e.get_cov()) < 1e-8
            # The final maps are compared BIT for bit below although the device's undistortion agrees with the host's only to ~1e-12
            # (device sin / cos / acos): that holds as long as no inserted point sits closer to a decision boundary of the insertion than
            # the device / host difference -- (a) a voxel seam of the map (key = short(float(p) / 1.0), lioOptimization.cpp:403-405) and
            # (b) a rounding boundary of the FP32 position that is stored (cloudMap.cpp:7).  Asserted here, per frame, not left to luck.
            pd, ph = np.asarray(f["point"], np.float64), np.asarray(fo["point"], np.float64)
            delta = float(np.max(np.abs(pd - ph)))
            seam = float(np.min(np.abs(ph - np.round(ph))))                          # voxel size 1.0: seams at the integers
            f32 = ph.astype(np.float32)
            up = np.nextafter(f32, np.float32(np.inf)).astype(np.float64); dn = np.nextafter(f32, np.float32(-np.inf)).astype(np.float64)
            mid = np.minimum(np.abs(ph - 0.5 * (f32.astype(np.float64) + up)), np.abs(ph - 0.5 * (f32.astype(np.float64) + dn)))
            flips = int(np.count_nonzero(pd.astype(np.float32) != f32))
            margins.append((delta, seam, float(np.min(mid)), flips))
            assert seam > 100.0 * delta, (delta, seam)                               # (a): no key can flip -- the maps have the same voxels
            assert flips <= 3 and (flips == 0 or float(np.min(mid)) <= delta)        # (b): a flip needs a point within delta of a boundary
        assert processed == ref.index_frame - 1 and processed >= 9
        print("device/host point difference, distance to the nearest voxel seam, to the nearest FP32 rounding boundary, FP32 flips per frame:", margins)
        total_flips = sum(m_[3] for m_ in margins)
        # the estimate follows the motion (odometry frame = first sensor pose)
        assert np.linalg.norm(got["state"][4:7] - gt[-1][1]) < 0.05
        kg, cg, xg = lio.ctx.map_download(); ko, co, xo = ref.m.export()
        assert np.array_equal(kg, ko) and np.array_equal(cg, co)
        # stored positions: bit-identical, except where a frame point's FP32 rounding differed between device and host (counted above:
        # ~1e-6 per coordinate at a 1e-11 device / host difference -- a handful per replay at most, none in most) -- there, one ulp
        differ = xg != xo
        assert int(np.count_nonzero(differ)) <= total_flips
        if differ.any():
            assert np.all(np.abs(xg[differ].astype(np.float64) - xo[differ].astype(np.float64)) <= np.spacing(np.abs(xo[differ]))), "more than one FP32 ulp"
    finally:
        lio.set_initial_flag(False)
        lio.close()


def test_build_residuals_overlap_runs_the_callback_once_beside_the_kernel(ctx_small, golden):
    """srl_build_residuals_overlap: same result as srl_build_residuals, the callback runs exactly once per call (also when the
    finite-max_num_residuals prefix pass has to be repeated over the whole sweep), and never for an empty sweep."""
    f = capi.make_frame(golden["q_pred"], golden["t_pred"], golden["t_last"])
    ctx_small.sweep_upload(golden["raw"])
    for max_res in (INT_MAX, 600, 2000):
        opts = srl.default_opts(max_num_residuals=max_res)
        a, _ = ctx_small.build_residuals(f, opts)
        calls = []
        b, rc = ctx_small.build_residuals_overlap(f, opts, lambda: calls.append(1))
        assert rc == 0 and len(calls) == 1
        assert np.array_equal(np.array(a.HtH), np.array(b.HtH)) and np.array_equal(np.array(a.Hth), np.array(b.Hth))
        assert a.num_residuals == b.num_residuals and a.last_visited == b.last_visited
    ctx_small.sweep_upload(np.zeros((0, 3)))
    calls = []
    b, rc = ctx_small.build_residuals_overlap(f, srl.default_opts(), lambda: calls.append(1))
    assert rc == 0 and b.num_residuals == 0 and not calls
    ctx_small.sweep_upload(golden["raw"])


def test_thread_pin_to_gpu_numa_restricts_the_calling_thread_to_the_local_cpus():
    """srl_thread_pin_to_gpu_numa: the calling thread ends up on CPUs of the NUMA node the call names (or the call reports
    SRL_ERR_UNSUPPORTED and changes nothing); the affinity is restored afterwards."""
    before = os.sched_getaffinity(0)
    ctx = srl.Context(0)
    try:
        node = ctx.pin_thread_to_gpu_numa()
        after = os.sched_getaffinity(0)
        if node is None:
            assert after == before
        else:
            assert node >= 0 and after and after <= before
            local = set()
            with open(f"/sys/devices/system/node/node{node}/cpulist") as fh:
                cpulist = fh.read().strip()
            for part in cpulist.split(","):
                a, _, b = part.partition("-")
                local.update(range(int(a), int(b or a) + 1))
            assert after <= local
    finally:
        os.sched_setaffinity(0, before)
        ctx.close()


def test_map_probe_checksum_matches_a_host_walk_of_the_same_map():
    """srl_map_probe_checksum (what integration/optimize_hip.cpp compares frame by frame with the node's own map): every probed point
    contributes srl_probe_mix(voxel key, points in the voxel, last stored point); recomputed here from the downloaded map"""
    M = (1 << 64) - 1

    def mix(kx, ky, kz, count, last):
        a, b, c = (int(x) for x in np.asarray(last, np.float32).view(np.uint32))
        h = (kx & 0xFFFF) | ((ky & 0xFFFF) << 16) | ((kz & 0xFFFF) << 32) | ((count & 0xFFFFFFFF) << 48)
        h &= M
        h = ((h ^ (h >> 31)) * 0x9E3779B97F4A7C15) & M
        h ^= (a << 32) | b
        h = ((h ^ (h >> 29)) * 0xBF58476D1CE4E5B9) & M
        h ^= c
        h = ((h ^ (h >> 32)) * 0x94D049BB133111EB) & M
        return h ^ (h >> 30)

    pts, L = synth.map_candidates(99, 20_000)
    ctx = srl.Context(0)
    try:
        ctx.map_insert(pts)
        keys, counts, xyz = ctx.map_download()
        vox = {tuple(int(v) for v in k): (int(c), xyz[i, c - 1]) for i, (k, c) in enumerate(zip(keys, counts)) if c > 0}
        rng = np.random.default_rng(3)
        probe = np.concatenate([pts[rng.choice(len(pts), 3000, replace=False)] + rng.normal(0, 0.2, (3000, 3)), rng.uniform(-500, 500, (200, 3))])
        for stride in (1, 7):
            want = 0
            for p in probe[::stride]:
                k = tuple(int(np.trunc(float(np.float32(v)) / 1.0)) for v in p)
                if k in vox:
                    want = (want + mix(k[0], k[1], k[2], vox[k][0], vox[k][1])) & M
            assert ctx.map_probe_checksum(probe, stride=stride) == want, stride
        # the committed frame's world points (NULL form): nothing committed on this context -> the empty sum
        assert ctx.map_probe_checksum(None) == 0
    finally:
        ctx.close()


@pytest.mark.parametrize("max_res", [INT_MAX, 600])
def test_state_parity_where_the_plane_normal_is_undetermined(oracle_lib, oracle_backend, max_res):
    """phase 2 decomposes the 3x3 covariance in closed form where the reference runs an iterative solver, and the
    single-pass tests compare keypoints with a vanishing eigen-gap on ids / a2D only -- but they DO enter the reference's sums (the
    neighbourhood term of the weight does not vanish with a2D).  Here a fifth of the keypoints sit next to thin cables (20 collinear
    neighbours, the two small eigenvalues 1e-7 of the large one and 1e-8 apart): the single pass must agree with the oracle on status,
    a2D and normals, and the full ESIKF solve must land on the oracle's state (1e-9) with the same iteration and residual counts, with
    and without the ordered cut."""
    pts, sw = synth.cable_scene(77)
    m = oracle_lib.Map(oracle_backend)
    m.add_points(pts)
    lio = srl.Lio(0)
    try:
        lio.ctx.map_upload(*m.export())
        e = oracle_lib.Eskf(oracle_backend)
        synth.eskf_prior(e, sw["q_pred"], sw["t_pred"], sw["vel"])
        lio.eskf_set_state(e.get_state()); lio.eskf_set_cov(e.get_cov())
        st = state16(sw)
        opts = srl.default_opts(max_num_residuals=max_res)
        # how ill-posed is it?  one pass with taps: a2D of the pole keypoints must be ~ 0 and there must be many of them
        g = gpu_pass(lio.ctx, sw["raw"], sw["q_pred"], sw["t_pred"], sw["t_last"], frame_id=100, max_num_residuals=INT_MAX)
        has_plane = (g["status"] == 1) | (g["status"] == 2)
        ill = has_plane & (g["a2D"] < 0.02)
        assert ill.sum() > 500, int(ill.sum())
        o = m.build_plane_residuals(oracle_lib.default_opts(max_num_residuals=INT_MAX), sw["raw"], sw["q_pred"], sw["t_pred"], sw["t_last"])
        da = np.abs(g["a2D"] - o["a2D"])[has_plane]
        assert np.array_equal(g["ids"], o["ids"]), "neighbour ids differ"
        # (round 5: with the closed form alone a2D came out as 0 here where the iterative solver finds 5e-5 -- the two small eigenvalues,
        # 4e-8 against 0.75, are a near-double root it cannot separate -- and the STATE was off by 1.5e-4 / 1.4e-3: such a keypoint keeps
        # the weight lambda_neighborhood * exp(...) (optimize.cpp:87-88) with an arbitrary normal.  Phase 2 now runs the Jacobi sweeps where
        # e1 - e0 < 1e-3 e2.)
        assert da.max() < 1e-9, (float(da.max()), int((da > 1e-9).sum()))
        acc = (g["status"] == 2) & (o["status"] == 2)
        assert np.array_equal(g["status"], o["status"])
        assert rel(g["normal"][acc], o["normal"][acc]) < 1e-6, rel(g["normal"][acc], o["normal"][acc])      # (the problem's own conditioning: gap 1e-8)
        r = lio.update_iekf(opts, sw["raw"], st, sw["t_last"], frame_id=100)
        u = oracle_lib.update_iekf(m, e, oracle_lib.opts_from_product(opts), sw["raw"], st, sw["t_last"], frame_id=100)
        assert r["rc"] == 0 and u["rc"] > 0
        assert r["iters"] == u["rc"], (r["iters"], u["rc"])
        assert rel(r["state"], u["state"]) < 1e-9, rel(r["state"], u["state"])
        assert rel(lio.eskf_get_cov(), e.get_cov()) < 1e-8
        assert r["num_residuals"] == u["num_residuals"]
    finally:
        lio.close()
