"""One launch per solve: srl_solve_iekf keeps the whole loop of updateIEKF (src/optimize.cpp:133-314) inside one persistent
kernel -- every buildPlaneResiduals pass, the 17-dim update on one wave of the finishing workgroup, step guard, convergence
rule, posterior covariance.  These tests run it on the GPU against (a) the per-iteration path (one srl_build_residuals call
per pass + the host mirror's algebra: the reference form), (b) the goldens produced by the reference's own translation units
and (c) the oracle; both forms of the second inverse (17 x 17 LU in the host's order / Schur complement) are exercised.
"""
import numpy as np
import pytest

import sr_livo_amd as srl
from sr_livo_amd import capi, synth

pytestmark = pytest.mark.gpu

INT_MAX = 2**31 - 1
TIGHT = 1e-9


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    if a.size == 0:
        return 0.0
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def _solve(lio, golden, prefix, frame_id, max_res, raw=None, persistent=True, exact_lu=False, log_iters=20, opts_kw=None,
           allow=(capi.SRL_ERR_NOT_ENOUGH_RESIDUALS,), state0=None):
    lio.set_persistent_solve(persistent)
    lio.ctx.set_iekf_exact_lu(exact_lu)
    lio.eskf_set_state(golden[f"{prefix}_eskf_state0"])
    lio.eskf_set_cov(golden[f"{prefix}_eskf_cov0"])
    opts = srl.default_opts(max_num_residuals=max_res, **(opts_kw or {}))
    r = lio.update_iekf(opts, golden["raw"] if raw is None else raw, golden[f"{prefix}_state0"] if state0 is None else state0,
                        golden["t_last"], frame_id=frame_id, log_iters=log_iters, allow=allow)
    r["eskf_state"] = lio.eskf_get_state()
    r["eskf_cov"] = lio.eskf_get_cov()
    r["launches"] = lio.last_solve_launches()
    return r


@pytest.fixture(scope="module")
def lio_small(golden):
    lio = srl.Lio(0)
    lio.ctx.map_upload(golden["map_keys"], golden["map_counts"], golden["map_xyz"])
    yield lio
    lio.close()


@pytest.fixture(scope="module")
def gref():
    import os
    path = os.path.join(os.path.dirname(__file__), "golden", "golden_ref_tu.npz")
    return dict(np.load(path, allow_pickle=False))


@pytest.mark.parametrize("exact_lu", [True, False])
@pytest.mark.parametrize("prefix,frame_id,max_res", [("full", 100, INT_MAX), ("cut600", 100, 600), ("init", 5, INT_MAX)])
def test_persistent_solve_equals_the_per_iteration_path(lio_small, golden, prefix, frame_id, max_res, exact_lu):
    ref = _solve(lio_small, golden, prefix, frame_id, max_res, persistent=False)
    got = _solve(lio_small, golden, prefix, frame_id, max_res, persistent=True, exact_lu=exact_lu)
    assert ref["rc"] == 0 and got["rc"] == 0
    assert got["launches"] == 1 and ref["launches"] == ref["iters"] >= 2           # one kernel for the whole solve
    assert got["iters"] == ref["iters"] and got["num_residuals"] == ref["num_residuals"]
    # per pass: the same normal equations (the same association on the same poses) ...
    assert rel(got["log"][:, :42], ref["log"][:, :42]) < (1e-12 if exact_lu else 1e-9)
    assert np.array_equal(got["log"][:, 59], ref["log"][:, 59])
    # ... and the same step.  LU form: the host's operations in the host's order (sin / cos / acos of the device library
    # apart); Schur form: the same matrix by another route
    tol = 1e-12 if exact_lu else 1e-9
    assert rel(got["log"][:, 42:59], ref["log"][:, 42:59]) < tol
    assert rel(got["state"], ref["state"]) < tol
    assert rel(got["eskf_state"], ref["eskf_state"]) < tol
    assert rel(got["eskf_cov"], ref["eskf_cov"]) < (1e-11 if exact_lu else 1e-8)


@pytest.mark.parametrize("prefix,frame_id,max_res", [("full", 100, INT_MAX), ("cut600", 100, 600), ("init", 5, INT_MAX)])
def test_persistent_solve_matches_the_reference_tu_and_oracle_goldens(lio_small, golden, gref, prefix, frame_id, max_res):
    r = _solve(lio_small, golden, prefix, frame_id, max_res)
    assert r["rc"] == 0 and r["launches"] == 1
    # the reference's own updateIEKF (src/optimize.cpp compiled in place)
    assert int(gref[f"{prefix}_ref_solve_rc"]) == 1
    assert r["num_residuals"] == int(gref[f"{prefix}_ref_solve_num_residuals"])
    assert rel(r["state"], gref[f"{prefix}_ref_solve_state"]) < TIGHT
    assert rel(r["eskf_state"], gref[f"{prefix}_ref_solve_eskf_state"]) < TIGHT
    assert rel(r["eskf_cov"], gref[f"{prefix}_ref_solve_eskf_cov"]) < 1e-8
    # the oracle's per-iteration log
    assert r["iters"] == int(golden[f"{prefix}_solve_rc"])
    log_ref = golden[f"{prefix}_solve_log"]
    assert rel(r["log"][:, :42], log_ref[:, :42]) < TIGHT and rel(r["log"][:, 42:59], log_ref[:, 42:59]) < 1e-8
    assert np.array_equal(r["log"][:, 59], log_ref[:, 59])


def test_convergence_rule_is_off_for_the_first_two_frames(lio_small, golden):
    """frame_id <= 1: no convergence test (optimize.cpp:265) and init mode -> max(15, num_iters_icp) + 1 = 16 passes."""
    ref = _solve(lio_small, golden, "init", 1, INT_MAX, persistent=False)
    got = _solve(lio_small, golden, "init", 1, INT_MAX, persistent=True)
    assert ref["rc"] == 0 and got["rc"] == 0 and ref["iters"] == 16 and got["iters"] == 16 and got["launches"] == 1
    assert rel(got["eskf_state"], ref["eskf_state"]) < 1e-9 and rel(got["eskf_cov"], ref["eskf_cov"]) < 1e-8


def test_raw_abi_call_and_its_refusals(golden):
    ctx = srl.Context(0)
    try:
        ctx.map_upload(golden["map_keys"], golden["map_counts"], golden["map_xyz"])
        ctx.sweep_upload(golden["raw"])
        st0 = golden["full_state0"]
        frame = capi.make_frame(st0[0:4], st0[4:7], golden["t_last"], frame_id=100)
        r = ctx.solve_iekf(frame, srl.default_opts(max_num_residuals=INT_MAX), 0.001, golden["full_eskf_state0"], golden["full_eskf_cov0"], log_iters=8)
        assert r["rc"] == 0 and r["verdict"] == capi.IEKF_DONE and r["covariance_updated"] == 1 and r["observed"] == r["iterations"]
        assert rel(r["state"], golden["full_solve_eskf_state"]) < TIGHT and rel(r["cov"], golden["full_solve_eskf_cov"]) < 1e-8
        # what the kernel does not cover comes back untouched
        for kw in (dict(max_num_residuals=-1), dict(max_num_residuals=0), dict(select_mode=2)):
            r = ctx.solve_iekf(frame, srl.default_opts(**kw), 0.001, golden["full_eskf_state0"], golden["full_eskf_cov0"])
            assert r["rc"] == capi.SRL_ERR_RETRY_PER_ITERATION and np.array_equal(r["state"], golden["full_eskf_state0"])
        ctx.set_taps(1)
        r = ctx.solve_iekf(frame, srl.default_opts(), 0.001, golden["full_eskf_state0"], golden["full_eskf_cov0"])
        assert r["rc"] == capi.SRL_ERR_RETRY_PER_ITERATION
        ctx.set_taps(0)
    finally:
        ctx.close()


def test_failed_solves_nan_planarity_and_the_short_prefix(golden):
    from test_gpu_parity import _nan_scene
    keys, counts, xyz = _nan_scene(golden)
    R = synth.quat_to_rot(golden["q_pred"] / np.linalg.norm(golden["q_pred"]))
    lio = srl.Lio(0)
    try:
        lio.ctx.map_upload(keys, counts, xyz)
        # (a) no keypoint sees the map: zero residuals < min_number_neighbors, summary.success = false on the first pass,
        # pose and filter untouched (optimize.cpp:110-123) -- decided by the kernel, not by a fallback
        far = golden["raw"] + np.array([0.0, 0.0, 900.0])
        r = _solve(lio, golden, "full", 100, INT_MAX, raw=far)
        assert r["rc"] == capi.SRL_ERR_NOT_ENOUGH_RESIDUALS and r["launches"] == 1 and r["num_residuals"] == 0
        assert np.array_equal(r["state"], golden["full_state0"]) and np.array_equal(r["eskf_state"], golden["full_eskf_state0"])
        # (b) NaN planarity among the visited keypoints: std::runtime_error("error") (optimize.cpp:348-350)
        raw = golden["raw"].copy()
        raw[7] = R.T @ (np.array([300.5, 300.5, 30.45]) - golden["t_pred"])
        lio.set_persistent_solve(True)
        lio.eskf_set_state(golden["full_eskf_state0"]); lio.eskf_set_cov(golden["full_eskf_cov0"])
        with pytest.raises(srl.SrlError) as ei:
            lio.update_iekf(srl.default_opts(max_num_residuals=INT_MAX), raw, golden["full_state0"], golden["t_last"])
        assert ei.value.status == capi.SRL_ERR_NAN_PLANARITY
        # ... behind the cut of the shipped max_num_residuals it is never reached
        raw2 = golden["raw"].copy()
        raw2[1900] = raw[7]
        r = _solve(lio, golden, "cut600", 100, 600, raw=raw2)
        assert r["rc"] == 0 and r["num_residuals"] == 600 and r["launches"] == 1
        # (c) finite max_num_residuals with a keypoint prefix that cannot hold it: the kernel says so, the host mirror repeats
        # the solve per iteration (whole shard) -- same result as never trying
    finally:
        lio.close()


def test_short_prefix_falls_back_to_the_per_iteration_path(oracle_lib, oracle_backend):
    pts, L = synth.map_candidates(101, 100_000)
    sw = synth.make_sweep(77, 20_000, L)
    far = sw["raw"].copy(); far[:6000] += np.array([0.0, 0.0, 500.0])      # the first 6000 keypoints find no neighbours
    lio = srl.Lio(0)
    try:
        lio.ctx.map_insert(pts)
        out = {}
        for persistent in (False, True):
            class A:
                def __init__(s, l): s.l = l
                def set_noise(s, *a): s.l.eskf_set_noise(*a)
                def scale_init_cov(s): s.l.eskf_scale_init_cov()
                def init_imu(s, a, g): s.l.eskf_init_imu(a, g)
                def predict(s, dt, a, g): s.l.eskf_predict(dt, a, g)
                def get_state(s): return s.l.eskf_get_state()
                def set_state(s, x): s.l.eskf_set_state(x)
            lio.eskf_set_state(np.r_[np.zeros(3), 1.0, np.zeros(12), 0, 0, 9.81]); lio.eskf_set_cov(np.eye(17))
            synth.eskf_prior(A(lio), sw["q_pred"], sw["t_pred"], sw["vel"])
            lio.set_persistent_solve(persistent)
            st = np.concatenate([sw["q_pred"], sw["t_pred"], sw["vel"], np.zeros(6)])
            r = lio.update_iekf(srl.default_opts(max_num_residuals=600), far, st, sw["t_last"], frame_id=100, log_iters=20)
            out[persistent] = (r, lio.eskf_get_state(), lio.eskf_get_cov(), lio.last_solve_launches())
        (r0, s0, P0, l0), (r1, s1, P1, l1) = out[False], out[True]
        assert r0["rc"] == 0 and r1["rc"] == 0 and r0["iters"] == r1["iters"] and r1["num_residuals"] == 600
        assert l1 == l0                                   # the persistent attempt was abandoned: one launch per pass (x 2: prefix + whole shard) inside srl_build_residuals
        assert np.array_equal(s0, s1) and np.array_equal(P0, P1) and np.array_equal(r0["log"], r1["log"])
    finally:
        lio.close()


def test_step_guard_on_the_device(golden):
    """optimize.cpp:248-251: a step of more than 100 m is skipped -- every pass, here: a 200 m voxel grid puts keypoints
    150 m above a plane of map points into the same voxel; the signed gate (optimize.cpp:98) accepts the -150 m residuals
    and the update asks for a 150 m jump.  The loop runs out without touching the filter and without a covariance update."""
    rng = np.random.default_rng(3)
    plane = np.column_stack([rng.uniform(20.0, 180.0, 20), rng.uniform(20.0, 180.0, 20), rng.normal(10.0, 0.01, 20)]).astype(np.float32)
    keys = np.array([[0, 0, 0]], np.int16); counts = np.array([20], np.int32)
    raw = np.column_stack([rng.uniform(40.0, 160.0, 512), rng.uniform(40.0, 160.0, 512), rng.uniform(159.0, 161.0, 512)])
    st0 = np.r_[1.0, 0, 0, 0, np.zeros(3), np.zeros(9)]
    es0 = np.r_[np.zeros(3), 1.0, np.zeros(12), 0, 0, 9.81]
    opts_kw = dict(size_voxel_map=200.0, num_iters_icp=4)
    out = {}
    lio = srl.Lio(0)
    try:
        lio.ctx.map_upload(keys, counts, plane[None, :, :])
        for persistent in (False, True):
            lio.set_persistent_solve(persistent)
            lio.eskf_set_state(es0); lio.eskf_set_cov(np.eye(17))
            r = lio.update_iekf(srl.default_opts(max_num_residuals=INT_MAX, **opts_kw), raw, st0, np.array([0.0, 0.0, 0.5]), frame_id=100, log_iters=10)
            out[persistent] = (r, lio.eskf_get_state(), lio.eskf_get_cov(), lio.last_solve_launches())
        (r0, s0, P0, l0), (r1, s1, P1, l1) = out[False], out[True]
        assert r0["rc"] == 0 and r1["rc"] == 0 and r0["iters"] == 5 and r1["iters"] == 5 and l1 == 1 and l0 == 5
        assert r0["num_residuals"] == 512 and r1["num_residuals"] == 512
        assert np.linalg.norm(r0["log"][0, 42:45]) > 100.0 and rel(r1["log"][:, 42:59], r0["log"][:, 42:59]) < 1e-9
        for s, P, r in ((s0, P0, r0), (s1, P1, r1)):
            assert np.array_equal(s, es0) and np.array_equal(P, np.eye(17)) and np.array_equal(r["state"], st0)
    finally:
        lio.close()


def test_more_tiles_than_compute_units(oracle_lib, oracle_backend):
    """A sweep of more than 256 x 256 keypoints: every workgroup of the persistent kernel walks several tiles per pass."""
    pts, L = synth.map_candidates(202, 60_000)
    sw = synth.make_sweep(203, 70_000, L)
    m = oracle_lib.Map(oracle_backend)
    m.add_points(pts)
    e = oracle_lib.Eskf()
    synth.eskf_prior(e, sw["q_pred"], sw["t_pred"], sw["vel"])
    st = np.concatenate([sw["q_pred"], sw["t_pred"], sw["vel"], np.zeros(6)])
    lio = srl.Lio(0)
    try:
        lio.ctx.map_insert(pts)
        res = {}
        for persistent in (False, True):
            lio.set_persistent_solve(persistent)
            lio.eskf_set_state(e.get_state()); lio.eskf_set_cov(e.get_cov())
            r = lio.update_iekf(srl.default_opts(max_num_residuals=INT_MAX), sw["raw"], st, sw["t_last"], frame_id=100, log_iters=20)
            res[persistent] = (r, lio.eskf_get_state(), lio.eskf_get_cov(), lio.last_solve_launches())
        (r0, s0, P0, l0), (r1, s1, P1, l1) = res[False], res[True]
        assert r0["rc"] == 0 and r1["rc"] == 0 and r0["iters"] == r1["iters"] and l1 == 1
        assert np.array_equal(r0["log"][:, 59], r1["log"][:, 59])
        assert rel(r1["log"][:, :42], r0["log"][:, :42]) < 1e-11 and rel(s1, s0) < 1e-9 and rel(P1, P0) < 1e-8
        u = oracle_lib.update_iekf(m, e, oracle_lib.default_opts(max_num_residuals=INT_MAX), sw["raw"], st, sw["t_last"], frame_id=100, log_iters=20)
        assert u["rc"] == r1["iters"] and rel(s1, e.get_state()) < TIGHT
    finally:
        lio.close()
