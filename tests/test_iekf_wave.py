"""The persistent solve kernel's 17-dim algebra (sr_livo_amd/csrc/srl_iekf_wave.h: prior error state, covariance projection,
the two 17 x 17 inverses as a one-wave LU with partial pivoting, gain, step guard, observe, convergence rule, posterior
covariance -- src/optimize.cpp:172-310) is written against a wave interface and instantiated for an emulated wave of 64 lanes:
the SAME source the kernel runs, on the CPU.  It must reproduce the host mirror's updateIEKF (host/lioOptimization.cpp, the
reference's own order of operations) BIT FOR BIT when both are fed the same normal equations."""
import numpy as np
import pytest

import sr_livo_amd as srl
from oracle import pyoracle as po
from sr_livo_amd import capi, synth
from test_host_logic import oracle_provider

INT_MAX = 2**31 - 1


def _prior(lio, sw):
    class A:
        def __init__(self, l): self.l = l
        def set_noise(self, *a): self.l.eskf_set_noise(*a)
        def scale_init_cov(self): self.l.eskf_scale_init_cov()
        def init_imu(self, a, g): self.l.eskf_init_imu(a, g)
        def predict(self, dt, a, g): self.l.eskf_predict(dt, a, g)
        def get_state(self): return self.l.eskf_get_state()
        def set_state(self, s): self.l.eskf_set_state(s)
    synth.eskf_prior(A(lio), sw["q_pred"], sw["t_pred"], sw["vel"])


def _both(m, raw, sw, frame_id, opts_p, provider=None, extra_predict=0, laser_cov=0.001, seed=None, exact_lu=True):
    opts_o = po.opts_from_product(opts_p)
    lio = srl.Lio(-1)
    lio.set_laser_point_cov(laser_cov)
    _prior(lio, sw)
    rng = np.random.default_rng(seed or 0)
    for _ in range(extra_predict):       # a covariance with off-diagonal structure and a moving filter
        lio.eskf_predict(0.01, np.array([0.0, 0.0, 9.81]) + rng.normal(0, 0.5, 3), rng.normal(0, 0.1, 3))
    if extra_predict:
        s = lio.eskf_get_state(); s[0:3] = sw["t_pred"]; s[3:7] = sw["q_pred"]; lio.eskf_set_state(s)
    es0, P0 = lio.eskf_get_state().copy(), lio.eskf_get_cov().copy()
    st = np.concatenate([sw["q_pred"], sw["t_pred"], sw["vel"], np.zeros(6)])
    fresh = (lambda: provider()) if getattr(provider, "__name__", "") == "mixed_factory" else (lambda: provider or oracle_provider(m, raw, opts_o))
    g = lio.update_iekf_provided(opts_p, fresh(), len(raw), st, sw["t_last"], frame_id=frame_id, log_iters=20,
                                 allow=(capi.SRL_ERR_NOT_ENOUGH_RESIDUALS, capi.SRL_ERR_NAN_PLANARITY))
    frame = capi.make_frame(sw["q_pred"], sw["t_pred"], sw["t_last"], frame_id=frame_id)
    w = capi.iekf_wave_solve(frame, opts_p, laser_cov, es0, P0, fresh(), log_iters=20, exact_lu=exact_lu)
    return lio, g, w


@pytest.mark.parametrize("frame_id,max_res,iters_icp,extra", [(100, INT_MAX, 5, 0), (100, 600, 5, 7), (1, INT_MAX, 3, 0), (5, 300, 5, 3),
                                                              (100, INT_MAX, 1, 11)])
def test_wave_algebra_equals_the_host_mirror_bitwise(small_scene, frame_id, max_res, iters_icp, extra):
    m, sw = small_scene["map"], small_scene["sweep"]
    raw = sw["raw"][:700]
    opts_p = srl.default_opts(max_num_residuals=max_res, num_iters_icp=iters_icp)
    lio, g, w = _both(m, raw, sw, frame_id, opts_p, extra_predict=extra, seed=frame_id + extra)
    assert g["rc"] == 0 and w["rc"] == 0
    assert w["iterations"] == g["iters"]
    assert w["verdict"] in (capi.IEKF_DONE, capi.IEKF_DONE_NO_COV)
    assert np.array_equal(w["log"], g["log"])                              # every d_x of every iteration
    assert np.array_equal(w["state"], lio.eskf_get_state())               # the filter
    assert np.array_equal(w["state"][[3, 4, 5, 6, 0, 1, 2]], g["state"][:7])   # p_frame->p_state (optimize.cpp:255-256)
    assert w["covariance_updated"] == 1
    assert np.array_equal(w["cov"], lio.eskf_get_cov())


def test_wave_algebra_step_guard_failure_and_nan_paths(small_scene):
    m, sw = small_scene["map"], small_scene["sweep"]
    raw = sw["raw"][:300]
    # (a) not enough residuals on the first pass: state untouched, verdict FAIL (optimize.cpp:110-123)
    opts_p = srl.default_opts(max_num_residuals=-1)
    lio, g, w = _both(m, raw, sw, 100, opts_p)
    assert g["rc"] == capi.SRL_ERR_NOT_ENOUGH_RESIDUALS and w["rc"] == 0 and w["verdict"] == capi.IEKF_FAIL_RESIDUALS
    assert w["iterations"] == 0 and np.array_equal(w["state"], lio.eskf_get_state())

    # (b) NaN planarity (optimize.cpp:348-350)
    def nan_provider(frame, opts, out):
        return capi.SRL_ERR_NAN_PLANARITY
    lio, g, w = _both(m, raw, sw, 100, srl.default_opts(), provider=nan_provider)
    assert g["rc"] == capi.SRL_ERR_NAN_PLANARITY and w["rc"] == capi.SRL_ERR_NAN_PLANARITY and w["verdict"] == capi.IEKF_NAN

    # (c) the step guard (optimize.cpp:248-251): normal equations that ask for a 1 km step are skipped on every pass, the
    # loop runs out without a covariance update and the filter is where it started
    def wild_provider(frame, opts, out):
        out.HtH[:] = list((np.eye(6) * 1e6).ravel()); out.Hth[:] = [1e9, 0, 0, 0, 0, 0]
        out.num_residuals = 500; out.success = 1; out.loss_sum = 1.0
        return 0
    lio, g, w = _both(m, raw, sw, 100, srl.default_opts(num_iters_icp=4), provider=wild_provider)
    assert g["rc"] == 0 and g["iters"] == 5 and w["iterations"] == 5 and w["verdict"] == capi.IEKF_DONE_NO_COV
    assert w["covariance_updated"] == 0
    assert np.array_equal(w["state"], lio.eskf_get_state()) and np.array_equal(w["log"], g["log"])

    # (d) a guarded first pass followed by normal ones (a fresh counter per solve)
    good = oracle_provider(m, raw, po.opts_from_product(srl.default_opts()))
    def mixed_factory():
        calls = {"n": 0}
        def mixed(frame, opts, out):
            calls["n"] += 1
            return wild_provider(frame, opts, out) if calls["n"] == 1 else good(frame, opts, out)
        return mixed
    lio, g, w = _both(m, raw, sw, 100, srl.default_opts(num_iters_icp=6), provider=mixed_factory)
    assert g["rc"] == 0 and g["iters"] >= 3 and w["iterations"] == g["iters"] and w["verdict"] == capi.IEKF_DONE
    assert np.array_equal(w["log"], g["log"])
    assert np.array_equal(w["state"], lio.eskf_get_state()) and np.array_equal(w["cov"], lio.eskf_get_cov())


def test_wave_lu_pivots_like_the_host_on_tied_and_permuted_columns():
    """Pivot order: exact ties of |value| in a column (the host's strict `>` scan keeps the first) and matrices that need
    real row exchanges.  Driven through the solve: covariance = a permutation-heavy SPD matrix, H^T H with equal entries."""
    rng = np.random.default_rng(5)
    A = rng.normal(size=(17, 17))
    P = A @ A.T * 1e-3 + np.eye(17) * 1e-4
    P[[0, 9]] = P[[9, 0]]; P[:, [0, 9]] = P[:, [9, 0]]
    lio = srl.Lio(-1)
    s = lio.eskf_get_state(); s[3:7] = synth.quat_from_rotvec([0.02, -0.01, 0.3]); s[0:3] = [1.0, -2.0, 0.5]
    lio.eskf_set_state(s); lio.eskf_set_cov(P)
    def prov(frame, opts, out):
        H = np.ones((6, 6)) * 50.0 + np.diag([400.0, 400.0, 400.0, 90.0, 90.0, 90.0])
        out.HtH[:] = list(H.ravel()); out.Hth[:] = [0.3, -0.2, 0.1, 0.01, -0.02, 0.005]
        out.num_residuals = 100; out.success = 1; out.loss_sum = 0.5
        return 0
    st = np.concatenate([s[3:7], s[0:3], np.zeros(9)])
    es0, P0 = lio.eskf_get_state().copy(), lio.eskf_get_cov().copy()
    g = lio.update_iekf_provided(srl.default_opts(num_iters_icp=3), prov, 100, st, np.zeros(3), frame_id=100, log_iters=10)
    w = capi.iekf_wave_solve(capi.make_frame(s[3:7], s[0:3], np.zeros(3)), srl.default_opts(num_iters_icp=3), 0.001, es0, P0, prov, log_iters=10)
    assert g["rc"] == 0 and w["rc"] == 0 and w["iterations"] == g["iters"]
    assert np.array_equal(w["log"], g["log"]) and np.array_equal(w["state"], lio.eskf_get_state())
    assert np.array_equal(w["cov"], lio.eskf_get_cov())


def _rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


@pytest.mark.parametrize("frame_id,max_res,iters_icp,extra", [(100, INT_MAX, 5, 0), (100, 600, 5, 7), (1, INT_MAX, 3, 0), (5, 300, 5, 3),
                                                              (100, INT_MAX, 1, 11)])
def test_schur_form_of_the_second_inverse_agrees_with_the_lu_form(small_scene, frame_id, max_res, iters_icp, extra):
    """The kernel's default: temp_inv.block<17,6>(0,0) = F (R G + H)^-1 (Schur complement of the H-independent block, a 6 x 6
    Cholesky behind the reduction) instead of the 17 x 17 partial-pivot LU.  Same matrix, other operation order: the solve
    must take the same number of passes and land on the same filter within 1e-9 (measured: ~1e-12)."""
    m, sw = small_scene["map"], small_scene["sweep"]
    raw = sw["raw"][:700]
    opts_p = srl.default_opts(max_num_residuals=max_res, num_iters_icp=iters_icp)
    lio, g, w = _both(m, raw, sw, frame_id, opts_p, extra_predict=extra, seed=frame_id + extra, exact_lu=False)
    assert g["rc"] == 0 and w["rc"] == 0 and w["iterations"] == g["iters"] and w["verdict"] == capi.IEKF_DONE
    assert _rel(w["log"][:, 42:59], g["log"][:, 42:59]) < 1e-9            # d_x of every pass
    assert _rel(w["state"], lio.eskf_get_state()) < 1e-11
    P_ref = lio.eskf_get_cov()
    assert _rel(w["cov"], P_ref) < 1e-9
    d = np.sqrt(np.abs(np.diag(P_ref)))
    assert np.max(np.abs(w["cov"] - P_ref) / np.outer(d, d)) < 1e-6      # element-wise, scaled by the standard deviations
