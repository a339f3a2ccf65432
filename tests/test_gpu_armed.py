"""Armed launches (include/srlivo_hip.h: srl_set_armed_launch): the kernel of the next buildPlaneResiduals pass
(src/optimize.cpp:153) is enqueued while the current pass runs and receives its pose through the pose box.  Nothing observable may
change: the solved state, covariance, iteration and residual counts are BIT-identical to one launch per ESIKF iteration, for every
budget, and every way an armed launch can end -- fired, cancelled by another call, cancelled because the arguments changed, too old
on the host's clock, expired on the device's clock -- leaves the right answer behind.
"""
import time

import numpy as np
import pytest

import sr_livo_amd as srl
from sr_livo_amd import synth

pytestmark = pytest.mark.gpu

INT_MAX = 2**31 - 1


class _EskfAdapter:
    def __init__(self, lio): self.lio = lio
    def set_noise(self, *a): self.lio.eskf_set_noise(*a)
    def scale_init_cov(self): self.lio.eskf_scale_init_cov()
    def init_imu(self, a, g): self.lio.eskf_init_imu(a, g)
    def predict(self, dt, a, g): self.lio.eskf_predict(dt, a, g)
    def get_state(self): return self.lio.eskf_get_state()
    def set_state(self, s): self.lio.eskf_set_state(s)


@pytest.fixture(scope="module")
def scene():
    n_kp, map_pts, pattern, seed = synth.CONFIGS["C1"]                     # 4 096 keypoints: sixteen-wave workgroups, one round
    cands, L = synth.map_candidates(seed, map_pts)
    sweep = synth.make_sweep(seed + 1000, n_kp, L, pattern=pattern)
    lio = srl.Lio(0)
    lio.add_points_to_map(cands)
    prior_state = synth.eskf_prior(_EskfAdapter(lio), sweep["q_pred"], sweep["t_pred"], sweep["vel"]).copy()
    prior_cov = lio.eskf_get_cov().copy()
    state0 = np.concatenate([sweep["q_pred"], sweep["t_pred"], sweep["vel"], np.zeros(6)])
    lio.resident_sweep(sweep["raw"])
    yield dict(lio=lio, sweep=sweep, prior_state=prior_state, prior_cov=prior_cov, state0=state0, n=n_kp, cands=cands)
    lio.close()


def _solver(sc, max_res=INT_MAX, frame_id=100, **kw):
    opts = srl.default_opts(max_num_residuals=max_res, **kw)
    return sc["lio"].bound_solver(opts, sc["prior_state"], sc["prior_cov"], sc["state0"], sc["sweep"]["t_last"], frame_id, sc["n"])


def _run(sc, solve, armed, n=4):
    lio = sc["lio"]
    lio.ctx.set_armed_launch(armed)
    out = []
    for _ in range(n):
        rc, it, nr = solve()
        out.append((rc, it, nr, solve.state.copy(), lio.eskf_get_state().copy(), lio.eskf_get_cov().copy()))
    return out


@pytest.mark.parametrize("box", [0, 1])
@pytest.mark.parametrize("max_res,frame_id", [(INT_MAX, 100), (600, 100), (37, 100), (INT_MAX, 5)])
def test_armed_launches_change_no_bit(scene, max_res, frame_id, box):
    lio = scene["lio"]
    try:
        lio.ctx.set_pose_box(box)
    except RuntimeError as e:
        if box == 1 and "CPU-visible" in str(e):
            pytest.skip("device memory is not CPU-visible on this box")
        raise
    try:
        solve = _solver(scene, max_res, frame_id)
        ref = _run(scene, solve, False, 2)
        s0 = lio.ctx.arm_stats()
        got = _run(scene, solve, True, 5)
        s1 = lio.ctx.arm_stats()
        assert ref[0][0] == 0 and ref[0][1] >= 2
        for g in got:
            assert g[:3] == ref[0][:3], (g[:3], ref[0][:3], lio.ctx.arm_stats())
            for a, b in zip(g[3:], ref[0][3:]):
                assert np.array_equal(a, b)
        iters = ref[0][1]
        # every pass arms its successor; every pass but the very first of the series finds one waiting and fires it
        assert s1["armed"] - s0["armed"] == 5 * iters and s1["fired"] - s0["fired"] == 5 * iters - 1
        assert s1["expired"] == s0["expired"]
    finally:
        lio.ctx.set_pose_box(-1)
        lio.ctx.set_armed_launch(True)


def test_another_call_or_other_arguments_cancel_the_armed_launch(scene):
    lio = scene["lio"]
    lio.ctx.set_armed_launch(True)
    a = _solver(scene, INT_MAX)
    b = _solver(scene, 600)
    c = _solver(scene, INT_MAX, weight_alpha=0.8, weight_neighborhood=0.2)
    ref = {}
    lio.ctx.set_armed_launch(False)
    for k, s in (("a", a), ("b", b), ("c", c)):
        s(); ref[k] = s.state.copy()
    lio.ctx.set_armed_launch(True)
    s0 = lio.ctx.arm_stats()
    for k, s in (("a", a), ("b", b), ("a", a), ("c", c), ("c", c), ("b", b)):     # other budget / other weights: the waiting launch is not theirs
        rc, it, nr = s()
        assert rc == 0 and np.array_equal(s.state, ref[k]), k
    s1 = lio.ctx.arm_stats()
    assert s1["cancelled"] - s0["cancelled"] >= 4
    # another entry point between two solves: map download, sweep re-upload, taps
    a(); lio.ctx.map_download(); a()
    assert np.array_equal(a.state, ref["a"])
    lio.resident_sweep(scene["sweep"]["raw"]); a()
    assert np.array_equal(a.state, ref["a"])
    lio.ctx.disarm(); a()
    assert np.array_equal(a.state, ref["a"]) and lio.ctx.arm_stats()["expired"] == s0["expired"]


def test_an_armed_launch_too_old_for_the_host_is_cancelled_not_fired(scene):
    lio = scene["lio"]
    lio.ctx.set_armed_launch(False)
    a = _solver(scene)
    a(); ref = a.state.copy()
    lio.ctx.set_armed_launch(True)
    lio.ctx.set_arm_linger(host_linger_us=0.0)                             # every waiting launch is "too old"
    try:
        s0 = lio.ctx.arm_stats()
        for _ in range(3):
            a()
            assert np.array_equal(a.state, ref)
        s1 = lio.ctx.arm_stats()
        assert s1["fired"] == s0["fired"] and s1["cancelled"] - s0["cancelled"] >= 3 * 2 - 1
    finally:
        lio.ctx.set_arm_linger()


def test_an_armed_launch_that_gave_up_on_the_device_is_replaced_by_a_normal_one(scene):
    """kernel-side bound 200 us, host-side bound an hour: the host fires a launch that has already left -- the mailbox says so and
    the pass is repeated with a normal launch"""
    lio = scene["lio"]
    lio.ctx.set_armed_launch(False)
    a = _solver(scene)
    a(); ref = a.state.copy()
    lio.ctx.set_armed_launch(True)
    lio.ctx.set_arm_linger(host_linger_us=3.6e9, kernel_linger_us=200.0)
    try:
        a()
        time.sleep(0.05)                                                   # the launch armed by the last pass expires meanwhile
        s0 = lio.ctx.arm_stats()
        rc, it, nr = a()
        s1 = lio.ctx.arm_stats()
        assert rc == 0 and np.array_equal(a.state, ref)
        assert s1["expired"] - s0["expired"] == 1
    finally:
        lio.ctx.set_arm_linger()
        lio.ctx.disarm()


def test_frame_pipeline_and_map_growth_between_armed_solves(scene):
    """insert points (the map's table and slabs may move: other kernel arguments), solve again: equal to the un-armed solve on the same map"""
    lio = scene["lio"]
    a = _solver(scene)
    lio.ctx.set_armed_launch(True)
    a(); a()
    rng = np.random.default_rng(5)
    extra = scene["cands"][rng.choice(len(scene["cands"]), 2000, replace=False)] + rng.normal(0, 0.3, (2000, 3))
    lio.add_points_to_map(extra)
    a(); got = a.state.copy(); a()
    assert np.array_equal(a.state, got)
    lio.ctx.set_armed_launch(False)
    a()
    assert np.array_equal(a.state, got)
    lio.ctx.set_armed_launch(True)
