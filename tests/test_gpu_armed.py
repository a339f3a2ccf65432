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


ALWAYS = 2      # srl_set_armed_launch(ctx, 2): a launch armed behind EVERY eligible pass (the mechanism, without the policy of mode 1)


def _run(sc, solve, armed, n=4):
    lio = sc["lio"]
    lio.ctx.set_armed_launch(ALWAYS if armed else 0)
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
    lio.ctx.set_armed_launch(ALWAYS)
    a = _solver(scene, INT_MAX)
    b = _solver(scene, 600)
    c = _solver(scene, INT_MAX, weight_alpha=0.8, weight_neighborhood=0.2)
    ref = {}
    lio.ctx.set_armed_launch(False)
    for k, s in (("a", a), ("b", b), ("c", c)):
        s(); ref[k] = s.state.copy()
    lio.ctx.set_armed_launch(ALWAYS)
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
    lio.ctx.set_armed_launch(True)


def test_an_armed_launch_too_old_for_the_host_is_cancelled_not_fired(scene):
    lio = scene["lio"]
    lio.ctx.set_armed_launch(False)
    a = _solver(scene)
    a(); ref = a.state.copy()
    lio.ctx.set_armed_launch(ALWAYS)
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
        lio.ctx.set_armed_launch(True)


def test_an_armed_launch_that_gave_up_on_the_device_is_replaced_by_a_normal_one(scene):
    """kernel-side bound 200 us, host-side bound an hour: the host fires a launch that has already left -- the mailbox says so and
    the pass is repeated with a normal launch"""
    lio = scene["lio"]
    lio.ctx.set_armed_launch(False)
    a = _solver(scene)
    a(); ref = a.state.copy()
    lio.ctx.set_armed_launch(ALWAYS)
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
        lio.ctx.set_armed_launch(True)


def test_frame_pipeline_and_map_growth_between_armed_solves(scene):
    """insert points (the map's table and slabs may move: other kernel arguments), solve again: equal to the un-armed solve on the same map"""
    lio = scene["lio"]
    a = _solver(scene)
    lio.ctx.set_armed_launch(ALWAYS)
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


# ---------------------------------------------------------------------------------------------------------------------
# The arming policy of mode 1 (the default) and launches that survive srl_sweep_swap
# ---------------------------------------------------------------------------------------------------------------------
def _sweeps(sc, count, sizes=None):
    """`count` distinct sweeps of the scene (own seeds, own predicted poses, hence own priors) in page-locked memory, each with the
    solver that starts from ITS prior; sizes: keypoints per sweep (default: the scene's)"""
    import gc
    gc.collect()
    lio = sc["lio"]
    n_kp, map_pts, pattern, seed = synth.CONFIGS["C1"]
    out = []
    for j in range(count):
        n = sizes[j] if sizes else n_kp
        sw = synth.make_sweep(seed + 2000 + j, n, sc["L"], pattern=pattern)
        ps = sc["prior_state"].copy()
        ps[0:3] = sw["t_pred"]; ps[3:7] = sw["q_pred"]; ps[7:10] = sw["vel"]
        st0 = np.concatenate([sw["q_pred"], sw["t_pred"], sw["vel"], np.zeros(6)])
        pin = srl.PinnedArray(sw["raw"].shape)
        pin.array[:] = sw["raw"]
        out.append(dict(sweep=sw, pin=pin, n=n, prior_state=ps, state0=st0))
    return out


def _stream(sc, sweeps, opts, rounds, frame_id=100, during_solve=False):
    """the node's loop over a stream of sweeps: the next sweep is uploaded while the current one is solved (srl_sweep_prefetch), made
    current behind the solve (srl_sweep_swap); returns the solved states in order"""
    lio = sc["lio"]
    solvers = [lio.bound_solver(opts, s["prior_state"], sc["prior_cov"], s["state0"], s["sweep"]["t_last"], frame_id, s["n"]) for s in sweeps]
    S = len(sweeps)
    lio.prefetch_sweep(sweeps[0]["pin"].array); lio.swap_sweep()
    states = []
    for k in range(rounds):
        # the next sweep's upload: before the solve, or issued by the solve itself beside its first kernel (srl_lio_prefetch_sweep_during_solve)
        (lio.prefetch_sweep_during_solve if during_solve else lio.prefetch_sweep)(sweeps[(k + 1) % S]["pin"].array)
        rc, it, nr = solvers[k % S]()
        assert rc == 0
        states.append((it, nr, solvers[k % S].state.copy(), lio.eskf_get_cov().copy()))
        lio.swap_sweep()
    lio.ctx.disarm()
    return states


@pytest.fixture(scope="module")
def scene_L(scene):
    n_kp, map_pts, pattern, seed = synth.CONFIGS["C1"]
    _, L = synth.map_candidates(seed, map_pts)
    scene["L"] = L
    return scene


@pytest.mark.parametrize("box", [-1, 0])
@pytest.mark.parametrize("max_res", [INT_MAX, 600])
def test_a_launch_armed_behind_the_last_pass_becomes_the_first_pass_of_the_next_sweep(scene_L, max_res, box):
    """four distinct sweeps streamed through prefetch / swap: in steady state every pass -- the first pass of a sweep included -- fires
    a waiting launch, nothing is cancelled, and every solved state and covariance equals, bit for bit, the one launch per
    iteration produces on the same sweep.  box = 0: the pose box in host memory, relayed by workgroup 0 (hosts without a large BAR) --
    the launch fired for the swapped-in sweep carries GO | ALT through the relay (the relay once dropped it, and every
    workgroup but the first ran on an empty tile; the host now also checks the visited count of every fused pass)"""
    sc = scene_L
    lio = sc["lio"]
    sw = _sweeps(sc, 4)
    opts = srl.default_opts(max_num_residuals=max_res)
    lio.ctx.set_pose_box(box)
    try:
        lio.ctx.set_armed_launch(False)
        ref = _stream(sc, sw, opts, 8)
        lio.ctx.set_armed_launch(True)
        _stream(sc, sw, opts, 4)                                            # (the pass count of a solve is learnt from the solve before it)
        s0 = lio.ctx.arm_stats()
        got = _stream(sc, sw, opts, 8) + _stream(sc, sw, opts, 8, during_solve=True)
        s1 = lio.ctx.arm_stats()
        for k, g in enumerate(got):
            r = ref[k % 8]
            assert g[0] == r[0] and g[1] == r[1], (k, g[:2], r[:2])
            assert np.array_equal(g[2], r[2]) and np.array_equal(g[3], r[3]), k
        passes = sum(g[0] for g in got)
        # the very first pass of the series has no launch waiting; the disarm at the end of _stream cancels the one armed behind the last pass
        assert s1["fired"] - s0["fired"] >= passes - 2, (s0, s1, passes)
        assert s1["cancelled"] - s0["cancelled"] <= 2, (s0, s1)
        assert s1["expired"] == s0["expired"]
    finally:
        lio.ctx.set_pose_box(-1)
        lio.ctx.set_armed_launch(True)
        lio.resident_sweep(sc["sweep"]["raw"])
        for s in sw:
            s["pin"].close()


def test_prefix_first_upload_lets_the_first_pass_fire_before_the_whole_sweep_has_landed(scene_L):
    """VERDICT r05 item 2.  max_num_residuals = 600 visits the first 4 480 keypoints of a sweep; the prefetch sends that prefix first (own
    DMA, own event) and srl_sweep_swap keeps the waiting launch once the PREFIX has landed.  Sweeps of 60 000 keypoints (1.4 MB each: the
    tail is still crossing PCIe when a 50 us solve ends): states bit-identical to launch-per-iteration, (almost) nothing cancelled.
    Sweep 2 of the stream starts with 6 000 keypoints far off the map: its prefix holds no residual, the pass is repeated over the whole
    sweep -- a normal launch ordered behind the FULL upload -- and still every bit agrees."""
    sc = scene_L
    lio = sc["lio"]
    sw = _sweeps(sc, 4, sizes=[60000, 60000, 60000, 60000])
    sw[2]["pin"].array[:6000] += np.array([0.0, 0.0, 500.0])                # (in place, page-locked: what the DMA reads)
    opts = srl.default_opts(max_num_residuals=600)
    try:
        lio.ctx.set_armed_launch(False)
        ref = _stream(sc, sw, opts, 8, during_solve=True)
        assert ref[2][1] == 600 and ref[0][1] == 600
        lio.ctx.set_armed_launch(True)
        _stream(sc, sw, opts, 4, during_solve=True)
        s0 = lio.ctx.arm_stats()
        got = _stream(sc, sw, opts, 16, during_solve=True)
        s1 = lio.ctx.arm_stats()
        for k, g in enumerate(got):
            r = ref[k % 8]
            assert g[0] == r[0] and g[1] == r[1], (k, g[:2], r[:2])
            assert np.array_equal(g[2], r[2]) and np.array_equal(g[3], r[3]), k
        passes = sum(g[0] for g in got)
        # sweep 2 costs a cancellation per visit (its whole-sweep repeat is not the waiting launch's pass) and the launch armed behind it
        # one more; every other pass of the stream -- the first pass of every other sweep included -- fires
        assert s1["cancelled"] - s0["cancelled"] <= 2 * 4 + 2, (s0, s1)
        assert s1["fired"] - s0["fired"] >= passes - 3 * 4 - 2, (s0, s1, passes)
        assert s1["expired"] == s0["expired"]
    finally:
        lio.ctx.set_armed_launch(True)
        lio.resident_sweep(sc["sweep"]["raw"])
        for s_ in sw:
            s_["pin"].close()


def test_a_waiting_launch_serves_a_shorter_sweep_and_is_cancelled_for_a_longer_one(scene_L):
    """keypoint counts differ from sweep to sweep: the launch armed on 4 096 keypoints (128 workgroups) is fired for 4 000 (its last
    workgroups find empty tiles), the one armed on 4 000 (125 workgroups) cannot hold 4 096 and is cancelled -- same bits either way"""
    sc = scene_L
    lio = sc["lio"]
    sw = _sweeps(sc, 4, sizes=[4096, 4000, 4096, 3990])
    opts = srl.default_opts(max_num_residuals=INT_MAX)
    try:
        lio.ctx.set_armed_launch(False)
        ref = _stream(sc, sw, opts, 4)
        lio.ctx.set_armed_launch(True)
        _stream(sc, sw, opts, 4)
        s0 = lio.ctx.arm_stats()
        got = _stream(sc, sw, opts, 12)
        s1 = lio.ctx.arm_stats()
        for k, g in enumerate(got):
            r = ref[k % 4]
            assert g[0] == r[0] and g[1] == r[1] and np.array_equal(g[2], r[2]) and np.array_equal(g[3], r[3]), k
        assert s1["fired"] > s0["fired"] and s1["cancelled"] > s0["cancelled"] and s1["expired"] == s0["expired"]
    finally:
        lio.ctx.set_armed_launch(True)
        lio.resident_sweep(sc["sweep"]["raw"])
        for s in sw:
            s["pin"].close()


def test_no_launch_is_left_waiting_behind_a_solve_without_a_prefetched_sweep(scene):
    """mode 1 on a resident sweep solved again and again: launches are armed only between the passes of a solve -- none behind the pass
    expected to be the last --, so every armed launch is fired and nothing is cancelled; srl_solve_end cancels what a longer-than-
    expected solve leaves behind"""
    import gc
    gc.collect()
    lio = scene["lio"]
    lio.resident_sweep(scene["sweep"]["raw"])
    a = _solver(scene)
    lio.ctx.set_armed_launch(False)
    a(); ref = a.state.copy()
    lio.ctx.set_armed_launch(True)
    a(); a()
    s0 = lio.ctx.arm_stats()
    its = 0
    for _ in range(6):
        rc, it, nr = a()
        its += it
        assert rc == 0 and np.array_equal(a.state, ref)
    s1 = lio.ctx.arm_stats()
    assert s1["armed"] - s0["armed"] == its - 6 and s1["fired"] - s0["fired"] == its - 6, (s0, s1, its)
    assert s1["cancelled"] == s0["cancelled"] and s1["expired"] == s0["expired"]


def test_a_second_context_on_the_device_keeps_launches_from_being_armed(scene):
    import gc
    gc.collect()
    lio = scene["lio"]
    lio.resident_sweep(scene["sweep"]["raw"])
    a = _solver(scene)
    lio.ctx.set_armed_launch(True)
    a(); ref = a.state.copy()
    other = srl.Context(0)
    try:
        s0 = lio.ctx.arm_stats()
        for _ in range(3):
            a()
            assert np.array_equal(a.state, ref)
        s1 = lio.ctx.arm_stats()
        assert s1["armed"] == s0["armed"]
    finally:
        other.close()
    a(); a()
    assert lio.ctx.arm_stats()["armed"] > s1["armed"] and np.array_equal(a.state, ref)


@pytest.mark.parametrize("n_big", [70_000, 140_000])
def test_a_sweep_of_more_workgroups_than_compute_units_stays_fused_and_armed(scene_L, n_big):
    """70 000 / 140 000 keypoints = 274 / 547 workgroups of 256 on 256 compute units: the pass runs in rounds -- still one kernel per ESIKF
    iteration (fused final reduction over up to 2 048 rows) and armed like any other pass: the first round waits resident for its pose,
    the later rounds find it in the box when they start.  Bit-identical to the same pass launched per iteration, and equal to the
    un-fused pass (reduce kernel) up to summation order"""
    sc = scene_L
    lio = sc["lio"]
    n_kp, map_pts, pattern, seed = synth.CONFIGS["C1"]
    sw = synth.make_sweep(seed + 3000, n_big, sc["L"], pattern=pattern)
    ps = sc["prior_state"].copy()
    ps[0:3] = sw["t_pred"]; ps[3:7] = sw["q_pred"]; ps[7:10] = sw["vel"]
    st0 = np.concatenate([sw["q_pred"], sw["t_pred"], sw["vel"], np.zeros(6)])
    opts = srl.default_opts(max_num_residuals=INT_MAX)
    try:
        lio.resident_sweep(sw["raw"])
        solve = lio.bound_solver(opts, ps, sc["prior_cov"], st0, sw["t_last"], 100, n_big)
        lio.ctx.set_armed_launch(False)
        rc, it, nr = solve()
        assert rc == 0 and it >= 2
        ref, ref_cov, launches = solve.state.copy(), lio.eskf_get_cov().copy(), lio.last_solve_launches()
        assert launches == it                                               # one kernel per iteration: fused although the grid exceeds the chip
        lio.ctx.set_fused_reduce(0)
        solve()
        unfused = solve.state.copy()
        lio.ctx.set_fused_reduce(1)
        assert np.max(np.abs(unfused - ref)) <= 1e-12 * np.max(np.abs(ref))
        lio.ctx.set_armed_launch(ALWAYS)
        s0 = lio.ctx.arm_stats()
        for _ in range(4):
            rc, it2, nr2 = solve()
            assert rc == 0 and (it2, nr2) == (it, nr) and np.array_equal(solve.state, ref) and np.array_equal(lio.eskf_get_cov(), ref_cov)
        s1 = lio.ctx.arm_stats()
        assert s1["fired"] - s0["fired"] == 4 * it - 1 and s1["expired"] == s0["expired"]
    finally:
        lio.ctx.set_armed_launch(True)
        lio.resident_sweep(sc["sweep"]["raw"])


def test_the_fused_stream_step_equals_the_separate_calls(scene_L):
    """srl_lio_stream_step (prior reset + upload of the next sweep registered with the solve + solve + swap in one C call: the body of a
    node's loop, what bench.py's stream calls) against the same stream through the separate entry points: same iteration and residual
    counts, states and covariances bit for bit, armed launches surviving the swaps either way"""
    sc = scene_L
    lio = sc["lio"]
    sw = _sweeps(sc, 4)
    opts = srl.default_opts(max_num_residuals=INT_MAX)
    try:
        lio.ctx.set_armed_launch(True)
        ref = _stream(sc, sw, opts, 8, during_solve=True)
        steps = [lio.bound_stream_step(opts, s["prior_state"], sc["prior_cov"], s["state0"], s["sweep"]["t_last"], 100, s["n"], sw[(j + 1) % 4]["pin"].array)
                 for j, s in enumerate(sw)]
        lio.prefetch_sweep(sw[0]["pin"].array); lio.swap_sweep()
        s0 = lio.ctx.arm_stats()
        for k in range(16):
            rc, it, nr = steps[k % 4]()
            r = ref[k % 8]
            assert rc == 0 and (it, nr) == (r[0], r[1]), (k, rc, it, nr, r[:2])
            assert np.array_equal(steps[k % 4].state, r[2]) and np.array_equal(lio.eskf_get_cov(), r[3]), k
        s1 = lio.ctx.arm_stats()
        lio.ctx.disarm()
        assert s1["cancelled"] - s0["cancelled"] <= 1 and s1["fired"] - s0["fired"] >= sum(r[0] for r in ref) * 2 - 2
    finally:
        lio.ctx.set_armed_launch(True)
        lio.resident_sweep(sc["sweep"]["raw"])
        for s_ in sw:
            s_["pin"].close()
