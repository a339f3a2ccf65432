"""Neighbourhood bounds (SrlAssocArgs::bound_in, include/srlivo_hip_debug.h: srl_debug_set_bound_culling): every pass leaves, per keypoint,
its world position and the exact squared distance of its K-th nearest neighbour; the next pass over the same sweep and map skips the
voxels that lie further from the keypoint than sqrt(tau) + |movement|.  searchNeighbors (optimize.cpp:365-426) keeps the K nearest of
whatever it visited, so NOTHING observable may change: neighbour ids, candidate counts (the reference's loop visits every voxel: P_k
counts them all), normal equations, solved states -- bit for bit against the same passes with the culling off; across pose jumps, map
changes, sweep changes, ties, finite budgets."""
import numpy as np
import pytest

import sr_livo_amd as srl
from sr_livo_amd import capi, synth

pytestmark = pytest.mark.gpu
INT_MAX = 2**31 - 1


def _poses(sw, count, seed, jump=None):
    """poses of consecutive ESIKF iterations: the predicted pose moving towards the ground truth (and `jump`: one far-off pose in between)"""
    rng = np.random.default_rng(seed)
    out = []
    for k in range(count):
        a = 0.7 ** k
        q = synth.quat_mul(sw["q_gt"], synth.quat_from_rotvec(a * rng.normal(0, 0.004, 3)))
        t = sw["t_gt"] + a * (sw["t_pred"] - sw["t_gt"]) + rng.normal(0, 1e-4, 3)
        out.append((q, t))
    if jump is not None:
        out.insert(jump, (synth.quat_mul(sw["q_gt"], synth.quat_from_rotvec([0.0, 0.0, 0.3])), sw["t_gt"] + np.array([1.5, -0.7, 0.2])))
    return out


def _passes(ctx, sw, poses, opts, taps):
    res = []
    ctx.set_taps(taps)
    for q, t in poses:
        neq, rc = ctx.build_residuals(capi.make_frame(q, t, sw["t_last"]), opts)
        ids = ctx.fetch_neighbors()[0].copy() if taps else None
        res.append((neq, ids))
    ctx.set_taps(False)
    return res


def _same(a, b, what):
    (na, ia), (nb, ib) = a, b
    assert na.num_residuals == nb.num_residuals and na.sum_candidates == nb.sum_candidates and na.last_visited == nb.last_visited, what
    assert np.array_equal(np.array(na.HtH), np.array(nb.HtH)) and np.array_equal(np.array(na.Hth), np.array(nb.Hth)) and na.loss_sum == nb.loss_sum, what
    if ia is not None:
        assert np.array_equal(ia, ib), what


@pytest.fixture(scope="module")
def scene():
    cands, L = synth.map_candidates(4401, 200_000)
    ctx = srl.Context(0)
    ctx.map_insert(cands)
    yield dict(ctx=ctx, cands=cands, L=L)
    ctx.close()


@pytest.mark.parametrize("taps", [True, False])
@pytest.mark.parametrize("n_kp,max_res", [(16_384, INT_MAX), (3_000, INT_MAX), (16_384, 600)])
def test_passes_with_and_without_the_bounds_agree_bit_for_bit(scene, n_kp, max_res, taps):
    ctx = scene["ctx"]
    sw = synth.make_sweep(4402 + n_kp, n_kp, scene["L"])
    opts = srl.default_opts(max_num_residuals=max_res)
    poses = _poses(sw, 5, 1, jump=3)
    ctx.set_armed_launch(0)
    ctx.set_bound_culling(0)
    ctx.sweep_upload(sw["raw"])
    ref = _passes(ctx, sw, poses, opts, taps)
    for armed in (0, 2):
        ctx.set_armed_launch(armed)
        ctx.set_bound_culling(1)
        ctx.sweep_upload(sw["raw"])
        got = _passes(ctx, sw, poses, opts, taps)
        for k, (g, r) in enumerate(zip(got, ref)):
            _same(g, r, (armed, k))
    ctx.set_armed_launch(1)


def test_init_mode_passes_agree_bit_for_bit(scene):
    """frame_id < init_num_frames: r = 2 (125 voxels probed, ~45 occupied), the looped fast path -- with the bounds a pass visits the few
    voxels within reach of the K nearest; ids, candidate counts and sums equal the passes without them"""
    ctx = scene["ctx"]
    sw = synth.make_sweep(4477, 6_000, scene["L"])
    opts = srl.default_opts(max_num_residuals=INT_MAX)
    poses = _poses(sw, 4, 5, jump=2)

    def run():
        ctx.sweep_upload(sw["raw"])
        ctx.set_taps(True)
        res = []
        for q, t in poses:
            neq, rc = ctx.build_residuals(capi.make_frame(q, t, sw["t_last"], frame_id=5), opts)
            res.append((neq, ctx.fetch_neighbors()[0].copy()))
        ctx.set_taps(False)
        return res
    ctx.set_armed_launch(0)
    ctx.set_bound_culling(0)
    ref = run()
    ctx.set_bound_culling(1)
    got = run()
    for k, (g, r) in enumerate(zip(got, ref)):
        _same(g, r, k)
    ctx.set_armed_launch(1)


def test_a_map_change_or_another_sweep_voids_the_bounds(scene):
    """points inserted right next to keypoints between two passes (their K-th neighbour moves closer AND voxels that were empty fill up);
    another sweep uploaded over the old one; the same sweep through prefetch / swap: every pass equals a context that never had bounds"""
    ctx = scene["ctx"]
    fresh = srl.Context(0)
    try:
        fresh.set_bound_culling(0)
        fresh.map_insert(scene["cands"])
        opts = srl.default_opts(max_num_residuals=INT_MAX)
        sw = synth.make_sweep(4499, 8_192, scene["L"])
        sw2 = synth.make_sweep(4498, 8_192, scene["L"])
        poses = _poses(sw, 3, 2)
        ctx.set_bound_culling(1)
        for c in (ctx, fresh):
            c.sweep_upload(sw["raw"])
        a = _passes(ctx, sw, poses[:2], opts, True)
        b = _passes(fresh, sw, poses[:2], opts, True)
        for k in range(2):
            _same(a[k], b[k], ("before", k))
        R = synth.quat_to_rot(poses[1][0] / np.linalg.norm(poses[1][0]))
        world = sw["raw"][::7] @ R.T + poses[1][1]
        extra = np.concatenate([world + np.array([0.0, 0.0, 0.02]), world + np.array([1.2, 0.0, 0.6])])      # next to keypoints, and in voxels beside them
        for c in (ctx, fresh):
            c.map_insert(extra)
        a = _passes(ctx, sw, poses[2:] + poses[:1], opts, True)
        b = _passes(fresh, sw, poses[2:] + poses[:1], opts, True)
        for k in range(2):
            _same(a[k], b[k], ("after insert", k))
        for c in (ctx, fresh):
            c.sweep_upload(sw2["raw"])                      # same keypoint count, other points
        p2 = _poses(sw2, 2, 3)
        a = _passes(ctx, sw2, p2, opts, True)
        b = _passes(fresh, sw2, p2, opts, True)
        for k in range(2):
            _same(a[k], b[k], ("other sweep", k))
        pin = srl.PinnedArray(sw["raw"].shape); pin.array[:] = sw["raw"]
        for c in (ctx, fresh):
            c.sweep_prefetch(pin.array); c.sweep_swap()
        a = _passes(ctx, sw, poses, opts, False)
        b = _passes(fresh, sw, poses, opts, False)
        for k in range(3):
            _same(a[k], b[k], ("swapped in", k))
        ctx.disarm(); fresh.disarm()
        pin.close()
    finally:
        fresh.close()


def test_tied_distances_with_bounds(golden):
    """the lattice scene (86 % of the keypoints tied across the cut: settled by the heap replay, which visits everything itself) over three
    passes with the same pose: ids equal to the goldens of the reference's own translation units on every pass"""
    ctx = srl.Context(0)
    try:
        ctx.map_upload(golden["tie_map_keys"], golden["tie_map_counts"], golden["tie_map_xyz"])
        ctx.sweep_upload(golden["tie_raw"])
        f = capi.make_frame(golden["tie_q"], golden["tie_t"], golden["tie_t_last"])
        ctx.set_taps(True)
        for _ in range(3):
            neq, rc = ctx.build_residuals(f, srl.default_opts(max_num_residuals=INT_MAX))
            ids = ctx.fetch_neighbors()[0]
            assert np.array_equal(ids, golden["tie_one_ids"]) and neq.num_residuals == int(golden["tie_one_num_residuals"])
            assert neq.sum_candidates == int(golden["tie_one_sum_candidates"])
    finally:
        ctx.close()


@pytest.mark.parametrize("max_res", [INT_MAX, 600])
def test_full_solves_agree_bit_for_bit(max_res):
    """updateIEKF on the C1 configuration, stream of sweeps included: states and covariances with the bounds equal those without"""
    n_kp, map_pts, pattern, seed = synth.CONFIGS["C1"]
    cands, L = synth.map_candidates(seed, map_pts)
    out = {}
    for mode in (0, 1):
        lio = srl.Lio(0)
        try:
            lio.ctx.set_bound_culling(mode)
            lio.add_points_to_map(cands)
            states = []
            for j in range(3):
                sw = synth.make_sweep(seed + 1000 + j, n_kp, L, pattern=pattern)

                class A:
                    def set_noise(s, *a): lio.eskf_set_noise(*a)
                    def scale_init_cov(s): lio.eskf_scale_init_cov()
                    def init_imu(s, a, g): lio.eskf_init_imu(a, g)
                    def predict(s, dt, a, g): lio.eskf_predict(dt, a, g)
                    def get_state(s): return lio.eskf_get_state()
                    def set_state(s, x): lio.eskf_set_state(x)
                synth.eskf_prior(A(), sw["q_pred"], sw["t_pred"], sw["vel"])
                st = np.concatenate([sw["q_pred"], sw["t_pred"], sw["vel"], np.zeros(6)])
                g = lio.update_iekf(srl.default_opts(max_num_residuals=max_res), sw["raw"], st, sw["t_last"])
                assert g["rc"] == 0 and g["iters"] >= 2
                states.append((g["iters"], g["num_residuals"], g["state"].copy(), lio.eskf_get_cov().copy()))
            out[mode] = states
        finally:
            lio.close()
    for a, b in zip(out[0], out[1]):
        assert a[0] == b[0] and a[1] == b[1] and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])
