"""Pins the CPU oracle (oracle/srl_oracle.cpp) on a CPU-only box.

The reference ships no tests / golden vectors for this path.  The oracle is pinned bitwise against
the reference's own translation units compiled in place (tests/test_reference_tu.py); this file holds the checks
that do not need them: the known-answer values probed from the reference's own definitions,
analytic cases, an independent NumPy/SciPy re-implementation (tests/np_reference.py), the build against the real
vendored tsl::robin_map (oracle/_ref), and the committed golden vectors.
"""
import os

import numpy as np
import pytest
from scipy.spatial.transform import Rotation

from oracle import pyoracle as po
import sr_livo_amd as srl
from sr_livo_amd import synth

import np_reference as npr

INT_MAX = 2**31 - 1


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


# ------------------------------------------------------------------ known answers
def test_voxel_hash_known_answers():
    lib = po.load()
    kat = {(0, 0, 0): 0, (1, 1, 1): 176698553, (-1, 2, 3): 215321618, (-1, -1, -1): 18446744073532853063,
           (100, -200, 3): 3766153873, (32767, -32768, -1): 1785909152748}
    for (x, y, z), h in kat.items():
        assert lib.orc_voxel_hash(x, y, z) == h


def test_voxel_key_truncates_toward_zero():
    lib = po.load()
    for v, k in [(-1.999, -1), (-1.0, -1), (-0.999, 0), (-0.0, 0), (0.999, 0), (1.0, 1), (1.5, 1), (-32.7, -32)]:
        assert lib.orc_voxel_coord(v, 1.0) == k
    assert lib.orc_voxel_coord(2.9, 1.5) == 1 and lib.orc_voxel_coord(-2.9, 1.5) == -1


def test_bounded_heap_tie_behaviour_is_libstdcxx():
    """optimize.cpp:397-404 with K = 4 and distances 1,2,2,2,2,0.5,2,1.5 (x 0.125): the literal
    std::priority_queue keeps visit #3 among the ties, and the tie is flagged."""
    m = po.Map()
    q = np.array([0.5, 0.5, 0.5])
    offs = [(0.125, 0, 0), (0.25, 0, 0), (-0.25, 0, 0), (0, 0.25, 0), (0, -0.25, 0), (0, 0, 0.0625), (0, 0, 0.25), (0, 0, -0.1875)]
    pts = np.array([q + np.array(o) for o in offs])
    assert m.add_points(pts, min_dist=0.01) == 8 and m.num_voxels() == 1
    r = m.search_neighbors(q, K=4)
    assert r["n"] == 4 and r["tie"] and r["num_candidates"] == 8
    assert list(r["ids"]) == [5, 0, 7, 3]
    assert np.allclose(r["dist"], [0.0625, 0.125, 0.1875, 0.25], rtol=0, atol=0)


def test_map_insert_rules():
    """lioOptimization.cpp:400-446: FP32 storage, first point unconditional, min-distance, cap, min_num_points."""
    m = po.Map()
    assert m.add_points([[0.1, 0.1, 0.1]]) == 1
    assert m.add_points([[0.1 + 0.15, 0.1, 0.1]]) == 0                 # not strictly farther than 0.15 (FP32 rounding)
    assert m.add_points([[0.1 + 0.1501, 0.1, 0.1]]) == 1
    assert m.add_points([[-0.4, 0.2, -0.3]]) == 1                      # voxel (0,0,0) is (-1,1)^3: same voxel
    assert m.num_voxels() == 1 and m.size() == 3
    assert m.add_points([[5.5, 5.5, 5.5]], min_num_points=1) == 0      # never creates a voxel when min_num_points > 0
    rng = np.random.default_rng(0)
    many = np.column_stack([rng.uniform(2.0, 3.0, 4000), rng.uniform(2.0, 3.0, 4000), rng.uniform(2.0, 3.0, 4000)])
    m.add_points(many, min_dist=0.0)
    keys, counts, xyz = m.export()
    assert counts.max() == 20 and m.size() == counts.sum()           # cap
    k = int(np.where((keys == [2, 2, 2]).all(1))[0][0])
    assert np.array_equal(xyz[k, 0], many[0].astype(np.float32))       # first come, FP32


# ------------------------------------------------------------------ third-party arithmetic restated (Appendix C)
def test_quaternion_and_so3_helpers_against_scipy():
    lib = po.load()
    rng = np.random.default_rng(1)
    for _ in range(200):
        w = rng.normal(size=3) * rng.choice([1e-6, 1e-3, 0.3, 2.0])
        R = np.empty(9); lib.orc_so3_to_rot(po._dp(w), po._dp(R))
        assert rel(R.reshape(3, 3), Rotation.from_rotvec(w).as_matrix()) < 1e-7 + 1e-2 * (np.linalg.norm(w) < 1e-4) * np.linalg.norm(w) ** 2
        q = np.empty(4); lib.orc_so3_to_quat(po._dp(w), po._dp(q))
        qs = Rotation.from_rotvec(w).as_quat()                         # x y z w
        assert rel(q, [qs[3], qs[0], qs[1], qs[2]]) < 1e-7
        R2 = np.empty(9); lib.orc_quat_to_rot(po._dp(q), po._dp(R2))
        assert rel(R2.reshape(3, 3), Rotation.from_rotvec(w).as_matrix()) < 1e-7
        if np.linalg.norm(w) < 3.0:
            w2 = np.empty(3); lib.orc_rot_to_so3(po._dp(R2), po._dp(w2))
            assert np.linalg.norm(w2 - w) < 1e-6 * max(1.0, np.linalg.norm(w))
        q3 = np.empty(4); lib.orc_rot_to_quat(po._dp(R2), po._dp(q3))
        assert min(np.linalg.norm(q3 - q), np.linalg.norm(q3 + q)) < 1e-9
    # un-normalised quaternion -> toRotationMatrix does NOT normalise (Appendix B.10)
    q = np.array([0.9, 0.1, -0.3, 0.2]); R = np.empty(9); lib.orc_quat_to_rot(po._dp(q), po._dp(R))
    assert rel(R.reshape(3, 3), npr.quat_to_rot(q)) < 1e-15
    assert abs(np.linalg.det(R.reshape(3, 3)) - 1.0) > 1e-3
    # AngularDistance (degrees)
    w = np.array([0.0, 0.0, 0.01])
    assert abs(lib.orc_angular_distance_so3(po._dp(w)) - np.degrees(0.01)) < 1e-6


def test_eig3_and_inverse17_against_numpy():
    lib = po.load()
    rng = np.random.default_rng(2)
    for _ in range(300):
        P = rng.normal(size=(20, 3)) * rng.uniform(0.01, 2.0, 3)
        A = (P - P.mean(0)).T @ (P - P.mean(0))
        ev = np.empty(3); V = np.empty(9)
        lib.orc_eig3(po._dp(np.ascontiguousarray(A)), po._dp(ev), po._dp(V))
        w, U = np.linalg.eigh(A)
        assert rel(ev, w) < 1e-12
        V = V.reshape(3, 3)
        for c in range(3):
            assert abs(abs(V[:, c] @ U[:, c]) - 1.0) < 1e-9
    lib.orc_eig3(po._dp(np.zeros(9)), po._dp(ev), po._dp(V.ravel()))
    assert np.all(ev == 0)
    for _ in range(20):
        B = rng.normal(size=(17, 17)); A = B @ B.T + np.eye(17) * 0.1
        Ai = np.empty(289)
        assert lib.orc_inverse17(po._dp(np.ascontiguousarray(A).ravel()), po._dp(Ai)) == 0
        assert rel(Ai.reshape(17, 17), np.linalg.inv(A)) < 1e-9


# ------------------------------------------------------------------ analytic cases
def test_exact_plane_gives_plane_normal_and_signed_offset():
    lib = po.load()
    rng = np.random.default_rng(3)
    n = np.array([0.3, -0.5, 0.81]); n /= np.linalg.norm(n)
    u = np.cross(n, [1, 0, 0]); u /= np.linalg.norm(u); v = np.cross(n, u)
    pts = np.array([2.0, 1.0, 0.5]) + np.outer(rng.uniform(-1, 1, 20), u) + np.outer(rng.uniform(-0.3, 0.3, 20), v)
    c = np.empty(3); nn = np.empty(3); cov = np.empty(9); a2d = po.C.c_double(); ev = np.empty(3)
    assert lib.orc_neighborhood(po._vp(np.ascontiguousarray(pts)), 20, po._dp(c), po._dp(nn), po._dp(cov), po.C.byref(a2d), po._dp(ev)) == 0
    assert abs(abs(nn @ n) - 1.0) < 1e-12 and rel(c, pts.mean(0)) < 1e-15
    assert abs(ev[0]) < 1e-12 * ev[2]
    assert abs(a2d.value - (np.sqrt(ev[1]) - np.sqrt(abs(ev[0]))) / np.sqrt(ev[2])) < 1e-15
    assert rel(cov.reshape(3, 3), (pts - pts.mean(0)).T @ (pts - pts.mean(0))) < 1e-12   # NOT divided by K (Appendix B.8)


def test_identity_pose_on_planar_map_has_zero_gradient_direction():
    """Keypoints sampled exactly on the (noise-free) ground of the map at the true pose: distances ~ 0,
    so H^T h is negligible against H^T H (SURVEY 8(c) analytic case)."""
    g = np.arange(-6, 6, 0.2)
    X, Y = np.meshgrid(g, g)
    ground = np.column_stack([X.ravel() + 0.03, Y.ravel() + 0.07, np.full(X.size, -1.7)])
    m = po.Map(); m.add_points(ground)
    kp = np.column_stack([np.linspace(-3, 3, 200), np.linspace(-2.5, 2.9, 200), np.full(200, -1.7)])
    o = m.build_plane_residuals(po.default_opts(max_num_residuals=INT_MAX), kp, [1, 0, 0, 0], [0, 0, 0], [0, 0, 5.0])
    assert (o["status"] == 2).all()
    assert np.abs(o["distance"]).max() < 1e-6
    assert np.abs(np.abs(o["normal"][:, 2]) - 1).max() < 1e-9
    assert np.abs(o["Hth"]).max() < 1e-6 * np.abs(o["HtH"]).max()


# ------------------------------------------------------------------ independent NumPy implementation
@pytest.mark.parametrize("frame_id,max_res", [(100, INT_MAX), (100, 150), (5, INT_MAX)])
def test_build_plane_residuals_vs_numpy(small_scene, frame_id, max_res):
    m, sw = small_scene["map"], small_scene["sweep"]
    raw = sw["raw"][:600]
    R_il = synth.quat_to_rot(synth.quat_from_rotvec([0.02, -0.01, 0.03])); t_il = np.array([0.05, -0.02, 0.01])
    o = m.build_plane_residuals(po.default_opts(max_num_residuals=max_res), raw, sw["q_pred"] * 1.0003, sw["t_pred"], sw["t_last"],
                                R_il=R_il, t_il=t_il, frame_id=frame_id)
    keys, counts, xyz = m.export()
    ref = npr.build_plane_residuals(keys, counts, xyz, raw, sw["q_pred"] * 1.0003, sw["t_pred"], sw["t_last"], R_il, t_il,
                                    frame_id=frame_id, max_num_residuals=max_res)
    assert o["neq"].num_ties == 0
    assert np.array_equal(o["status"], ref["status"])
    vis = ref["status"] != 3
    assert np.array_equal(o["ids"][vis], ref["ids"][vis])
    hp = (ref["status"] == 1) | (ref["status"] == 2)
    for k in ("a2D", "weight", "norm_offset", "distance"):
        assert rel(o[k][hp], ref[k][hp]) < 1e-9, k
    assert rel(o["normal"][hp], ref["normal"][hp]) < 1e-9
    acc = ref["status"] == 2
    assert rel(o["jacobian"][acc], ref["jacobian"][acc]) < 1e-9
    assert rel(o["HtH"], ref["HtH"]) < 1e-10 and rel(o["Hth"], ref["Hth"]) < 1e-9
    assert o["neq"].num_residuals == ref["num_residuals"] and rel(o["neq"].loss_sum, ref["loss"]) < 1e-10


def test_update_iekf_vs_numpy(small_scene):
    m, sw = small_scene["map"], small_scene["sweep"]
    raw = sw["raw"][:800]
    opts = po.default_opts(max_num_residuals=INT_MAX)
    e = po.Eskf(); synth.eskf_prior(e, sw["q_pred"], sw["t_pred"], sw["vel"])
    s0, P0 = e.get_state().copy(), e.get_cov().copy()
    st = np.concatenate([sw["q_pred"], sw["t_pred"], sw["vel"], np.zeros(6)])
    u = po.update_iekf(m, e, opts, raw, st, sw["t_last"], log_iters=10)
    keys, counts, xyz = m.export()
    ref = npr.update_iekf(keys, counts, xyz, raw, s0, P0, st, sw["t_last"], max_iter=5)
    assert u["rc"] == ref["iters"] >= 2
    assert rel(u["log"][:, 42:59], np.array(ref["dx"])) < 1e-7
    assert rel(u["state"], ref["state"]) < 1e-9
    assert rel(e.get_state(), ref["eskf_state"]) < 1e-9
    assert rel(e.get_cov(), ref["eskf_cov"]) < 1e-7


def test_eskf_predict_vs_numpy():
    e = po.Eskf()
    e.set_noise(0.1, 0.1, 1e-4, 1e-4); e.scale_init_cov()
    s = e.get_state(); s[3:7] = synth.quat_from_rotvec([0.1, -0.2, 0.05]); s[7:10] = [0.3, -0.1, 0.05]; s[10:13] = [0.01, 0.02, -0.01]; s[13:16] = [1e-3, -2e-3, 5e-4]
    e.set_state(s)
    e.init_imu([0.1, 0.2, 9.7], [0.01, -0.02, 0.03])
    P = e.get_cov().copy(); st = e.get_state().copy()
    acc1, gyr1 = np.array([0.15, 0.1, 9.9]), np.array([0.02, 0.0, -0.01])
    e.predict(0.01, acc1, gyr1)
    ref_s, ref_P = npr.eskf_predict(st, P, 0.01, np.array([0.1, 0.2, 9.7]), np.array([0.01, -0.02, 0.03]), acc1, gyr1,
                                    np.diag([0.1] * 6 + [1e-4] * 6))
    assert rel(e.get_state(), ref_s) < 1e-12 and rel(e.get_cov(), ref_P) < 1e-12


# ------------------------------------------------------------------ real tsl::robin_map build + golden regression
def test_tsl_backend_equals_plain_backend(small_scene):
    if not os.path.exists(po.LIB_TSL):
        pytest.skip("oracle/_ref not built (reference tree absent)")
    assert po.load("tsl").orc_map_backend() == b"tsl::robin_map"
    pts = small_scene["candidates"]; sw = small_scene["sweep"]
    a, b = po.Map("plain"), po.Map("tsl")
    assert a.add_points(pts) == b.add_points(pts)
    for x, y in zip(a.export(), b.export()):
        assert np.array_equal(x, y)
    oa = a.build_plane_residuals(po.default_opts(max_num_residuals=600), sw["raw"], sw["q_pred"], sw["t_pred"], sw["t_last"])
    ob = b.build_plane_residuals(po.default_opts(po.load("tsl"), max_num_residuals=600), sw["raw"], sw["q_pred"], sw["t_pred"], sw["t_last"])
    for k in ("status", "ids", "distance", "jacobian", "HtH", "Hth"):
        assert np.array_equal(oa[k], ob[k]), k


@pytest.mark.parametrize("prefix,frame_id,max_res", [("full", 100, INT_MAX), ("cut600", 100, 600), ("init", 5, INT_MAX), ("neg1", 100, -1)])
def test_oracle_reproduces_golden_vectors(golden, small_scene, oracle_backend, prefix, frame_id, max_res):
    m = small_scene["map"]
    keys, counts, xyz = m.export()
    assert np.array_equal(keys, golden["map_keys"]) and np.array_equal(counts, golden["map_counts"]) and np.array_equal(xyz, golden["map_xyz"])
    assert np.array_equal(small_scene["sweep"]["raw"], golden["raw"])
    opts = po.default_opts(max_num_residuals=max_res)
    o = m.build_plane_residuals(opts, golden["raw"], golden["q_pred"], golden["t_pred"], golden["t_last"], frame_id=frame_id)
    for k in ("status", "ids", "normal", "distance", "weight", "jacobian", "HtH", "Hth"):
        assert np.array_equal(o[k], golden[f"{prefix}_one_{k}"]), k
    e = po.Eskf(oracle_backend); e.set_state(golden[f"{prefix}_eskf_state0"]); e.set_cov(golden[f"{prefix}_eskf_cov0"])
    u = po.update_iekf(m, e, opts, golden["raw"], golden[f"{prefix}_state0"], golden["t_last"], frame_id=frame_id, log_iters=20)
    assert u["rc"] == int(golden[f"{prefix}_solve_rc"])
    assert np.array_equal(u["state"], golden[f"{prefix}_solve_state"])
    assert np.array_equal(e.get_cov(), golden[f"{prefix}_solve_eskf_cov"])


# ----------------------------------------------------------------------------- frame side (utility.cpp:167-201,314-318)
def test_transform_points_and_grid_sampling(oracle_lib):
    rng = np.random.default_rng(77)
    raw = rng.uniform(-30, 30, (20000, 3))
    q = np.array([0.9, 0.1, -0.2, 0.3]) * 1.01                # un-normalised on purpose
    t = np.array([1.0, -2.0, 0.5]); t_il = np.array([0.1, 0.2, -0.05])
    a = 0.3; R_il = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1.0]])
    world = oracle_lib.transform_points(raw, q, t, R_il, t_il)
    w, x, y, z = q                                             # Eigen's toRotationMatrix on the raw coefficients
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                  [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                  [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    assert np.allclose(world, (raw @ R_il.T + t_il) @ R.T + t, rtol=0, atol=1e-12)
    for size in (0.5, 1.5):
        idx = oracle_lib.grid_sampling(world, size)
        keys = np.trunc(world / size).astype(np.int64)         # static_cast<short> truncates toward zero
        first = {}
        for i, k in enumerate(map(tuple, keys)):
            first.setdefault(k, i)
        assert sorted(idx.tolist()) == sorted(first.values())  # first point of every voxel, nothing else
        assert np.array_equal(idx, oracle_lib.grid_sampling(world, size))
        assert np.array_equal(idx, srl.grid_sampling(world, size))      # host mirror uses the same container type
    assert len(oracle_lib.grid_sampling(np.zeros((0, 3)), 1.0)) == 0


# ----------------------------------------------------------------------------- sweep reconstruction (row f4)
def _imu_track(rng, S, t0, dt):
    """S imu states along a smooth motion; rows of 17 doubles."""
    st = np.zeros((S, 17))
    q = Rotation.from_rotvec([0.1, -0.05, 0.3])
    p = np.array([1.0, 2.0, 0.3]); v = np.array([1.5, -0.4, 0.1])
    for k in range(S):
        st[k, 0] = t0 + k * dt
        st[k, 1:4] = rng.normal(0, 0.5, 3)                     # un_acc (world, gravity-free here)
        st[k, 4:7] = rng.normal(0, 0.3, 3)                     # un_gyr
        st[k, 7:10] = p
        x, y, z, w = q.as_quat(); st[k, 10:14] = [w, x, y, z]
        st[k, 14:17] = v
        q = q * Rotation.from_rotvec(st[k, 4:7] * dt); p = p + v * dt; v = v + st[k, 1:4] * dt
    return st


def test_mt19937_64_known_answer(oracle_lib):
    # C++11 [rand.predef]: the 10000th consecutive invocation of a default-constructed mt19937_64
    # (boost::mt19937_64 documents the same check value)
    assert oracle_lib.mt19937_64_nth(10000) == 9981545732273789042


def test_distort_by_constant_velocity_vs_scipy_slerp(oracle_lib):
    from scipy.spatial.transform import Slerp
    rng = np.random.default_rng(3)
    st = _imu_track(rng, 12, 50.0, 0.01)
    n = 4000
    raw = rng.uniform(-20, 20, (n, 3)); rel = np.sort(rng.uniform(0, 110.0, n)); rel[0] = 0.0; rel[-1] = 110.0
    R_il = Rotation.from_rotvec([0.01, 0.02, -0.03]).as_matrix(); t_il = np.array([0.05, -0.02, 0.1])
    imu, k = oracle_lib.distort_frame(raw, rel, st, 50.0, 1, R_il, t_il)
    assert k == n
    tb, te = 50.0, st[-1, 0]
    tp = tb + rel / 1000.0
    tp = np.where(np.abs(tp - tb) < 1e-6, tb + 1e-6, tp); tp = np.where(np.abs(tp - te) < 1e-6, te - 1e-6, tp)
    a = np.clip((tp - tb) / (te - tb), 0, 1)
    wxyz = st[[0, -1], 10:14]
    sl = Slerp([0, 1], Rotation.from_quat(wxyz[:, [1, 2, 3, 0]]))
    Rm = sl(a).as_matrix()
    want = np.einsum("nij,nj->ni", Rm, raw @ R_il.T + t_il) + (1 - a)[:, None] * st[0, 7:10] + a[:, None] * st[-1, 7:10]
    assert np.max(np.abs(imu - want)) < 1e-11
    # transformAllImuPoint: back into the lidar frame at the sweep end
    back = oracle_lib.transform_all_imu_point(imu, st, R_il, t_il)
    Re = Rotation.from_quat(st[-1, [11, 12, 13, 10]]).as_matrix()
    want_raw = ((imu - st[-1, 7:10]) @ Re - t_il) @ R_il
    assert np.max(np.abs(back - want_raw)) < 1e-11
    # points at the very end of the sweep are their own correction
    last = rel >= (te - tb) * 1000.0 - 1e-3
    assert np.max(np.abs(back[last] - raw[last])) < 1e-4


def test_distort_by_imu_interval_walk(oracle_lib):
    rng = np.random.default_rng(4)
    st = _imu_track(rng, 9, 10.0, 0.0125)
    n = 3000
    raw = rng.uniform(-15, 15, (n, 3)); rel = np.sort(rng.uniform(0, 100.0, n))
    rel[5] = 12.5; rel[6] = 12.5 + 5e-4                      # on / within 1e-6 of an interval boundary
    rel = np.sort(rel)
    imu, k = oracle_lib.distort_frame(raw, rel, st, 10.0, 0)
    assert k == n
    # independent: interval of each point, first-order propagation from its start state
    tp = 10.0 + rel / 1000.0
    seg = np.clip(np.searchsorted(st[:, 0], tp, side="right") - 1, 0, len(st) - 2)
    want = np.empty_like(raw)
    for i in range(n):
        a, b = st[seg[i]], st[seg[i] + 1]
        t = tp[i]
        if abs(t - a[0]) < 1e-6: t = a[0] + 1e-6
        if abs(t - b[0]) < 1e-6: t = b[0] - 1e-6
        dt = t - a[0]
        Rq = Rotation.from_quat(a[[11, 12, 13, 10]]) * Rotation.from_rotvec(b[4:7] * dt)
        want[i] = Rq.apply(raw[i]) + a[7:10] + a[14:17] * dt + 0.5 * b[1:4] * dt * dt
    bad = np.abs(imu - want).max(axis=1) > 1e-9
    # only points within 1e-6 s of a boundary may legitimately sit in the neighbouring interval
    near = np.min(np.abs(tp[:, None] - st[None, :, 0]), axis=1) < 2e-6
    assert not np.any(bad & ~near)
    # a point that goes back in time stops the walk: everything behind it keeps its old imu_point
    rel2 = rel.copy(); rel2[1000] = rel2[10]
    sentinel = np.full_like(raw, 7.0)
    imu2, k2 = oracle_lib.distort_frame(raw, rel2, st, 10.0, 0, imu_point_in=sentinel)
    assert k2 == 1000 and np.array_equal(imu2[1000:], sentinel[1000:]) and np.array_equal(imu2[:1000], imu[:1000])


def test_build_frame_order_and_point_timestamps(oracle_lib):
    rng = np.random.default_rng(8)
    pts = rng.uniform(-10, 10, (6000, 3))
    full = oracle_lib.build_frame_order(pts, 0.5, do_subsample=False)
    assert sorted(full.tolist()) == list(range(6000)) and not np.array_equal(full, np.arange(6000))
    sub = oracle_lib.build_frame_order(pts, 0.5)
    # one point per voxel: the first of the voxel in the shuffled order
    pos = np.empty(6000, int); pos[full] = np.arange(6000)
    keys = np.trunc(pts / 0.5).astype(int)
    first = {}
    for i in full:
        first.setdefault(tuple(keys[i]), i)
    assert sorted(sub.tolist()) == sorted(first.values())
    assert np.array_equal(sub, oracle_lib.build_frame_order(pts, 0.5))
    ts = np.array([9.99, 10.0, 10.05, 10.1, 10.11])
    rel, alpha, keep = oracle_lib.make_point_timestamp(ts, 10.0, 10.1, True)
    assert keep.all() and np.allclose(rel, (ts - 10.0) * 1000) and alpha[-1] == 1.0 - 1e-5 and alpha[0] < 0
    rel, alpha, keep = oracle_lib.make_point_timestamp(ts, 10.0, 10.1, False)
    assert keep.tolist() == [False, True, True, True, False] and np.allclose(alpha[1:4], [0, 0.5, 1.0])
