"""The oracle against the REFERENCE'S OWN translation units (CPU; row (c) of SURVEY.md 8).

oracle/_ref/libref_path.so = /root/reference/src/{optimize,lioOptimization,eskfEstimator,utility,state,cloudMap,parameters}.cpp
compiled where they lie behind a C ABI (oracle/ref_harness.cpp; third-party headers the image lacks are the stand-ins of oracle/ref_shim/, the voxel
map is the real vendored tsl::robin_map).  Two kinds of tests:

  * golden: tests/golden/golden_ref_tu.npz holds outputs of that library on the scenes of golden_small.npz
    (tests/golden/make_golden_ref.py).  The oracle must reproduce them BITWISE -- these run wherever the repository is,
    including the GPU box where neither /root/reference nor the library's sources exist.
  * live (skipped where the prebuilt library is absent): restatement and reference side by side on more inputs --
    searchNeighbors on the tie scene (the neighbour list the real std::priority_queue leaves), computeNeighborhoodDistribution
    incl. the NaN throw, eskfEstimator::predict / observe / tryInit, gridSampling, transformPoint, distortFrameBy*,
    transformAllImuPoint, numType helpers, optimize(), km-scale coordinates, the truncation seam, empty sweeps, and the node
    itself: its own constructor / readParameters / imuHandler / getMeasurements / run / process / buildFrame / stateEstimation
    / addPointsToMap on 40 sweeps of sensor streams, stateInitialization, makePointTimestamp.

Bitwise everywhere: with the same third-party arithmetic underneath, the restatement and the reference's source must not
differ in a single operation.
"""
import os

import numpy as np
import pytest

from oracle import pyoracle as po
from oracle import pyref as pr
from sr_livo_amd import synth

INT_MAX = 2**31 - 1
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
live = pytest.mark.skipif(not pr.available(), reason="oracle/_ref/libref_path.so not built (needs /root/reference at build time)")


@pytest.fixture(scope="module")
def gref():
    return dict(np.load(os.path.join(ROOT, "tests", "golden", "golden_ref_tu.npz"), allow_pickle=False))


def accepted(o):
    return o["status"] == 2


def assert_pass_equals(o, r):
    """o: pyoracle build_plane_residuals (per keypoint), r: reference (accepted list in push order)."""
    acc = accepted(o)
    assert int(acc.sum()) == len(r["distance"])
    assert np.array_equal(o["point_world"], r["point_world"])
    for k in ("normal", "jacobian", "norm_offset", "distance", "weight"):
        assert np.array_equal(o[k][acc], r[k]), k
    assert o["neq"].num_residuals == r["num_residuals"] and o["neq"].success == r["success"]
    assert o["neq"].loss_sum == r["loss"]


# ----------------------------------------------------------------------------- golden (no reference library needed)
@pytest.mark.parametrize("prefix,frame_id,max_res", [("full", 100, INT_MAX), ("cut600", 100, 600), ("init", 5, INT_MAX), ("neg1", 100, -1)])
def test_oracle_reproduces_reference_tu_goldens_bitwise(golden, gref, oracle_backend, prefix, frame_id, max_res):
    m = po.Map(oracle_backend)
    m.import_(golden["map_keys"], golden["map_counts"], golden["map_xyz"])
    opts = po.default_opts(max_num_residuals=max_res)
    o = m.build_plane_residuals(opts, golden["raw"], golden["q_pred"], golden["t_pred"], golden["t_last"], frame_id=frame_id)
    r = {k[len(prefix) + 5:]: v for k, v in gref.items() if k.startswith(prefix + "_ref_")}
    r["success"] = int(r["success"]); r["num_residuals"] = int(r["num_residuals"]); r["loss"] = float(r["loss"])
    assert_pass_equals(o, r)
    # the location the reference stores per residual (R_il raw + t_il; identity extrinsics in this scene) = the raw point
    assert np.array_equal(golden["raw"][accepted(o)], r["location"])
    e = po.Eskf(oracle_backend)
    e.set_state(golden[f"{prefix}_eskf_state0"]); e.set_cov(golden[f"{prefix}_eskf_cov0"])
    u = po.update_iekf(m, e, opts, golden["raw"], golden[f"{prefix}_state0"], golden["t_last"], frame_id=frame_id)
    assert (u["rc"] > 0) == (int(r["solve_rc"]) == 1) and u["num_residuals"] == int(r["solve_num_residuals"])
    assert np.array_equal(u["state"], r["solve_state"])
    assert np.array_equal(e.get_state(), r["solve_eskf_state"]) and np.array_equal(e.get_cov(), r["solve_eskf_cov"])


@pytest.mark.parametrize("prefix,kw,frame_id", [("tie", {}, 100), ("tie5", dict(max_number_neighbors=5, min_number_neighbors=5), 100), ("tieinit", {}, 5)])
def test_oracle_reproduces_reference_tu_tie_neighbours(golden, gref, oracle_backend, prefix, kw, frame_id):
    """Neighbour lists on the tie scene: which tied candidates survive, and in which order, is libstdc++'s heap inside the
    reference's searchNeighbors (src/optimize.cpp:394-422)."""
    m = po.Map(oracle_backend)
    m.import_(golden["tie_map_keys"], golden["tie_map_counts"], golden["tie_map_xyz"])
    opts = po.default_opts(max_num_residuals=INT_MAX, **kw)
    o = m.build_plane_residuals(opts, golden["tie_raw"], golden["tie_q"], golden["tie_t"], golden["tie_t_last"], frame_id=frame_id)
    r = {k[len(prefix) + 5:]: v for k, v in gref.items() if k.startswith(prefix + "_ref_")}
    r["success"] = int(r["success"]); r["num_residuals"] = int(r["num_residuals"]); r["loss"] = float(r["loss"])
    assert_pass_equals(o, r)
    assert o["neq"].num_ties > 1000                                   # the scene is there for its ties
    xyz = golden["tie_map_xyz"].reshape(-1, 3)                        # id = voxel * cap + slot
    K = opts.max_number_neighbors
    cnt = r["num_neighbors"]
    assert np.array_equal((o["ids"] >= 0).sum(axis=1), cnt)
    for i in range(len(cnt)):
        assert np.array_equal(xyz[o["ids"][i, : cnt[i]]], r["neighbors"][i, : cnt[i]]), i
    assert K == r["neighbors"].shape[1]


# ----------------------------------------------------------------------------- live: restatement and reference side by side
@pytest.fixture(scope="module")
def scene(small_scene):
    return dict(om=small_scene["map"], rm=pr.Map.from_oracle(small_scene["map"]), sweep=small_scene["sweep"], L=small_scene["L"])


@live
def test_voxel_hash_is_the_reference_functor():
    lib = pr.load(); olib = po.load()
    rng = np.random.default_rng(1)
    for x, y, z in rng.integers(-32768, 32768, (2000, 3)).tolist() + [[0, 0, 0], [-1, -1, -1], [32767, -32768, 1]]:
        assert lib.ref_voxel_hash(x, y, z) == olib.orc_voxel_hash(x, y, z)


@live
@pytest.mark.parametrize("frame_id,max_res,kw", [(100, INT_MAX, {}), (100, 600, {}), (100, 37, {}), (5, INT_MAX, {}), (100, -1, {}), (100, 0, {}),
                                                  (100, INT_MAX, dict(max_number_neighbors=5, min_number_neighbors=5)),
                                                  (100, INT_MAX, dict(max_number_neighbors=32, min_number_neighbors=20)),
                                                  (100, INT_MAX, dict(threshold_voxel_occupancy=5, max_dist_to_plane_icp=0.05)),
                                                  (100, INT_MAX, dict(power_planarity=1.5, weight_alpha=0.5, weight_neighborhood=0.7))])
def test_build_plane_residuals_bitwise(scene, frame_id, max_res, kw):
    sw = scene["sweep"]
    opts = po.default_opts(max_num_residuals=max_res, **kw)
    a = 0.3
    R_il = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1.0]]); t_il = np.array([0.05, -0.02, 0.1])
    for ext in ((None, None), (R_il, t_il)):
        q = sw["q_pred"] * (1.0 if ext[0] is None else 1.003)           # un-normalised: quirk B.10 (optimize.cpp:35 vs :95)
        o = scene["om"].build_plane_residuals(opts, sw["raw"], q, sw["t_pred"], sw["t_last"], ext[0], ext[1], frame_id=frame_id)
        r = scene["rm"].build_plane_residuals(opts, sw["raw"], q, sw["t_pred"], sw["t_last"], ext[0], ext[1], frame_id=frame_id)
        assert_pass_equals(o, r)


@live
def test_empty_sweep_and_empty_map_fail_like_the_reference(scene):
    opts = po.default_opts(max_num_residuals=INT_MAX)
    sw = scene["sweep"]
    r = scene["rm"].build_plane_residuals(opts, np.zeros((0, 3)), sw["q_pred"], sw["t_pred"], sw["t_last"])
    o = scene["om"].build_plane_residuals(opts, np.zeros((0, 3)), sw["q_pred"], sw["t_pred"], sw["t_last"])
    assert r["rc"] == 0 and r["success"] == 0 == o["neq"].success and r["num_residuals"] == 0 == o["neq"].num_residuals
    empty = pr.Map(np.zeros((0, 3), np.int16), np.zeros(0, np.int32), np.zeros((0, 20, 3), np.float32))
    r = empty.build_plane_residuals(opts, sw["raw"][:64], sw["q_pred"], sw["t_pred"], sw["t_last"])
    assert r["rc"] == 0 and r["success"] == 0


@live
@pytest.mark.parametrize("frame_id,max_res,iters", [(100, INT_MAX, 5), (100, 600, 5), (5, INT_MAX, 5), (100, -1, 5), (100, INT_MAX, 1), (1, INT_MAX, 3)])
def test_update_iekf_bitwise(scene, oracle_backend, frame_id, max_res, iters):
    sw = scene["sweep"]
    opts = po.default_opts(max_num_residuals=max_res, num_iters_icp=iters)
    e = po.Eskf(oracle_backend); synth.eskf_prior(e, sw["q_pred"], sw["t_pred"], sw["vel"])
    re_ = pr.Eskf(); re_.set_state(e.get_state()); re_.set_cov(e.get_cov())
    st = np.concatenate([sw["q_pred"], sw["t_pred"], sw["vel"], np.zeros(6)])
    u = po.update_iekf(scene["om"], e, opts, sw["raw"], st, sw["t_last"], frame_id=frame_id)
    r = pr.update_iekf(scene["rm"], re_, opts, sw["raw"], st, sw["t_last"], frame_id=frame_id)
    assert (u["rc"] > 0) == (r["rc"] == 1) and u["num_residuals"] == r["num_residuals"]
    assert np.array_equal(u["state"], r["state"])
    assert np.array_equal(e.get_state(), re_.get_state()) and np.array_equal(e.get_cov(), re_.get_cov())


@live
def test_search_neighbors_on_the_tie_scene_is_the_reference_heap(oracle_backend):
    tpts, tsw = synth.lattice_scene(4711, 512)
    om = po.Map(oracle_backend); om.add_points(tpts)
    rm = pr.Map.from_oracle(om)
    keys, counts, xyz = om.export()
    flat = xyz.reshape(-1, 3)
    world = po.transform_points(tsw["raw"], tsw["q_pred"], tsw["t_pred"])
    ties = 0
    for nb, K, thr in ((1, 20, 1), (1, 5, 1), (2, 20, 1), (1, 20, 7), (1, 32, 1), (1, 1, 1)):
        for p in world[:256]:
            o = om.search_neighbors(p, nb=nb, K=K, thr=thr)
            r = rm.search_neighbors(p, nb=nb, K=K, thr=thr)
            assert o["n"] == r["n"]
            assert np.array_equal(flat[o["ids"]].astype(np.float64), r["xyz"])
            assert np.array_equal(keys[o["ids"] // 20], r["voxels"])            # the `voxels` out-parameter
            ties += int(o["tie"])
    assert ties > 500


@live
def test_compute_neighborhood_distribution_bitwise_and_nan_throw():
    rng = np.random.default_rng(9)
    for trial in range(300):
        n = int(rng.integers(3, 33))
        kind = trial % 5
        pts = rng.normal(0, 1.0, (n, 3)) * np.array([2.0, 1.5, [1.0, 0.01, 1e-4, 0.0, 0.3][kind]])
        pts = (pts @ synth.quat_to_rot(synth.quat_from_rotvec(rng.normal(0, 1, 3))).T + rng.uniform(-80, 80, 3)).astype(np.float32).astype(np.float64)
        r = pr.neighborhood(pts)
        c = np.zeros(3); nrm = np.zeros(3); cov = np.zeros(9); ev = np.zeros(3)
        import ctypes as C
        a2d = C.c_double()
        rc = po.load().orc_neighborhood(po._vp(pts), n, po._dp(c), po._dp(nrm), po._dp(cov), C.byref(a2d), po._dp(ev))
        assert rc == r["rc"] == 0
        assert np.array_equal(c, r["center"]) and np.array_equal(cov.reshape(3, 3), r["cov"])
        assert np.array_equal(nrm, r["normal"]) and a2d.value == r["a2D"]
    same = np.tile(np.array([[1.25, -2.5, 0.75]]), (20, 1))                  # sigma_1 = 0: 0 / 0 -> the reference throws (optimize.cpp:348-350)
    assert pr.neighborhood(same)["rc"] == -1


@live
def test_eskf_predict_observe_bitwise(oracle_backend):
    rng = np.random.default_rng(11)
    e = po.Eskf(oracle_backend); r = pr.Eskf()
    for f in (e, r):
        f.set_noise(0.1, 0.2, 1e-4, 2e-4)
    e.scale_init_cov()
    s = e.get_state(); s[3:7] = synth.quat_from_rotvec([0.1, -0.2, 0.05]); s[7:10] = [0.3, -0.1, 0.05]; s[10:13] = [0.01, 0.02, -0.01]; s[13:16] = [1e-3, -2e-3, 5e-4]
    e.set_state(s); r.set_state(s); r.set_cov(e.get_cov())
    e.init_imu([0.1, 0.2, 9.7], [0.01, -0.02, 0.03]); r.init_imu([0.1, 0.2, 9.7], [0.01, -0.02, 0.03])
    for k in range(50):
        acc, gyr = np.array([0.1, 0.0, 9.8]) + rng.normal(0, 0.3, 3), rng.normal(0, 0.2, 3)
        e.predict(0.005, acc, gyr); r.predict(0.005, acc, gyr)
        if k % 7 == 3:
            dx = rng.normal(0, 1e-2, 17) * (1e-3 if k % 14 == 3 else 1.0)      # both branches of so3ToQuat / so3ToRotation
            e.observe(dx); r.observe(dx)
        assert np.array_equal(e.get_state(), r.get_state()) and np.array_equal(e.get_cov(), r.get_cov()), k


@live
@pytest.mark.parametrize("sigma_g,sigma_a,want", [(0.01, 0.05, 1), (0.6, 0.05, -1), (0.01, 0.7, -2)])
def test_try_init_bitwise(oracle_backend, sigma_g, sigma_a, want):
    rng = np.random.default_rng(5)
    g_dir = np.array([0.05, -0.02, 1.0]); g_dir /= np.linalg.norm(g_dir)
    pr.reset_globals()
    e = po.Eskf(oracle_backend); r = pr.Eskf()
    e.set_noise(0.1, 0.2, 1e-4, 2e-4); r.set_noise(0.1, 0.2, 1e-4, 2e-4)
    po.load().orc_eskf_set_g_norm(e.h, 9.81)
    t = 100.0
    last = 0
    for _ in range(8):
        ts = t + 0.005 * np.arange(1, 101); t = ts[-1]
        gyr = np.array([0.002, -0.001, 0.0005]) + rng.normal(0, sigma_g, (100, 3)); acc = 9.79 * g_dir + rng.normal(0, sigma_a, (100, 3))
        last = e.try_init(ts, gyr, acc)
        became, st = r.try_init(ts, gyr, acc, 9.81)
        so = e.init_stats()
        for k in ("mean_gyr", "mean_acc", "gyr_cov", "acc_cov"):
            assert np.array_equal(so[k], st[k]), k
        assert so["num_init_meas"] == st["num_init_meas"] and so["initial_flag"] == st["initial_flag"] and became == (1 if last == 1 else 0)
        if last != 0:
            break
    assert last == want
    assert np.array_equal(e.get_state(), r.get_state()) and np.array_equal(e.get_cov(), r.get_cov())
    if want == 1:
        acc, gyr = np.array([0.1, 0.0, 9.8]), np.array([0.01, 0.0, -0.01])
        e.predict(0.005, acc, gyr); r.predict(0.005, acc, gyr)
        assert np.array_equal(e.get_state(), r.get_state()) and np.array_equal(e.get_cov(), r.get_cov())
    pr.reset_globals()


@live
def test_transform_point_and_grid_sampling_order():
    rng = np.random.default_rng(77)
    raw = rng.uniform(-30, 30, (20000, 3))
    q = np.array([0.9, 0.1, -0.2, 0.3]) * 1.01
    t = np.array([1.0, -2.0, 0.5]); t_il = np.array([0.1, 0.2, -0.05])
    a = 0.3; R_il = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1.0]])
    w_o = po.transform_points(raw, q, t, R_il, t_il); w_r = pr.transform_points(raw, q, t, R_il, t_il)
    assert np.array_equal(w_o, w_r)
    for size in (0.5, 1.0, 1.5):
        assert np.array_equal(po.grid_sampling(w_o, size), pr.grid_sampling(w_o, size))     # std::tr1::unordered_map iteration order
    assert len(pr.grid_sampling(np.zeros((0, 3)), 1.0)) == 0


def _imu_track(rng, S, t0, dt):
    st = np.zeros((S, 17))
    q = synth.quat_from_rotvec([0.1, -0.05, 0.3]); p = np.array([1.0, 2.0, 0.3]); v = np.array([1.5, -0.4, 0.1])
    for k in range(S):
        st[k, 0] = t0 + k * dt; st[k, 1:4] = rng.normal(0, 0.5, 3); st[k, 4:7] = rng.normal(0, 0.3, 3)
        st[k, 7:10] = p; st[k, 10:14] = q; st[k, 14:17] = v
        q = synth.quat_mul(q, synth.quat_from_rotvec(st[k, 4:7] * dt)); p = p + v * dt; v = v + st[k, 1:4] * dt
    return st


@live
def test_sweep_reconstruction_bitwise():
    rng = np.random.default_rng(3)
    R_il = synth.quat_to_rot(synth.quat_from_rotvec([0.01, 0.02, -0.03])); t_il = np.array([0.05, -0.02, 0.1])
    st = _imu_track(rng, 12, 50.0, 0.01)
    n = 4000
    raw = rng.uniform(-20, 20, (n, 3)); rel = np.sort(rng.uniform(0, 110.0, n)); rel[0] = 0.0; rel[-1] = 110.0
    o, _ = po.distort_frame(raw, rel, st, 50.0, 1, R_il, t_il)
    r = pr.distort_frame(raw, rel, st, 50.0, 1, R_il, t_il)
    assert np.array_equal(o, r)
    st2 = st.copy(); st2[-1, 10:14] = st2[0, 10:14]                      # q_begin == q_end: slerp's absD >= 1 - eps branch
    assert np.array_equal(po.distort_frame(raw, rel, st2, 50.0, 1, R_il, t_il)[0], pr.distort_frame(raw, rel, st2, 50.0, 1, R_il, t_il))
    st3 = st.copy(); st3[-1, 10:14] = -st3[-1, 10:14]                    # d < 0: the sign flip of scale1
    assert np.array_equal(po.distort_frame(raw, rel, st3, 50.0, 1, R_il, t_il)[0], pr.distort_frame(raw, rel, st3, 50.0, 1, R_il, t_il))
    assert np.array_equal(po.transform_all_imu_point(o, st, R_il, t_il), pr.transform_all_imu_point(o, st, R_il, t_il))
    # IMU mode: the sequential interval walk, boundary timestamps, a point going back in time stops everything behind it
    st = _imu_track(rng, 9, 10.0, 0.0125)
    rel = np.sort(rng.uniform(0, 100.0, n)); rel[5] = 12.5; rel[6] = 12.5 + 5e-4; rel = np.sort(rel)
    sentinel = np.full_like(raw, 7.0)
    o, k = po.distort_frame(raw, rel, st, 10.0, 0, R_il, t_il, imu_point_in=sentinel)
    assert k == n and np.array_equal(o, pr.distort_frame(raw, rel, st, 10.0, 0, R_il, t_il, imu_point_in=sentinel))
    rel2 = rel.copy(); rel2[1000] = rel2[10]
    o2, k2 = po.distort_frame(raw, rel2, st, 10.0, 0, R_il, t_il, imu_point_in=sentinel)
    assert k2 == 1000 and np.array_equal(o2, pr.distort_frame(raw, rel2, st, 10.0, 0, R_il, t_il, imu_point_in=sentinel))


@live
def test_numtype_helpers_bitwise():
    import ctypes as C
    rng = np.random.default_rng(2)
    lib = po.load()
    for k in range(400):
        w = rng.normal(0, 1, 3) * [1.0, 1e-3, 1e-6, 3.0][k % 4]
        R = np.zeros(9); lib.orc_so3_to_rot(po._dp(w), po._dp(R))
        assert np.array_equal(R.reshape(3, 3), pr.so3_to_rot(w))
        q = np.zeros(4); lib.orc_so3_to_quat(po._dp(w), po._dp(q))
        assert np.array_equal(q, pr.so3_to_quat(w))
        back = np.zeros(3); lib.orc_rot_to_so3(po._dp(R), po._dp(back))
        assert np.array_equal(back, pr.rot_to_so3(R.reshape(3, 3)))
        assert lib.orc_angular_distance_so3(po._dp(w)) == pr.angular_distance_so3(w) or (np.isnan(lib.orc_angular_distance_so3(po._dp(w))) and np.isnan(pr.angular_distance_so3(w)))
        g = rng.normal(0, 1, 3) * 9.81
        B = np.zeros(6); lib.orc_derivative_s2(po._dp(g), po._dp(B))
        assert np.array_equal(B.reshape(3, 2), pr.derivative_s2(g))


@live
def test_optimize_end_to_end_bitwise(scene, oracle_backend):
    """lioOptimization::optimize (src/optimize.cpp:428-447): gridSampling on point_frame, updateIEKF, re-transform of the frame."""
    sw = scene["sweep"]
    frame_raw = sw["raw"]
    frame_point = po.transform_points(frame_raw, sw["q_pred"], sw["t_pred"])          # what stateEstimation leaves in .point
    for frame_id, sample in ((100, 1.5), (5, 1.0)):
        opts = po.default_opts(max_num_residuals=600)
        e = po.Eskf(oracle_backend); synth.eskf_prior(e, sw["q_pred"], sw["t_pred"], sw["vel"])
        re_ = pr.Eskf(); re_.set_state(e.get_state()); re_.set_cov(e.get_cov())
        st = np.concatenate([sw["q_pred"], sw["t_pred"], sw["vel"], np.zeros(6)])
        r = pr.optimize(scene["rm"], re_, opts, frame_raw, frame_point, sample, st, sw["t_last"], frame_id=frame_id)
        idx = po.grid_sampling(frame_point, sample)
        u = po.update_iekf(scene["om"], e, opts, frame_raw[idx], st, sw["t_last"], frame_id=frame_id)
        assert r["rc"] == 1 and u["rc"] > 0 and r["num_residuals"] == u["num_residuals"]
        assert np.array_equal(u["state"], r["state"]) and np.array_equal(e.get_cov(), re_.get_cov())
        assert np.array_equal(po.transform_points(frame_raw, u["state"][0:4], u["state"][4:7]), r["frame_point"])


@live
def test_far_coordinates_and_truncation_seam_bitwise(oracle_backend):
    """+-20 km (FP32 map coordinates have a 2 mm ulp there) and keypoints within 1e-7 of a voxel boundary through zero
    (static_cast<short> truncates toward zero, optimize.cpp:372-374)."""
    pts, L = synth.map_candidates(31, 20_000)
    for off in (np.array([20000.0, -20000.0, 150.0]), np.zeros(3)):
        om = po.Map(oracle_backend); om.add_points(pts + off)
        rm = pr.Map.from_oracle(om)
        sw = synth.make_sweep(32, 1024, L)
        t_pred = sw["t_pred"] + off; t_last = sw["t_last"] + off
        raw = sw["raw"].copy()
        if not off.any():
            # put keypoints on / next to the planes x = 0, y = 0, z = -1 in the world frame
            Rm = synth.quat_to_rot(sw["q_pred"])
            world = raw @ Rm.T + t_pred
            for i, (ax, val) in enumerate([(0, 0.0), (0, 1e-7), (0, -1e-7), (1, 0.0), (1, -1e-7), (2, -1.0), (2, -1.0 + 1e-7), (2, -1.0 - 1e-7)] * 8):
                world[i, ax] = val
            raw = (world - t_pred) @ Rm
        opts = po.default_opts(max_num_residuals=INT_MAX)
        o = om.build_plane_residuals(opts, raw, sw["q_pred"], t_pred, t_last)
        r = rm.build_plane_residuals(opts, raw, sw["q_pred"], t_pred, t_last)
        assert_pass_equals(o, r)
        assert o["neq"].num_residuals > 300


# ============================================================================== the node itself (src/lioOptimization.cpp)
def _sorted_map(keys, counts, xyz):
    order = np.lexsort((keys[:, 2], keys[:, 1], keys[:, 0]))
    return keys[order], counts[order], xyz[order]


@pytest.mark.parametrize("mc", [1, 0])
def test_oracle_replay_reproduces_the_reference_node_run_bitwise(gref, oracle_backend, mc):
    """golden `run{mc}_*`: the reference's own node -- constructor, readParameters, imuHandler, getMeasurements, run, process,
    stateInitialization, buildFrame (mt19937_64 shuffles, subSampleFrame), stateEstimation, optimize, addPointsToMap -- fed 40
    sweeps of sensor streams.  The same loop restated on the oracle's pieces (tests/replay_reference.py) must give the same
    poses, filter, frames and final map, bit for bit."""
    from replay_reference import REPLAY_OO, REPLAY_SEQ, OracleReplay, replay_inputs
    st, parts, gt = replay_inputs()
    oo = dict(REPLAY_OO, motion_compensation=mc)
    ref = OracleReplay(po, oracle_backend, oo, po.default_opts(max_num_residuals=REPLAY_SEQ["max_num_residuals"]))
    pre = f"run{mc}"
    row = 0
    for i, ms in enumerate(parts):
        out = ref.run_measurement(ms)
        if out is None:
            continue
        assert i == int(gref[f"{pre}_measurement"][row]) and ref.frames[-1]["frame_id"] == int(gref[f"{pre}_frame_id"][row])
        f = ref.frames[-1]
        assert np.array_equal(f["state"], gref[f"{pre}_state"][row])
        assert np.array_equal(ref.e.get_state(), gref[f"{pre}_eskf_state"][row]) and np.array_equal(ref.e.get_cov(), gref[f"{pre}_eskf_cov"][row])
        assert len(f["raw"]) == int(gref[f"{pre}_frame_points"][row]) and ref.m.size() == int(gref[f"{pre}_map_points"][row])
        assert np.array_equal(f["raw"].sum(0), gref[f"{pre}_raw_sum"][row]) and np.array_equal(f["point"].sum(0), gref[f"{pre}_point_sum"][row])
        assert np.array_equal(f["imu_point"].sum(0), gref[f"{pre}_imu_sum"][row])
        row += 1
    assert row == len(gref[f"{pre}_measurement"]) == 9
    k, c, x = _sorted_map(*ref.m.export())
    assert np.array_equal(k, gref[f"{pre}_map_keys"]) and np.array_equal(c, gref[f"{pre}_map_counts"]) and np.array_equal(x, gref[f"{pre}_map_xyz"])
    # and the estimate follows the motion (odometry frame = first sensor pose)
    assert np.linalg.norm(ref.frames[-1]["state"][4:7] - gt[len(parts) - 1][1]) < 0.08


@live
@pytest.mark.parametrize("mc,init", [(1, 0), (0, 0), (1, 1)])
def test_reference_node_run_live(oracle_backend, mc, init):
    """The same comparison against the live library with full frame contents (and INIT_CONSTANT_VELOCITY)."""
    from replay_reference import REPLAY_OO, REPLAY_SEQ, OracleReplay, replay_inputs
    st, parts, _ = replay_inputs()
    oo = dict(REPLAY_OO, motion_compensation=mc, initialization=init)
    icp = po.default_opts(max_num_residuals=REPLAY_SEQ["max_num_residuals"])
    pr.set_params(*pr.params_from_options(oo, icp))
    node = pr.Node(True)
    try:
        node.push_imu(st["imu_t"], st["imu_acc"], st["imu_gyr"])
        node.push_points(st["pts_raw"], st["pts_timestamp"])
        for t in st["image_times"]:
            node.push_image_time(t)
        ref = OracleReplay(po, oracle_backend, oo, icp)
        processed = 0
        for ms in parts:
            want = ref.run_measurement(ms)
            info = node.run()
            assert info["rc"] == 0 and info["index_frame"] == ref.index_frame and info["initial_flag"] == ref.initial_flag
            assert info["map_points"] == ref.m.size()
            if want is None:
                continue
            processed += 1
            f = node.last_frame(); fo = ref.frames[-1]
            assert f["frame_id"] == fo["frame_id"] and f["time_sweep_end"] == fo["time_sweep_end"]
            for a, b in ((f["state"], fo["state"]), (f["raw_point"], fo["raw"]), (f["imu_point"], fo["imu_point"]), (f["point"], fo["point"])):
                assert np.array_equal(a, b)
            s, P = node.eskf()
            assert np.array_equal(s, ref.e.get_state()) and np.array_equal(P, ref.e.get_cov())
        assert processed == 9
        do = pr.map_as_dict(*ref.m.export()); dr = pr.map_as_dict(*node.map_export())
        assert set(do) == set(dr) and all(np.array_equal(do[k], dr[k]) for k in do)
    finally:
        node.close()
        pr.set_params()


@live
def test_add_points_to_map_is_the_reference_insert(oracle_backend):
    """lioOptimization::addPointsToMap -> addPointToMap (src/lioOptimization.cpp:399-446,520-554): batches, full voxels, the
    minimum-distance rule against residents and earlier points of the same batch, min_num_points > 0, +-20 km."""
    pts, L = synth.map_candidates(556, 80_000)
    rng = np.random.default_rng(4)
    for off, min_num in ((np.zeros(3), 0), (np.array([20000.0, -20000.0, 150.0]), 0), (np.zeros(3), 3)):
        pr.set_params()
        node = pr.Node(True)
        om = po.Map(oracle_backend)
        try:
            for b in range(5):
                chunk = pts[rng.permutation(len(pts))[:20_000]] + off
                mn = min_num if b >= 2 else 0
                a = om.add_points(chunk, voxel_size=1.0, cap=20, min_dist=0.1, min_num_points=mn)
                r = node.add_points_to_map(chunk, voxel_size=1.0, cap=20, min_dist=0.1, min_num_points=mn)
                assert a == r and a > 0
            do = pr.map_as_dict(*om.export()); dr = pr.map_as_dict(*node.map_export())
            assert set(do) == set(dr) and all(np.array_equal(do[k], dr[k]) for k in do)
            assert max(len(v) for v in do.values()) == 20                 # some voxels filled up
        finally:
            node.close()


@live
def test_state_initialization_and_point_timestamps_are_the_reference_members(oracle_backend):
    rng = np.random.default_rng(8)
    for init in (0, 1):        # the `else` arm (copy the last pose) cannot be selected through readParameters: an unknown string keeps the default
        pr.set_params(strs={"odometry_options/initialization": pr.INITIALIZATION.get(init, "NONE")})
        node = pr.Node(True)
        try:
            for index_frame in (1, 2, 3, 4, 25):
                for flag in (False, True):
                    p2 = np.r_[synth.quat_from_rotvec(rng.normal(0, 0.3, 3)), rng.normal(0, 2, 3)]
                    p1 = np.r_[synth.quat_from_rotvec(rng.normal(0, 0.3, 3)), rng.normal(0, 2, 3)]
                    eq = synth.quat_from_rotvec(rng.normal(0, 0.3, 3)); et = rng.normal(0, 2, 3)
                    qo, to = po.state_initialization(index_frame, init, flag, p2, p1, eq, et, backend=oracle_backend)
                    qr, tr = node.state_initialization(index_frame, flag, p2, p1, eq, et)
                    assert np.array_equal(qo, qr) and np.array_equal(to, tr), (init, index_frame, flag)
        finally:
            node.close()
    ts = np.sort(rng.uniform(99.9, 100.25, 5000)); ts[10] = 100.0; ts[-10] = 100.2
    for enable in (True, False):
        pr.set_params()
        node = pr.Node(enable)
        try:
            rel_o, al_o, keep_o = po.make_point_timestamp(ts, 100.0, 100.2, enable, backend=oracle_backend)
            rel_r, al_r, keep_r = node.make_point_timestamp(ts, 100.0, 100.2)
            assert np.array_equal(np.flatnonzero(keep_o), keep_r)
            assert np.array_equal(rel_o[keep_o], rel_r) and np.array_equal(al_o[keep_o], al_r)
            assert (len(keep_r) == len(ts)) == enable
        finally:
            node.close()
    pr.set_params()


@live
@pytest.mark.parametrize("pos,max_res,expect_throw", [(100, INT_MAX, True), (100, 600, True), (1500, 600, False), (1500, INT_MAX, True),
                                                      (2047, 2040, False), (0, -1, True), (5, -1, False)])
def test_nan_planarity_throw_is_visited_only_in_the_reference(golden, oracle_backend, pos, max_res, expect_throw):
    """The reference throws std::runtime_error("error") at optimize.cpp:348-350 only for keypoints its sequential loop reaches
    before the break at :107 (and, with max_num_residuals <= 0, before the first keypoint that has a plane).  Same scene and
    cases as the GPU test of that name: the reference's own throw, the oracle's nan_error and -- in test_gpu_parity.py -- the
    device's SRL_ERR_NAN_PLANARITY must agree case by case."""
    keys = np.concatenate([golden["map_keys"], np.array([[300, 300, 30]], np.int16)])
    counts = np.concatenate([golden["map_counts"], np.array([20], np.int32)])
    xyz = np.concatenate([golden["map_xyz"], np.full((1, 20, 3), [300.5, 300.5, 30.5], np.float32)])       # 20 IDENTICAL points
    om = po.Map(oracle_backend); om.import_(keys, counts, xyz)
    rm = pr.Map(keys, counts, xyz)
    raw = golden["raw"].copy()
    R = synth.quat_to_rot(golden["q_pred"] / np.linalg.norm(golden["q_pred"]))
    raw[pos] = R.T @ (np.array([300.5, 300.5, 30.45]) - golden["t_pred"])       # lands next to the degenerate voxel
    opts = po.default_opts(max_num_residuals=max_res)
    o = om.build_plane_residuals(opts, raw, golden["q_pred"], golden["t_pred"], golden["t_last"])
    r = rm.build_plane_residuals(opts, raw, golden["q_pred"], golden["t_pred"], golden["t_last"])
    assert (r["rc"] == -2) == expect_throw == bool(o["neq"].nan_error)
    if not expect_throw:
        assert_pass_equals(o, r)
    # and through updateIEKF: the solve aborts (-2) or runs to the same state
    e = po.Eskf(oracle_backend); e.set_state(golden["full_eskf_state0"]); e.set_cov(golden["full_eskf_cov0"])
    re_ = pr.Eskf(); re_.set_state(golden["full_eskf_state0"]); re_.set_cov(golden["full_eskf_cov0"])
    u = po.update_iekf(om, e, opts, raw, golden["full_state0"], golden["t_last"])
    ru = pr.update_iekf(rm, re_, opts, raw, golden["full_state0"], golden["t_last"])
    if expect_throw:
        assert u["rc"] == -2 and ru["rc"] == -2
    else:
        assert (u["rc"] > 0) == (ru["rc"] == 1) and np.array_equal(u["state"], ru["state"])
