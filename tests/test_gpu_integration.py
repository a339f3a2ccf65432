"""The drop-in proof (north_star: "so the ROS node drops it in unchanged").

oracle/_ref/libref_node_hip.so (recipe: oracle/Makefile, target refnode_hip) is the REFERENCE NODE -- its own
src/{lioOptimization, eskfEstimator, utility, state, cloudMap, parameters}.cpp compiled where they lie, no header edited -- with
exactly one file exchanged: integration/optimize_hip.cpp instead of src/optimize.cpp, linked against libsrlivo_hip.so.  The
reference's run() loop (imuHandler, getMeasurements, process, buildFrame, stateEstimation, addPointsToMap: all its own code)
is fed the 40-sweep sensor streams of the replay goldens and must land where the all-CPU reference node landed
(tests/golden/golden_ref_tu.npz, produced by the same harness around the reference's own src/optimize.cpp): per frame the solved
state, the filter, its covariance, the frame contents and the map size; at the end the map, host copy and device copy.
"""
import ctypes as C
import os

import numpy as np
import pytest

import sr_livo_amd as srl
from sr_livo_amd import capi

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_NODE_HIP = os.path.join(os.path.dirname(HERE), "oracle", "_ref", "libref_node_hip.so")
TIGHT = 1e-9


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


@pytest.fixture(scope="module")
def hip_node_lib():
    if not os.path.exists(LIB_NODE_HIP):
        pytest.skip("oracle/_ref/libref_node_hip.so not built (needs /root/reference at build time)")
    srl.load_library()                       # libsrlivo_hip.so first: the node library resolves its srl_* symbols from it
    from oracle import pyref as pr
    saved = (pr.LIB_REF, pr._lib)
    pr.LIB_REF, pr._lib = LIB_NODE_HIP, None
    lib = pr.load()
    lib.ref_node_lio_ptr.argtypes = [C.c_void_p]; lib.ref_node_lio_ptr.restype = C.c_void_p
    lib.srl_integration_ctx.argtypes = [C.c_void_p]; lib.srl_integration_ctx.restype = C.c_void_p
    lib.srl_integration_release.argtypes = [C.c_void_p]
    lib.ref_optimize_times.argtypes = [C.POINTER(C.c_double), C.c_int]; lib.ref_optimize_times.restype = C.c_int
    yield pr, lib
    pr.LIB_REF, pr._lib = saved


@pytest.mark.parametrize("mc", [1, 0])
def test_reference_node_with_the_hip_binding_matches_the_all_cpu_reference_node(hip_node_lib, mc):
    pr, lib = hip_node_lib
    from oracle import pyoracle as po
    from replay_reference import REPLAY_OO, REPLAY_SEQ, replay_inputs
    gref = dict(np.load(os.path.join(HERE, "golden", "golden_ref_tu.npz"), allow_pickle=False))
    st, parts, _ = replay_inputs()
    oo = dict(REPLAY_OO, motion_compensation=mc)
    icp = po.default_opts(max_num_residuals=REPLAY_SEQ["max_num_residuals"])
    pr.set_params(*pr.params_from_options(oo, icp))
    node = pr.Node(True)
    node_ptr = lib.ref_node_lio_ptr(node.h)
    try:
        node.push_imu(st["imu_t"], st["imu_acc"], st["imu_gyr"])
        node.push_points(st["pts_raw"], st["pts_timestamp"])
        for t in st["image_times"]:
            node.push_image_time(t)
        pre = f"run{mc}"
        row, last_fid = 0, None
        for i in range(len(parts)):
            info = node.run()
            assert info["rc"] == 0
            f = node.last_frame()
            if f is None or f["frame_id"] == last_fid:
                continue
            last_fid = f["frame_id"]
            s, P = node.eskf()
            assert i == int(gref[f"{pre}_measurement"][row]) and f["frame_id"] == int(gref[f"{pre}_frame_id"][row])
            assert len(f["raw_point"]) == int(gref[f"{pre}_frame_points"][row])
            assert info["map_points"] == int(gref[f"{pre}_map_points"][row])
            assert rel(f["state"], gref[f"{pre}_state"][row]) < TIGHT
            assert rel(s, gref[f"{pre}_eskf_state"][row]) < TIGHT
            assert rel(P, gref[f"{pre}_eskf_cov"][row]) < 1e-8
            assert rel(f["raw_point"].sum(0), gref[f"{pre}_raw_sum"][row]) < 1e-11 and rel(f["point"].sum(0), gref[f"{pre}_point_sum"][row]) < TIGHT
            row += 1
        assert row == len(gref[f"{pre}_measurement"]) == 9
        # the node's own map (tsl::robin_map filled by its own addPointsToMap) ...
        k, c, x = node.map_export()
        order = np.lexsort((k[:, 2], k[:, 1], k[:, 0]))
        assert np.array_equal(k[order], gref[f"{pre}_map_keys"]) and np.array_equal(c[order], gref[f"{pre}_map_counts"])
        assert np.array_equal(x[order], gref[f"{pre}_map_xyz"])
        # ... and the device map the binding keeps in step with it: optimize() commits a solved frame on the device itself, from the
        # world points the node inserts right after the call -- the two maps are EQUAL, voxel by voxel, point by point
        h = lib.srl_integration_ctx(node_ptr)
        assert h, "the binding was never entered: the node did not call into integration/optimize_hip.cpp"
        ctx = srl.Context(handle=C.c_void_p(h))
        dk, dc, dx = ctx.map_download()
        npts, _ = ctx.map_size()
        assert npts == int(gref[f"{pre}_map_points"][row - 1]) == info["map_points"]
        dev = {tuple(key): (cnt, xyz[:cnt].tobytes()) for key, cnt, xyz in zip(dk.tolist(), dc.tolist(), dx)}
        host = {tuple(key): (cnt, xyz[:cnt].tobytes()) for key, cnt, xyz in zip(k.tolist(), c.tolist(), x)}
        assert dev == host
    finally:
        lib.srl_integration_release(node_ptr)
        node.close()


def test_the_binding_rebuilds_a_device_map_that_fell_out_of_step(hip_node_lib):
    """The device map is a mirror of the node's voxel_map.  If the two ever differ (here: half of the device map is thrown away behind the
    binding's back in the middle of the replay), the next optimize() rebuilds the device copy from voxel_map instead of failing, and the
    node lands where the all-CPU reference node landed."""
    pr, lib = hip_node_lib
    from oracle import pyoracle as po
    from replay_reference import REPLAY_OO, REPLAY_SEQ, replay_inputs
    lib.srl_integration_resyncs.argtypes = [C.c_void_p]; lib.srl_integration_resyncs.restype = C.c_long
    gref = dict(np.load(os.path.join(HERE, "golden", "golden_ref_tu.npz"), allow_pickle=False))
    st, parts, _ = replay_inputs()
    pr.set_params(*pr.params_from_options(dict(REPLAY_OO, motion_compensation=1), po.default_opts(max_num_residuals=REPLAY_SEQ["max_num_residuals"])))
    node = pr.Node(True)
    node_ptr = lib.ref_node_lio_ptr(node.h)
    try:
        node.push_imu(st["imu_t"], st["imu_acc"], st["imu_gyr"])
        node.push_points(st["pts_raw"], st["pts_timestamp"])
        for t in st["image_times"]:
            node.push_image_time(t)
        row, last_fid, damaged = 0, None, False
        for i in range(len(parts)):
            info = node.run()
            assert info["rc"] == 0
            f = node.last_frame()
            if f is None or f["frame_id"] == last_fid:
                continue
            last_fid = f["frame_id"]
            assert rel(f["state"], gref["run1_state"][row]) < TIGHT, f"row {row}"
            assert info["map_points"] == int(gref["run1_map_points"][row])
            row += 1
            if row == 4 and not damaged:
                ctx = srl.Context(handle=C.c_void_p(lib.srl_integration_ctx(node_ptr)))
                k, c, x = ctx.map_download()
                half = len(k) // 2
                ctx.map_upload(k[:half], c[:half], x[:half])
                assert ctx.map_size()[0] < info["map_points"]
                damaged = True
        assert row == 9 and damaged and lib.srl_integration_resyncs(node_ptr) == 1
        k, c, x = node.map_export()
        ctx = srl.Context(handle=C.c_void_p(lib.srl_integration_ctx(node_ptr)))
        dk, dc, dx = ctx.map_download()
        dev = {tuple(key): (cnt, xyz[:cnt].tobytes()) for key, cnt, xyz in zip(dk.tolist(), dc.tolist(), dx)}
        host = {tuple(key): (cnt, xyz[:cnt].tobytes()) for key, cnt, xyz in zip(k.tolist(), c.tolist(), x)}
        assert dev == host
    finally:
        lib.srl_integration_release(node_ptr)
        node.close()


# the replay stream at BASELINE scale: 24k-point sweeps, every point a keypoint candidate at a 0.2 m sampling voxel, every accepted
# residual counts (BASELINE.json configs[1]: "R3Live Livox Avia sweep (~24k pts after reconstruction) ... full ESIKF solve")
HEAVY_SEQ = dict(map_seed=556, map_target=400_000, seq_seed=32, n_moving=4, n_pts=24_000, max_num_residuals=2**31 - 1)
HEAVY_OO = dict(init_sample_voxel_size=0.2, sample_voxel_size=0.2)


def _optimize_times(pr, lib_path, mc, seq=None, oo_over=None):
    """wall time (us) of every lioOptimization::optimize call of a replay of the reference's node, through the harness's --wrap timer"""
    from oracle import pyoracle as po
    from replay_reference import REPLAY_OO, REPLAY_SEQ, replay_inputs
    seq = REPLAY_SEQ if seq is None else seq
    saved = (pr.LIB_REF, pr._lib)
    pr.LIB_REF, pr._lib = lib_path, None
    try:
        lib = pr.load()
        lib.ref_optimize_times.argtypes = [C.POINTER(C.c_double), C.c_int]; lib.ref_optimize_times.restype = C.c_int
        lib.ref_optimize_times_reset()
        st, parts, _ = replay_inputs(seq)
        oo = dict(REPLAY_OO, motion_compensation=mc, **(oo_over or {}))
        pr.set_params(*pr.params_from_options(oo, po.default_opts(max_num_residuals=seq["max_num_residuals"])))
        node = pr.Node(True)
        try:
            node.push_imu(st["imu_t"], st["imu_acc"], st["imu_gyr"])
            node.push_points(st["pts_raw"], st["pts_timestamp"])
            for t in st["image_times"]:
                node.push_image_time(t)
            states, map_points = [], 0
            for _ in range(len(parts)):
                info = node.run()
                assert info["rc"] == 0
                map_points = info["map_points"]
                f = node.last_frame()
                if f is not None:
                    states.append(f["state"].copy())
            out = (C.c_double * 256)()
            n = lib.ref_optimize_times(out, 256)
            if hasattr(lib, "srl_integration_release"):
                lib.ref_node_lio_ptr.argtypes = [C.c_void_p]; lib.ref_node_lio_ptr.restype = C.c_void_p
                lib.srl_integration_release.argtypes = [C.c_void_p]
                lib.srl_integration_release(lib.ref_node_lio_ptr(node.h))
            return np.array(out[:n]), states[-1], map_points
        finally:
            node.close()
    finally:
        pr.LIB_REF, pr._lib = saved


def _record(name, payload):
    import json
    out = os.path.join(os.path.dirname(HERE), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, name), "w") as fh:
        json.dump(payload, fh, indent=1)


def test_optimize_through_the_binding_against_the_reference_optimize_on_the_replay_streams(hip_node_lib):
    """Node-level figures of the drop-in: the reference's own node, fed the same sweeps, once with its own
    src/optimize.cpp (all CPU, oracle/_ref/libref_path.so) and once with integration/optimize_hip.cpp -- wall time of every
    lioOptimization::optimize call (gridSampling + updateIEKF + re-transform; in the binding also the device-side map insertion), same
    results.  Two streams: the 40-sweep replay of the goldens (6 000-point sweeps, the shipped sampling of 1.5 m and
    max_num_residuals = 600: a few hundred keypoints per solve -- the reference's optimize() itself takes well under a millisecond
    there, so the binding can only win a small factor) and the same scene at BASELINE scale (24 000-point sweeps, 0.2 m sampling, every
    residual counts), where the path is the node's bottleneck: >= 20x asserted there."""
    pr, _lib = hip_node_lib
    cpu_lib = os.path.join(os.path.dirname(HERE), "oracle", "_ref", "libref_path.so")
    if not os.path.exists(cpu_lib):
        pytest.skip("oracle/_ref/libref_path.so not built")
    rec = {}
    for name, seq, oo in (("replay_40_sweeps", None, None), ("baseline_scale_24k", HEAVY_SEQ, HEAVY_OO)):
        _optimize_times(pr, LIB_NODE_HIP, 1, seq, oo)                      # first run: context creation, allocations, first launches
        t_hip, s_hip, m_hip = _optimize_times(pr, LIB_NODE_HIP, 1, seq, oo)
        t_cpu, s_cpu, m_cpu = _optimize_times(pr, cpu_lib, 1, seq, oo)
        assert len(t_cpu) == len(t_hip) >= 3 and m_hip == m_cpu
        assert rel(s_hip, s_cpu) < TIGHT
        ratio = float(np.median(t_cpu) / np.median(t_hip))
        print(f"{name}: optimize() per call, median us: reference {np.median(t_cpu):.0f}, binding {np.median(t_hip):.0f}; x{ratio:.1f}")
        rec[name] = {"reference_optimize_us": t_cpu.tolist(), "binding_optimize_us": t_hip.tolist(), "median_ratio": ratio, "map_points": int(m_hip)}
    rec["what"] = ("wall time of every lioOptimization::optimize call of the reference's node (its own run() loop): src/optimize.cpp on one host core vs "
                   "integration/optimize_hip.cpp; same final state and map size")
    _record("integration_optimize_times.json", rec)
    assert rec["replay_40_sweeps"]["median_ratio"] >= 1.5
    assert rec["baseline_scale_24k"]["median_ratio"] >= 20.0
