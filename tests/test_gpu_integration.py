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
        # ... and the device map the binding kept in step with it (one frame behind: the last frame is inserted at the next solve)
        h = lib.srl_integration_ctx(node_ptr)
        assert h, "the binding was never entered: the node did not call into integration/optimize_hip.cpp"
        ctx = srl.Context(handle=C.c_void_p(h))
        dk, dc, dx = ctx.map_download()
        npts, _ = ctx.map_size()
        assert npts == int(gref[f"{pre}_map_points"][row - 2])            # the map as the last solve saw it
        dev = {tuple(key): (cnt, xyz[:cnt].tobytes()) for key, cnt, xyz in zip(dk.tolist(), dc.tolist(), dx)}
        host = {tuple(key): (cnt, xyz[:cnt]) for key, cnt, xyz in zip(k.tolist(), c.tolist(), x)}
        for key, (cnt, blob) in dev.items():      # every device voxel is a prefix of the host voxel (first-come order kept)
            assert key in host and cnt <= host[key][0] and blob == host[key][1][:cnt].tobytes()
    finally:
        lib.srl_integration_release(node_ptr)
        node.close()
