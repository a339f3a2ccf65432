"""Generates tests/golden/golden_ref_tu.npz from the REFERENCE'S OWN translation units (run in the build container):

    make -C oracle refpath && python tests/golden/make_golden_ref.py

oracle/_ref/libref_path.so is /root/reference/src/{optimize,lioOptimization,eskfEstimator,utility,state,cloudMap,
parameters}.cpp compiled where they lie (oracle/ref_harness.cpp, oracle/Makefile); every array written here is an output of
lioOptimization::buildPlaneResiduals / updateIEKF / searchNeighbors / optimize, and -- the `run*` arrays -- of the node's own
constructor / readParameters / imuHandler / getMeasurements / run / process / buildFrame / stateEstimation / addPointsToMap
on a 40-sweep sequence, as the reference wrote them.  Third-party arithmetic (Eigen is absent from the image)
is the stand-in of oracle/ref_shim/Eigen/Core -- see its header for what that means for low-order bits.
The inputs are those of golden_small.npz (same scenes), so the two files are read side by side:
  * tests/test_reference_tu.py (CPU): the oracle must reproduce these vectors BITWISE (also on the GPU box, where neither
    /root/reference nor a compiler for it exists);
  * tests/test_gpu_parity.py (GPU): the HIP path against the same vectors.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402
from oracle import pyref as pr  # noqa: E402
from sr_livo_amd import synth  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = (("full", 100, 2**31 - 1), ("cut600", 100, 600), ("init", 5, 2**31 - 1), ("neg1", 100, -1))
TIE_CASES = (("tie", {}, 100), ("tie5", dict(max_number_neighbors=5, min_number_neighbors=5), 100), ("tieinit", {}, 5))


def one_pass(rm, opts, raw, q, t, t_last, frame_id, prefix, data):
    r = rm.build_plane_residuals(opts, raw, q, t, t_last, frame_id=frame_id)
    assert r["rc"] >= 0
    for k in ("point_world", "location", "normal", "jacobian", "norm_offset", "distance", "weight"):
        data[f"{prefix}_ref_{k}"] = r[k]
    data[f"{prefix}_ref_success"] = r["success"]
    data[f"{prefix}_ref_num_residuals"] = r["num_residuals"]
    data[f"{prefix}_ref_loss"] = r["loss"]
    return r


sys.path.insert(0, os.path.join(ROOT, "tests"))
from replay_reference import REPLAY_OO, REPLAY_SEQ, replay_inputs  # noqa: E402


def replay(data):
    """lioOptimization's own constructor / readParameters / imuHandler / getMeasurements / run / process / buildFrame /
    stateEstimation / addPointsToMap (src/lioOptimization.cpp) on a 40-sweep sequence: per processed frame the solved state,
    the filter, the frame size and the map size; at the end the map."""
    st, parts, _ = replay_inputs()
    for mc in (1, 0):
        oo = dict(REPLAY_OO, motion_compensation=mc)
        icp = po.default_opts(max_num_residuals=REPLAY_SEQ["max_num_residuals"])
        pr.set_params(*pr.params_from_options(oo, icp))
        node = pr.Node(True)
        node.push_imu(st["imu_t"], st["imu_acc"], st["imu_gyr"])
        node.push_points(st["pts_raw"], st["pts_timestamp"])
        for t in st["image_times"]:
            node.push_image_time(t)
        rows = []
        for i in range(len(parts)):
            before = node.run()
            assert before["rc"] == 0
            f = node.last_frame()
            if f is None or (rows and f["frame_id"] == rows[-1]["frame_id"]):
                continue
            s, P = node.eskf()
            rows.append(dict(measurement=i, frame_id=f["frame_id"], state=f["state"], eskf_state=s, eskf_cov=P, frame_points=len(f["raw_point"]),
                             map_points=before["map_points"], raw_sum=f["raw_point"].sum(0), point_sum=f["point"].sum(0), imu_sum=f["imu_point"].sum(0)))
        k, c, x = node.map_export()
        order = np.lexsort((k[:, 2], k[:, 1], k[:, 0]))
        pre = f"run{mc}"
        for name in ("measurement", "frame_id", "state", "eskf_state", "eskf_cov", "frame_points", "map_points", "raw_sum", "point_sum", "imu_sum"):
            data[f"{pre}_{name}"] = np.array([r[name] for r in rows])
        data[f"{pre}_map_keys"] = k[order]; data[f"{pre}_map_counts"] = c[order]; data[f"{pre}_map_xyz"] = x[order]
        node.close()
        print(pre, "frames", len(rows), "first processed at measurement", rows[0]["measurement"], "map points", rows[-1]["map_points"], "voxels", len(k))


def main():
    g = np.load(os.path.join(HERE, "golden_small.npz"), allow_pickle=False)
    data = dict(source=pr.load().ref_describe().decode())
    rm = pr.Map(g["map_keys"], g["map_counts"], g["map_xyz"])
    raw, q, t, t_last, vel = g["raw"], g["q_pred"], g["t_pred"], g["t_last"], g["vel"]
    for name, frame_id, max_res in CASES:
        opts = po.default_opts(max_num_residuals=max_res)
        r = one_pass(rm, opts, raw, q, t, t_last, frame_id, name, data)
        e = pr.Eskf()
        e.set_state(g[f"{name}_eskf_state0"]); e.set_cov(g[f"{name}_eskf_cov0"])
        u = pr.update_iekf(rm, e, opts, raw, g[f"{name}_state0"], t_last, frame_id=frame_id)
        data[f"{name}_ref_solve_rc"] = u["rc"]
        data[f"{name}_ref_solve_state"] = u["state"]
        data[f"{name}_ref_solve_num_residuals"] = u["num_residuals"]
        data[f"{name}_ref_solve_eskf_state"] = e.get_state()
        data[f"{name}_ref_solve_eskf_cov"] = e.get_cov()
        print(name, "residuals", r["num_residuals"], "success", r["success"], "solve rc", u["rc"], "t err", np.linalg.norm(u["state"][4:7] - g["t_gt"]))
    # tie scene: the neighbour lists are what the real std::priority_queue leaves (src/optimize.cpp:394-422)
    tm = pr.Map(g["tie_map_keys"], g["tie_map_counts"], g["tie_map_xyz"])
    for name, kw, fid in TIE_CASES:
        opts = po.default_opts(max_num_residuals=2**31 - 1, **kw)
        r = one_pass(tm, opts, g["tie_raw"], g["tie_q"], g["tie_t"], g["tie_t_last"], fid, name, data)
        K = opts.max_number_neighbors
        nb = 2 if fid < opts.init_num_frames else opts.voxel_neighborhood
        thr = 1 if fid < opts.init_num_frames else opts.threshold_voxel_occupancy
        nbr = np.full((len(g["tie_raw"]), K, 3), np.nan, dtype=np.float32)
        cnt = np.zeros(len(g["tie_raw"]), dtype=np.int32)
        for i, p in enumerate(r["point_world"]):
            s = tm.search_neighbors(p, nb=nb, size=opts.size_voxel_map, K=K, thr=thr)
            cnt[i] = s["n"]
            nbr[i, : s["n"]] = s["xyz"].astype(np.float32)          # map points are FP32 (cloudMap.h:54): exact
            assert np.array_equal(nbr[i, : s["n"]].astype(np.float64), s["xyz"])
        data[f"{name}_ref_neighbors"] = nbr
        data[f"{name}_ref_num_neighbors"] = cnt
        print(name, "residuals", r["num_residuals"])
    replay(data)
    path = os.path.join(HERE, "golden_ref_tu.npz")
    np.savez_compressed(path, **data)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
