"""Parity report on the GPU box: for every BASELINE configuration, one buildPlaneResiduals pass and one full
updateIEKF solve by
    QL   the oracle with the restated Eigen 3.3.7 SelfAdjointEigenSolver (what the reference executes),
    JAC  the oracle with the independent FP64 cyclic Jacobi solver,
    GPU  the HIP path through the C-ABI (closed-form eigen-decomposition in the kernel),
    REF  the reference's OWN translation units (oracle/_ref/libref_path.so: src/optimize.cpp & co. compiled in place against
         stand-in third-party headers; prebuilt) -- where that library travelled with the tree,
and the maximum deviations between them (relative to the field's largest magnitude unless noted).  QL vs REF must be
bitwise (`ref_tu.*_bitwise`); QL vs JAC bounds how far results can move with the eigen-solver.

    python tools/parity_report.py [CONFIG ...] > gpurun_out/parity_report.json
"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import sr_livo_amd as srl  # noqa: E402
from sr_livo_amd import capi, synth  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
from oracle import pyref as pr  # noqa: E402

INT_MAX = 2**31 - 1


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)) if a.size else 0.0


def fields(x, y, has, acc):
    out = {k: rel(x[k][has], y[k][has]) for k in ("normal", "a2D", "weight", "norm_offset", "distance")}
    out["jacobian"] = rel(x["jacobian"][acc], y["jacobian"][acc])
    out["normal_max_angle_rad"] = float(np.max(np.arccos(np.clip(np.abs(np.sum(x["normal"][has] * y["normal"][has], 1)), -1, 1)))) if has.any() else 0.0
    return out


def solve_oracle(m, backend, sw, opts, frame_id=100):
    e = po.Eskf(backend)
    synth.eskf_prior(e, sw["q_pred"], sw["t_pred"], sw["vel"])
    s0, P0 = e.get_state().copy(), e.get_cov().copy()
    st = np.concatenate([sw["q_pred"], sw["t_pred"], sw["vel"], np.zeros(6)])
    u = po.update_iekf(m, e, opts, sw["raw"], st, sw["t_last"], frame_id=frame_id)
    return u, e.get_state(), e.get_cov(), s0, P0, st


def one_config(name, sample=None):
    import os
    backend = "tsl" if os.path.exists(po.LIB_TSL) else "plain"
    n_kp, map_pts, pattern, seed = synth.CONFIGS[name]
    t0 = time.time()
    pts, L = synth.map_candidates(seed, map_pts)
    sw = synth.make_sweep(seed + 1000, n_kp, L, pattern=pattern)
    if sample:
        sw = dict(sw, raw=sw["raw"][:sample])
    m = po.Map(backend)
    m.add_points(pts)
    lio = srl.Lio(0)
    lio.add_points_to_map(pts)
    rep = {"config": name, "keypoints": len(sw["raw"]), "map_points": int(m.size()), "map_voxels": int(m.num_voxels())}
    rm = pr.Map.from_oracle(m) if pr.available() else None
    for label, frame_id, max_res in (("r1_all", 100, INT_MAX), ("r1_cut600", 100, 600), ("init_r2", 5, INT_MAX)):
        if label == "init_r2" and len(sw["raw"]) > 70000:
            continue
        oo = po.default_opts(max_num_residuals=max_res)
        o_q = m.build_plane_residuals(oo, sw["raw"], sw["q_pred"], sw["t_pred"], sw["t_last"], frame_id=frame_id)
        with po.eig_solver(po.EIG_JACOBI):
            o_j = m.build_plane_residuals(oo, sw["raw"], sw["q_pred"], sw["t_pred"], sw["t_last"], frame_id=frame_id)
        ctx = lio.ctx
        ctx.sweep_upload(sw["raw"]); ctx.set_taps(1)
        neq, rc = ctx.build_residuals(capi.make_frame(sw["q_pred"], sw["t_pred"], sw["t_last"], frame_id=frame_id), srl.default_opts(max_num_residuals=max_res))
        ids, status, ncand = ctx.fetch_neighbors()
        g = ctx.fetch_residuals()
        ctx.set_taps(0)
        has = (o_q["status"] == 1) | (o_q["status"] == 2)
        acc = o_q["status"] == 2
        vis = o_q["status"] != 3
        r = {"ids_equal_gpu_vs_oracle": bool(np.array_equal(ids[vis], o_q["ids"][vis])), "status_equal": bool(np.array_equal(status, o_q["status"])),
             "keypoints_with_tied_distances": int(o_q["neq"].num_ties), "gpu_off_fast_path": int(neq.num_fallback),
             "residuals": int(o_q["neq"].num_residuals), "residuals_gpu": int(neq.num_residuals),
             "QL_vs_JAC": fields(o_q, o_j, has, acc), "GPU_vs_QL": fields(g, o_q, has, acc), "GPU_vs_JAC": fields(g, o_j, has, acc),
             "HtH": {"QL_vs_JAC": rel(o_q["HtH"], o_j["HtH"]), "GPU_vs_QL": rel(np.array(neq.HtH).reshape(6, 6), o_q["HtH"]),
                     "GPU_vs_JAC": rel(np.array(neq.HtH).reshape(6, 6), o_j["HtH"])},
             "Hth": {"QL_vs_JAC": rel(o_q["Hth"], o_j["Hth"]), "GPU_vs_QL": rel(np.array(neq.Hth), o_q["Hth"])}}
        # full solve
        oo2 = po.default_opts(max_num_residuals=max_res)
        u_q, es_q, P_q, s0, P0, st = solve_oracle(m, backend, sw, oo2, frame_id)
        with po.eig_solver(po.EIG_JACOBI):
            u_j, es_j, P_j, _, _, _ = solve_oracle(m, backend, sw, oo2, frame_id)
        lio.eskf_set_state(s0); lio.eskf_set_cov(P0)
        gs = lio.update_iekf(srl.default_opts(max_num_residuals=max_res), sw["raw"], st, sw["t_last"], frame_id=frame_id)
        r["solve"] = {"iterations": {"QL": int(u_q["rc"]), "JAC": int(u_j["rc"]), "GPU": int(gs["iters"])},
                      "state": {"QL_vs_JAC": rel(u_q["state"], u_j["state"]), "GPU_vs_QL": rel(gs["state"], u_q["state"]), "GPU_vs_JAC": rel(gs["state"], u_j["state"])},
                      "eskf_state": {"QL_vs_JAC": rel(es_q, es_j), "GPU_vs_QL": rel(lio.eskf_get_state(), es_q)},
                      "covariance": {"QL_vs_JAC": rel(P_q, P_j), "GPU_vs_QL": rel(lio.eskf_get_cov(), P_q)}}
        if rm is not None:
            # the reference's own buildPlaneResiduals / updateIEKF on the same inputs: the restatement must equal it bit for bit
            rr = rm.build_plane_residuals(oo, sw["raw"], sw["q_pred"], sw["t_pred"], sw["t_last"], frame_id=frame_id)
            re_ = pr.Eskf(); re_.set_state(s0); re_.set_cov(P0)
            ru = pr.update_iekf(rm, re_, oo2, sw["raw"], st, sw["t_last"], frame_id=frame_id)
            one_bitwise = bool(rr["rc"] == int(acc.sum()) and all(np.array_equal(o_q[k][acc], rr[k]) for k in ("normal", "jacobian", "norm_offset", "distance", "weight"))
                               and np.array_equal(o_q["point_world"], rr["point_world"]) and o_q["neq"].loss_sum == rr["loss"])
            r["ref_tu"] = {"one_pass_bitwise": one_bitwise, "residuals": int(rr["num_residuals"]),
                           "solve_state_bitwise": bool(np.array_equal(u_q["state"], ru["state"])),
                           "solve_covariance_bitwise": bool(np.array_equal(P_q, re_.get_cov())),
                           "GPU_vs_REF_state": rel(gs["state"], ru["state"]),
                           "GPU_vs_REF_jacobian": rel(g["jacobian"][acc], rr["jacobian"]), "GPU_vs_REF_distance": rel(g["distance"][acc], rr["distance"])}
        rep[label] = r
    rep["seconds"] = round(time.time() - t0, 1)
    lio.close()
    return rep


def main():
    names = sys.argv[1:] or ["C1", "C2", "C3", "HEADLINE", "C4"]
    out = {"what": __doc__.strip().split("\n\n")[0], "tolerance_north_star": 1e-5, "configs": []}
    for n in names:
        out["configs"].append(one_config(n, sample=32768 if n == "C4" else None))
        print(json.dumps(out["configs"][-1]), file=sys.stderr)
    worst = {}
    for c in out["configs"]:
        for lab in ("r1_all", "r1_cut600", "init_r2"):
            if lab not in c:
                continue
            for pair in ("QL_vs_JAC", "GPU_vs_QL"):
                for k, v in c[lab][pair].items():
                    worst.setdefault(pair, {}).setdefault(k, 0.0)
                    worst[pair][k] = max(worst[pair][k], v)
                worst[pair]["HtH"] = max(worst[pair].get("HtH", 0.0), c[lab]["HtH"][pair])
                worst[pair]["solve_state"] = max(worst[pair].get("solve_state", 0.0), c[lab]["solve"]["state"][pair])
    out["worst_over_all_configs"] = worst
    refs = [c[lab]["ref_tu"] for c in out["configs"] for lab in ("r1_all", "r1_cut600", "init_r2") if lab in c and "ref_tu" in c[lab]]
    if refs:
        out["oracle_equals_reference_tu_bitwise_everywhere"] = bool(all(x["one_pass_bitwise"] and x["solve_state_bitwise"] and x["solve_covariance_bitwise"] for x in refs))
        out["worst_GPU_vs_reference_tu"] = {k: max(x[k] for x in refs) for k in ("GPU_vs_REF_state", "GPU_vs_REF_jacobian", "GPU_vs_REF_distance")}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
