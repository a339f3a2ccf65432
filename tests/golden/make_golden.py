"""Generates the committed golden fixtures from the CPU oracle (run in the build container):

    python tests/golden/make_golden.py

The reference ships no golden vectors for this path; these vectors are outputs of oracle/ (the
line-by-line restatement), built against the real vendored tsl::robin_map when /root/reference is present, with
per-keypoint detail (ids, status, taps) the reference's interfaces do not expose.  Their reference-side counterpart is
golden_ref_tu.npz (make_golden_ref.py): outputs of the reference's OWN translation units compiled in place, on the same
scenes -- tests/test_reference_tu.py requires the two to agree bitwise.  Both travel to the GPU box, where the HIP path
is compared with them.
The symmetric eigen-solver is the oracle's restatement of Eigen 3.3.7's algorithm (eig3_eigen_ql).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402
from sr_livo_amd import synth  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def solve_case(m, sweep, frame_id, max_res, backend):
    opts = po.default_opts(max_num_residuals=max_res)
    one = m.build_plane_residuals(opts, sweep["raw"], sweep["q_pred"], sweep["t_pred"], sweep["t_last"], frame_id=frame_id)
    e = po.Eskf(backend)
    synth.eskf_prior(e, sweep["q_pred"], sweep["t_pred"], sweep["vel"])
    P0 = e.get_cov().copy()
    s0 = e.get_state().copy()
    st = np.concatenate([sweep["q_pred"], sweep["t_pred"], sweep["vel"], np.zeros(6)])
    u = po.update_iekf(m, e, opts, sweep["raw"], st, sweep["t_last"], frame_id=frame_id, log_iters=20)
    out = {f"one_{k}": v for k, v in one.items() if isinstance(v, np.ndarray)}
    out.update(one_num_residuals=one["neq"].num_residuals, one_loss=one["neq"].loss_sum,
               one_sum_candidates=one["neq"].sum_candidates, one_num_ties=one["neq"].num_ties,
               one_num_visited=one["neq"].num_visited, one_success=one["neq"].success,
               eskf_state0=s0, eskf_cov0=P0, state0=st, solve_rc=u["rc"], solve_state=u["state"],
               solve_num_residuals=u["num_residuals"], solve_log=u["log"] if u["log"] is not None else np.zeros((0, 61)),
               solve_eskf_state=e.get_state(), solve_eskf_cov=e.get_cov())
    return out


def main():
    backend = "tsl" if os.path.exists(po.LIB_TSL) else "plain"
    pts, L = synth.map_candidates(777, 30_000)
    m = po.Map(backend)
    m.add_points(pts)
    keys, counts, xyz = m.export()
    sweep = synth.make_sweep(778, 2048, L)
    data = dict(map_keys=keys, map_counts=counts, map_xyz=xyz, map_seed=777, map_target=30_000, L=L,
                raw=sweep["raw"], q_pred=sweep["q_pred"], t_pred=sweep["t_pred"], t_last=sweep["t_last"], vel=sweep["vel"],
                q_gt=sweep["q_gt"], t_gt=sweep["t_gt"], backend=backend)
    for name, frame_id, max_res in (("full", 100, 2**31 - 1), ("cut600", 100, 600), ("init", 5, 2**31 - 1), ("neg1", 100, -1)):
        for k, v in solve_case(m, sweep, frame_id, max_res, backend).items():
            data[f"{name}_{k}"] = v
    # the tie scene (synth.lattice_scene): most keypoints have exactly tied candidate distances, so ids / order are the
    # real libstdc++ heap's; one pass with the shipped K = 20 and one with K = 5 (ties across the cut in almost every keypoint)
    tpts, tsw = synth.lattice_scene(4711, 1536)
    tm = po.Map(backend)
    tm.add_points(tpts)
    tk, tc, tx = tm.export()
    data.update(tie_map_keys=tk, tie_map_counts=tc, tie_map_xyz=tx, tie_raw=tsw["raw"], tie_q=tsw["q_pred"], tie_t=tsw["t_pred"],
                tie_t_last=tsw["t_last"])
    for name, kw, fid in (("tie", {}, 100), ("tie5", dict(max_number_neighbors=5, min_number_neighbors=5), 100), ("tieinit", {}, 5)):
        one = tm.build_plane_residuals(po.default_opts(max_num_residuals=2**31 - 1, **kw), tsw["raw"], tsw["q_pred"], tsw["t_pred"],
                                       tsw["t_last"], frame_id=fid)
        for k, v in one.items():
            if isinstance(v, np.ndarray):
                data[f"{name}_one_{k}"] = v
        data.update({f"{name}_one_num_residuals": one["neq"].num_residuals, f"{name}_one_loss": one["neq"].loss_sum,
                     f"{name}_one_num_ties": one["neq"].num_ties, f"{name}_one_success": one["neq"].success,
                     f"{name}_one_sum_candidates": one["neq"].sum_candidates, f"{name}_one_num_visited": one["neq"].num_visited})
        print(name, "keypoints", len(tsw["raw"]), "with ties", one["neq"].num_ties, "residuals", one["neq"].num_residuals)
    data["eig_solver"] = "eigen-3.3.7-ql"
    path = os.path.join(HERE, "golden_small.npz")
    np.savez_compressed(path, **data)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB; map", m.size(), "pts", m.num_voxels(), "voxels; backend", backend)
    for name in ("full", "cut600", "init", "neg1"):
        print(name, "one-pass residuals", data[f"{name}_one_num_residuals"], "ties", data[f"{name}_one_num_ties"],
              "solve iters", data[f"{name}_solve_rc"], "final t err", np.linalg.norm(data[f"{name}_solve_state"][4:7] - sweep["t_gt"]))


if __name__ == "__main__":
    main()
