"""Time line of armed passes (srl_debug_pass_stamps): where one ESIKF iteration goes, from the device's 100 MHz clock (workgroup 0
and the finishing workgroup) and the host's steady clock.  Medians over the fired passes of back-to-back solves.

The product build carries no stamp sites (they cost 0.2-0.4 us per pass): build the stamped variant first and point the library at it,

    tools/build_variant.sh stamps "-DSRL_ARM_STAMPS -DSRL_STAMP_DETAIL"
    SRL_LIB_PATH=$PWD/gpurun_in/lib_stamps.so python tools/arm_timeline.py [--configs HEADLINE,C2,C3,HEADLINE@600] [--box 1]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401

import sr_livo_amd as srl  # noqa: E402
from sr_livo_amd import synth  # noqa: E402
sys.path.insert(0, "tools")
from benchlib.stream import _EskfAdapter  # noqa: E402

INT_MAX = 2**31 - 1
PLAN = {"HEADLINE": ("HEADLINE", INT_MAX, 100), "C1": ("C1", INT_MAX, 100), "C2": ("C2", INT_MAX, 100), "C3": ("C3", INT_MAX, 100),
        "HEADLINE@600": ("HEADLINE", 600, 100)}
SLOTS = ("entry", "pose", "tile", "phase0", "phase1", "phase2", "published", "finished")


def run(name, box):
    wl, max_res, frame_id = PLAN[name]
    n_kp, map_pts, pattern, seed = synth.CONFIGS[wl]
    cands, L = synth.map_candidates(seed, map_pts)
    sweep = synth.make_sweep(seed + 1000, n_kp, L, pattern=pattern)
    lio = srl.Lio(0)
    try:
        lio.ctx.pin_thread_to_gpu_numa()
        lio.add_points_to_map(cands)
        prior_state = synth.eskf_prior(_EskfAdapter(lio), sweep["q_pred"], sweep["t_pred"], sweep["vel"]).copy()
        prior_cov = lio.eskf_get_cov().copy()
        state0 = np.concatenate([sweep["q_pred"], sweep["t_pred"], sweep["vel"], np.zeros(6)])
        lio.resident_sweep(sweep["raw"])
        if box is not None:
            lio.ctx.set_pose_box(box)
        solve = lio.bound_solver(srl.default_opts(max_num_residuals=max_res), prior_state, prior_cov, state0, sweep["t_last"], frame_id, n_kp)
        lio.ctx.set_armed_launch(2)      # a launch armed behind EVERY pass (the sweep is re-solved in place: the policy of mode 1 would leave each solve's first pass un-armed)
        lio.ctx.pass_stamps(True, read=False)
        rows = []
        for rep in range(12):
            for _ in range(24):                      # 48+ passes: the 64-row ring holds them
                rc, it, nr = solve()
            g, h = lio.ctx.pass_stamps(True)
            order = np.argsort(h[:, 0])
            g, h = g[order], h[order]
            ok = (h[:, 0] > 0) & (h[:, 3] == 1) & (g[:, 1] > 0) & (g[:, 15] > 0)
            idx = np.nonzero(ok)[0]
            for i in idx:
                if i + 1 < len(h) and ok[i + 1]:
                    w0 = g[i, :8] * 10.0        # ns
                    fn = g[i, 8:16] * 10.0
                    fx = g[i, 16:24] * 10.0
                    nxt = g[i + 1]
                    rows.append(dict(
                        w0_wait=(w0[1] - w0[0]), w0_phase0=(w0[3] - w0[1]), w0_phase1=(w0[4] - w0[3]), w0_phase2=(w0[5] - w0[4]), w0_publish=(w0[6] - w0[5]),
                        fin_wait=(fn[1] - fn[0]), fin_phase0=(fn[3] - fn[1]), fin_phase1=(fn[4] - fn[3]), fin_phase2=(fn[5] - fn[4]), fin_publish=(fn[6] - fn[5]),
                        fin_gather_mailbox=(fn[7] - fn[6]), fin_loads=(fx[0] - fn[6]), fin_bar1=(fx[1] - fx[0]), fin_sum=(fx[2] - fx[1]), fin_stores=(fx[3] - fx[2]), fin_drain=(fx[4] - fx[3]), fin_seq=(fn[7] - fx[4]), p2_loops=(fx[5] - fn[4]), p2_eigen=(fx[6] - fx[5]), p2_gate_J=(fx[7] - fx[6]), p2_rows_walk=(g[i, 24] * 10.0 - fx[7]), p2_tail=(fn[5] - g[i, 24] * 10.0), active=(fn[7] - min(w0[1], fn[1])),
                        hop_gpu=(min(nxt[1], nxt[9]) * 10.0 - fn[7]),               # mailbox written -> next pose received (device clock)
                        period_gpu=(nxt[15] - g[i, 15]) * 10.0,
                        pose_skew=(fn[1] - w0[1]),
                        host_compute=(h[i + 1, 1] - h[i, 2]),                       # result seen -> next pose written (host clock)
                        host_call=(h[i, 2] - h[i, 0])))
        med = {k: float(np.median([r[k] for r in rows])) / 1e3 for k in rows[0]} if rows else {}
        return {"name": name, "passes": len(rows), "median_us": med}
    finally:
        lio.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="HEADLINE,C2,C3,HEADLINE@600")
    ap.add_argument("--box", type=int, default=None)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "arm_timeline.json"))
    args = ap.parse_args()
    res = [run(n, args.box) for n in args.configs.split(",")]
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(res, open(args.out, "w"), indent=1)
    keys = list(res[0]["median_us"].keys()) if res and res[0]["median_us"] else []
    print("%-20s" % "us (median)" + "".join("%14s" % r["name"] for r in res))
    for k in keys:
        print("%-20s" % k + "".join("%14.2f" % r["median_us"].get(k, float("nan")) for r in res))
    print("%-20s" % "passes" + "".join("%14d" % r["passes"] for r in res))


if __name__ == "__main__":
    main()
