"""Host-side time line of the stream loop (bench.py: prefetch -> solve -> swap): medians of the three calls' wall times, the whole step,
and the same with the sweep resident (no prefetch / swap), per configuration.  python tools/stream_probe.py [--configs C2,HEADLINE@600,HEADLINE]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401

import sr_livo_amd as srl  # noqa: E402
from sr_livo_amd import synth  # noqa: E402
sys.path.insert(0, "tools")
from benchlib.stream import _EskfAdapter, make_stream  # noqa: E402

INT_MAX = 2**31 - 1
PLAN = {"HEADLINE": ("HEADLINE", INT_MAX), "C1": ("C1", INT_MAX), "C2": ("C2", INT_MAX), "C3": ("C3", INT_MAX), "HEADLINE@600": ("HEADLINE", 600)}


def run(name, n_solves=400):
    wl, max_res = PLAN[name]
    n_kp, map_pts, pattern, seed = synth.CONFIGS[wl]
    cands, L = synth.map_candidates(seed, map_pts)
    sweep = synth.make_sweep(seed + 1000, n_kp, L, pattern=pattern)
    lio = srl.Lio(0)
    out = {"name": name}
    try:
        lio.ctx.pin_thread_to_gpu_numa()
        lio.add_points_to_map(cands)
        prior_state = synth.eskf_prior(_EskfAdapter(lio), sweep["q_pred"], sweep["t_pred"], sweep["vel"]).copy()
        prior_cov = lio.eskf_get_cov().copy()
        opts = srl.default_opts(max_num_residuals=max_res)
        stream = make_stream(sweep, prior_state, seed + 1000, n_kp, L, pattern, 2)
        for e in stream:
            e["solve"] = lio.bound_solver(opts, e["prior_state"], prior_cov, e["state0"], e["sweep"]["t_last"], 100, n_kp)
        S = len(stream)
        for mode, label in ((1, "armed"), (0, "unarmed")):
            lio.ctx.set_armed_launch(mode)
            lio.prefetch_sweep(stream[0]["pin"].array); lio.swap_sweep()
            T = np.zeros((n_solves, 4))
            its = 0
            for k in range(n_solves + 50):
                t0 = time.perf_counter()
                lio.prefetch_sweep_during_solve(stream[(k + 1) % S]["pin"].array)
                t1 = time.perf_counter()
                rc, it, nr = stream[k % S]["solve"]()
                t2 = time.perf_counter()
                lio.swap_sweep()
                t3 = time.perf_counter()
                if k >= 50:
                    T[k - 50] = (t1 - t0, t2 - t1, t3 - t2, t3 - t0)
                    its += it
            lio.ctx.disarm()
            med = np.median(T, axis=0) * 1e6
            out[label + "_stream_us"] = dict(prefetch=med[0], solve=med[1], swap=med[2], step=med[3], iters=its / n_solves, arm=lio.ctx.arm_stats())
            # resident: sweep 0 re-solved
            lio.resident_sweep(stream[0]["sweep"]["raw"])
            R = np.zeros(n_solves)
            for k in range(n_solves + 50):
                t0 = time.perf_counter()
                stream[0]["solve"]()
                if k >= 50:
                    R[k - 50] = time.perf_counter() - t0
            lio.ctx.disarm()
            out[label + "_resident_us"] = float(np.median(R) * 1e6)
        # resident with a launch armed behind every pass (rounds 1-4's loop)
        lio.ctx.set_armed_launch(2)
        R = np.zeros(n_solves)
        for k in range(n_solves + 50):
            t0 = time.perf_counter()
            stream[0]["solve"]()
            if k >= 50:
                R[k - 50] = time.perf_counter() - t0
        lio.ctx.disarm()
        out["always_armed_resident_us"] = float(np.median(R) * 1e6)
        for e in stream:
            e["pin"].close()
    finally:
        lio.close()
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="C2,HEADLINE@600,HEADLINE")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "stream_probe.json"))
    a = ap.parse_args()
    res = [run(n) for n in a.configs.split(",")]
    json.dump(res, open(a.out, "w"), indent=1, default=float)
    for r in res:
        print(json.dumps(r, default=float))
