"""Per-stage wall time of the frame-resident pipeline (upload -> keypoint selection -> solve -> commit) on a 1 M-point map, for frames
spread over the scene like a reconstructed sweep.  Stage times come from srl_debug_frame_timing (stream synchronised at every stage
boundary: the sum of the stages is an upper bound of the un-instrumented call, printed beside it).

    python tools/pipeline_probe.py [--frames 6000,24000,65536]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sr_livo_amd as srl  # noqa: E402
from sr_livo_amd import capi, synth  # noqa: E402


def run(ctx, cands, n_frame, reps=7, want_index=False):
    rng = np.random.default_rng(3 + n_frame)
    frame = cands[rng.choice(len(cands), n_frame, replace=False)] + rng.normal(0, 0.03, (n_frame, 3))
    pin = srl.PinnedArray(frame.shape)
    pin.array[:] = frame
    pin_world = srl.PinnedArray(frame.shape)
    q, t = np.array([1.0, 0, 0, 0]), np.zeros(3)
    f = capi.make_frame(q, t, t)
    opts = srl.default_opts(max_num_residuals=2**31 - 1)
    out = {"frame_points": n_frame}

    def one(timing):
        ctx.frame_timing(timing)
        t0 = time.perf_counter()
        ctx.frame_upload(pin.array)
        t1 = time.perf_counter()
        k = ctx.frame_select_keypoints(q, t, 1.5, want_index=want_index)
        t2 = time.perf_counter()
        neq, _ = ctx.build_residuals(f, opts)
        neq, _ = ctx.build_residuals(f, opts)
        ctx.solve_end()              # (like the host mirror: the arming policy learns that a solve is two passes -> no launch left waiting behind it)
        t3 = time.perf_counter()
        ctx.frame_commit(q, t, want_world=True, want_added=False, world_out=pin_world.array)
        t4 = time.perf_counter()
        st = ctx.frame_timing(False)
        return (len(k) if want_index else k), (t1 - t0, t2 - t1, t3 - t2, t4 - t3), st

    one(False); one(True)
    t_loop = time.perf_counter()
    plain = np.array([one(False)[1] for _ in range(reps)]) * 1e6
    ctx.map_size()                                # the last (deferred) insertion belongs to the loop
    out["loop_us_per_frame"] = (time.perf_counter() - t_loop) * 1e6 / reps
    stages = [one(True) for _ in range(reps)]
    out["keypoints"] = stages[0][0]
    med = np.median(plain, axis=0)
    out["us"] = dict(upload=float(med[0]), select=float(med[1]), two_passes=float(med[2]), commit=float(med[3]), total=float(med.sum()))
    out["sweeps_per_s"] = 1e6 / out["loop_us_per_frame"]
    out["stage_us"] = {k: float(np.median([s[2][k] for s in stages])) for k in stages[0][2]}
    ctx.map_size()
    pin.close(); pin_world.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", default="6000,24000,65536")
    ap.add_argument("--want-index", action="store_true", help="ask srl_frame_select_keypoints for the index list (the host mirror does not need it)")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "pipeline_probe.json"))
    args = ap.parse_args()
    cands, L = synth.map_candidates(7, 1_000_000)
    lio = srl.Lio(0)
    lio.ctx.pin_thread_to_gpu_numa()
    lio.add_points_to_map(cands)
    res = []
    for n in [int(x) for x in args.frames.split(",")]:
        r = run(lio.ctx, cands, n, want_index=args.want_index)
        res.append(r)
        u = r["us"]
        print(f"frame {n:6d} pts -> {r['keypoints']:5d} keypoints: upload {u['upload']:.0f}  select {u['select']:.0f}  two passes {u['two_passes']:.0f}  commit {u['commit']:.0f}"
              f"  = {u['total']:.0f} us; loop {r['loop_us_per_frame']:.0f} us per frame ({r['sweeps_per_s']:.0f} frames/s)")
        print("      stages (synchronised):", "  ".join(f"{k} {v:.0f}" for k, v in r["stage_us"].items()))
    lio.close()
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
