"""Kernel time of srl_build_residuals per workload, no result checks (for timing-only experiment builds):
SRL_LIB_PATH=... python tools/ktime.py [HEADLINE C2 C3 H600 C1]   -> us per launch (HIP events inside the library), fused and unfused"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sr_livo_amd as srl
from sr_livo_amd import capi, synth

import numpy as np
REPS = 60
SORTED = "--sorted" in sys.argv          # keypoints pre-sorted by map voxel at the predicted pose (locality experiment)
names = [a for a in sys.argv[1:] if not a.startswith("--")]
for name in (names or ["HEADLINE", "C2", "C3", "H600"]):
    cap = 600 if name == "H600" else 2**31 - 1
    n_kp, map_pts, pattern, seed = synth.CONFIGS["HEADLINE" if name == "H600" else name]
    cands, L = synth.map_candidates(seed, map_pts)
    sw = synth.make_sweep(seed + 1000, n_kp, L, pattern=pattern)
    if SORTED:
        pw = sw["raw"] @ synth.quat_to_rot(sw["q_pred"]).T + sw["t_pred"]
        key = np.trunc(pw).astype(np.int64) + 32768
        order = np.argsort(key[:, 0] | (key[:, 1] << 16) | (key[:, 2] << 32), kind="stable")
        sw["raw"] = np.ascontiguousarray(sw["raw"][order])
    ctx = srl.Context(0)
    ctx.map_insert(cands)
    ctx.sweep_upload(sw["raw"])
    opts = srl.default_opts(max_num_residuals=cap)
    f = capi.make_frame(sw["q_pred"], sw["t_pred"], sw["t_last"])
    res = []
    for fused in (0, 1):
        ctx.set_fused_reduce(fused)
        for _ in range(5):
            ctx.build_residuals(f, opts)
        ctx.set_profiling(1)
        t0 = ctx.timing()
        for _ in range(REPS):
            out, _rc = ctx.build_residuals(f, opts)
        t1 = ctx.timing()
        ctx.set_profiling(0)
        res.append(((t1.sum_assoc_ms - t0.sum_assoc_ms) / REPS * 1e3, out.num_residuals, out.num_fallback))
    print(f"{os.path.basename(os.environ.get('SRL_LIB_PATH', 'default')):18s} {name:9s} assoc-only {res[0][0]:7.2f} us  fused {res[1][0]:7.2f} us   residuals {res[1][1]} fallback {res[1][2]}{' sorted' if SORTED else ''}", flush=True)
    ctx.close()
