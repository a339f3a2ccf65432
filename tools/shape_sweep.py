"""Launch-shape sweep of the association kernel on the BASELINE configurations (tuning experiment; GPU box).
    python tools/shape_sweep.py C2 C3 ...   ->  kernel us per (keypoints per wave, waves per workgroup, fused)"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import sr_livo_amd as srl
from sr_livo_amd import capi, synth

MAX_RES = 2**31 - 1
args = sys.argv[1:]
if args and args[0].startswith("--max="):
    MAX_RES = int(args.pop(0)[6:])
for wl in (args or ["C1", "C2", "C3", "HEADLINE"]):
    n_kp, map_pts, pattern, seed = synth.CONFIGS[wl]
    cands, L = synth.map_candidates(seed, map_pts)
    sw = synth.make_sweep(seed + 1000, n_kp, L, pattern=pattern)
    ctx = srl.Context(0)
    ctx.map_insert(cands)
    ctx.sweep_upload(sw["raw"])
    f = capi.make_frame(sw["q_pred"], sw["t_pred"], sw["t_last"])
    opts = srl.default_opts(max_num_residuals=MAX_RES)
    res = {}
    for fused in (0, 1):
        ctx.set_fused_reduce(fused)
        for kpw in (0, 2, 3, 4, 6, 8, 12, 16):
            for wpb in ((0,) if kpw == 0 else ((16,) if kpw in (2, 3, 6, 12) else (4, 16))):
                ctx.set_launch_shape(kpw, wpb)
                print(wl, 'fused', fused, 'kpw', kpw, 'wpb', wpb, file=sys.stderr, flush=True)
                for _ in range(5):
                    ctx.build_residuals(f, opts)
                ctx.set_profiling(2)
                t0 = time.perf_counter()
                for _ in range(40):
                    ctx.build_residuals(f, opts)
                wall = (time.perf_counter() - t0) / 40 * 1e6
                tm = ctx.timing()
                ctx.set_profiling(0)
                res[(fused, kpw, wpb)] = (tm.sum_assoc_ms / max(tm.calls, 1) * 1e3, wall)
    print(wl, n_kp, {f"fused{k[0]}_kpw{k[1]}_wpb{k[2]}": (round(v[0], 1), round(v[1], 1)) for k, v in res.items()})
    ctx.close()
