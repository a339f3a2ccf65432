"""Two-stage voxel visit (near voxels first): kernel time and bit-parity against the one-stage visit, per near radius.
usage: python tools/near_probe.py [fractions...]      (fraction of size_voxel_map; 0 = one stage)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sr_livo_amd as srl
from sr_livo_amd import capi, synth

fracs = [float(x) for x in sys.argv[1:]] or [0.0, 0.4, 0.5, 0.6, 0.75, 1.0]
REPS = 40


def run(ctx, f, opts, frac):
    ctx.set_near_radius(frac)
    for _ in range(5):
        ctx.build_residuals(f, opts)
    ctx.set_profiling(1)
    t0 = ctx.timing()
    for _ in range(REPS):
        out, _rc = ctx.build_residuals(f, opts)
    t1 = ctx.timing()
    ctx.set_profiling(0)
    ctx.set_taps(1)
    ctx.build_residuals(f, opts)
    ids, status, ncand = ctx.fetch_neighbors()
    ctx.set_taps(0)
    us = (t1.sum_assoc_ms - t0.sum_assoc_ms) / REPS * 1e3
    return us, ids, status, ncand, np.array(out.HtH[:] + out.Hth[:] + [out.loss_sum, out.num_residuals, out.sum_candidates])


for name, cap in (("HEADLINE", 2**31 - 1), ("C2", 2**31 - 1), ("C3", 2**31 - 1), ("HEADLINE", 600), ("C1", 2**31 - 1)):
    n_kp, map_pts, pattern, seed = synth.CONFIGS[name]
    cands, L = synth.map_candidates(seed, map_pts)
    sw = synth.make_sweep(seed + 1000, n_kp, L, pattern=pattern)
    ctx = srl.Context(0)
    ctx.map_insert(cands)
    ctx.sweep_upload(sw["raw"])
    opts = srl.default_opts(max_num_residuals=cap)
    f = capi.make_frame(sw["q_pred"], sw["t_pred"], sw["t_last"])
    base = None
    for fr in fracs:
        us, ids, status, ncand, hth = run(ctx, f, opts, fr)
        if base is None:
            base = (ids, status, ncand, hth)
            note = "reference"
        else:
            same_ids = np.array_equal(ids, base[0]); same_st = np.array_equal(status, base[1]); same_nc = np.array_equal(ncand, base[2])
            note = f"ids {'==' if same_ids else '!='} status {'==' if same_st else '!='} ncand {'==' if same_nc else '!='}"
            if hth is not None:
                note += f" HtH {'==' if np.array_equal(hth, base[3]) else '!='}"
        print(f"{name:9s} cap {cap if cap < 2**30 else 'inf':>4} near {fr:4.2f}: kernel {us:7.2f} us  {note}", flush=True)
    ctx.close()
