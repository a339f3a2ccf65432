"""Debug: distribution of workgroup start/end times inside one association launch (SRL_ABLATE=128 build hook)."""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sr_livo_amd as srl
from sr_livo_amd import capi, synth

n_kp, map_pts, pattern, seed = synth.CONFIGS["HEADLINE"]
cands, L = synth.map_candidates(seed, map_pts)
sw = synth.make_sweep(seed + 1000, n_kp, L, pattern=pattern)
ctx = srl.Context(0)
ctx.map_insert(cands)
ctx.sweep_upload(sw["raw"])
ctx.lib.srl_debug_set_ablate(ctx.h, 128)      # workgroup start / end stamps (results stay valid: nothing is switched off)
opts = srl.default_opts(max_num_residuals=2**31 - 1)
f = capi.make_frame(sw["q_pred"], sw["t_pred"], sw["t_last"])
lib = ctx.lib
lib.srl_debug_block_times.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
for rep in range(4):
    ctx.build_residuals(f, opts)
    out = np.zeros((2048, 3)); nb = C.c_int()
    lib.srl_debug_block_times(ctx.h, out.ctypes.data_as(C.c_void_p), 2048, C.byref(nb))
    t = out[: nb.value]
    t0 = t[:, 0].min()
    st = (t[:, 0] - t0) * 0.01; en = (t[:, 1] - t0) * 0.01; du = en - st          # microseconds
    print(f"rep {rep}: blocks {nb.value}  start spread {st.max():.2f} us  duration mean {du.mean():.2f} min {du.min():.2f} "
          f"p50 {np.median(du):.2f} p90 {np.percentile(du, 90):.2f} p99 {np.percentile(du, 99):.2f} max {du.max():.2f}  "
          f"end: mean {en.mean():.2f} p50 {np.median(en):.2f} p99 {np.percentile(en, 99):.2f} last {en.max():.2f}")
    if rep == 3:
        for x in range(8):
            sel = (t[:, 2].astype(int) & 15) == x
            if sel.any():
                print(f"  xcc {x}: {sel.sum()} blocks, start mean {st[sel].mean():.2f}, duration mean {du[sel].mean():.2f}, last end {en[sel].max():.2f}")
# ---- extra diagnostics on the last repetition
h, edges = np.histogram(du, bins=16)
print("duration histogram (us):", " ".join(f"{e:.0f}:{c}" for e, c in zip(edges[:-1], h)))
idx = np.arange(nb.value)
for lo in range(0, nb.value, 128):
    sel = (idx >= lo) & (idx < lo + 128)
    print(f"  blocks {lo:4d}..{lo + 127:4d}: start {st[sel].mean():5.2f}  duration mean {du[sel].mean():5.2f} max {du[sel].max():5.2f}  xcc set {sorted(set((t[sel, 2].astype(int) & 15).tolist()))}")
print("corr(duration, start) =", np.corrcoef(du, st)[0, 1])
# candidates per block (cost proxy) vs duration
ids, status, ncand = None, None, None
