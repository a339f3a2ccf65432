"""Does srl_sweep_prefetch overlap the H2D of sweep k+1 with the solve of sweep k?  Context-level host timings (us)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sr_livo_amd as srl
from sr_livo_amd import capi, synth

n_kp, map_pts, pattern, seed = synth.CONFIGS["HEADLINE"]
pts, L = synth.map_candidates(seed, map_pts)
sw = synth.make_sweep(seed + 1000, n_kp, L, pattern=pattern)
lio = srl.Lio(0); lio.add_points_to_map(pts); ctx = lio.ctx
f = capi.make_frame(sw["q_pred"], sw["t_pred"], sw["t_last"]); opts = srl.default_opts(max_num_residuals=2**31 - 1)
pin = [srl.PinnedArray(sw["raw"].shape) for _ in range(2)]
for p in pin: p.array[:] = sw["raw"]
ctx.sweep_upload(pin[0].array)
for _ in range(20): ctx.build_residuals(f, opts)
N = 300
def med(fn):
    ts = np.empty((N, 3))
    for k in range(N):
        ts[k] = fn(k)
    tot = ts.sum(1) * 1e6
    return np.median(ts, 0) * 1e6, tot.mean(), np.percentile(tot, [10, 50, 90, 99, 100]), tot
def solve_only(k):
    t0 = time.perf_counter(); ctx.build_residuals(f, opts); return (0, time.perf_counter() - t0, 0)
def sequential(k):
    t0 = time.perf_counter(); ctx.sweep_upload(pin[k & 1].array); t1 = time.perf_counter(); ctx.build_residuals(f, opts); return (t1 - t0, time.perf_counter() - t1, 0)
def pipelined(k):
    t0 = time.perf_counter(); ctx.sweep_prefetch(pin[(k + 1) & 1].array); t1 = time.perf_counter(); ctx.build_residuals(f, opts); t2 = time.perf_counter(); ctx.sweep_swap(); return (t1 - t0, t2 - t1, time.perf_counter() - t2)
def pipelined_late(k):      # prefetch issued, then 40 us of host time pass before the solve (copy has a head start)
    t0 = time.perf_counter(); ctx.sweep_prefetch(pin[(k + 1) & 1].array)
    while time.perf_counter() - t0 < 60e-6: pass
    t1 = time.perf_counter(); ctx.build_residuals(f, opts); t2 = time.perf_counter(); ctx.sweep_swap(); return (0, t2 - t1, time.perf_counter() - t2)
for name, fn in (("solve_only", solve_only), ("sequential", sequential), ("pipelined", pipelined), ("pipelined_headstart", pipelined_late), ("solve_only", solve_only)):
    m, mean, pc, tot = med(fn)
    print(f"{name:20s} median upload/prefetch {m[0]:6.1f}  solve {m[1]:6.1f}  swap {m[2]:5.1f}   mean step {mean:7.1f} us  p10/50/90/99/max {np.round(pc, 0)}  slow steps (>2x median) at {np.nonzero(tot > 2 * pc[1])[0][:12]}")
