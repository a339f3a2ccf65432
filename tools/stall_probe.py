"""Where do the HIP runtime's millisecond stalls fall?  Long runs of upload+solve steps; prints every step > 1 ms."""
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
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
T0 = time.perf_counter()
def run(name, fn):
    ts = np.empty((N, 2)); at = np.empty(N)
    for k in range(N):
        t0 = time.perf_counter(); fn(k); t1 = time.perf_counter(); ctx.build_residuals(f, opts); t2 = time.perf_counter()
        ts[k] = (t1 - t0, t2 - t1); at[k] = t0 - T0
    slow = np.nonzero(ts.sum(1) > 1e-3)[0]
    print(f"{name:10s} median {np.median(ts.sum(1))*1e6:6.1f} us mean {ts.sum(1).mean()*1e6:6.1f} us; slow steps: " +
          ", ".join(f"#{k} at {at[k]:.2f}s upload {ts[k,0]*1e3:.1f} ms solve {ts[k,1]*1e3:.1f} ms" for k in slow[:10]), flush=True)
for rep in range(2):
    run("solve", lambda k: None)
    run("pinned", lambda k: ctx.sweep_upload(pin[k & 1].array))
    run("pageable", lambda k: ctx.sweep_upload(sw["raw"]))
