"""Build the 1M-point headline map, then insert one 24k-point sweep (for rocprofv3 --kernel-trace --stats)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sr_livo_amd as srl
from sr_livo_amd import synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
pts, L = synth.map_candidates(7, N)
lio = srl.Lio(0)
t0 = time.perf_counter(); lio.add_points_to_map(pts); t1 = time.perf_counter()
rng = np.random.default_rng(3)
frame = pts[rng.choice(len(pts), 24_000, replace=False)] + rng.normal(0, 0.03, (24_000, 3))
lio.add_points_to_map(frame)
t2 = time.perf_counter(); lio.add_points_to_map(frame + 0.05); t3 = time.perf_counter()
print(f"build {N}: {(t1 - t0) * 1e3:.2f} ms (first call: includes allocation), 24k-point insert: {(t3 - t2) * 1e3:.3f} ms")
