"""Load-imbalance probe: the headline sweep vs the same sweep with every keypoint replaced by ONE of its points
(identical per-keypoint cost in every wave).  Prints the association kernel time (HIP events) of both."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sr_livo_amd as srl
from sr_livo_amd import capi, synth

n_kp, map_pts, pattern, seed = synth.CONFIGS["HEADLINE"]
cands, L = synth.map_candidates(seed, map_pts)
sw = synth.make_sweep(seed + 1000, n_kp, L, pattern=pattern)
ctx = srl.Context(0)
ctx.map_insert(cands)
opts = srl.default_opts(max_num_residuals=2**31 - 1)
f = capi.make_frame(sw["q_pred"], sw["t_pred"], sw["t_last"])
def run(raw, tag):
    ctx.sweep_upload(raw)
    for _ in range(3): ctx.build_residuals(f, opts)
    ctx.set_profiling(1)
    for _ in range(20): neq, _ = ctx.build_residuals(f, opts)
    t = ctx.timing(); ctx.set_profiling(0)
    print(tag, "assoc_us", round(t.sum_assoc_ms / t.calls * 1e3, 2), "cands/kp", t.sum_algorithmic_bytes / t.calls / n_kp, "res", neq.num_residuals, flush=True)
run(sw["raw"], "real")
rng = np.random.default_rng(0)
for i in rng.integers(0, n_kp, 6):
    run(np.repeat(sw["raw"][i:i + 1], n_kp, 0), f"uniform[{i}]")
# shuffled order (same multiset of keypoints): does the assignment of keypoints to waves matter?
run(sw["raw"][rng.permutation(n_kp)], "shuffled")
