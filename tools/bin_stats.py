"""Keypoints per map voxel (the bins an LDS-staged search would work on) for the BASELINE sweeps, CPU only.
Prints, per workload: occupied bins, keypoints per bin (mean / median / p90), share of keypoints in bins of >= 4 / >= 8 / >= 16,
and the bytes a bin stages (27 voxels x found slabs) against the bytes its keypoints read today."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sr_livo_amd import synth, capi

for name in (sys.argv[1:] or ["HEADLINE", "C2", "C3", "C1"]):
    n_kp, map_pts, pattern, seed = synth.CONFIGS[name]
    cands, L = synth.map_candidates(seed, map_pts)
    sw = synth.make_sweep(seed + 1000, n_kp, L, pattern=pattern)
    R = synth.quat_to_rot(sw["q_pred"])
    pw = sw["raw"] @ R.T + sw["t_pred"]
    key = np.trunc(pw).astype(np.int64)
    packed = (key[:, 0] + 32768) | ((key[:, 1] + 32768) << 16) | ((key[:, 2] + 32768) << 32)
    _, counts = np.unique(packed, return_counts=True)
    per_kp = np.repeat(counts, counts)            # bin size seen by each keypoint
    print(f"{name:9s} keypoints {n_kp:6d}  bins {len(counts):6d}  keypoints/bin mean {counts.mean():6.2f} median {np.median(counts):4.0f} p90 {np.percentile(counts, 90):5.0f} max {counts.max():5d}"
          f" | keypoints in bins >=4: {np.mean(per_kp >= 4) * 100:5.1f} %  >=8: {np.mean(per_kp >= 8) * 100:5.1f} %  >=16: {np.mean(per_kp >= 16) * 100:5.1f} %  >=64: {np.mean(per_kp >= 64) * 100:5.1f} %")
