"""Per-stage wall time of the frame-resident pipeline on a 1M-point map (informational; not the headline metric)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sr_livo_amd as srl
from sr_livo_amd import capi, synth

n_frame = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
cands, L = synth.map_candidates(7, 1_000_000)
sw = synth.make_sweep(8, n_frame, L)
lio = srl.Lio(0)
lio.add_points_to_map(cands)
ctx = lio.ctx
q, t = sw["q_pred"], sw["t_pred"]
st = np.zeros((12, 17)); st[:, 0] = 100.0 + 0.01 * np.arange(12); st[:, 10] = 1.0
rel = np.sort(np.random.default_rng(0).uniform(0, 100, n_frame))
def timed(f, reps=5):
    f(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); r = f(); ts.append(time.perf_counter() - t0)
    return r, 1e3 * float(np.median(ts))
_, t_und = timed(lambda: ctx.frame_undistort(sw["raw"], rel, st, 100.0, capi.MC_CONSTANT_VELOCITY))
_, t_up = timed(lambda: ctx.frame_upload(sw["raw"]))
kidx, t_sel = timed(lambda: ctx.frame_select_keypoints(q, t, 1.5))
f = capi.make_frame(q, t, sw["t_last"]); opts = srl.default_opts(max_num_residuals=2**31 - 1)
_, t_it = timed(lambda: ctx.build_residuals(f, opts))
_, t_commit = timed(lambda: ctx.frame_commit(sw["q_gt"], sw["t_gt"], want_world=False), reps=3)
print(f"frame {n_frame} pts: undistort {t_und:.2f} ms, upload {t_up:.2f} ms, select_keypoints {t_sel:.2f} ms -> {len(kidx)} keypoints, "
      f"one ESIKF pass {t_it * 1e3:.0f} us, commit (transform + map insert) {t_commit:.2f} ms")

# a frame spread over the scene like a real reconstructed sweep (24k points drawn from the map's surface candidates)
rng = np.random.default_rng(3)
frame = cands[rng.choice(len(cands), 24_000, replace=False)] + rng.normal(0, 0.03, (24_000, 3))
ident_q = np.array([1.0, 0, 0, 0]); zero = np.zeros(3)
_, t_up2 = timed(lambda: ctx.frame_upload(frame))
k2, t_sel2 = timed(lambda: ctx.frame_select_keypoints(ident_q, zero, 1.5))
_, t_commit2 = timed(lambda: ctx.frame_commit(ident_q, zero, want_world=False), reps=3)
print(f"scene-spread frame 24000 pts: upload {t_up2:.2f} ms, select_keypoints {t_sel2:.2f} ms -> {len(k2)} keypoints, commit {t_commit2:.2f} ms")
