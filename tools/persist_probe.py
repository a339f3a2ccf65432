"""Debug / timing probe for the persistent solve: one golden solve through srl_solve_iekf, printed step by step."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sr_livo_amd as srl
from sr_livo_amd import capi

g = dict(np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "golden_small.npz"), allow_pickle=False))
prefix = sys.argv[1] if len(sys.argv) > 1 else "full"
max_res = {"full": 2**31 - 1, "cut600": 600, "init": 2**31 - 1}[prefix]
frame_id = 5 if prefix == "init" else 100
exact = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ctx = srl.Context(0)
print("ctx ok", flush=True)
ctx.map_upload(g["map_keys"], g["map_counts"], g["map_xyz"])
ctx.sweep_upload(g["raw"])
ctx.set_iekf_exact_lu(bool(exact))
st0 = g[f"{prefix}_state0"]
frame = capi.make_frame(st0[0:4], st0[4:7], g["t_last"], frame_id=frame_id)
print("launching", prefix, "exact", exact, flush=True)
t0 = time.perf_counter()
r = ctx.solve_iekf(frame, srl.default_opts(max_num_residuals=max_res), 0.001, g[f"{prefix}_eskf_state0"], g[f"{prefix}_eskf_cov0"], log_iters=20)
print("returned in %.3f ms" % ((time.perf_counter() - t0) * 1e3), {k: r[k] for k in ("rc", "verdict", "iterations", "covariance_updated", "observed", "num_residuals")}, flush=True)
ref = g[f"{prefix}_solve_eskf_state"]
print("state err", float(np.max(np.abs(r["state"] - ref))), "cov err", float(np.max(np.abs(r["cov"] - g[f"{prefix}_solve_eskf_cov"]))), "iters ref", int(g[f"{prefix}_solve_rc"]), flush=True)
if r["log"] is not None and len(r["log"]):
    print("d_x log err", float(np.max(np.abs(r["log"][:, 42:59] - g[f"{prefix}_solve_log"][: len(r["log"]), 42:59]))), flush=True)
ctx.close()
