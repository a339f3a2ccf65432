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
# ---- time line of a headline-size solve (synthetic 64k sweep on the 1M map)
if len(sys.argv) > 3:
    from sr_livo_amd import synth
    wl = sys.argv[3]
    n_kp, map_pts, pattern, seed = synth.CONFIGS[wl]
    cands, L = synth.map_candidates(seed, map_pts)
    sw = synth.make_sweep(seed + 1000, n_kp, L, pattern=pattern)
    lio = srl.Lio(0)
    lio.add_points_to_map(cands)
    class A:
        def __init__(s, l): s.l = l
        def set_noise(s, *a): s.l.eskf_set_noise(*a)
        def scale_init_cov(s): s.l.eskf_scale_init_cov()
        def init_imu(s, a, g): s.l.eskf_init_imu(a, g)
        def predict(s, dt, a, g): s.l.eskf_predict(dt, a, g)
        def get_state(s): return s.l.eskf_get_state()
        def set_state(s, x): s.l.eskf_set_state(x)
    ps = synth.eskf_prior(A(lio), sw["q_pred"], sw["t_pred"], sw["vel"]).copy()
    pc = lio.eskf_get_cov().copy()
    st0 = np.concatenate([sw["q_pred"], sw["t_pred"], sw["vel"], np.zeros(6)])
    max_res = int(sys.argv[4]) if len(sys.argv) > 4 else 2**31 - 1
    lio.ctx.set_iekf_exact_lu(bool(exact))
    lio.resident_sweep(sw["raw"])
    solve = lio.bound_solver(srl.default_opts(max_num_residuals=max_res), ps, pc, st0, sw["t_last"], 100, n_kp)
    for _ in range(5):
        solve()
    lio.ctx.solve_stamps(True)
    for rep in range(3):
        t0 = time.perf_counter(); rc, it, nr = solve(); dt = time.perf_counter() - t0
        st = lio.ctx.solve_stamps(True, fetch=True)
        base = st[0][0]
        print(wl, "solve %.1f us, %d passes" % (dt * 1e6, it))
        for i in range(it):
            row = st[i]
            print("  pass %d (us from first stamp): prior %6.2f..%6.2f | own tiles done %6.2f | rows summed %6.2f | update done %6.2f | handed over %6.2f || wg0: row out %6.2f pose seen %6.2f"
                  % ((i,) + tuple((row[k] - base) / 100.0 for k in (0, 1, 2, 3, 4, 5, 8, 9))))
            print("         wg0 tile: start %6.2f | phase 0 done %6.2f | phase 1 done %6.2f | phase 2 done %6.2f" % tuple((row[k] - base) / 100.0 for k in (10, 11, 12, 13)))
            print("    finisher tile: start %6.2f | phase 0 done %6.2f | phase 1 done %6.2f | phase 2 done %6.2f" % tuple((row[k] - base) / 100.0 for k in (15, 6, 7, 14)))
    lio.ctx.solve_stamps(False)
    N = 200
    t0 = time.perf_counter()
    for _ in range(N): solve()
    print("persistent: %.1f us per solve" % ((time.perf_counter() - t0) / N * 1e6))
    lio.set_persistent_solve(False)
    for _ in range(5): solve()
    t0 = time.perf_counter()
    for _ in range(N): solve()
    print("per-iteration: %.1f us per solve" % ((time.perf_counter() - t0) / N * 1e6))
    lio.close()
ctx.close()
