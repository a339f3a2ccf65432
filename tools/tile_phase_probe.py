"""A/B probe: workgroup 0's phase times inside the persistent solve kernel (SRL_LIB_PATH selects the build)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sr_livo_amd as srl
from sr_livo_amd import synth

wl = sys.argv[1] if len(sys.argv) > 1 else "HEADLINE"
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
lio.resident_sweep(sw["raw"])
solve = lio.bound_solver(srl.default_opts(max_num_residuals=2**31 - 1), ps, pc, st0, sw["t_last"], 100, n_kp)
lio.ctx.solve_stamps(True)
rows = []
for rep in range(4):
    solve()
    st = lio.ctx.solve_stamps(True, fetch=True)
    for i in range(2):
        r = st[i]
        if r[10] and r[13]:
            rows.append(((r[11] - r[10]) / 100.0, (r[12] - r[11]) / 100.0, (r[13] - r[12]) / 100.0))
            print("rep %d pass %d wg0: phase0 %5.2f  phase1 %5.2f  phase2 %5.2f  total %5.2f us" % ((rep, i) + rows[-1] + (sum(rows[-1]),)), flush=True)
print(os.environ.get("SRL_LIB_PATH", "default"), "median phases", np.median(np.array(rows), axis=0))
lio.close()
