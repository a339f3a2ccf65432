"""experiment: host-side stamps of srl_build_residuals (no events): argument preparation, the launch call, the wait; and the
solve loop's time per ESIKF iteration around them"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import sr_livo_amd as srl
from sr_livo_amd import capi, synth
n_kp, map_pts, pattern, seed = synth.CONFIGS["HEADLINE"]
cands, L = synth.map_candidates(seed, map_pts)
sw = synth.make_sweep(seed + 1000, n_kp, L, pattern=pattern)
lio = srl.Lio(0)
print("numa", lio.ctx.pin_thread_to_gpu_numa())
lio.add_points_to_map(cands)
class E:  # minimal eskf-like for synth.eskf_prior
    pass
from oracle import pyoracle as po
eo = po.Eskf(); synth.eskf_prior(eo, sw["q_pred"], sw["t_pred"], sw["vel"])
s0, P0 = eo.get_state().copy(), eo.get_cov().copy()
st = np.concatenate([sw["q_pred"], sw["t_pred"], sw["vel"], np.zeros(6)])
opts = srl.default_opts(max_num_residuals=2**31 - 1)
lio.resident_sweep(sw["raw"]) if hasattr(lio, "resident_sweep") else None
for rep in range(3):
    lio.ctx.set_profiling(3)
    t0 = time.perf_counter(); iters = 0
    for _ in range(200):
        lio.eskf_set_state(s0); lio.eskf_set_cov(P0)
        r = lio.update_iekf(opts, None if hasattr(lio, "resident_sweep") else sw["raw"], st, sw["t_last"], n_resident=n_kp)
        iters += r["iters"]
    wall = (time.perf_counter() - t0) / iters * 1e6
    t = lio.ctx.timing(); lio.ctx.set_profiling(0)
    c = max(t.calls, 1)
    print(f"per ESIKF iteration: wall(py loop incl. state resets) {wall:.1f} us | prep {t.sum_assoc_ms / c * 1e3:.2f} launch call {t.sum_host_launch_us / c:.2f} "
          f"enqueue-rest+overlap cb {t.sum_reduce_ms / c * 1e3:.2f} wait {t.sum_host_wait_us / c:.2f} total in call {t.sum_host_total_us / c:.2f} us", flush=True)
lio.close()
