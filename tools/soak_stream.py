"""Soak test of the STREAM (GPU box): distinct sweeps of several sizes through prefetch (by the solve or before it) -> solve -> swap, armed
launches surviving the swaps; every solved state and covariance must equal, bit for bit, the one the same sweep gave with one launch per
iteration.  A stale read of a prefetched sweep (the waiting launch reads the staging buffer the DMA filled while it waited), a pose box
granule mixed up across sweeps, a launch fired for the wrong buffer: any of them shows as a mismatch.
    python tools/soak_stream.py [solves] [config]      config: C1 (default) | HEADLINE | C1@600"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import sr_livo_amd as srl  # noqa: E402
from sr_livo_amd import synth  # noqa: E402
sys.path.insert(0, "tools")
from benchlib.stream import _EskfAdapter  # noqa: E402

solves = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
cfg = sys.argv[2] if len(sys.argv) > 2 else "C1"
wl, max_res = (cfg.split("@")[0], int(cfg.split("@")[1])) if "@" in cfg else (cfg, 2**31 - 1)
n_kp, map_pts, pattern, seed = synth.CONFIGS[wl]
cands, L = synth.map_candidates(seed, map_pts)
lio = srl.Lio(0)
lio.ctx.pin_thread_to_gpu_numa()
lio.add_points_to_map(cands)
sizes = [n_kp, n_kp - 96, n_kp, n_kp - 7, n_kp, n_kp]          # shorter sweeps are served by the waiting launch, longer ones cancel it
sweeps = []
base = None
for j, n in enumerate(sizes):
    sw = synth.make_sweep(seed + 5000 + j, n, L, pattern=pattern)
    if base is None:
        base = synth.eskf_prior(_EskfAdapter(lio), sw["q_pred"], sw["t_pred"], sw["vel"]).copy()
        cov = lio.eskf_get_cov().copy()
    ps = base.copy(); ps[0:3] = sw["t_pred"]; ps[3:7] = sw["q_pred"]; ps[7:10] = sw["vel"]
    pin = srl.PinnedArray(sw["raw"].shape); pin.array[:] = sw["raw"]
    st0 = np.concatenate([sw["q_pred"], sw["t_pred"], sw["vel"], np.zeros(6)])
    sweeps.append(dict(sw=sw, pin=pin, n=n, solve=lio.bound_solver(srl.default_opts(max_num_residuals=max_res), ps, cov, st0, sw["t_last"], 100, n)))
S = len(sweeps)


def run(count, during, check):
    lio.prefetch_sweep(sweeps[0]["pin"].array); lio.swap_sweep()
    bad = 0
    for k in range(count):
        e = sweeps[k % S]
        (lio.prefetch_sweep_during_solve if during(k) else lio.prefetch_sweep)(sweeps[(k + 1) % S]["pin"].array)
        rc, it, nr = e["solve"]()
        assert rc == 0, (k, rc)
        got = (it, nr, e["solve"].state.copy(), lio.eskf_get_cov().copy())
        if check is None:
            e.setdefault("ref", got)
        else:
            r = e["ref"]
            if not (got[0] == r[0] and got[1] == r[1] and np.array_equal(got[2], r[2]) and np.array_equal(got[3], r[3])):
                bad += 1
                if bad < 5:
                    print(f"MISMATCH solve {k} sweep {k % S}: iters {got[0]} / {r[0]}, residuals {got[1]} / {r[1]}, max state diff {np.max(np.abs(got[2] - r[2])):.3e}", flush=True)
        lio.swap_sweep()
    lio.ctx.disarm()
    return bad


lio.ctx.set_armed_launch(False)
lio.ctx.set_bound_culling(0)
run(2 * S, lambda k: False, None)                      # references: one launch per iteration, every pass visits every found voxel
lio.ctx.set_bound_culling(1)                           # (round 6) ... the soak: passes after the first start from the neighbourhood bounds
lio.ctx.set_armed_launch(True)
s0 = lio.ctx.arm_stats()
t0 = time.time()
bad = run(solves, lambda k: (k // 97) % 2 == 0, True)   # the upload issued by the solve / before it, in alternating stretches
el = time.time() - t0
s1 = lio.ctx.arm_stats()
d = {k: s1[k] - s0[k] for k in s1}
print(f"soak_stream {cfg}: {solves} solves of {S} distinct sweeps (sizes {sizes}) in {el:.1f} s = {solves / el:.0f} sweeps/s; mismatches vs one launch per iteration: {bad}; "
      f"armed launches {d}")
for e in sweeps:
    e["pin"].close()
lio.close()
raise SystemExit(1 if bad else 0)
