"""The timed workload: a stream of distinct sweeps and the node's loop over it (prefetch -> solve -> swap)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import sr_livo_amd as srl
from sr_livo_amd import synth

class _EskfAdapter:
    """lets synth.eskf_prior drive the product's eskfEstimator through the srl_lio handle"""

    def __init__(self, lio):
        self.lio = lio

    def set_noise(self, *a): self.lio.eskf_set_noise(*a)
    def scale_init_cov(self): self.lio.eskf_scale_init_cov()
    def init_imu(self, a, g): self.lio.eskf_init_imu(a, g)
    def predict(self, dt, a, g): self.lio.eskf_predict(dt, a, g)
    def get_state(self): return self.lio.eskf_get_state()
    def set_state(self, s): self.lio.eskf_set_state(s)


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def make_stream(sweep0, prior_state0, sweep_seed, n_kp, L, pattern, count, iterations_of=None, gen=None):
    """`count` distinct sweeps of one scene for the timed stream (one solve per sweep, never the same sweep twice in a row):
    own seeds (sweep_seed + 100 j), own ground-truth and predicted poses, hence own priors (the prior covariance is the scene's); raw
    points in page-locked host memory.  Entry 0 is the given sweep.
    iterations_of(entry) -> ESIKF iterations of its solve: when given, a candidate whose solve takes another number of iterations than
    sweep 0's is skipped (its seed is recorded), so that "one step" is the same amount of algorithmic work for every sweep of the stream
    and the rate stays comparable with the single-sweep figure of rounds 1-4.  At most 4 x count candidates are drawn."""
    stream = [dict(sweep=sweep0, prior_state=prior_state0, seed=sweep_seed,
                   state0=np.concatenate([sweep0["q_pred"], sweep0["t_pred"], sweep0["vel"], np.zeros(6)]))]
    want = iterations_of(stream[0]) if iterations_of else None
    skipped = []
    j = 0
    while len(stream) < max(int(count), 1) and j < 4 * max(int(count), 1):
        j += 1
        sw = gen(sweep_seed + 100 * j) if gen else synth.make_sweep(sweep_seed + 100 * j, n_kp, L, pattern=pattern)
        ps = prior_state0.copy()
        ps[0:3] = sw["t_pred"]; ps[3:7] = sw["q_pred"]; ps[7:10] = sw["vel"]
        e = dict(sweep=sw, prior_state=ps, seed=sweep_seed + 100 * j, state0=np.concatenate([sw["q_pred"], sw["t_pred"], sw["vel"], np.zeros(6)]))
        if iterations_of is not None:
            it = iterations_of(e)
            if it != want:
                skipped.append({"seed": e["seed"], "iterations": it})
                continue
        stream.append(e)
    stream[0]["skipped"] = skipped
    stream[0]["iterations"] = want
    for e in stream:
        e["pin"] = srl.PinnedArray(e["sweep"]["raw"].shape)
        e["pin"].array[:] = e["sweep"]["raw"]
    return stream


class Streamer:
    """the node's loop over a stream of sweeps on one context: prefetch of the next sweep (copy stream) -> full ESIKF solve of the current
    one from its own prior -> swap.  Every sweep crosses PCIe exactly once per solve; no host synchronisation.  One C call per step
    (srl_lio_stream_step: what the body of a C++ node's loop is -- between separate calls this harness' own language cost ~3 us per solve)."""

    def __init__(self, lio, stream, opts, prior_cov, frame_id, n_kp):
        self.lio, self.stream, self.S, self.pos = lio, stream, len(stream), 0
        for k, e in enumerate(stream):
            nx = stream[(k + 1) % self.S]
            e["step"] = lio.bound_stream_step(opts, e["prior_state"], prior_cov, e["state0"], e["sweep"]["t_last"], frame_id, n_kp, nx["pin"].array)

    def begin(self):
        self.lio.prefetch_sweep(self.stream[self.pos % self.S]["pin"].array)
        self.lio.swap_sweep()

    def step(self):
        k = self.pos
        e = self.stream[k % self.S]
        # sweep k + 1 arrives during the solve of sweep k: its upload is issued by the solve itself, beside the kernel of the first pass
        rc, it, nr = e["step"]()
        if rc:
            lib = self.lio.lib
            why = (lib.srl_lio_last_error(self.lio.h) or b"").decode(errors="replace") or (lib.srl_last_error(self.lio.ctx.h) or b"").decode(errors="replace")
            raise RuntimeError(f"stream step failed with status {rc} on sweep {k % self.S} of the stream: {why}")
        self.pos = k + 1
        return {"iters": it, "num_residuals": nr, "state": e["step"].state, "sweep": k % self.S}

    def close(self):
        for e in self.stream:
            e["pin"].close()

