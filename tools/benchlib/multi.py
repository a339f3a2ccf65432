"""Legs of `bench.py --gpus N` (N > 1) BEHIND the timed region: the OTHER transport on the same contexts, BASELINE config 4 (the only
configuration DEFINED as sharded over 8 GPUs: 262 144 keypoints, 10 M-pt map) sharded over the N ranks, and the same N
GPUs as independent replicas (config 5).  The driver's 8-GPU tier gets ONE run: the line it prints must carry all of this."""
import time

import numpy as np
import torch

import sr_livo_amd as srl
from sr_livo_amd import synth

from .stream import Streamer, _EskfAdapter, make_stream


def _timed_stream(run, streamer, steps, warm):
    streamer.begin()
    for _ in range(warm):
        streamer.step()
    run.barrier()
    arm0 = run.lio.ctx.arm_stats()
    its = 0
    t = time.perf_counter()
    for _ in range(steps):
        its += streamer.step()["iters"]
    arm1 = run.lio.ctx.arm_stats()
    run.barrier()
    el_rank = time.perf_counter() - t
    el = run.max_over_ranks(el_rank)
    return {"sweeps_per_s": steps / el, "us_per_esikf_iter": el * 1e6 / max(its, 1), "us_per_esikf_iter_this_rank": el_rank * 1e6 / max(its, 1),
            "esikf_iterations_per_solve": its / max(steps, 1), "arm_stats": {k: arm1[k] - arm0[k] for k in arm1}}


def other_transport_leg(run, steps):
    """the SAME stream through the transport the timed region did not use (both in one line).  The timed transport is
    re-attached afterwards; a failure here is reported, never fatal (the fallback chain of the launcher covers the timed one)."""
    timed = run.transport
    other = "peer" if timed != "peer" else "rccl"
    try:
        run.detach()
        run.attach(other)
        res = _timed_stream(run, run.streamer, max(steps, 50), 6)
        res["transport"] = other
        info = run.lio.ctx.comm_info()
        res["ranks_seen"] = info.get("ranks_seen")
    except Exception as e:  # noqa: BLE001
        res = {"transport": other, "error": repr(e)[:200]}
    try:
        run.detach()
        run.attach(timed)
    except Exception as e:  # noqa: BLE001
        res["reattach_error"] = repr(e)[:200]
    return res


def sharded_config_leg(run, transport, workload="C4", steps=20, stream_sweeps=2):
    """BASELINE config 4 sharded by point range over the N ranks of this run (32 768 keypoints per rank at N = 8), on a context of its own
    (the run's own context is closed by now: a second live context on the device would keep launches from being armed) with the run's
    transport: a stream of `stream_sweeps` distinct sweeps, sweeps/s of the whole job and us per ESIKF iteration."""
    n_kp, map_pts, pattern, seed = synth.CONFIGS[workload]
    cands, L = synth.map_candidates(seed, map_pts)
    sweep = synth.make_sweep(seed + 1000, n_kp, L, pattern=pattern)
    lio = srl.Lio(run.local_rank)
    streamer = None
    try:
        lio.add_points_to_map(cands)
        del cands
        run.lio = lio
        run.attach(transport)
        prior_state = synth.eskf_prior(_EskfAdapter(lio), sweep["q_pred"], sweep["t_pred"], sweep["vel"]).copy()
        prior_cov = lio.eskf_get_cov().copy()
        opts = srl.default_opts(max_num_residuals=2**31 - 1)
        streamer = Streamer(lio, make_stream(sweep, prior_state, seed + 1000, n_kp, L, pattern, stream_sweeps, None), opts, prior_cov, 100, n_kp)
        res = _timed_stream(run, streamer, steps, 4)
        shard = lio.ctx.sweep_shard()
        res.update(workload=f"{workload}: {n_kp} keypoints ({pattern}) sharded over {run.world} ranks ({shard[1]} on rank 0), {lio.map_size()}-pt map replicated, "
                            f"max_num_residuals=INT_MAX, stream of {streamer.S} sweeps", transport=transport, map_points=lio.map_size())
        return res
    except Exception as e:  # noqa: BLE001
        return {"workload": workload, "error": repr(e)[:200]}
    finally:
        try:
            lio.ctx.disarm(); torch.cuda.synchronize()
            run.detach()
        except Exception:  # noqa: BLE001
            pass
        if streamer is not None:
            streamer.close()
        lio.close()
        run.lio = None


def replicas_leg(run, steps):
    """the other way to use N GPUs (BASELINE config 5: one sweep per GPU, independent maps, no collective): the communicator is parked and
    every rank solves the WHOLE sweep; informational, never `value`"""
    lio = run.lio
    run.lio.ctx.disarm()
    if run.transport == "peer":
        run.detach()
    else:
        lio.ctx.comm_suspend(True)           # keep the communicator, run the whole sweep locally
    lio.resident_sweep(run.sweep["raw"])
    run.solve()
    run.barrier()
    n = max(steps, 100)
    t = time.perf_counter()
    for _ in range(n):
        run.solve()
    lio.ctx.disarm()
    torch.cuda.synchronize()
    rate = run.world * n / run.max_over_ranks(time.perf_counter() - t)
    return {"value": rate, "what": "the same N GPUs each solving its own whole 64k sweep (replicas, no collective; BASELINE config 5), measured after the timed region"}
