"""Legs of bench.py BEHIND the timed region, on the timed context: the kernel's duration under HIP events (the roofline's denominator),
the long stamped stream, A/B forms, PCIe forms, the unfiltered stream.  None of them contributes to `value`."""
import os
import time
import types

import numpy as np
import torch

import sr_livo_amd as srl
from sr_livo_amd import synth

from .launcher import c_stdout_to_stderr
from .profiles import HBM_PEAK_GBS
from .stream import Streamer, make_stream

EVENT_PERIOD = 5      # of every 5 association launches ONE is timed (two event records); odd: first and second iterations are sampled alike


def kernel_time_leg(run, min_launches=200):
    """The association kernel's average duration: the SAME stream loop as the timed region, run again right behind it with HIP events
    on the context's stream -- one launch in EVENT_PERIOD timed -- until >= min_launches durations exist.  (Inside the timed region the
    event records cost the loop ~5 us per solve: VERDICT r05 item 4 moved them here.)  An armed launch's event pair opens when the pass
    before it ends: its wait for the host's pose is inside the duration."""
    ctx, step = run.lio.ctx, run.streamer.step
    ctx.set_profiling(2)
    ctx.set_profiling_period(EVENT_PERIOD)
    for _ in range(8):
        step()                            # (first use of the event pairs; clocks back up behind the read-back above)
    ctx.timing_mark()
    solves, t0 = 0, time.perf_counter()
    tim = None
    # N > 1: every rank runs the SAME number of solves (each one is a collective) -- the count of durations read back can differ by a launch
    # between ranks (a cancelled armed launch voids its event pair), so it must not decide when a rank leaves the loop
    fixed_batches = None if run.dist is None else max(1, (min_launches * EVENT_PERIOD + 99) // 100)
    for b in range(40):
        for _ in range(50):
            step()
        solves += 50
        tim = ctx.timing()                # (reads the completed pairs back; cancels the waiting launch: once per 50 solves)
        if (fixed_batches is None and (tim.calls >= min_launches or solves >= 2000)) or (fixed_batches is not None and b + 1 >= fixed_batches):
            break
    el = time.perf_counter() - t0
    ctx.set_profiling(0)
    ctx.set_profiling_period(1)
    return types.SimpleNamespace(calls=tim.calls, sum_assoc_ms=tim.sum_assoc_ms, sum_algorithmic_bytes=tim.sum_algorithmic_bytes,
                                 sum_passes=tim.sum_passes, sum_keypoints=tim.sum_keypoints, solves=solves, elapsed_s=el)


def long_stream_leg(run, n_long):
    """the same loop for >= 1 000 solves with a stamp per solve (a 20-step region lasts 2 ms): mean- and median-based rates, and the states
    every sweep of the stream was solved to (compared with the oracle and with the launch-per-iteration form)"""
    ctx, step = run.lio.ctx, run.streamer.step
    per, states = np.empty(n_long), {}
    run.barrier()
    arm0 = ctx.arm_stats()
    t0 = time.perf_counter()
    for k in range(n_long):
        ta = time.perf_counter()
        rr = step()
        per[k] = time.perf_counter() - ta
        if rr["sweep"] not in states:
            states[rr["sweep"]] = (rr["iters"], rr["num_residuals"], rr["state"].copy())
    arm1 = ctx.arm_stats()
    run.barrier()
    el = run.max_over_ranks(time.perf_counter() - t0)
    return {"solves": n_long, "sweeps_per_s_mean": n_long / el, "sweeps_per_s_median": 1.0 / float(np.median(per)),
            "us_per_solve_p10_p50_p90_max": [float(np.percentile(per, q)) * 1e6 for q in (10, 50, 90, 100)],
            "arm_stats": {k: arm1[k] - arm0[k] for k in arm1}, "solves_over_1ms": int(np.count_nonzero(per > 1e-3))}, states


def unfiltered_stream_leg(run, sweeps=8, solves=400):
    """the timed stream keeps sweeps that take sweep 0's number of ESIKF iterations (selection on the outcome).  The same
    loop over the FIRST `sweeps` seeds, none skipped: sweeps/s and the iteration-weighted time per ESIKF iteration."""
    a = run.args
    st = make_stream(run.sweep, run.prior_state, run.sweep_seed, run.n_kp, run.L, run.pattern, sweeps, None, run.gen)
    sm = Streamer(run.lio, st, run.opts, run.prior_cov, a.frame_id, run.n_kp)
    try:
        sm.begin()
        for _ in range(2 * len(st)):
            sm.step()
        run.barrier()
        its, per_sweep = 0, {}
        t0 = time.perf_counter()
        for _ in range(solves):
            r = sm.step()
            its += r["iters"]
            per_sweep[r["sweep"]] = r["iters"]
        run.barrier()
        el = run.max_over_ranks(time.perf_counter() - t0)
        return {"sweeps": len(st), "solves": solves, "sweeps_per_s": solves / el, "us_per_esikf_iter": el * 1e6 / max(its, 1),
                "esikf_iterations_by_sweep": [per_sweep.get(j) for j in range(len(st))],
                "what": "the stream WITHOUT the iteration-count filter (seeds in order, none skipped): the unbiased rate"}
    finally:
        run.lio.ctx.disarm(); torch.cuda.synchronize()
        sm.close()


def launch_ab_leg(run, stream_states, armed_us_per_iter):
    """A/B: the same stream with one launch call per ESIKF iteration on the critical path (armed launches off: round 3's form), with the
    kernel's HIP-event duration in that form; every sweep of the stream must come out with the same bits either way"""
    ctx, step, S = run.lio.ctx, run.streamer.step, run.S
    ctx.set_armed_launch(False)
    for _ in range(3):
        step()
    ctx.set_profiling(2)
    run.barrier()
    t0 = time.perf_counter()
    un_states, un_iters = {}, 0
    for _ in range(max(run.args.steps, 2 * S, 100)):
        r = step()
        un_iters += r["iters"]
        un_states.setdefault(r["sweep"], r["state"].copy())
    run.barrier()
    el = time.perf_counter() - t0
    tim = ctx.timing()
    ctx.set_profiling(0)
    ctx.set_armed_launch(True)
    ab = {"armed_us_per_iter": armed_us_per_iter, "launch_per_iteration_us_per_iter": el * 1e6 / max(un_iters, 1),
          "state_bitwise_equal": bool(all(j in stream_states and np.array_equal(un_states[j], stream_states[j][2]) for j in un_states)),
          "sweeps_compared": len(un_states)}
    return ab, tim


def resident_legs(run):
    """sweep 0 resident in HBM: (a) re-solved back to back from the same prior (rounds 1-4 printed this as `value`; no node can do it),
    (b) the per-call host breakdown under full event profiling, (c) the sharded code path with ONE rank (1-rank RCCL communicator,
    collectives forced), (d) the association work alone (final reduction in its own kernel).  Returns (r0, dict)."""
    a, lio, solve = run.args, run.lio, run.solve
    lio.resident_sweep(run.sweep["raw"])
    for _ in range(3):
        r0 = solve()
    r0 = dict(r0, state=r0["state"].copy())       # sweep 0 solved from its prior: what the CPU legs and the parity figures refer to
    out = {}
    if not a.no_aux_legs:
        run.barrier()
        t = time.perf_counter()
        for _ in range(max(a.steps, 100)):
            rr = solve()
        run.barrier()
        el = run.max_over_ranks(time.perf_counter() - t) / max(a.steps, 100)
        out["resident_resolve"] = {"sweeps_per_s": 1.0 / el, "us_per_esikf_iter": el * 1e6 / max(rr["iters"], 1),
                                   "what": "sweep 0 re-solved back to back, resident in HBM (no upload, no swap): the first pass of every solve pays a launch"}
    lio.ctx.set_profiling(1)
    for _ in range(max(3, min(10, a.steps))):
        solve()
    out["tim_full"] = lio.ctx.timing()
    lio.ctx.set_profiling(0)
    if run.world == 1 and run.dist is None and not a.no_aux_legs:
        try:
            os.environ["SRL_FORCE_COLLECTIVES"] = "1"
            with c_stdout_to_stderr():
                lio.ctx.comm_init_rank(1, 0, srl.Context.comm_unique_id())
                lio.resident_sweep(run.sweep["raw"])
                for _ in range(3):
                    solve()
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(max(a.steps, 100)):
                rc_ = solve()
            torch.cuda.synchronize()
            el = (time.perf_counter() - t) / max(a.steps, 100)
            st = lio.ctx.arm_stats()
            out["sharded_path_one_rank"] = {"ms_per_esikf_iter": el * 1e3 / max(rc_["iters"], 1), "sweeps_per_s": 1.0 / el, "armed_launches_fired": st["fired"],
                                            "what": "1-rank RCCL communicator with the collectives forced (fused pass -> device mailbox -> ncclAllReduce of 50 doubles -> "
                                                    "publish kernel; the next pass's launch armed behind them)"}
        except Exception as e:  # noqa: BLE001
            out["sharded_path_one_rank"] = {"error": repr(e)}
        finally:
            os.environ.pop("SRL_FORCE_COLLECTIVES", None)
            lio.ctx.comm_destroy()
            lio.resident_sweep(run.sweep["raw"])
            solve()
    if run.world == 1 and not a.no_fused_reduce and not a.no_aux_legs:
        lio.ctx.set_fused_reduce(0)
        solve()
        lio.ctx.set_profiling(2)
        for _ in range(max(5, min(20, a.steps))):
            solve()
        out["tim_unfused"] = lio.ctx.timing()
        lio.ctx.set_profiling(0)
        lio.ctx.set_fused_reduce(1)
        solve()
    return r0, out


def pcie_legs(run, pipelined):
    """The other ways a sweep can reach the solve (`value` = the pipelined stream: the next sweep crosses PCIe on the copy stream while the
    current one is solved): upload and solve back to back on ONE stream, no overlap -- (a) from page-locked memory, (b) from pageable
    memory through the context's pinned ring.  >= 200 solves per leg."""
    a, lio, solve = run.args, run.lio, run.solve
    n = max(200, a.steps)
    rates = {"pinned": None, "pageable": None, "pipelined": pipelined["sweeps_per_s_mean"] if pipelined else None}
    medians, stalls = ({"pipelined": pipelined["sweeps_per_s_median"]} if pipelined else {}), {}
    mult = run.world if (run.world > 1 and not run.sharded) else 1
    for label, src in (() if a.no_aux_legs else (("pinned", run.stream[0]["pin"].array), ("pageable", run.sweep["raw"]))):
        lio.resident_sweep(run.stream[0]["pin"].array)
        for _ in range(4):
            lio.resident_sweep(src); solve()
        run.barrier()
        per = np.empty(n)
        t = time.perf_counter()
        for k in range(n):
            ta = time.perf_counter()
            lio.resident_sweep(src); solve()
            per[k] = time.perf_counter() - ta
        run.barrier()
        rates[label] = mult * n / run.max_over_ranks(time.perf_counter() - t)
        medians[label] = mult / float(np.median(per))
        stalls[label] = [{"step": int(k), "ms": round(float(per[k]) * 1e3, 2)} for k in np.nonzero(per > 1e-3)[0][:8]]
    if not a.no_aux_legs:
        lio.resident_sweep(run.sweep["raw"]); solve()
    lio.ctx.disarm()
    torch.cuda.synchronize()
    return rates, medians, stalls, n


def roofline_block(run, tim, iters_timed, nb, extra):
    """`roofline` of the bench line: achieved = algorithmic bytes per launch / average HIP-event duration of the launch"""
    from .profiles import compulsory_bytes, issue_roofline, load_profile, traffic_from_profile
    a = run.args
    calls, passes = max(tim.calls, 1), max(tim.sum_passes, 1)
    assoc_ms = tim.sum_assoc_ms / calls
    bytes_per_launch = tim.sum_algorithmic_bytes / calls
    achieved = bytes_per_launch / (assoc_ms * 1e-3) / 1e9 if assoc_ms > 0 else 0.0
    headline_default = a.workload == "HEADLINE" and run.world == 1 and a.frame_id >= 20 and a.max_num_residuals == 2**31 - 1
    traffic, traffic_src = traffic_from_profile() if headline_default else (None, None)
    armed_used = (run.lio.ctx.arm_stats()["fired"] > 0) and not a.no_armed
    roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
            "traffic_source": traffic_src, "kernel": f"srl_assoc_armed_kernel<{nb}>" if armed_used else f"srl_assoc_kernel<{nb}>",
            "avg_launch_ms": assoc_ms, "launches": tim.calls, "event_period": EVENT_PERIOD,
            "measured_over": f"{tim.calls} launches of {tim.solves} solves of the same stream loop, timed with HIP events on the context's stream right "
                             f"behind the K-step region (event records inside the region cost it ~5 us per solve)",
            "passes_per_launch": passes / calls, "avg_pass_ms": tim.sum_assoc_ms / passes,
            "algorithmic_bytes_per_launch": bytes_per_launch, "algorithmic_bytes_per_pass": tim.sum_algorithmic_bytes / passes,
            "profile_stale": bool(load_profile()[1]) if headline_default else None,
            "note": "achieved = ALGORITHMIC bytes (24 + 12 (2r+1)^3 + 12 P_k per keypoint, SURVEY 8(d)) / launch time: the rate at which the "
                    "reference's byte stream is consumed.  The working set is L2/MALL resident, so real HBM traffic (`traffic`, "
                    "`hbm_measured_GBs`) is far below it and the kernel is bound by instruction issue: see `issue`."}
    if armed_used:
        roof["launch_duration_includes"] = "the armed launch's wait for the host's pose (its event pair opens when the pass before it ends)"
    tf = extra.get("tim_full")
    if tf is not None:
        roof["reduce_kernel_avg_ms"] = tf.sum_reduce_ms / max(tf.calls, 1)
        roof["device_total_avg_ms"] = tf.sum_total_ms / max(tf.calls, 1)
    for key, t, what in (("unarmed", extra.get("tim_unarmed"), "the same kernel launched per iteration (armed launches off): duration without any wait inside"),
                         ("association_only", extra.get("tim_unfused"), "the same launches with the final reduction left to the separate reduce kernel "
                          "(srl_debug_set_fused_reduce(0)): the association kernel proper")):
        if t is not None and t.calls > 0:
            ms = t.sum_assoc_ms / t.calls
            roof[key] = {"avg_launch_ms": ms, "launches": t.calls, "what": what,
                         "frac": (t.sum_algorithmic_bytes / t.calls) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS if ms > 0 else None}
    if run.rank == 0 and run.world == 1:
        try:
            keys, counts, _ = run.lio.ctx.map_download()
            R = synth.quat_to_rot(run.sweep["q_pred"] / np.linalg.norm(run.sweep["q_pred"]))
            comp, s_u, p_u = compulsory_bytes(keys, counts, run.sweep["raw"] @ R.T + run.sweep["t_pred"], nb)
            roof["compulsory_bytes_per_launch"] = comp
            roof["compulsory"] = {"unique_slots_probed": s_u, "unique_points_touched": p_u,
                                  "what": "24 N + 12 S_unique + 12 P_unique at the first iteration's pose (SURVEY 8(d))"}
            if traffic:
                roof["hbm_measured_GBs"] = traffic / (assoc_ms * 1e-3) / 1e9
                roof["hbm_measured_frac_of_peak"] = roof["hbm_measured_GBs"] / HBM_PEAK_GBS
                roof["traffic_over_compulsory"] = traffic / comp
        except Exception as e:  # noqa: BLE001
            roof["compulsory_error"] = str(e)
        if headline_default:
            roof["issue"] = issue_roofline(assoc_ms)
    return roof
