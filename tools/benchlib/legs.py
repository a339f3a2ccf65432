"""Legs of bench.py that run BEHIND the timed region on one GPU: every BASELINE configuration, the frame pipeline."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

import sr_livo_amd as srl
from sr_livo_amd import synth

from .profiles import HBM_PEAK_GBS, INT_MAX, profile_entry  # noqa: F401
from .stream import Streamer, _EskfAdapter, make_stream, rel

def oracle_solve(po, backend, lio, opts, sweep, prior_state, prior_cov, state0, frame_id, threads):
    """the oracle's updateIEKF on the map the device holds (imported voxel by voxel: device-side insertion is tested
    bit-identical to the sequential addPointsToMap)"""
    omap = po.Map(backend)
    omap.import_(*lio.ctx.map_download())
    eo = po.Eskf(backend)
    eo.set_state(prior_state); eo.set_cov(prior_cov)
    with po.threads(threads):
        u = po.update_iekf(omap, eo, po.opts_from_product(opts), sweep["raw"], state0, sweep["t_last"], frame_id=frame_id)
    return u, omap


CONFIG_CLOCK_WARMUP_S = 0.05


def run_config(name, workload, max_res, frame_id, steps, warmup, device, po, backend, threads, stream_sweeps=4, spread=False):
    """one BASELINE configuration on this GPU, measured like the headline: a stream of distinct sweeps (prefetch -> solve -> swap, every
    sweep crossing PCIe once per solve); rate, per-iteration time, association-kernel time and roofline fraction, parity of the solved
    state of sweep 0 against the oracle"""
    n_kp, map_pts, pattern, seed = synth.CONFIGS[workload]
    cands, L = synth.map_candidates(seed, map_pts)
    # spread: the OFF-CACHE sweep -- keypoints area-uniform over the whole scene in random order instead of a lidar cone
    gen = (lambda sd: synth.make_spread_sweep(sd, n_kp, cands, L)) if spread else None
    sweep = gen(seed + 1000) if spread else synth.make_sweep(seed + 1000, n_kp, L, pattern=pattern)
    if spread:
        pattern = "spread (area-uniform over the scene, random order)"
    lio = srl.Lio(device)
    streamer = None
    try:
        lio.add_points_to_map(cands)
        if not spread:
            del cands
        prior_state = synth.eskf_prior(_EskfAdapter(lio), sweep["q_pred"], sweep["t_pred"], sweep["vel"]).copy()
        prior_cov = lio.eskf_get_cov().copy()
        state0 = np.concatenate([sweep["q_pred"], sweep["t_pred"], sweep["vel"], np.zeros(6)])
        opts = srl.default_opts(max_num_residuals=max_res)
        def iterations_of(e):
            lio.resident_sweep(e["sweep"]["raw"])
            rc_, it_, _ = lio.bound_solver(opts, e["prior_state"], prior_cov, e["state0"], e["sweep"]["t_last"], frame_id, n_kp)()
            return it_ if rc_ == 0 else -1

        streamer = Streamer(lio, make_stream(sweep, prior_state, seed + 1000, n_kp, L, pattern, stream_sweeps, iterations_of, gen), opts, prior_cov, frame_id, n_kp)
        step = streamer.step
        streamer.begin()
        for _ in range(warmup):
            step()
        # ... and the same time-based clock warm-up as the headline leg (a timed region of a few milliseconds straight after an idle
        # phase ran on ramping clocks: the A/B leg behind it measured 5 % faster on identical code)
        t_w = time.perf_counter()
        while time.perf_counter() - t_w < CONFIG_CLOCK_WARMUP_S:
            step()
        # timed region: no events on the stream (the light profiling's event pair costs ~1.5 us per launch); per-solve stamps
        # on the host besides the total, so that one scheduling hiccup in a region of a few milliseconds shows as what it is.
        # NO device synchronisation in front of it (unlike the headline's timed region, whose contract demands one): every step returns
        # with its results, so the loop is at a step boundary anyway -- and the idle gap of a synchronisation costs the short solves of the
        # small configurations a clock transient of a dozen steps (measured: C2@600 56 instead of 51 us per solve over 200 steps)
        arm0 = lio.ctx.arm_stats()
        per = np.empty(steps)
        its = 0
        states = {}
        t = time.perf_counter()
        for k in range(steps):
            tk = time.perf_counter()
            rr = step()
            per[k] = time.perf_counter() - tk
            its += rr["iters"]
            if rr["sweep"] not in states:
                states[rr["sweep"]] = (rr["iters"], rr["num_residuals"], rr["state"].copy())
        el = time.perf_counter() - t
        arm1 = lio.ctx.arm_stats()
        lio.ctx.disarm()
        torch.cuda.synchronize()
        it, nr, state = states[0] if 0 in states else (rr["iters"], rr["num_residuals"], rr["state"].copy())
        # kernel time of the same solves: a second pass with one event pair around every association launch
        lio.ctx.set_profiling(2)     # (first use on this context: a thousand event creations, milliseconds of idle GPU ...)
        t_w = time.perf_counter()
        while time.perf_counter() - t_w < 0.01:
            step()                   # (... so the clocks are brought back up before the launches that count)
        lio.ctx.timing_mark()
        for _ in range(min(steps, 20)):
            step()
        tim = lio.ctx.timing()
        lio.ctx.set_profiling(0)
        calls = max(tim.calls, 1)
        assoc_ms = tim.sum_assoc_ms / calls
        bytes_per_launch = tim.sum_algorithmic_bytes / calls
        passes_per_launch = max(tim.sum_passes, 1) / calls
        launches_per_solve = lio.last_solve_launches()
        # A/B: one launch per ESIKF iteration (armed launches off: round 3's form), same stream
        lio.ctx.set_armed_launch(False)
        step(); step()
        torch.cuda.synchronize()
        its_un, un_equal = 0, True
        t_un = time.perf_counter()
        for _ in range(steps):
            ru = step()
            its_un += ru["iters"]
            if ru["sweep"] in states:
                un_equal = un_equal and bool(np.array_equal(ru["state"], states[ru["sweep"]][2]))
        torch.cuda.synchronize()
        el_un = time.perf_counter() - t_un
        lio.ctx.set_armed_launch(True)
        # the association work alone (final reduction in its own kernel, launch shape chosen for the kernel's own time), sweep 0 resident
        lio.resident_sweep(sweep["raw"])
        solve = lio.bound_solver(opts, prior_state, prior_cov, state0, sweep["t_last"], frame_id, n_kp)
        solve()
        # sweep 0 re-solved in HBM (rounds 1-4 measured the configurations this way)
        lio.ctx.disarm(); torch.cuda.synchronize()
        t_r = time.perf_counter()
        for _ in range(steps):
            rc_r, it_r, _nr = solve()
        lio.ctx.disarm(); torch.cuda.synchronize()
        el_r = time.perf_counter() - t_r
        # ... and with a launch armed behind EVERY pass (srl_set_armed_launch(2)): the loop rounds 1-4 quoted, in which the launch armed by the
        # last pass of a solve is fired by the first pass of the next solve of the SAME sweep -- kept for comparison with those rounds only
        lio.ctx.set_armed_launch(2)
        solve(); solve()
        t_r2 = time.perf_counter()
        for _ in range(steps):
            solve()
        el_r2 = time.perf_counter() - t_r2
        lio.ctx.disarm(); torch.cuda.synchronize()
        lio.ctx.set_armed_launch(True)
        lio.ctx.set_fused_reduce(0)
        solve()
        lio.ctx.set_profiling(2)
        for _ in range(min(max(3, steps // 2), 20)):
            solve()
        tu = lio.ctx.timing()
        lio.ctx.set_profiling(0)
        lio.ctx.set_fused_reduce(1)
        ms_u = tu.sum_assoc_ms / max(tu.calls, 1)
        arm = {k: arm1[k] - arm0[k] for k in arm1}
        ent = {"name": name, "workload": f"{workload}: stream of {streamer.S} distinct sweeps of {n_kp} keypoints ({pattern}), {lio.map_size()}-pt map, max_num_residuals={max_res}, frame_id={frame_id}"
                                         f" (r={2 if frame_id < 20 else 1}); every sweep crosses PCIe once per solve",
               "sweeps_per_s": steps / el, "ms_per_solve": el / steps * 1e3, "esikf_iterations": it, "ms_per_esikf_iter": el * 1e3 / max(its, 1),
               "steps": steps, "ms_per_solve_median": float(np.median(per)) * 1e3, "ms_per_solve_max": float(per.max()) * 1e3,
               "residuals_used": nr, "kernel_launches_per_solve": launches_per_solve,
               "arm_stats": arm, "armed": bool(arm["fired"] > 0), "stream_sweeps": streamer.S, "stream_seeds_skipped": streamer.stream[0].get("skipped"),
               "launch_per_iteration_ab": {"ms_per_esikf_iter": el_un * 1e3 / max(its_un, 1), "state_bitwise_equal": un_equal,
                                           "what": "armed launches off (srl_set_armed_launch(0)), same stream"},
               "resident_resolve_us_per_iter": el_r / steps * 1e6 / max(it_r, 1),
               "resident_resolve_always_armed_us_per_iter": el_r2 / steps * 1e6 / max(it_r, 1),
               "kernel_us": assoc_ms * 1e3, "passes_per_launch": passes_per_launch, "kernel_us_per_pass": assoc_ms * 1e3 / passes_per_launch,
               "assoc_kernel_us": assoc_ms * 1e3 / passes_per_launch, "assoc_launches": tim.calls,
               "keypoints_per_launch": tim.sum_keypoints / calls, "algorithmic_MB_per_launch": bytes_per_launch / 1e6,
               "hbm_roofline_frac": bytes_per_launch / (assoc_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if assoc_ms > 0 else None,
               "association_only_us": ms_u * 1e3,
               "association_only_hbm_roofline_frac": (tu.sum_algorithmic_bytes / max(tu.calls, 1)) / (ms_u * 1e-3) / 1e9 / HBM_PEAK_GBS if ms_u > 0 else None,
               "note": "kernel_us = HIP-event duration of the association kernel with the fused final reduction (an armed launch's event pair opens "
                       "when the pass before it ends: its wait for the host's pose is inside); association_only_* = one pass with the "
                       "reduction in its own kernel, launched per iteration"}
        ent["profile"] = profile_entry(name, assoc_ms / passes_per_launch)
        if po is not None:
            u, _ = oracle_solve(po, backend, lio, opts, sweep, prior_state, prior_cov, state0, frame_id, threads)
            ent["parity"] = {"state_rel_err_vs_oracle": rel(state, u["state"]), "iterations_oracle": int(u["rc"]),
                             "residuals_oracle": int(u["num_residuals"]), "ok": bool(u["rc"] == it and u["num_residuals"] == nr and rel(state, u["state"]) < 1e-5)}
        return ent
    finally:
        if streamer is not None:
            try:
                lio.ctx.disarm()
                torch.cuda.synchronize()
            except Exception:  # noqa: BLE001
                pass
            streamer.close()
        lio.close()


def run_pipeline(device, frame_points=(24_000, 65_536, 262_144), reps=9):
    """The frame-resident pipeline either side of the solve: per frame upload of the raw points (page-locked) ->
    keypoint selection on the device in gridSampling order (1.5 m sampling) -> two ESIKF passes on the selected keypoints -> commit
    (re-transform + addPointsToMap on the device, world points downloaded; the insertion itself is only enqueued -- num_added = NULL -- and
    the next frame's passes are ordered behind it on the stream, so a frame's time contains the previous frame's insertion wherever the
    device is the bottleneck), on a 1 M-point map, for frames spread over the scene.
    Wall time per stage (median), frames/s of the whole chain, and the synchronised stage breakdown of srl_debug_frame_timing."""
    from sr_livo_amd import capi
    cands, L = synth.map_candidates(7, 1_000_000)
    lio = srl.Lio(device)
    out = []
    try:
        lio.add_points_to_map(cands)
        ctx = lio.ctx
        q, t = np.array([1.0, 0, 0, 0]), np.zeros(3)
        f = capi.make_frame(q, t, t)
        opts = srl.default_opts(max_num_residuals=INT_MAX)
        for n_frame in frame_points:
            rng = np.random.default_rng(3 + n_frame)
            frame = cands[rng.choice(len(cands), n_frame, replace=False)] + rng.normal(0, 0.03, (n_frame, 3))
            pin = srl.PinnedArray(frame.shape)
            pin.array[:] = frame
            pin_world = srl.PinnedArray(frame.shape)          # point3D::point comes back into page-locked memory (as in integration/optimize_hip.cpp)

            def one(timing):
                ctx.frame_timing(timing)
                t0 = time.perf_counter()
                ctx.frame_upload(pin.array)
                t1 = time.perf_counter()
                k = ctx.frame_select_keypoints(q, t, 1.5, want_index=False)      # like the host mirror: the selection stays on the device
                t2 = time.perf_counter()
                ctx.build_residuals(f, opts)
                ctx.build_residuals(f, opts)
                ctx.solve_end()                                   # (like the host mirror: the arming policy learns that a solve is two passes -> no launch left waiting)
                t3 = time.perf_counter()
                ctx.frame_commit(q, t, want_world=True, want_added=False, world_out=pin_world.array)      # addPointsToMap returns nothing either
                t4 = time.perf_counter()
                return int(k), (t1 - t0, t2 - t1, t3 - t2, t4 - t3), ctx.frame_timing(False)

            one(False); one(True)
            t_loop = time.perf_counter()
            plain = np.array([one(False)[1] for _ in range(reps)]) * 1e6
            ctx.map_size()                                    # the last (deferred) insertion belongs to the loop
            loop_us = (time.perf_counter() - t_loop) * 1e6 / reps
            staged = [one(True) for _ in range(5)]
            med = np.median(plain, axis=0)
            out.append({"frame_points": n_frame, "keypoints": staged[0][0], "map_points": lio.map_size(), "frames_per_s": 1e6 / loop_us, "loop_us_per_frame": loop_us,
                        "us": {"upload": float(med[0]), "select": float(med[1]), "two_passes": float(med[2]), "commit": float(med[3]), "total": float(med.sum())},
                        "stage_us_synchronised": {k: float(np.median([s_[2][k] for s_ in staged])) for k in staged[0][2]}})
            ctx.map_size()                                    # (settles the last deferred insertion before the buffers go)
            pin.close(); pin_world.close()
    finally:
        lio.close()
    return out


def all_configs(device, po):
    """every BASELINE configuration + the shipped setting + init mode + the off-cache sweep, then the frame pipeline (rank 0, N = 1 only);
    -> the blocks `configs`, `pipeline`, `pipeline_detail` of the bench result"""
    backend, threads = None, 1
    if po is not None:
        backend = "tsl" if os.path.exists(po.LIB_TSL) else "plain"
        threads = min(os.cpu_count() or 1, 64)
    # small configurations get enough solves that the timed region spans tens of milliseconds.  SURVEY 8(d): every configuration twice --
    # max_num_residuals = INT_MAX (throughput) and = 600 (config/r3live.yaml:69, the shipped value: ordered cut); C1 is the plumbing scale,
    # C4 the 8-GPU configuration on one GPU; SPREAD = 65 536 keypoints one-per-voxel-ish over the 10 M-pt map (never a lidar pattern: what
    # the kernel does when its working set is NOT cache resident)
    plan = [("C1", "C1", INT_MAX, 100, 200, False), ("C2", "C2", INT_MAX, 100, 200, False), ("C3", "C3", INT_MAX, 100, 200, False),
            ("C4", "C4", INT_MAX, 100, 20, False), ("HEADLINE@600", "HEADLINE", 600, 100, 200, False), ("C2@600", "C2", 600, 100, 200, False),
            ("C3@600", "C3", 600, 100, 200, False), ("INIT(frame_id=5)", "HEADLINE", INT_MAX, 5, 20, False), ("SPREAD", "SPREAD", INT_MAX, 100, 40, True)]
    out = {"configs": []}
    for name, wl, mr, fid, st, spread in plan:
        try:
            out["configs"].append(run_config(name, wl, mr, fid, st, 2, device, po, backend, threads, stream_sweeps=2 if st <= 40 else 4, spread=spread))
        except Exception as e:  # noqa: BLE001
            out["configs"].append({"name": name, "error": repr(e)})
    try:
        pl = run_pipeline(device)
        out["pipeline_detail"] = pl
        out["pipeline"] = {"what": "frames/s (wall time of back-to-back frames) of upload + device keypoint selection + two passes + device commit, 1M-pt map; "
                                   "us = median host time per stage (the map insertion is enqueued by commit and runs on under the next frame's upload/select)",
                           "frames": [{"points": e["frame_points"], "keypoints": e["keypoints"], "frames_per_s": e["frames_per_s"], "us": e["us"]} for e in pl]}
    except Exception as e:  # noqa: BLE001
        out["pipeline"] = {"error": repr(e)[:160]}
    return out
