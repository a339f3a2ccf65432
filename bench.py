#!/usr/bin/env python
"""bench.py -- headline benchmark of the SR-LIVO LIO scan-matching hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N = 1, or N > 1: becomes the launcher of its own N ranks)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" = one full lioOptimization::updateIEKF solve (all ESIKF iterations until convergence, each iteration = one association /
plane-fit / residual pass over the whole sweep + the ordered reduction + the host's 17-dim update; optimize.cpp:133-314) of ONE sweep of a
STREAM of distinct synthetic 64k-point Livox-like sweeps against a 1M-point voxel map (SURVEY.md 8(d) "headline").  Every sweep crosses
PCIe once: it is uploaded on the copy stream during the solve before it and is resident in HBM when its own solve starts.
metric = sweeps/s.  The timed region below contains NOTHING but that loop -- no event record, no stamp beyond one clock read per step.

N > 1: one process per GPU; the sweep is sharded by point range, the map is replicated, and the only exchange step is the sum of the 6x6
normal equations each iteration (RCCL all-reduce by default, `--transport peer`: direct stores over xGMI) -- strong scaling of one sweep,
the design north_star names.  `--mode replay` runs N independent sweeps instead (config 5).  torch.distributed is control plane only (gloo).

Everything else the line carries is measured BEHIND the timed region by tools/benchlib (the file the driver hashes holds the loop that
produces `value`; the legs live beside it):
  roofline      aux_legs.kernel_time_leg: the same loop again with HIP events on the context's stream (>= 200 launch durations) -> algorithmic
                bytes / duration vs the HBM roofline; counter-measured traffic and the instruction-issue roofline from profiles/r06_*
  stream        >= 1 000 solves with a stamp per solve; the unfiltered-seed, iteration-weighted rate
  configs       every BASELINE configuration (C1..C4), the shipped max_num_residuals = 600, init mode, the off-cache "spread" sweep
  cpu_baseline  the same solve through the reference's OWN src/optimize.cpp (oracle/_ref, kind "reference") or the oracle port -- the only
                use of oracle/ in this process, as checker and reported baseline
  N > 1         the other transport, BASELINE config 4 sharded over the N ranks, the N GPUs as replicas (tools/benchlib/multi.py)
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "tools")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402,F401  (device memory / streams / torch.distributed plumbing only)

from sr_livo_amd import synth  # noqa: E402
from benchlib import aux_legs, launcher, line as line_mod, multi  # noqa: E402
from benchlib.profiles import INT_MAX  # noqa: E402
from benchlib.setup import Run  # noqa: E402


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="HEADLINE", choices=sorted(synth.CONFIGS))
    ap.add_argument("--mode", default="sharded", choices=["sharded", "replay"])
    ap.add_argument("--max-num-residuals", type=int, default=INT_MAX, help="2^31-1 = throughput headline (every keypoint contributes); 600 = shipped yaml value")
    ap.add_argument("--frame-id", type=int, default=100, help="< 20: init mode (r = 2, >= 16 iterations)")
    ap.add_argument("--stream-sweeps", type=int, default=4, help="distinct sweeps (own seeds, poses, priors) the timed stream cycles through")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the per-configuration legs (C1..C4, @600, init mode, spread; N > 1: config 4 sharded)")
    ap.add_argument("--select-mode", type=int, default=0)
    ap.add_argument("--no-fused-reduce", action="store_true", help="A/B: always run the separate reduce kernel")
    ap.add_argument("--no-numa-pin", action="store_true", help="A/B: do not pin the process to the GPU-local NUMA node")
    ap.add_argument("--no-aux-legs", action="store_true", help="only the timed configuration runs on the GPU (profiling: nothing else in the trace)")
    ap.add_argument("--clock-warmup-ms", type=float, default=50.0, help="setup, before the W warm-up steps: solve for this long (steady clocks; 0 = off)")
    ap.add_argument("--no-armed", action="store_true", help="A/B, profiling: armed launches off -- every ESIKF iteration pays its launch (round 3's form)")
    ap.add_argument("--transport", choices=("rccl", "peer"), default="rccl",
                    help="sharded mode: how the 50-double rows of the ranks are summed.  rccl: ncclAllReduce on the library's own communicator; "
                         "peer: direct stores into the peers' inboxes over xGMI (srl_peer_attach, HIP IPC handles exchanged over gloo)")
    ap.add_argument("--sharded-config", default="C4", choices=sorted(synth.CONFIGS), help="N > 1: the configuration of the sharded-config leg")
    ap.add_argument("--no-other-transport", action="store_true", help="N > 1: do not run the stream on the transport the timed region did not use")
    ap.add_argument("--force-comm", action="store_true", help="attach an RCCL communicator even at world size 1")
    return ap.parse_args()


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launcher.self_launch(args.gpus))          # `python bench.py --gpus N`: this process becomes the launcher of N ranks
    run = Run(args)                       # ranks, device, map (device-side addPointsToMap), transport, the stream of sweeps in page-locked memory
    lio, world, rank, sharded = run.lio, run.world, run.rank, run.sharded
    step = run.streamer.step              # one step: upload of sweep k + 1 issued beside the first kernel -> full ESIKF solve of sweep k -> swap
    n_cw = run.clock_warmup()             # setup, untimed, disclosed in the line (`clock_warmup`)
    for _ in range(args.warmup):
        r = step()

    # ================================================== THE TIMED REGION ==================================================
    # K steps of the stream bracketed by barrier + device synchronisation; max over ranks.  Nothing else touches the stream.
    step_end = np.empty(args.steps)
    run.barrier()
    arm_before = lio.ctx.arm_stats()
    t1 = time.perf_counter()
    iters_timed = 0
    for k in range(args.steps):
        r = step()
        iters_timed += r["iters"]
        step_end[k] = time.perf_counter()
    arm_after = lio.ctx.arm_stats()       # (before the barrier's own disarm of the launch armed behind the last pass)
    run.barrier()
    elapsed = time.perf_counter() - t1
    # ======================================================================================================================
    elapsed_rank = elapsed
    elapsed = run.max_over_ranks(elapsed)
    sweeps_per_step = world if (world > 1 and not sharded) else 1
    value = sweeps_per_step * args.steps / elapsed
    ms_per_step = elapsed / args.steps * 1e3
    launches_per_solve = lio.last_solve_launches()
    arm_stats = {k: arm_after[k] - arm_before[k] for k in arm_after}
    comm_state = lio.ctx.comm_info()      # (read while the transport of the timed region is still attached)
    if comm_state.get("transport_used") == "peer":
        rep, failed = lio.ctx.peer_stats()
        comm_state.update(passes_repeated_for_a_late_rank=rep, session_failed=failed)

    # ---------------- behind the region: the kernel's duration (roofline), the long stamped stream, A/B and PCIe forms
    tim = aux_legs.kernel_time_leg(run, 8 if args.no_aux_legs else 200)        # (profiling runs keep the trace to the timed form: --no-aux-legs)
    stream_long, stream_states = (None, {}) if args.no_aux_legs else aux_legs.long_stream_leg(run, max(1000, args.steps))
    unfiltered = None if (args.no_aux_legs or sharded) else aux_legs.unfiltered_stream_leg(run)
    extra, launch_ab = {}, None
    if world == 1 and not args.no_aux_legs and not args.no_armed:
        launch_ab, extra["tim_unarmed"] = aux_legs.launch_ab_leg(run, stream_states, elapsed * 1e6 / max(iters_timed, 1))
    other_transport = multi.other_transport_leg(run, args.steps) if (sharded and world > 1 and not args.no_aux_legs and not args.no_other_transport) else None
    r0, res_legs = aux_legs.resident_legs(run)
    extra.update(res_legs)
    rates, medians, stalls, n_pcie = aux_legs.pcie_legs(run, stream_long)
    nb = 2 if args.frame_id < 20 else 1
    roof = aux_legs.roofline_block(run, tim, iters_timed, nb, extra)
    replicas = multi.replicas_leg(run, args.steps) if sharded else None

    S, stream = run.S, run.stream
    tf = extra["tim_full"]
    fcalls = max(tf.calls, 1)
    out = {
        "metric": "sweeps/s (full ESIKF solve of a 64k-pt Livox sweep vs 1M-pt voxel map)",
        "value": value, "unit": "sweeps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong" if sharded or world == 1 else "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{args.workload}: stream of {S} distinct {run.n_kp}-keypoint {run.pattern} sweeps (own poses and priors), {run.n_map}-pt voxel map "
                               f"({run.map_pts} target), max_num_residuals={args.max_num_residuals}, r={nb}, K=20; one full ESIKF solve per sweep, every sweep crosses "
                               f"PCIe once (uploaded on the copy stream during the solve before it: resident in HBM when its solve starts); sweeps drawn with seeds "
                               f"{stream[0]['seed']} + 100 j, keeping those that take sweep 0's {stream[0]['iterations']} ESIKF iterations (skipped seeds: "
                               f"{[x['seed'] for x in stream[0]['skipped']]}; the unfiltered rate: `stream.unfiltered`)",
                   "parallelism": ("point-range shards x%d + %s of the 6x6 normal equations" % (world, "direct peer exchange" if args.transport == "peer" else "RCCL all-reduce")) if sharded
                                  else ("replicas x%d" % world if world > 1 else "single GPU"),
                   "esikf_iterations_per_solve": iters_timed / max(args.steps, 1), "residuals_used": r0["num_residuals"], "kernel_launches_per_solve": launches_per_solve,
                   "launch_mode": "one launch per ESIKF iteration" if (args.no_armed or arm_stats["fired"] == 0)
                                  else "armed launches: the kernel of pass k+1 is enqueued while pass k runs and receives its pose through the pose box -- across "
                                       "srl_sweep_swap too (the launch armed behind a sweep's last pass is the next sweep's first pass)",
                   "value_is": "SURVEY 8(d)'s metric: sweeps/s of a stream of distinct sweeps, H2D of every sweep included (overlapped with the solve before it); "
                               "the K-step region holds no event record.  Upload-then-solve without overlap:",
                   "pcie_inclusive_sweeps_per_s": {"pipelined_prefetch": rates["pipelined"], "pinned_upload_then_solve": rates["pinned"],
                                                   "pageable_upload_then_solve": rates["pageable"]}},
        "ms_per_esikf_iter": elapsed * 1e3 / max(iters_timed, 1),
        "roofline": roof, "launch_ab": launch_ab,
        "per_step_us": [round(float(x), 1) for x in np.diff(np.concatenate([[t1], step_end])) * 1e6],
        "arm_stats_timed_region": arm_stats,
        "stream": dict(stream_long or {}, sweeps=S, unfiltered=unfiltered,
                       value_over_long_mean=(value / stream_long["sweeps_per_s_mean"]) if stream_long else None,
                       state_of_sweep0_equals_resident_solve=bool(0 in stream_states and np.array_equal(stream_states[0][2], r0["state"]))),
        "resident_resolve": extra.get("resident_resolve"), "sharded_path_one_rank": extra.get("sharded_path_one_rank"),
        "host_us_per_iter": {"enqueue": tf.sum_host_launch_us / fcalls, "wait_results": tf.sum_host_wait_us / fcalls, "build_residuals_call": tf.sum_host_total_us / fcalls,
                             "whole_iteration": ms_per_step * 1e3 / max(r0["iters"], 1),
                             "note": "first three: extra solves after the timed region with full event profiling (adds ~20 us/iter); whole_iteration: the timed region"},
        "host_placement": dict(run.pin_info, what="the process runs on the CPUs of the GPU's NUMA node (srl_thread_pin_to_gpu_numa)"),
        "pcie": {"from_pinned_host_memory_sweeps_per_s": rates["pinned"], "from_pageable_host_memory_sweeps_per_s": rates["pageable"],
                 "pipelined_prefetch_sweeps_per_s": rates["pipelined"], "median_based": medians, "solves_per_leg": n_pcie, "steps_over_1ms": stalls,
                 "bytes_h2d_per_sweep": int(run.sweep["raw"].nbytes)},
        "setup_s": run.setup_s, "clock_warmup": {"ms": args.clock_warmup_ms, "solves": n_cw},
    }
    if run.comm_info is not None or world > 1:
        # every --gpus N line explains what it ran on: the transport the library actually used, the ranks the communicator itself counts,
        # each rank's own per-iteration time, the time behind the association kernel, the OTHER transport on the same stream
        ci = dict(run.comm_info or {}, **comm_state)
        us_rank = elapsed_rank * 1e6 / max(iters_timed, 1)
        if run.dist is not None:
            per_rank = [None] * world
            run.dist.all_gather_object(per_rank, us_rank)
            ci["us_per_iter_per_rank"] = {"min": float(min(per_rank)), "max": float(max(per_rank))}
        ci["behind_association_kernel_us"] = tf.sum_reduce_ms / fcalls * 1e3
        ci[args.transport] = {"us_per_esikf_iter": elapsed * 1e6 / max(iters_timed, 1), "sweeps_per_s": value, "arm_stats": arm_stats, "timed_region": True}
        if other_transport is not None:
            ci[other_transport.pop("transport")] = other_transport
        out["comm"] = ci
    if world > 1:
        out["multi_gpu_note"] = "no multi-GPU scaling curve has been measured by the builder (gpurun boxes expose one GPU); this line is it"
    if replicas is not None:
        out["aux_independent_sweeps_per_s"] = replicas

    # ---------------- CPU baselines + parity (rank 0, N = 1 only): the ONLY use of oracle/ in this process, behind the timed region
    po = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from benchlib import cpu
        blocks, po = cpu.cpu_baselines_and_parity(run, r0, stream_states)
        out.update(blocks)
    run.close()

    # ---------------- every BASELINE configuration + shipped setting + init mode + the off-cache sweep + the frame pipeline (rank 0, N = 1);
    # N > 1: BASELINE config 4 -- the configuration DEFINED as sharded over 8 GPUs -- sharded over the N ranks (every rank takes part)
    if rank == 0 and world == 1 and not args.no_configs:
        from benchlib import legs
        out.update(legs.all_configs(run.local_rank, po))
    if sharded and world > 1 and not args.no_configs:
        out["sharded_config"] = multi.sharded_config_leg(run, args.transport, args.sharded_config)
    if rank == 0:
        line_mod.write_detail(out)
        print(line_mod.compact_line(out), flush=True)
    if run.dist is not None:
        with launcher.c_stdout_to_stderr():
            run.dist.destroy_process_group()


if __name__ == "__main__":
    main()
