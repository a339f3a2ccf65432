#!/usr/bin/env python
"""bench.py -- headline benchmark of the SR-LIVO LIO scan-matching hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N = 1)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" = one full lioOptimization::updateIEKF solve (all ESIKF iterations until convergence, each
iteration = one association/plane-fit/residual pass over the whole sweep + the ordered reduction +
the host 17-dim update; optimize.cpp:133-314) of ONE synthetic 64k-point Livox-like sweep against a
1M-point voxel map (SURVEY.md 8(d) "headline", BASELINE.json configs[1] scaled to the metric's 64k
sweep).  Map and sweep are resident in HBM before the timed region.  metric = sweeps/s.

N > 1: one process per GPU; the sweep is sharded by point range, the map is replicated, and the only
exchange step is the RCCL all-reduce of the 6x6 normal equations each iteration (strong scaling of
one sweep -- the design north_star names).  --mode replay instead runs N independent sweeps (config 5).

The JSON line also carries
  roofline     : the association kernel vs the HBM roofline (algorithmic bytes / HIP-event time)
  cpu_baseline : the CPU oracle (single thread, like the reference) timed on this box's host cores.
The oracle is used ONLY for that leg and for the parity figure printed next to it.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402  (device memory / streams / torch.distributed plumbing only)

import sr_livo_amd as srl  # noqa: E402
from sr_livo_amd import synth  # noqa: E402

HBM_PEAK_GBS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="HEADLINE", choices=sorted(synth.CONFIGS))
    ap.add_argument("--mode", default="sharded", choices=["sharded", "replay"])
    ap.add_argument("--max-num-residuals", type=int, default=2**31 - 1,
                    help="2^31-1 = throughput headline (every keypoint contributes); 600 = shipped yaml value")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--select-mode", type=int, default=0)
    ap.add_argument("--force-comm", action="store_true",
                    help="attach an RCCL communicator even at world size 1 (exercises the sharded code path on a 1-GPU box)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one process per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:
        import torch.distributed as dist_mod
        dist = dist_mod
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    n_kp, map_pts, pattern, seed = synth.CONFIGS[args.workload]
    sharded = ((world > 1 or (args.force_comm and "RANK" in os.environ)) and args.mode == "sharded")
    sweep_seed = seed + 1000 + (rank if (world > 1 and not sharded) else 0)
    map_seed = seed + (rank if (world > 1 and not sharded) else 0)

    # ---------------- inputs: map built by the product's device-side addPointsToMap, sweep pinned in HBM
    t0 = time.time()
    cands, L = synth.map_candidates(map_seed, map_pts)
    sweep = synth.make_sweep(sweep_seed, n_kp, L, pattern=pattern)
    lio = srl.Lio(local_rank)
    lio.add_points_to_map(cands)
    n_map = lio.map_size()
    if sharded or (args.force_comm and dist is not None):
        uid = [srl.Context.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        lio.ctx.comm_init_rank(world, rank, uid[0])
    prior_state = synth.eskf_prior(_EskfAdapter(lio), sweep["q_pred"], sweep["t_pred"], sweep["vel"]).copy()
    prior_cov = lio.eskf_get_cov().copy()
    state0 = np.concatenate([sweep["q_pred"], sweep["t_pred"], sweep["vel"], np.zeros(6)])
    opts = srl.default_opts(max_num_residuals=args.max_num_residuals, select_mode=args.select_mode)
    lio.resident_sweep(sweep["raw"])
    setup_s = time.time() - t0

    # one step = eskf_set_state + eskf_set_cov (reset the prior) + update_iekf on the resident sweep, through a closure
    # that converts its arguments once (the per-call numpy/ctypes marshalling of the generic wrappers costs ~10 us)
    _solve = lio.bound_solver(opts, prior_state, prior_cov, state0, sweep["t_last"], 100, n_kp)

    def solve():
        rc, it, nr = _solve()
        if rc:
            raise SystemExit(f"update_iekf failed with status {rc}")
        return {"iters": it, "num_residuals": nr, "state": _solve.state}

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        r = solve()
    # HIP events on the context's own stream, inside the timed region: one pair around every association launch, read
    # back lazily after the region (mode 2) -- the full per-call breakdown (mode 1: four events + a sync per call, ~20 us
    # of host time per iteration) is taken on a few extra solves after the timed region instead.
    lio.ctx.set_profiling(2)
    barrier()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        r = solve()
    barrier()
    elapsed = time.perf_counter() - t1
    tim = lio.ctx.timing()
    lio.ctx.set_profiling(1)
    for _ in range(max(3, min(10, args.steps))):
        solve()
    tim_full = lio.ctx.timing()
    lio.ctx.set_profiling(0)
    if dist is not None:
        te = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())

    # PCIe-inclusive rate (never `value`): the sweep crosses the host boundary on every solve
    # (srl_sweep_upload: 24 B/keypoint H2D + SoA transpose), the map stays resident.
    n_pcie = max(3, min(10, args.steps))
    barrier()
    t2 = time.perf_counter()
    for _ in range(n_pcie):
        lio.resident_sweep(sweep["raw"])
        solve()
    barrier()
    pcie_elapsed = time.perf_counter() - t2

    # N > 1, sharded: also report the other way to use N GPUs (BASELINE config 5: one sweep per GPU, no collective),
    # measured after the timed region on the same contexts; informational, never `value`.
    replicas_rate = None
    if sharded:
        r_sharded = r
        lio.ctx.comm_suspend(True)           # keep the communicator, run the whole sweep locally
        lio.resident_sweep(sweep["raw"])
        solve()
        barrier()
        t3 = time.perf_counter()
        for _ in range(args.steps):
            solve()
        torch.cuda.synchronize()
        te = torch.tensor([time.perf_counter() - t3], device="cuda", dtype=torch.float64)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        replicas_rate = world * args.steps / float(te.item())
        r = r_sharded

    iters = r["iters"]
    sweeps_per_step = world if (world > 1 and not sharded) else 1
    value = sweeps_per_step * args.steps / elapsed
    ms_per_step = elapsed / args.steps * 1e3

    calls = max(tim.calls, 1)
    fcalls = max(tim_full.calls, 1)
    assoc_ms = tim.sum_assoc_ms / calls
    bytes_per_launch = tim.sum_algorithmic_bytes / calls
    achieved = bytes_per_launch / (assoc_ms * 1e-3) / 1e9 if assoc_ms > 0 else 0.0

    # HBM traffic per launch from the committed rocprofv3 PMC passes of this same command (bench.py cannot
    # profile itself): (2 x FETCH_SIZE + WRITE_SIZE) KB, the x2 being the gfx950 FETCH_SIZE correction for wide
    # coalesced reads (MI355X_MICROARCH.md, HBM section).  null when no matching profile is committed.
    traffic, traffic_src = None, None
    try:
        prof_path = os.path.join(ROOT, "profiles", "r01_final_rocprofv3_summary.json")
        pm = json.load(open(prof_path))["pmc_per_dispatch"]
        k = [v for n, v in pm.items() if "assoc" in n][0]
        if args.workload == "HEADLINE" and world == 1:
            traffic = (2.0 * k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024.0
            traffic_src = "profiles/r01_final_rocprofv3_summary.json (separate --pmc passes of this command)"
    except Exception:
        pass

    out = {
        "metric": "sweeps/s (full ESIKF solve of a 64k-pt Livox sweep vs 1M-pt voxel map)",
        "value": value, "unit": "sweeps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "strong" if sharded or world == 1 else "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{args.workload}: {n_kp}-keypoint {pattern} sweep, {n_map}-pt voxel map "
                               f"({map_pts} target), max_num_residuals={args.max_num_residuals}, "
                               f"r=1, K=20; inputs resident in HBM",
                   "parallelism": ("point-range shards x%d + RCCL all-reduce of 6x6 normal equations" % world) if sharded
                                  else ("replicas x%d" % world if world > 1 else "single GPU"),
                   "esikf_iterations_per_solve": iters, "residuals_used": r["num_residuals"]},
        "ms_per_esikf_iter": ms_per_step / max(iters, 1),
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                     "kernel": "srl_assoc_kernel<1>", "avg_launch_ms": assoc_ms, "launches": tim.calls,
                     "algorithmic_bytes_per_launch": bytes_per_launch,
                     "reduce_kernel_avg_ms": tim_full.sum_reduce_ms / fcalls, "device_total_avg_ms": tim_full.sum_total_ms / fcalls},
        "host_us_per_iter": {"enqueue": tim_full.sum_host_launch_us / fcalls, "wait_results": tim_full.sum_host_wait_us / fcalls,
                             "build_residuals_call": tim_full.sum_host_total_us / fcalls,
                             "whole_iteration": ms_per_step * 1e3 / max(iters, 1),
                             "note": "first three: extra solves after the timed region with full event profiling (adds ~20 us/iter); "
                                     "whole_iteration: the timed region"},
        "pcie_inclusive_sweeps_per_s": (world if (world > 1 and not sharded) else 1) * n_pcie / pcie_elapsed,
        "setup_s": setup_s,
    }
    if replicas_rate is not None:
        out["aux_independent_sweeps_per_s"] = {"value": replicas_rate, "what": "the same N GPUs each solving its own 64k sweep "
                                               "(replicas, no collective; BASELINE config 5), measured after the timed region"}

    # ---------------- CPU baseline + parity figure (rank 0, N = 1 only)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import pyoracle as po
        backend = "tsl" if os.path.exists(po.LIB_TSL) else "plain"
        omap = po.Map(backend)
        omap.add_points(cands)
        oo = po.opts_from_product(opts)
        times = []
        ou = None
        budget_s, t_start = 25.0, time.perf_counter()
        while len(times) < 5 and (time.perf_counter() - t_start) < budget_s:
            eo = po.Eskf(backend)
            eo.set_state(prior_state)
            eo.set_cov(prior_cov)
            tc = time.perf_counter()
            ou = po.update_iekf(omap, eo, oo, sweep["raw"], state0, sweep["t_last"], frame_id=100)
            times.append(time.perf_counter() - tc)
        cpu_s = float(np.median(times))
        state_err = float(np.max(np.abs(r["state"] - ou["state"])) / np.max(np.abs(ou["state"])))
        out["cpu_baseline"] = {"value": 1.0 / cpu_s, "unit": "sweeps/s", "cores": 1, "kind": "port",
                               "sample": f"{len(times)} full solves ({ou['rc']} ESIKF iterations each) of the same sweep and map; "
                                         f"oracle restatement of optimize.cpp, single thread like the reference, "
                                         f"voxel map = {backend}; host has {os.cpu_count()} cores",
                               "ms_per_solve": cpu_s * 1e3, "ms_per_esikf_iter": cpu_s * 1e3 / max(ou["rc"], 1)}
        out["parity"] = {"state_rel_err_vs_oracle": state_err, "iterations_gpu": iters, "iterations_oracle": ou["rc"],
                         "residuals_gpu": r["num_residuals"], "residuals_oracle": ou["num_residuals"]}
    if rank == 0:
        print(json.dumps(out))
    lio.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
