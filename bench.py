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
torch.distributed is control plane only (gloo: barrier, unique-id broadcast, max of the elapsed times); the
data-path collective is the library's own communicator on the process's single RCCL instance.

The JSON line also carries
  roofline     : the association kernel vs the HBM roofline (algorithmic bytes / HIP-event time), the compulsory
                 HBM floor of the launch, the counter-measured traffic, and -- because the working set is cache resident
                 and the kernel is bound by instruction issue -- an instruction-issue roofline from the committed PMC pass
  configs      : every BASELINE configuration (C1..C4), the headline at the shipped max_num_residuals = 600 and the
                 init mode (frame_id < 20: r = 2, >= 16 iterations), each with kernel time, roofline fraction, rate, parity
  cpu_baseline : the same solve through the reference's OWN src/optimize.cpp (oracle/_ref/libref_path.so, prebuilt; kind
                 "reference") where that library is present, else the CPU oracle (kind "port"); the oracle legs are always
                 reported too: cpu_baseline_port (single thread like the reference), cpu_baseline_all_cores (OpenMP).
The oracle is used ONLY for those legs and for the parity figures printed next to the timings.
"""
import argparse
import gc
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

HBM_PEAK_GBS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)
CLOCK_HZ = 2.4e9          # max engine clock (same guide)
N_CU, N_SIMD = 256, 1024
INT_MAX = 2**31 - 1
PROFILE_ROUND = "r05"    # the committed rocprofv3 summaries the line may quote: profiles/<round>_<config>_rocprofv3_summary.json
KERNEL_SOURCES = ("sr_livo_amd/csrc/srl_kernels.hip", "sr_livo_amd/csrc/srl_device.h")


def kernel_source_sha():
    """sha256 over the kernel sources: tools/profile_gpu.sh stamps every profile summary with it, and a summary whose stamp
    differs from the tree's is stale -- its counters describe another kernel and are not carried into the bench line."""
    import hashlib
    h = hashlib.sha256()
    for rel_path in KERNEL_SOURCES:
        with open(os.path.join(ROOT, rel_path), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def profile_path(tag="headline"):
    return os.path.join(ROOT, "profiles", f"{PROFILE_ROUND}_{tag}_rocprofv3_summary.json")


def load_profile(tag="headline"):
    """(pmc counters of the dominant kernel, stale?) from the committed rocprofv3 summary of this command (tag: which configuration)"""
    try:
        prof = json.load(open(profile_path(tag)))
        # the association kernel of that run: the armed instantiation where the run used armed launches (the one with the most dispatches)
        disp = prof.get("pmc_dispatches", {})
        names = sorted((n for n in prof["pmc_per_dispatch"] if "assoc" in n), key=lambda n: -max(disp.get(n, {"": 0}).values()))
        return prof["pmc_per_dispatch"][names[0]], prof.get("kernel_source_sha256") != kernel_source_sha()
    except Exception:
        return None, True


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


def compulsory_bytes(keys, counts, world_pts, nb):
    """SURVEY 8(d) compulsory floor of one association launch: 24 N (raw points) + 12 S_unique (distinct hash slots probed)
    + 12 P_unique (distinct map points inside the probed voxels) -- what HBM would have to deliver if nothing were read twice."""
    k = np.trunc(world_pts).astype(np.int64)                      # voxel key by truncation (size_voxel_map = 1.0)
    r = np.arange(-nb, nb + 1)
    off = np.stack(np.meshgrid(r, r, r, indexing="ij"), -1).reshape(-1, 3)
    pk = lambda a: (a[..., 0] + 32768) | ((a[..., 1] + 32768) << 16) | ((a[..., 2] + 32768) << 32)   # noqa: E731
    probed = np.unique(pk(k[:, None, :] + off[None, :, :]).ravel())
    mk = pk(keys.astype(np.int64))
    order = np.argsort(mk)
    pos = np.searchsorted(mk[order], probed)
    pos[pos >= len(mk)] = 0
    hit = mk[order][pos] == probed
    p_unique = int(counts[order][pos][hit].sum())
    return 24 * len(world_pts) + 12 * len(probed) + 12 * p_unique, int(len(probed)), p_unique


def issue_roofline(assoc_ms, tag="headline"):
    """Instruction-issue roofline of the association kernel from the committed PMC pass of this command (bench.py cannot
    count its own instructions).  The working set is cache resident and the kernel is bound by VALU issue, so this -- not
    the HBM figure -- says how close the kernel runs to the machine.  SQ_ACTIVE_INST_VALU counts the quad-cycles (4 shader
    cycles) the SIMDs spent issuing VALU work: 1.01 per VALU instruction in this kernel, i.e. one wave64 VALU instruction
    occupies its SIMD for 4 cycles.  floor = busy cycles / (SIMDs x clock): the time the same instruction stream would take
    with every SIMD issuing VALU back to back; frac = floor / measured launch time."""
    k, stale = load_profile(tag)
    if k is None or stale:
        return None
    valu, salu, lds = k.get("SQ_INSTS_VALU"), k.get("SQ_INSTS_SALU"), k.get("SQ_INSTS_LDS")
    if not valu:
        return None
    f64 = sum(k.get(c, 0.0) for c in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64"))
    busy_quads = k.get("SQ_ACTIVE_INST_VALU") or valu
    t_valu = busy_quads * 4.0 / (N_SIMD * CLOCK_HZ) * 1e6
    t_salu = (salu or 0.0) * 4.0 / (N_SIMD * CLOCK_HZ) * 1e6      # one scalar issue per SIMD per quad-cycle
    t_lds = (k.get("SQ_ACTIVE_INST_LDS") or 0.0) * 4.0 / (N_SIMD * CLOCK_HZ) * 1e6
    floor_us = max(t_valu, t_salu, t_lds)
    waves = k.get("SQ_WAVES") or 1.0
    return {"bound": "valu-issue", "valu_insts": valu, "valu_f64_insts": f64 or None, "salu_insts": salu, "lds_insts": lds,
            "vmem_rd_insts": k.get("SQ_INSTS_VMEM_RD"), "valu_busy_quad_cycles": busy_quads,
            "valu_floor_us": t_valu, "salu_floor_us": t_salu, "lds_floor_us": t_lds, "floor_us": floor_us,
            "achieved_us": assoc_ms * 1e3, "frac": floor_us / (assoc_ms * 1e3) if assoc_ms > 0 else None,
            "valu_busy_share_of_wave_lifetime_x_waves_per_simd": busy_quads / max(k.get("SQ_WAVE_CYCLES") or 1.0, 1.0) * (waves / N_SIMD),
            "wait_inst_any_share": (k.get("SQ_WAIT_INST_ANY") or 0.0) / max(k.get("SQ_WAVE_CYCLES") or 1.0, 1.0),
            "lds_bank_conflict_cycles": k.get("SQ_LDS_BANK_CONFLICT"), "source": os.path.relpath(profile_path(tag), ROOT),
            "model": "floor = SQ_ACTIVE_INST_VALU quad-cycles x 4 / (1024 SIMDs x 2.4 GHz); measured time = the live HIP-event average"}


def traffic_from_profile(tag="headline"):
    k, stale = load_profile(tag)
    if k is None or stale or "FETCH_SIZE" not in k:
        return None, None
    # (2 x FETCH_SIZE + WRITE_SIZE) KB: x2 = the gfx950 FETCH_SIZE correction for wide coalesced reads (MI355X_MICROARCH.md)
    return (2.0 * k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024.0, os.path.relpath(profile_path(tag), ROOT) + " (separate --pmc passes of this command)"


PROFILE_TAG = {"C1": "c1", "C2": "c2", "C3": "c3", "C4": "c4", "HEADLINE@600": "headline600", "C2@600": "c2_600", "C3@600": "c3_600", "INIT(frame_id=5)": "init"}


def profile_entry(name, assoc_ms):
    """what the committed rocprofv3 passes of this configuration say (profiles/r03_<tag>_*): HBM traffic per launch, hit rate,
    instruction mix and the issue floor -- dropped when the kernel sources changed since (profile_stale)"""
    tag = PROFILE_TAG.get(name)
    if tag is None or not os.path.exists(profile_path(tag)):
        return None
    k, stale = load_profile(tag)
    ent = {"source": os.path.relpath(profile_path(tag), ROOT), "profile_stale": bool(stale)}
    if k is None or stale:
        return ent
    traffic, _ = traffic_from_profile(tag)
    hit, miss = k.get("TCC_HIT_sum"), k.get("TCC_MISS_sum")
    ent.update({"traffic_bytes_per_launch": traffic, "l2_hit_rate": (hit / (hit + miss)) if hit and miss is not None and (hit + miss) > 0 else None,
                "hbm_measured_GBs": (traffic / (assoc_ms * 1e-3) / 1e9) if traffic and assoc_ms > 0 else None, "issue": issue_roofline(assoc_ms, tag)})
    return ent


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


def make_stream(sweep0, prior_state0, sweep_seed, n_kp, L, pattern, count, iterations_of=None):
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
        sw = synth.make_sweep(sweep_seed + 100 * j, n_kp, L, pattern=pattern)
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
    one from its own prior -> swap.  Every sweep crosses PCIe exactly once per solve; no host synchronisation."""

    def __init__(self, lio, stream, opts, prior_cov, frame_id, n_kp):
        import ctypes
        self.lio, self.stream, self.S, self.pos = lio, stream, len(stream), 0
        for e in stream:
            e["solve"] = lio.bound_solver(opts, e["prior_state"], prior_cov, e["state0"], e["sweep"]["t_last"], frame_id, n_kp)
            e["ptr"] = e["pin"].array.ctypes.data_as(ctypes.c_void_p)          # arguments converted once: the loop below calls the C entry points directly
            e["n"] = int(len(e["pin"].array))
        self._prefetch, self._swap, self._h = lio.lib.srl_lio_prefetch_sweep_during_solve, lio.lib.srl_lio_swap_sweep, lio.h

    def begin(self):
        self.lio.prefetch_sweep(self.stream[self.pos % self.S]["pin"].array)
        self.lio.swap_sweep()

    def step(self):
        k = self.pos
        e = self.stream[k % self.S]
        # sweep k + 1 arrives during the solve of sweep k: its upload is issued by the solve itself, beside the kernel of the first pass
        nx = self.stream[(k + 1) % self.S]
        rc = self._prefetch(self._h, nx["ptr"], nx["n"])
        rc2, it, nr = e["solve"]()
        rc = rc or rc2 or self._swap(self._h)
        if rc:
            lib = self.lio.lib
            why = (lib.srl_lio_last_error(self.lio.h) or b"").decode(errors="replace") or (lib.srl_last_error(self.lio.ctx.h) or b"").decode(errors="replace")
            raise RuntimeError(f"stream step failed with status {rc} on sweep {k % self.S} of the stream: {why}")
        self.pos = k + 1
        return {"iters": it, "num_residuals": nr, "state": e["solve"].state, "sweep": k % self.S}

    def close(self):
        for e in self.stream:
            e["pin"].close()


def run_config(name, workload, max_res, frame_id, steps, warmup, device, po, backend, threads, stream_sweeps=4):
    """one BASELINE configuration on this GPU, measured like the headline: a stream of distinct sweeps (prefetch -> solve -> swap, every
    sweep crossing PCIe once per solve); rate, per-iteration time, association-kernel time and roofline fraction, parity of the solved
    state of sweep 0 against the oracle"""
    n_kp, map_pts, pattern, seed = synth.CONFIGS[workload]
    cands, L = synth.map_candidates(seed, map_pts)
    sweep = synth.make_sweep(seed + 1000, n_kp, L, pattern=pattern)
    lio = srl.Lio(device)
    streamer = None
    try:
        lio.add_points_to_map(cands)
        del cands
        prior_state = synth.eskf_prior(_EskfAdapter(lio), sweep["q_pred"], sweep["t_pred"], sweep["vel"]).copy()
        prior_cov = lio.eskf_get_cov().copy()
        state0 = np.concatenate([sweep["q_pred"], sweep["t_pred"], sweep["vel"], np.zeros(6)])
        opts = srl.default_opts(max_num_residuals=max_res)
        def iterations_of(e):
            lio.resident_sweep(e["sweep"]["raw"])
            rc_, it_, _ = lio.bound_solver(opts, e["prior_state"], prior_cov, e["state0"], e["sweep"]["t_last"], frame_id, n_kp)()
            return it_ if rc_ == 0 else -1

        streamer = Streamer(lio, make_stream(sweep, prior_state, seed + 1000, n_kp, L, pattern, stream_sweeps, iterations_of), opts, prior_cov, frame_id, n_kp)
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
        # on the host besides the total, so that one scheduling hiccup in a region of a few milliseconds shows as what it is
        lio.ctx.disarm()             # (a device-wide synchronisation would otherwise wait for the launch the last pass armed to leave by itself)
        torch.cuda.synchronize()
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
        lio.ctx.disarm()
        torch.cuda.synchronize()
        el = time.perf_counter() - t
        arm1 = lio.ctx.arm_stats()
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


def run_pipeline(device, frame_points=(24_000, 65_536), reps=9):
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


def eo_last_cov(po, backend, omap, oo, prior_state, prior_cov, sweep, state0, frame_id):
    """covariance the oracle leaves after the same solve (for the bitwise oracle-vs-reference-TU flag of the bench line)"""
    eo = po.Eskf(backend)
    eo.set_state(prior_state); eo.set_cov(prior_cov)
    po.update_iekf(omap, eo, oo, sweep["raw"], state0, sweep["t_last"], frame_id=frame_id)
    return eo.get_cov()


def pin_to_gpu_numa_node(device_index):
    """Run this process on the CPUs of the NUMA node the GPU hangs off (2-socket hosts: the mailbox read and the doorbell
    write of every ESIKF iteration otherwise cross the socket interconnect -- measured +3 us per iteration, tools/numa_probe.py)
    through the library's own helper, srl_thread_pin_to_gpu_numa (the main thread is pinned before any other thread exists, so
    the whole process follows).  Standard placement for a latency-bound host loop; INTEGRATION.md says the same for the node.
    Never fatal."""
    info = {"pinned": False}
    try:
        import sr_livo_amd as srl
        ctx = srl.Context(device_index)
        try:
            node = ctx.pin_thread_to_gpu_numa()
        finally:
            ctx.close()
        if node is not None:
            info.update(pinned=True, numa_node=int(node), cpus=len(os.sched_getaffinity(0)))
    except Exception as e:  # noqa: BLE001
        info["error"] = repr(e)
    return info


class c_stdout_to_stderr:
    """RCCL prints a version banner through C stdio on communicator creation; the bench contract is ONE JSON line on stdout.
    Inside this block file descriptor 1 points at stderr, and the C buffers are flushed before it is pointed back."""

    def __enter__(self):
        import ctypes
        sys.stdout.flush()
        self.libc = ctypes.CDLL(None)
        self.libc.fflush(None)
        self.saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        self.libc.fflush(None)
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


LINE_LIMIT_BYTES = 8000     # the driver keeps an 8 KB tail of stdout: the ONE line it parses must fit with room to spare
DETAIL_PATH = os.path.join(ROOT, "gpurun_out", "bench_detail.json")


def _num(x, sig=6):
    """floats to `sig` significant digits (the line is read by a parser and by people: 17 digits help neither); NaN / inf -> None
    (strict JSON has neither)"""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, (float, np.floating)):
        x = float(x)
        if x != x or x in (float("inf"), float("-inf")):
            return None
        return float(f"{x:.{sig}g}")
    if isinstance(x, np.integer):
        return int(x)
    if isinstance(x, dict):
        return {k: _num(v, sig) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_num(v, sig) for v in x]
    return x


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def compact_line(out):
    """The ONE stdout line of the bench contract, from the full result dictionary: contract keys, config.workload, a compact roofline
    and cpu_baseline, parity, the PCIe-inclusive rates and one short tuple per BASELINE configuration.  Everything else (per-config
    profile blocks, A/B legs, notes) goes to gpurun_out/bench_detail.json (`detail`).  Strict JSON, < LINE_LIMIT_BYTES."""
    line = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                     "vs_baseline", "dtype", "data")}
    cfg = out.get("config", {})
    line["config"] = _pick(cfg, ("workload", "parallelism", "esikf_iterations_per_solve", "residuals_used", "kernel_launches_per_solve", "launch_mode"))
    line["ms_per_esikf_iter"] = out.get("ms_per_esikf_iter")
    r = out.get("roofline") or {}
    roof = {k: r.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic")}
    roof.update(_pick(r, ("kernel", "avg_launch_ms", "launches", "launches_in_region", "event_period", "algorithmic_bytes_per_launch", "compulsory_bytes_per_launch",
                          "hbm_measured_GBs", "traffic_source", "profile_stale", "launch_duration_includes")))
    if isinstance(r.get("issue"), dict):
        roof["issue"] = _pick(r["issue"], ("bound", "frac", "floor_us", "valu_floor_us", "salu_floor_us", "lds_floor_us"))
    for k in ("association_only", "unarmed"):
        if isinstance(r.get(k), dict):
            roof[k] = _pick(r[k], ("avg_launch_ms", "frac", "launches"))
    line["roofline"] = roof
    for k in ("cpu_baseline", "cpu_baseline_port", "cpu_baseline_all_cores"):
        c = out.get(k)
        if isinstance(c, dict):
            e = _pick(c, ("value", "unit", "cores", "kind", "ms_per_solve", "note"))
            if k == "cpu_baseline" and "sample" in c:
                e["sample"] = str(c["sample"])[:160]
            line[k] = e
    if isinstance(out.get("parity"), dict):
        line["parity"] = out["parity"]
    pc = cfg.get("pcie_inclusive_sweeps_per_s") or {}
    line["pcie_inclusive_sweeps_per_s"] = {"pipelined_prefetch": pc.get("pipelined_prefetch"), "pinned": pc.get("pinned_upload_then_solve"),
                                           "pageable": pc.get("pageable_upload_then_solve")}
    if isinstance(out.get("stream"), dict):
        line["stream"] = _pick(out["stream"], ("sweeps", "solves", "sweeps_per_s_mean", "sweeps_per_s_median", "arm_stats", "state_of_sweep0_equals_resident_solve"))
    if isinstance(out.get("arm_stats_timed_region"), dict):
        line["arm_stats"] = out["arm_stats_timed_region"]
    if isinstance(out.get("resident_resolve"), dict):
        line["resident_resolve"] = _pick(out["resident_resolve"], ("sweeps_per_s", "us_per_esikf_iter"))
    if isinstance(out.get("clock_warmup"), dict):
        line["clock_warmup"] = out["clock_warmup"]          # untimed solves before the W warm-up steps (steady clocks): disclosed in the line
    for k in ("launch_ab", "pipeline", "comm", "aux_independent_sweeps_per_s", "multi_gpu_note", "fallback"):
        if out.get(k) is not None:
            line[k] = out[k]
    cfgs = []
    for c in out.get("configs") or []:
        if "error" in c:
            cfgs.append({"name": c.get("name"), "error": str(c["error"])[:120]})
            continue
        issue = ((c.get("profile") or {}).get("issue") or {}).get("frac")
        cfgs.append({"name": c["name"], "us_per_iter": c["ms_per_esikf_iter"] * 1e3, "kernel_us": c.get("kernel_us", c.get("assoc_kernel_us")), "frac": c.get("hbm_roofline_frac"),
                     "issue_frac": issue, "sweeps_per_s": c["sweeps_per_s"], "iters": c["esikf_iterations"], "armed": c.get("armed"),
                     "us_per_iter_r04_loop": c.get("resident_resolve_always_armed_us_per_iter"),
                     "parity_ok": (c.get("parity") or {}).get("ok")})
    if cfgs:
        line["configs"] = cfgs
    line["detail"] = os.path.relpath(DETAIL_PATH, ROOT)
    line = _num(line)
    # value and ms_per_step keep their full precision: the driver cross-checks one against the other
    line["value"], line["ms_per_step"] = out.get("value"), out.get("ms_per_step")
    text = json.dumps(line, allow_nan=False, separators=(",", ":"))
    if len(text) > LINE_LIMIT_BYTES:      # never print a line the driver cannot parse: shed the optional blocks, largest first
        for k in ("pipeline", "launch_ab", "configs", "cpu_baseline_all_cores", "cpu_baseline_port", "comm"):
            line.pop(k, None)
            text = json.dumps(line, allow_nan=False, separators=(",", ":"))
            if len(text) <= LINE_LIMIT_BYTES:
                break
    return text


def write_detail(out):
    try:
        os.makedirs(os.path.dirname(DETAIL_PATH), exist_ok=True)
        with open(DETAIL_PATH, "w") as f:
            json.dump(_num(out, 9), f, indent=1)
    except OSError as e:
        print(f"bench.py: could not write {DETAIL_PATH}: {e}", file=sys.stderr)


SELF_LAUNCH_TIMEOUT_S = 600


def self_launch(n_gpus):
    """`python bench.py --gpus N` without a launcher around it: re-run this command line as N ranks under torch.distributed.run
    (one process per GPU, rendezvous over loopback) and pass rank 0's line through.  The sharded path has never met a real N-GPU node
    (the build boxes have one GPU): should the chosen exchange fail or hang there, the run falls back -- RCCL all-reduce -> direct peer
    exchange -> independent replicas (BASELINE config 5, no collective) -- and says so in the line (`fallback`)."""
    import socket
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    base = [a for a in sys.argv[1:]]
    explicit = any(a in ("--transport", "--mode") or a.startswith("--transport=") or a.startswith("--mode=") for a in base)
    attempts = [([], None)] if explicit else [([], None), (["--transport", "peer"], "RCCL form failed or hung: direct peer exchange"),
                                              (["--mode", "replay"], "sharded forms failed or hung: independent replicas, one sweep per GPU")]
    last_rc = 1
    for extra, note in attempts:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + base + extra
        # the launcher and its ranks form a process group of their own: an attempt that hangs is ended as a whole (killing only the launcher
        # would leave its ranks spinning on the GPUs under the next attempt)
        p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, start_new_session=True)
        try:
            out, _ = p.communicate(timeout=SELF_LAUNCH_TIMEOUT_S)
        except subprocess.TimeoutExpired:
            import signal
            try:
                os.killpg(p.pid, signal.SIGKILL)               # exactly the group started above
            except ProcessLookupError:
                pass
            p.communicate()
            print(f"bench.py: {' '.join(extra) or 'default transport'} timed out after {SELF_LAUNCH_TIMEOUT_S} s", file=sys.stderr)
            continue
        last_rc = p.returncode
        lines = [ln for ln in out.decode(errors="replace").splitlines() if ln.startswith("{")]
        if p.returncode == 0 and lines:
            line = lines[-1]
            if note:
                try:
                    d = json.loads(line)
                    d["fallback"] = note
                    line = json.dumps(d, allow_nan=False, separators=(",", ":"))
                except ValueError:
                    pass
            print(line, flush=True)
            return 0
        print(f"bench.py: {' '.join(extra) or 'default transport'} failed with exit code {p.returncode}", file=sys.stderr)
    return last_rc or 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="HEADLINE", choices=sorted(synth.CONFIGS))
    ap.add_argument("--mode", default="sharded", choices=["sharded", "replay"])
    ap.add_argument("--event-period", type=int, default=0,
                    help="HIP-event timing of the association kernel inside the timed region: every N-th launch (0 = 5 when --steps >= 10, else every launch)")
    ap.add_argument("--max-num-residuals", type=int, default=INT_MAX,
                    help="2^31-1 = throughput headline (every keypoint contributes); 600 = shipped yaml value")
    ap.add_argument("--frame-id", type=int, default=100, help="< 20: init mode (r = 2, >= 16 iterations)")
    ap.add_argument("--stream-sweeps", type=int, default=4,
                    help="distinct sweeps (own seeds, poses, priors) the timed stream cycles through; every sweep crosses PCIe once per solve")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the per-configuration array (C1..C4, headline@600, init mode)")
    ap.add_argument("--select-mode", type=int, default=0)
    ap.add_argument("--no-fused-reduce", action="store_true", help="A/B: always run the separate reduce kernel")
    ap.add_argument("--no-numa-pin", action="store_true", help="A/B: do not pin the process to the GPU-local NUMA node")
    ap.add_argument("--no-aux-legs", action="store_true",
                    help="only the timed configuration runs on the GPU (profiling: no association-only / PCIe legs in the trace)")
    ap.add_argument("--clock-warmup-ms", type=float, default=50.0,
                    help="setup, before the W warm-up steps: solve for this long so that the timed region starts at steady clocks (0 = off)")
    ap.add_argument("--no-armed", action="store_true",
                    help="A/B, profiling: armed launches off -- every ESIKF iteration pays its launch call, dispatch and ramp (round 3's form)")
    ap.add_argument("--transport", choices=("rccl", "peer"), default="rccl",
                    help="sharded mode: how the 50-double rows of the ranks are summed.  rccl: ncclAllReduce on the library's own communicator; "
                         "peer: direct stores into the peers' inboxes over xGMI (srl_peer_attach, HIP IPC handles exchanged over gloo)")
    ap.add_argument("--force-comm", action="store_true",
                    help="attach an RCCL communicator even at world size 1 (exercises the sharded code path on a 1-GPU box)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("SRL_BENCH_ALL_ON_DEVICE0") == "1":
        # test hook (1-GPU boxes): every rank on device 0 -- lets `--transport peer` run its N > 1 path end to end (the inboxes travel
        # as HIP IPC handles exactly as between GPUs); RCCL refuses two ranks on one device.  Never a performance figure.
        local_rank = 0
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))           # `python bench.py --gpus N`: this process becomes the launcher of N ranks
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    try:
        os.nice(-10)                  # a 20-step region lasts 2 ms: one pre-emption of the polling thread (80 us) is 4 % of it.  Best effort.
    except OSError:
        pass
    torch.cuda.set_device(local_rank)
    torch.cuda.synchronize()          # torch's lazy CUDA initialisation happens HERE, not inside the barrier in front of the timed region
                                      # (hundreds of ms of idle GPU there: the first timed step then ran at 147 us instead of 100)
    pin_info = {"pinned": False, "disabled": True} if args.no_numa_pin else pin_to_gpu_numa_node(local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:
        import datetime
        import torch.distributed as dist_mod
        dist = dist_mod
        # control plane only (barrier, 128-byte id broadcast, one max-reduce): gloo over loopback.  No torch NCCL process
        # group is created -- the only communicator on the GPUs is the library's own, on the process's one RCCL instance.
        with c_stdout_to_stderr():          # gloo announces its connections on stdout
            dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=300))

    n_kp, map_pts, pattern, seed = synth.CONFIGS[args.workload]
    sharded = ((world > 1 or (args.force_comm and "RANK" in os.environ)) and args.mode == "sharded")
    sweep_seed = seed + 1000 + (rank if (world > 1 and not sharded) else 0)
    map_seed = seed + (rank if (world > 1 and not sharded) else 0)

    # ---------------- inputs: map built by the product's device-side addPointsToMap, sweep pinned in HBM
    t0 = time.time()
    cands, L = synth.map_candidates(map_seed, map_pts)
    sweep = synth.make_sweep(sweep_seed, n_kp, L, pattern=pattern)
    lio = srl.Lio(local_rank)
    if args.no_fused_reduce:
        lio.ctx.set_fused_reduce(0)
    lio.add_points_to_map(cands)
    n_map = lio.map_size()
    comm_info = None
    if sharded and args.transport == "peer" and world > 1:
        handles = [None] * world
        dist.all_gather_object(handles, lio.ctx.peer_export()[0])
        lio.ctx.peer_attach(world, rank, handles=handles)
        comm_info = {"transport": "direct peer exchange (srl_peer_attach): rows stored into the peers' inboxes, summed in rank order inside the "
                                  "association kernel's finishing workgroup; no RCCL call on the data path"}
    elif sharded or (args.force_comm and dist is not None):
        uid = [srl.Context.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        with c_stdout_to_stderr():
            lio.ctx.comm_init_rank(world, rank, uid[0])
        origin, ver, pre = srl.comm_backend_info()
        comm_info = {"rccl": origin, "version": ver, "instance": "already loaded in the process" if pre else "dlopen'ed by libsrlivo_hip.so"}
    prior_state = synth.eskf_prior(_EskfAdapter(lio), sweep["q_pred"], sweep["t_pred"], sweep["vel"]).copy()
    prior_cov = lio.eskf_get_cov().copy()
    state0 = np.concatenate([sweep["q_pred"], sweep["t_pred"], sweep["vel"], np.zeros(6)])
    opts = srl.default_opts(max_num_residuals=args.max_num_residuals, select_mode=args.select_mode)
    # THE STREAM (one full solve per sweep, "incl. H2D of the sweep"; src/lioOptimization.cpp:1003-1027 never solves a sweep
    # twice): S distinct sweeps of the scene -- own seeds, own ground-truth and predicted poses, hence own priors -- in page-locked host
    # memory.  Sweep 0 is the sweep every other leg (CPU baselines, parity, profiles) uses.
    def iterations_of(e):
        lio.resident_sweep(e["sweep"]["raw"])
        sv = lio.bound_solver(opts, e["prior_state"], prior_cov, e["state0"], e["sweep"]["t_last"], args.frame_id, n_kp)
        rc, it, _ = sv()
        return it if rc == 0 else -1

    stream = make_stream(sweep, prior_state, sweep_seed, n_kp, L, pattern, max(int(args.stream_sweeps), 1), iterations_of)
    S = len(stream)
    lio.resident_sweep(sweep["raw"])
    setup_s = time.time() - t0
    # The interpreter's cyclic collector is host noise, not part of the path: with torch imported one full collection costs
    # ~40 ms (measured: exactly one 38-43 ms step per run, at a fixed step index, gone without torch in the process).  Move
    # everything allocated so far out of the collector's reach; the collector itself stays on.
    gc.collect()
    gc.freeze()

    # one step = eskf_set_state + eskf_set_cov (reset the prior) + update_iekf on the resident sweep, through a closure
    # that converts its arguments once (the per-call numpy/ctypes marshalling of the generic wrappers costs ~10 us)
    _solve = lio.bound_solver(opts, prior_state, prior_cov, state0, sweep["t_last"], args.frame_id, n_kp)
    streamer = Streamer(lio, stream, opts, prior_cov, args.frame_id, n_kp)
    if args.no_armed:
        lio.ctx.set_armed_launch(False)

    def solve():
        rc, it, nr = _solve()
        if rc:
            why = (lio.lib.srl_lio_last_error(lio.h) or b"").decode(errors="replace") or (lio.lib.srl_last_error(lio.ctx.h) or b"").decode(errors="replace")
            raise SystemExit(f"update_iekf failed with status {rc}: {why}")
        return {"iters": it, "num_residuals": nr, "state": _solve.state}

    # one step of the stream: the NEXT sweep starts crossing PCIe (copy stream) -> full ESIKF solve of the current one from its own prior ->
    # the next sweep becomes current (class Streamer)
    stream_begin, stream_step = streamer.begin, streamer.step

    def barrier():
        lio.ctx.disarm()      # (the launch the last pass armed would hold a device-wide synchronisation until it leaves by itself)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if dist is None:
            return x
        te = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        return float(te.item())

    # setup, untimed: bring GPU and host core to their steady clocks (a timed region of 20 solves lasts 2 ms: measured right after the
    # map build, its steps kept getting faster until the end -- 107 -> 103 us per solve); the W warm-up steps of the contract follow
    lio.ctx.set_profiling(2)          # the event pairs of the timed region exist and have been recorded once before it starts (first use of an
                                      # event costs; 1 024 creations right before the region left the GPU idle long enough to drop its clocks)
    t_cw = time.perf_counter()
    n_cw = 0
    stream_begin()
    if dist is None:
        while time.perf_counter() - t_cw < args.clock_warmup_ms * 1e-3:
            stream_step()
            n_cw += 1
    else:
        for _ in range(int(args.clock_warmup_ms * 4)):        # ranks solve in lock-step (the exchange is collective): a count, not a clock
            stream_step()
            n_cw += 1
    # HIP events on the context's own stream, inside the timed region: one pair around every association launch, read
    # back lazily after the region (mode 2) -- the full per-call breakdown (mode 1: four events + a sync per call, ~20 us
    # of host time per iteration) is taken on a few extra solves after the timed region instead.
    lio.ctx.set_profiling(2)          # (reads back the clock warm-up's several hundred event pairs -- milliseconds of idle GPU -- BEFORE the W warm-up steps)
    # ... of every `event_period` launches ONE is timed (two event records: the launch before it leaves its end event as the start); an
    # event record behind every launch costs the loop ~2.5 us per launch (HISTORY.md, round 5).  Odd period: first and second iterations
    # of the solves are sampled alike.  Short regions time every launch.
    event_period = args.event_period if args.event_period > 0 else (5 if args.steps >= 10 else 1)
    lio.ctx.set_profiling_period(event_period)
    for _ in range(args.warmup):
        r = stream_step()
    lio.ctx.timing_mark()             # the figures are those of the timed region alone: the warm-up's launches are left out when their events
                                      # are read -- WITHOUT reading anything back here (srl_get_timing waits for every event and cancels the
                                      # armed launch: the GPU then idled for ~300 us in front of the region and its first step ran at 147 us)
    tim_w = {k: 0 for k in ("calls", "sum_assoc_ms", "sum_algorithmic_bytes", "sum_passes", "sum_keypoints")}
    step_end = np.empty(args.steps)
    barrier()
    arm_before = lio.ctx.arm_stats()
    t1 = time.perf_counter()
    iters_timed = 0
    for k in range(args.steps):
        r = stream_step()
        iters_timed += r["iters"]
        step_end[k] = time.perf_counter()
    arm_after = lio.ctx.arm_stats()             # (before the barrier's own disarm of the launch armed behind the last pass)
    barrier()
    elapsed = time.perf_counter() - t1
    per_step_us = np.diff(np.concatenate([[t1], step_end])) * 1e6
    import types
    tim_t = lio.ctx.timing()
    tim = types.SimpleNamespace(**{k: getattr(tim_t, k) - v for k, v in tim_w.items()})
    launches_timed = lio.last_solve_launches()
    arm_stats = {k: arm_after[k] - arm_before[k] for k in arm_after}
    lio.ctx.set_profiling(0)
    lio.ctx.set_profiling_period(1)   # (the legs behind the timed region time every launch)
    # the same loop for >= 1 000 solves with a stamp per solve (a 20-step region lasts 2 ms): median- and mean-based rates, the states
    # every sweep of the stream was solved to (compared with the oracle and with the launch-per-iteration form further down)
    n_long = 0 if args.no_aux_legs else max(1000, args.steps)
    per_long = np.empty(n_long)
    stream_states = {}
    barrier()
    arm_b2 = lio.ctx.arm_stats()
    t_long = time.perf_counter()
    for k in range(n_long):
        ta = time.perf_counter()
        rr = stream_step()
        per_long[k] = time.perf_counter() - ta
        if rr["sweep"] not in stream_states:
            stream_states[rr["sweep"]] = (rr["iters"], rr["num_residuals"], rr["state"].copy())
    arm_a2 = lio.ctx.arm_stats()
    barrier()
    el_long = max_over_ranks(time.perf_counter() - t_long) if n_long else None
    stream_long = None
    if n_long:
        stream_long = {"solves": n_long, "sweeps_per_s_mean": n_long / el_long, "sweeps_per_s_median": 1.0 / float(np.median(per_long)),
                       "us_per_solve_p10_p50_p90_max": [float(np.percentile(per_long, q)) * 1e6 for q in (10, 50, 90, 100)],
                       "arm_stats": {k: arm_a2[k] - arm_b2[k] for k in arm_a2},
                       "solves_over_1ms": int(np.count_nonzero(per_long > 1e-3))}
    elapsed_rank = elapsed                    # this rank's own clock (the line carries min / max over the ranks)
    comm_state = lio.ctx.comm_info()          # (read while the communicator / peer table of the timed region is still attached)
    if comm_state.get("transport_used") == "peer":
        rep, failed = lio.ctx.peer_stats()    # passes repeated because a rank's row was late (srl_peer_set_deadline_ms), session state
        comm_state.update(passes_repeated_for_a_late_rank=rep, session_failed=failed)
    elapsed = max_over_ranks(elapsed)
    launches_per_solve = launches_timed
    # A/B: the same stream with one launch call per ESIKF iteration on the critical path (armed launches off: round 3's form), with the
    # kernel's HIP-event duration in that form (an armed launch's event pair also brackets its wait for the host's pose); every sweep of
    # the stream must come out with the same bits either way
    launch_ab, tim_unarmed = None, None
    if world == 1 and not args.no_aux_legs and not args.no_armed:
        lio.ctx.set_armed_launch(False)
        for _ in range(3):
            stream_step()
        lio.ctx.set_profiling(2)
        barrier()
        t_un = time.perf_counter()
        un_states, un_iters = {}, 0
        for _ in range(max(args.steps, 2 * S)):
            r_un = stream_step()
            un_iters += r_un["iters"]
            un_states.setdefault(r_un["sweep"], r_un["state"].copy())
        barrier()
        el_un = time.perf_counter() - t_un
        tim_unarmed = lio.ctx.timing()
        lio.ctx.set_profiling(0)
        lio.ctx.set_armed_launch(True)
        launch_ab = {"armed_us_per_iter": elapsed * 1e6 / max(iters_timed, 1),
                     "launch_per_iteration_us_per_iter": el_un * 1e6 / max(un_iters, 1),
                     "state_bitwise_equal": bool(all(j in stream_states and np.array_equal(un_states[j], stream_states[j][2]) for j in un_states)),
                     "sweeps_compared": len(un_states)}
    # the remaining legs work on sweep 0 resident in HBM
    lio.resident_sweep(sweep["raw"])
    for _ in range(3):
        r0 = solve()
    r0 = dict(r0, state=r0["state"].copy())          # sweep 0 solved from its prior: what the CPU legs and the parity figures refer to
    # ... first: sweep 0 solved again and again from the same prior without leaving HBM (rounds 1-4 printed this as `value`; no node can
    # do it -- a new sweep arrives for every solve -- so it is an auxiliary figure now: what the solve costs without the stream around it)
    resident = None
    if not args.no_aux_legs:
        barrier()
        t_r = time.perf_counter()
        for _ in range(args.steps):
            r_res = solve()
        barrier()
        el_r = max_over_ranks(time.perf_counter() - t_r)
        resident = {"sweeps_per_s": args.steps / el_r, "us_per_esikf_iter": el_r / args.steps * 1e6 / max(r_res["iters"], 1),
                    "what": "sweep 0 re-solved back to back, resident in HBM (no upload, no swap): the first pass of every solve pays a launch"}
    lio.ctx.set_profiling(1)
    for _ in range(max(3, min(10, args.steps))):
        solve()
    tim_full = lio.ctx.timing()
    lio.ctx.set_profiling(0)
    # the sharded code path with ONE rank (all a 1-GPU box can run of it): 1-rank RCCL communicator, collectives forced --
    # fused pass into a device-side mailbox, ncclAllReduce of 50 doubles, publish kernel.  What the exchange step costs per
    # ESIKF iteration when there is nobody to exchange with; not a scaling figure.
    comm_1rank = None
    if world == 1 and dist is None and not args.no_aux_legs:
        try:
            os.environ["SRL_FORCE_COLLECTIVES"] = "1"
            with c_stdout_to_stderr():
                lio.ctx.comm_init_rank(1, 0, srl.Context.comm_unique_id())
                lio.resident_sweep(sweep["raw"])
                for _ in range(3):
                    solve()
            torch.cuda.synchronize()
            t_c = time.perf_counter()
            for _ in range(args.steps):
                r_c = solve()
            torch.cuda.synchronize()
            el_c = time.perf_counter() - t_c
            comm_1rank = {"ms_per_esikf_iter": el_c / args.steps * 1e3 / max(r_c["iters"], 1), "sweeps_per_s": args.steps / el_c,
                          "extra_us_per_iter_vs_timed_region": (el_c - elapsed) / args.steps * 1e6 / max(r_c["iters"], 1),
                          "what": "1-rank RCCL communicator with the collectives forced (fused pass -> device mailbox -> ncclAllReduce of 50 doubles -> publish kernel)"}
        except Exception as e:  # noqa: BLE001
            comm_1rank = {"error": repr(e)}
        finally:
            os.environ.pop("SRL_FORCE_COLLECTIVES", None)
            lio.ctx.comm_destroy()
            lio.resident_sweep(sweep["raw"])
            solve()
    # the association work alone: the same launches with the final reduction in its own kernel (the fused tail -- row
    # publish, arrival counters, final sum by the last workgroup -- is part of the kernel the timed region runs)
    tim_unfused = None
    if world == 1 and not args.no_fused_reduce and not args.no_aux_legs:
        lio.ctx.set_fused_reduce(0)
        solve()
        lio.ctx.set_profiling(2)
        for _ in range(max(5, min(20, args.steps))):
            solve()
        tim_unfused = lio.ctx.timing()
        lio.ctx.set_profiling(0)
        lio.ctx.set_fused_reduce(1)
        solve()

    # The other ways a sweep can reach the solve (`value` = the pipelined stream above: the next sweep crosses PCIe on the copy stream
    # while the current one is solved): upload and solve back to back on ONE stream, no overlap, no host synchronisation --
    # (a) from page-locked memory (srl_pinned_alloc), (b) from ordinary pageable memory through the context's pinned ring.
    # Each leg runs >= 200 solves after its own warm-up; the median-based rate and every step over 1 ms are kept beside the mean.
    n_pcie = max(200, args.steps)
    rates = {"pinned": None, "pageable": None, "pipelined": stream_long["sweeps_per_s_mean"] if stream_long else None}
    medians, stalls = {}, {}
    if stream_long:
        medians["pipelined"] = stream_long["sweeps_per_s_median"]

    def step_sequential(src):
        return lambda k: (lio.resident_sweep(src), solve())

    legs = (("pinned", step_sequential(stream[0]["pin"].array)), ("pageable", step_sequential(sweep["raw"])))
    for label, step in (() if args.no_aux_legs else legs):
        lio.resident_sweep(stream[0]["pin"].array)
        for k in range(4):
            step(k)
        barrier()
        per = np.empty(n_pcie)
        t2 = time.perf_counter()
        for k in range(n_pcie):
            ta = time.perf_counter()
            step(k)
            per[k] = time.perf_counter() - ta
        barrier()
        mult = world if (world > 1 and not sharded) else 1
        rates[label] = mult * n_pcie / max_over_ranks(time.perf_counter() - t2)
        medians[label] = mult / float(np.median(per))
        stalls[label] = [{"step": int(k), "ms": round(float(per[k]) * 1e3, 2)} for k in np.nonzero(per > 1e-3)[0][:8]]
    if not args.no_aux_legs:
        lio.resident_sweep(sweep["raw"]); solve()
    lio.ctx.disarm()
    torch.cuda.synchronize()

    # N > 1, sharded: also report the other way to use N GPUs (BASELINE config 5: one sweep per GPU, no collective),
    # measured after the timed region on the same contexts; informational, never `value`.
    replicas_rate = None
    if sharded:
        r_sharded = r0
        lio.ctx.comm_suspend(True)           # keep the communicator, run the whole sweep locally
        lio.resident_sweep(sweep["raw"])
        solve()
        barrier()
        t3 = time.perf_counter()
        for _ in range(args.steps):
            solve()
        torch.cuda.synchronize()
        replicas_rate = world * args.steps / max_over_ranks(time.perf_counter() - t3)
        r0 = r_sharded

    r_stream_last = r
    r = r0
    iters = r["iters"]
    sweeps_per_step = world if (world > 1 and not sharded) else 1
    value = sweeps_per_step * args.steps / elapsed
    ms_per_step = elapsed / args.steps * 1e3

    calls = max(tim.calls, 1)
    fcalls = max(tim_full.calls, 1)
    passes = max(tim.sum_passes, 1)
    assoc_ms = tim.sum_assoc_ms / calls
    bytes_per_launch = tim.sum_algorithmic_bytes / calls
    achieved = bytes_per_launch / (assoc_ms * 1e-3) / 1e9 if assoc_ms > 0 else 0.0
    nb = 2 if args.frame_id < 20 else 1

    headline_default = args.workload == "HEADLINE" and world == 1 and args.frame_id >= 20 and args.max_num_residuals == INT_MAX
    traffic, traffic_src = traffic_from_profile() if headline_default else (None, None)
    roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
            "kernel": f"srl_assoc_armed_kernel<{nb}>" if (not args.no_armed and world == 1) else f"srl_assoc_kernel<{nb}>",
            "avg_launch_ms": assoc_ms, "launches": tim.calls, "launches_in_region": iters_timed, "event_period": event_period,
            "passes_per_launch": passes / calls, "avg_pass_ms": tim.sum_assoc_ms / passes,
            "algorithmic_bytes_per_launch": bytes_per_launch, "algorithmic_bytes_per_pass": tim.sum_algorithmic_bytes / passes,
            "profile_stale": bool(load_profile()[1]) if headline_default else None,
            "reduce_kernel_avg_ms": tim_full.sum_reduce_ms / fcalls, "device_total_avg_ms": tim_full.sum_total_ms / fcalls,
            "note": "achieved = ALGORITHMIC bytes (24 + 12 (2r+1)^3 + 12 P_k per keypoint, SURVEY 8(d)) / launch time: the rate at which the "
                    "reference's byte stream is consumed.  The working set is L2/MALL resident, so real HBM traffic (`traffic`, "
                    "`hbm_measured_GBs`) is far below it and the kernel is bound by instruction issue: see `issue`."}
    armed_used = (lio.ctx.arm_stats()["fired"] > 0) and not args.no_armed
    if armed_used:
        roof["launch_duration_includes"] = "the armed launch's wait for the host's pose (its event pair opens when the pass before it ends)"
    if tim_unarmed is not None and tim_unarmed.calls > 0:
        ms_n = tim_unarmed.sum_assoc_ms / tim_unarmed.calls
        roof["unarmed"] = {"avg_launch_ms": ms_n, "launches": tim_unarmed.calls,
                           "frac": (tim_unarmed.sum_algorithmic_bytes / tim_unarmed.calls) / (ms_n * 1e-3) / 1e9 / HBM_PEAK_GBS if ms_n > 0 else None,
                           "what": "the same kernel launched per iteration (armed launches off): duration without any wait inside"}
    if tim_unfused is not None and tim_unfused.calls > 0:
        ms_u = tim_unfused.sum_assoc_ms / tim_unfused.calls
        roof["association_only"] = {"avg_launch_ms": ms_u, "launches": tim_unfused.calls,
                                    "frac": (tim_unfused.sum_algorithmic_bytes / tim_unfused.calls) / (ms_u * 1e-3) / 1e9 / HBM_PEAK_GBS if ms_u > 0 else None,
                                    "what": "the same launches with the final reduction left to the separate reduce kernel (srl_debug_set_fused_reduce(0)), "
                                            "measured after the timed region: the association kernel proper.  In the timed region the last workgroup also "
                                            "sums the block partials and publishes the result (one kernel per ESIKF iteration)"}
    if rank == 0 and world == 1:
        try:
            keys, counts, _ = lio.ctx.map_download()
            R = synth.quat_to_rot(sweep["q_pred"] / np.linalg.norm(sweep["q_pred"]))
            comp, s_u, p_u = compulsory_bytes(keys, counts, sweep["raw"] @ R.T + sweep["t_pred"], nb)
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

    out = {
        "metric": "sweeps/s (full ESIKF solve of a 64k-pt Livox sweep vs 1M-pt voxel map)",
        "value": value, "unit": "sweeps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "strong" if sharded or world == 1 else "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{args.workload}: stream of {S} distinct {n_kp}-keypoint {pattern} sweeps (own poses and priors), {n_map}-pt voxel map "
                               f"({map_pts} target), max_num_residuals={args.max_num_residuals}, r={nb}, K=20; one full ESIKF solve per sweep, "
                               f"every sweep crosses PCIe once (uploaded on the copy stream during the solve before it: resident in HBM when its solve starts); "
                               f"sweeps drawn with seeds {stream[0]['seed']} + 100 j, keeping those that take sweep 0's {stream[0]['iterations']} ESIKF iterations "
                               f"(skipped seeds: {[x['seed'] for x in stream[0]['skipped']]})",
                   "parallelism": ("point-range shards x%d + %s of the 6x6 normal equations" % (world, "direct peer exchange" if args.transport == "peer" else "RCCL all-reduce")) if sharded
                                  else ("replicas x%d" % world if world > 1 else "single GPU"),
                   "esikf_iterations_per_solve": iters_timed / max(args.steps, 1), "residuals_used": r["num_residuals"],
                   "kernel_launches_per_solve": launches_per_solve,
                   "launch_mode": ("one launch per ESIKF iteration" + ("" if args.no_armed else " (sharded passes behind an RCCL all-reduce are not armed)")) if (args.no_armed or (sharded and arm_stats["fired"] == 0))
                                  else "armed launches: the kernel of pass k+1 is enqueued while pass k runs and receives its pose through the pose box -- across srl_sweep_swap too (the launch armed behind a sweep's last pass is the next sweep's first pass)",
                   "value_is": "SURVEY 8(d)'s metric: sweeps/s of a stream of distinct sweeps, H2D of every sweep included (overlapped with the solve before it). "
                               "Rounds 1-4 printed the rate of ONE sweep re-solved in HBM: now `resident_resolve`; upload-then-solve without overlap:",
                   "pcie_inclusive_sweeps_per_s": {"pipelined_prefetch": rates["pipelined"], "pinned_upload_then_solve": rates["pinned"],
                                                   "pageable_upload_then_solve": rates["pageable"]}},
        "ms_per_esikf_iter": elapsed * 1e3 / max(iters_timed, 1),
        "roofline": roof,
        "launch_ab": launch_ab,
        "per_step_us": [round(float(x), 1) for x in per_step_us],
        "arm_stats_timed_region": arm_stats,
        "stream": dict(stream_long or {}, sweeps=S, state_of_sweep0_equals_resident_solve=bool(0 in stream_states and np.array_equal(stream_states[0][2], r0["state"]))),
        "resident_resolve": resident,
        "sharded_path_one_rank": comm_1rank,
        "host_us_per_iter": {"enqueue": tim_full.sum_host_launch_us / fcalls, "wait_results": tim_full.sum_host_wait_us / fcalls,
                             "build_residuals_call": tim_full.sum_host_total_us / fcalls,
                             "whole_iteration": ms_per_step * 1e3 / max(iters, 1),
                             "note": "first three: extra solves after the timed region with full event profiling (adds ~20 us/iter); "
                                     "whole_iteration: the timed region"},
        "host_placement": dict(pin_info, what="the process runs on the CPUs of the GPU's NUMA node (srl_thread_pin_to_gpu_numa: "
                               "/sys/bus/pci/devices/<gpu>/local_cpulist); on the other socket every ESIKF iteration costs ~3 us more"),
        "pcie_inclusive_sweeps_per_s": rates["pinned"],
        "pcie": {"from_pinned_host_memory_sweeps_per_s": rates["pinned"], "from_pageable_host_memory_sweeps_per_s": rates["pageable"],
                 "pipelined_prefetch_sweeps_per_s": rates["pipelined"], "median_based": medians, "solves_per_leg": n_pcie, "steps_over_1ms": stalls,
                 "value_over_pipelined": (value / rates["pipelined"]) if rates["pipelined"] else None,
                 "value_over_pinned": (value / rates["pinned"]) if rates["pinned"] else None, "bytes_h2d_per_sweep": int(sweep["raw"].nbytes),
                 "note": "`value` = the pipelined stream (the next sweep is uploaded on the copy stream while the current one is solved: "
                         "srl_sweep_prefetch / srl_sweep_swap); pinned / pageable = upload and solve back to back on one stream"},
        "setup_s": setup_s, "clock_warmup": {"ms": args.clock_warmup_ms, "solves": n_cw},
    }
    if comm_info is not None or world > 1:
        # every --gpus N line explains what it ran on: the transport the library actually used after bench.py's fallback chain, the ranks
        # the communicator itself counts, each rank's own per-iteration time (gloo all-gather) and the time behind the association kernel
        # (reduce / exchange / publish) from the post-region profiling pass
        ci = dict(comm_info or {})
        try:
            ci.update(comm_state)
        except Exception as e:  # noqa: BLE001
            ci["info_error"] = repr(e)
        us_rank = elapsed_rank * 1e6 / max(iters_timed, 1)
        if dist is not None:
            per_rank = [None] * world
            dist.all_gather_object(per_rank, us_rank)
            ci["us_per_iter_per_rank"] = {"min": float(min(per_rank)), "max": float(max(per_rank))}
        ci["behind_association_kernel_us"] = tim_full.sum_reduce_ms / fcalls * 1e3
        out["comm"] = ci
    if world > 1:
        out["multi_gpu_note"] = "no multi-GPU scaling curve has been measured by the builder (gpurun boxes expose one GPU); this line is it"
    if replicas_rate is not None:
        out["aux_independent_sweeps_per_s"] = {"value": replicas_rate, "what": "the same N GPUs each solving its own 64k sweep "
                                               "(replicas, no collective; BASELINE config 5), measured after the timed region"}

    # ---------------- CPU baselines + parity figure (rank 0, N = 1 only)
    po = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import pyoracle as po
        backend = "tsl" if os.path.exists(po.LIB_TSL) else "plain"
        ncores = os.cpu_count() or 1
        omap = po.Map(backend)
        omap.add_points(cands)
        oo = po.opts_from_product(opts)

        def cpu_leg(threads, budget_s, max_runs):
            times, ou = [], None
            t_start = time.perf_counter()
            with po.threads(threads):
                while len(times) < max_runs and (time.perf_counter() - t_start) < budget_s:
                    eo = po.Eskf(backend)
                    eo.set_state(prior_state); eo.set_cov(prior_cov)
                    tc = time.perf_counter()
                    ou = po.update_iekf(omap, eo, oo, sweep["raw"], state0, sweep["t_last"], frame_id=args.frame_id)
                    times.append(time.perf_counter() - tc)
            return float(np.median(times)), len(times), ou

        cpu_s, n1, ou = cpu_leg(1, 20.0, 5)
        state_err = rel(r["state"], ou["state"])
        out["cpu_baseline"] = {"value": 1.0 / cpu_s, "unit": "sweeps/s", "cores": 1, "kind": "port",
                               "sample": f"{n1} full solves ({ou['rc']} ESIKF iterations each) of the same sweep and map; "
                                         f"oracle restatement of optimize.cpp, single thread like the reference, "
                                         f"voxel map = {backend}; host has {ncores} cores",
                               "ms_per_solve": cpu_s * 1e3, "ms_per_esikf_iter": cpu_s * 1e3 / max(ou["rc"], 1)}
        # all cores: the thread count that runs fastest on this host (oversubscribing a 256-core box is slower than 64 threads)
        tried = {}
        for nt in sorted({ncores, min(ncores, 128), min(ncores, 64), min(ncores, 32)}, reverse=True):
            tried[nt] = cpu_leg(nt, 4.0, 5)
        best = min(tried, key=lambda k: tried[k][0])
        cpu_all, na, oa = tried[best]
        out["cpu_baseline_all_cores"] = {"value": 1.0 / cpu_all, "unit": "sweeps/s", "cores": best, "kind": "port",
                                         "threads_tried_ms_per_solve": {str(k): round(v[0] * 1e3, 2) for k, v in tried.items()},
                                         "sample": f"{na} full solves of the same sweep and map; the oracle's keypoint loop visited in parallel "
                                                   f"(OpenMP, {best} threads = the fastest of those tried on this {ncores}-core host), committed in keypoint "
                                                   f"order: results bit-identical to the single-thread run ({bool(np.array_equal(oa['state'], ou['state']))})",
                                         "ms_per_solve": cpu_all * 1e3, "ms_per_esikf_iter": cpu_all * 1e3 / max(oa["rc"], 1),
                                         "speedup_over_1_core": cpu_s / cpu_all}
        out["parity"] = {"state_rel_err_vs_oracle": state_err, "iterations_gpu": iters, "iterations_oracle": ou["rc"],
                         "residuals_gpu": r["num_residuals"], "residuals_oracle": ou["num_residuals"]}
        # every OTHER sweep of the timed stream against the oracle as well (OpenMP keypoint loop: bit-identical to the single-thread run)
        worst, ok_all = 0.0, True
        for j in sorted(stream_states):
            if j == 0:
                continue
            e = stream[j]
            with po.threads(best):
                eo = po.Eskf(backend)
                eo.set_state(e["prior_state"]); eo.set_cov(prior_cov)
                oj = po.update_iekf(omap, eo, oo, e["sweep"]["raw"], e["state0"], e["sweep"]["t_last"], frame_id=args.frame_id)
            gj = stream_states[j]
            worst = max(worst, rel(gj[2], oj["state"]))
            ok_all = ok_all and gj[0] == oj["rc"] and gj[1] == oj["num_residuals"]
        out["parity"]["stream_sweeps_checked"] = len(stream_states)
        out["parity"]["stream_state_rel_err_vs_oracle_max"] = max(worst, state_err if 0 in stream_states else 0.0)
        out["parity"]["stream_counts_equal"] = bool(ok_all)
        # the reference's OWN translation units (oracle/_ref/libref_path.so = /root/reference/src/optimize.cpp & co. compiled in
        # place against stand-in third-party headers; prebuilt, travels with the tree): the same solve through
        # lioOptimization::updateIEKF as the reference wrote it.  Checker + baseline only.
        try:
            from oracle import pyref as pr
            if pr.available():
                rmap = pr.Map.from_oracle(omap)
                rtimes, ru, re_ = [], None, None
                t_start = time.perf_counter()
                while len(rtimes) < 3 and (time.perf_counter() - t_start) < 12.0:
                    re_ = pr.Eskf()
                    re_.set_state(prior_state); re_.set_cov(prior_cov)
                    tc = time.perf_counter()
                    ru = pr.update_iekf(rmap, re_, oo, sweep["raw"], state0, sweep["t_last"], frame_id=args.frame_id)
                    rtimes.append(time.perf_counter() - tc)
                ref_s = float(np.median(rtimes))
                out["cpu_baseline_reference_tu"] = {
                    "value": 1.0 / ref_s, "unit": "sweeps/s", "cores": 1, "kind": "reference",
                    "sample": f"{len(rtimes)} full solves of the same sweep and map through the reference's own lioOptimization::updateIEKF "
                              f"(src/optimize.cpp compiled in place; third-party arithmetic = the stand-in Eigen of oracle/ref_shim, so this is "
                              f"not an Eigen-vectorised build); single thread",
                    "note": "stand-in Eigen, eager (un-vectorised): overstates the cost of the reference with real Eigen",
                    "ms_per_solve": ref_s * 1e3}
                out["parity"]["state_rel_err_vs_reference_tu"] = rel(r["state"], ru["state"])
                out["parity"]["oracle_equals_reference_tu_bitwise"] = bool(np.array_equal(ou["state"], ru["state"]) and
                                                                           np.array_equal(re_.get_cov(), eo_last_cov(po, backend, omap, oo, prior_state, prior_cov, sweep, state0, args.frame_id)))
                out["parity"]["residuals_reference_tu"] = ru["num_residuals"]
                del rmap
                # the reference's own code is the baseline of record where its library travelled with the tree; the oracle
                # restatement (bitwise equal to it) stays beside it as the port
                out["cpu_baseline_port"] = out["cpu_baseline"]
                out["cpu_baseline"] = dict(out["cpu_baseline_reference_tu"])
        except Exception as e:  # noqa: BLE001
            out["cpu_baseline_reference_tu"] = {"error": repr(e)}
        del omap
    try:
        lio.ctx.disarm(); torch.cuda.synchronize()
    except Exception:  # noqa: BLE001
        pass
    streamer.close()
    lio.close()

    # ---------------- every BASELINE configuration + shipped setting + init mode (rank 0, N = 1 only)
    if rank == 0 and world == 1 and not args.no_configs:
        backend = None
        threads = 1
        if po is not None:
            backend = "tsl" if os.path.exists(po.LIB_TSL) else "plain"
            threads = min(os.cpu_count() or 1, 64)
        del cands
        # small configurations get enough solves that the timed region spans tens of milliseconds
        # every configuration twice -- max_num_residuals = INT_MAX (throughput) and = 600 (config/r3live.yaml:69, the shipped
        # value: ordered cut); C1 is the plumbing scale, C4 the 8-GPU configuration on one GPU
        plan = [("C1", "C1", INT_MAX, 100, 200), ("C2", "C2", INT_MAX, 100, 200), ("C3", "C3", INT_MAX, 100, 200),
                ("C4", "C4", INT_MAX, 100, 20), ("HEADLINE@600", "HEADLINE", 600, 100, 200), ("C2@600", "C2", 600, 100, 200),
                ("C3@600", "C3", 600, 100, 200), ("INIT(frame_id=5)", "HEADLINE", INT_MAX, 5, 20)]
        cfgs = []
        for name, wl, mr, fid, st in plan:
            try:
                cfgs.append(run_config(name, wl, mr, fid, st, 2, local_rank, po, backend, threads, stream_sweeps=2 if st <= 20 else 4))
            except Exception as e:  # noqa: BLE001
                cfgs.append({"name": name, "error": repr(e)})
        out["configs"] = cfgs
        try:
            pl = run_pipeline(local_rank)
            out["pipeline_detail"] = pl
            out["pipeline"] = {"what": "frames/s (wall time of back-to-back frames) of upload + device keypoint selection + two passes + device commit, 1M-pt map; "
                                       "us = median host time per stage (the map insertion is enqueued by commit and runs on under the next frame's upload/select)",
                               "frames": [{"points": e["frame_points"], "keypoints": e["keypoints"], "frames_per_s": e["frames_per_s"],
                                           "us": e["us"]} for e in pl]}
        except Exception as e:  # noqa: BLE001
            out["pipeline"] = {"error": repr(e)[:160]}
    if rank == 0:
        write_detail(out)
        print(compact_line(out), flush=True)
    if dist is not None:
        with c_stdout_to_stderr():
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
