"""The ONE stdout line of the bench contract (< 8 KB, strict JSON) cut from the full result dictionary; the rest goes to gpurun_out/bench_detail.json."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

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
    roof.update(_pick(r, ("kernel", "avg_launch_ms", "launches", "measured_over", "event_period", "algorithmic_bytes_per_launch", "compulsory_bytes_per_launch",
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
        line["stream"] = _pick(out["stream"], ("sweeps", "solves", "sweeps_per_s_mean", "sweeps_per_s_median", "value_over_long_mean", "arm_stats",
                                               "state_of_sweep0_equals_resident_solve"))
        if isinstance(out["stream"].get("unfiltered"), dict):
            line["stream"]["unfiltered"] = _pick(out["stream"]["unfiltered"], ("sweeps", "solves", "sweeps_per_s", "us_per_esikf_iter", "esikf_iterations_by_sweep"))
    if isinstance(out.get("arm_stats_timed_region"), dict):
        line["arm_stats"] = out["arm_stats_timed_region"]
    if isinstance(out.get("resident_resolve"), dict):
        line["resident_resolve"] = _pick(out["resident_resolve"], ("sweeps_per_s", "us_per_esikf_iter"))
    if isinstance(out.get("clock_warmup"), dict):
        line["clock_warmup"] = out["clock_warmup"]          # untimed solves before the W warm-up steps (steady clocks): disclosed in the line
    for k in ("launch_ab", "pipeline", "comm", "sharded_config", "aux_independent_sweeps_per_s", "multi_gpu_note", "fallback"):
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
        if c["name"] == "SPREAD":          # the off-cache leg: what HBM and the L2 did (committed counter passes of this workload)
            pr = c.get("profile") or {}
            cfgs[-1].update(_pick(pr, ("l2_hit_rate", "hbm_measured_GBs", "traffic_bytes_per_launch")), algorithmic_MB=c.get("algorithmic_MB_per_launch"),
                            probed_voxels=c.get("probed_voxels"))
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

