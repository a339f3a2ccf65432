"""What the committed rocprofv3 summaries (profiles/<round>_<config>_rocprofv3_summary.json) say about the association kernel: counter-measured
HBM traffic, the instruction-issue roofline, the compulsory floor.  bench.py quotes them only when their stamp equals the kernel sources'."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

HBM_PEAK_GBS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)
CLOCK_HZ = 2.4e9          # max engine clock (same guide)
N_CU, N_SIMD = 256, 1024
INT_MAX = 2**31 - 1
PROFILE_ROUND = "r06"    # the committed rocprofv3 summaries the line may quote: profiles/<round>_<config>_rocprofv3_summary.json
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


PROFILE_TAG = {"C1": "c1", "C2": "c2", "C3": "c3", "C4": "c4", "HEADLINE@600": "headline600", "C2@600": "c2_600", "C3@600": "c3_600", "INIT(frame_id=5)": "init", "SPREAD": "spread"}


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

