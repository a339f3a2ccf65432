"""Instruction counts per section of the association kernel: run under
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU --output-format csv -d DIR -o pmc -- python tools/ablate_pmc.py
then python tools/ablate_pmc.py --report DIR : the launches are grouped by ablation setting (3 per setting, in BITS order)."""
import sys, os, glob, csv, re
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
BITS = (0, 1, 2 | 1, 4 | 1, 8 | 4 | 1, 32 | 8 | 4 | 1, 64 | 32 | 8 | 4, 16)
NAMES = {0: "full", 1: "no phase 2", 3: "no FP64 finish, no phase 2", 5: "no selection (probes only), no phase 2", 13: "no probe_finish either", 45: "no probe issue either",
         108: "phase 0 + pair-loop skeleton", 16: "empty kernel"}
REPS = 3
if len(sys.argv) > 2 and sys.argv[1] == "--report":
    rows = []
    for f in glob.glob(sys.argv[2] + "/**/*counter_collection.csv", recursive=True):
        rows += [r for r in csv.DictReader(open(f)) if "srl_assoc_kernel" in r["Kernel_Name"]]
    disp = sorted({int(r["Dispatch_Id"]) for r in rows})
    per = {d: {} for d in disp}
    for r in rows:
        per[int(r["Dispatch_Id"])][r["Counter_Name"]] = per[int(r["Dispatch_Id"])].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    n = 65536
    prev = None
    for i, b in enumerate(BITS):
        ds = disp[i * REPS:(i + 1) * REPS]
        c = {k: sum(per[d].get(k, 0.0) for d in ds) / len(ds) for k in per[ds[0]]}
        keys = sorted(c)
        line = f"{NAMES[b]:42s} " + "  ".join(f"{k.replace('SQ_', '')} {c[k] / n:8.2f}" for k in keys) + "  per keypoint"
        if prev:
            line += "   | removed vs previous row: " + " ".join(f"{k.replace('SQ_', '')} {(prev.get(k, 0) - c[k]) / n:7.2f}" for k in keys)
        print(line)
        prev = c
    sys.exit(0)
import sr_livo_amd as srl
from sr_livo_amd import capi, synth
n_kp, map_pts, pattern, seed = synth.CONFIGS["HEADLINE"]
cands, L = synth.map_candidates(seed, map_pts)
sw = synth.make_sweep(seed + 1000, n_kp, L, pattern=pattern)
ctx = srl.Context(0)
ctx.map_insert(cands)
ctx.sweep_upload(sw["raw"])
f = capi.make_frame(sw["q_pred"], sw["t_pred"], sw["t_last"])
opts = srl.default_opts(max_num_residuals=2**31 - 1)
ctx.set_fused_reduce(0)
for bits in BITS:
    ctx.lib.srl_debug_set_ablate(ctx.h, bits)
    for _ in range(REPS):
        ctx.build_residuals(f, opts)
ctx.close()
