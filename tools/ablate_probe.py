"""Phase breakdown of the association kernel by ablation (tuning experiment; GPU box):
    python tools/ablate_probe.py [WORKLOAD ...]   ->  kernel us with parts of the kernel switched off
bits (srl_kernels.hip): 1 no plane fit (phase 2), 2 no FP64 finish, 4 no selection, 8 no probe_finish, 16 empty kernel,
32 no probe issue, 64 return after phase 1, 256 keypoint pairs selected one after the other (no paired loads)."""
import sys
sys.path.insert(0, ".")
import sr_livo_amd as srl
from sr_livo_amd import capi, synth

import os
BITS = [int(x) for x in os.environ["BITS"].split(",")] if os.environ.get("BITS") else (0, 1, 2, 2 | 1, 4, 4 | 1, 8 | 4 | 1, 32 | 8 | 4 | 1, 64, 64 | 4, 64 | 32 | 8 | 4, 16)
for wl in (sys.argv[1:] or ["HEADLINE"]):
    n_kp, map_pts, pattern, seed = synth.CONFIGS[wl]
    cands, L = synth.map_candidates(seed, map_pts)
    sw = synth.make_sweep(seed + 1000, n_kp, L, pattern=pattern)
    ctx = srl.Context(0)
    ctx.map_insert(cands)
    ctx.sweep_upload(sw["raw"])
    f = capi.make_frame(sw["q_pred"], sw["t_pred"], sw["t_last"])
    opts = srl.default_opts(max_num_residuals=2**31 - 1)
    ctx.set_fused_reduce(0)
    out = {}
    for bits in BITS:
        ctx.lib.srl_debug_set_ablate(ctx.h, bits)
        for _ in range(5):
            ctx.build_residuals(f, opts)
        ctx.set_profiling(1)
        for _ in range(30):
            ctx.build_residuals(f, opts)
        t = ctx.timing(); ctx.set_profiling(0)
        out[bits] = round(t.sum_assoc_ms / t.calls * 1e3, 2)
        print(wl, "ablate", bits, "assoc_us", out[bits], flush=True)
    ctx.lib.srl_debug_set_ablate(ctx.h, 0)
    ctx.close()
