"""Soak test of the fused final reduction (GPU box): many back-to-back launches at alternating sweep sizes; every result must
equal the first one of its size bit for bit and the separate-reduce result up to summation order."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import sr_livo_amd as srl
from sr_livo_amd import capi, synth

pts, L = synth.map_candidates(31, 300_000)
ctx = srl.Context(0)
ctx.map_insert(pts)
opts = srl.default_opts(max_num_residuals=2**31 - 1)
sizes = [65536, 2048, 16384, 40000, 24576, 3000, 131072]
sweeps = {n: synth.make_sweep(100 + n, n, L) for n in sizes}
ref, first = {}, {}
for n in sizes:
    sw = sweeps[n]
    ctx.sweep_upload(sw["raw"])
    f = capi.make_frame(sw["q_pred"], sw["t_pred"], sw["t_last"])
    ctx.set_fused_reduce(0); ref[n] = np.array(ctx.build_residuals(f, opts)[0].HtH)
    ctx.set_fused_reduce(1); first[n] = np.array(ctx.build_residuals(f, opts)[0].HtH)
    assert np.max(np.abs(first[n] - ref[n])) <= 1e-12 * np.max(np.abs(ref[n])), n
t0 = time.time(); launches = 0
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for r in range(rounds):
    for n in sizes:
        sw = sweeps[n]
        ctx.sweep_upload(sw["raw"])
        f = capi.make_frame(sw["q_pred"], sw["t_pred"], sw["t_last"])
        for _ in range(5):
            h = np.array(ctx.build_residuals(f, opts)[0].HtH)
            launches += 1
            if not np.array_equal(h, first[n]):
                raise SystemExit(f"MISMATCH at round {r} size {n}")
st = ctx.arm_stats()
print(f"soak ok: {launches} fused launches over {len(sizes)} sweep sizes in {time.time() - t0:.1f} s, every result bitwise equal to the first of its size; "
      f"armed launches: {st}")
