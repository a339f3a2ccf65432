"""Soak test of the frame pipeline (GPU box): frames of five sizes in a long back-to-back loop -- page-locked upload on the copy stream,
keypoint selection (grouping, first-occurrence ranks and the std::tr1::unordered_map iteration order on the device; the host waits on a
tagged word for the count and fetches the index list), one pass, deferred commit (own radix passes, re-transform fused into the first
insertion kernel) with the world points coming back on the copy stream.  Every frame's keypoint list and world points must equal the first ones of its size bit for bit
(they depend on the frame and the pose only, not on the map), the map after the first CHECK frames must equal the map a second context
builds with the synchronous form, and the loop is long enough for the scratch tables' 16-bit epoch to wrap (65 535 frames) at least once
when run with the default count.

    python tools/soak_pipeline.py [frames]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import sr_livo_amd as srl
from sr_livo_amd import capi, synth

FRAMES = int(sys.argv[1]) if len(sys.argv) > 1 else 140_000
CHECK = 400
pts, L = synth.map_candidates(41, 200_000)
sizes = [24_000, 3_000, 9_000, 700, 16_000]
sweeps = [synth.make_sweep(500 + n, n, L) for n in sizes]
pins = [srl.PinnedArray((n, 3)) for n in sizes]
worlds = [srl.PinnedArray((n, 3)) for n in sizes]
for p, sw in zip(pins, sweeps):
    p.array[:] = sw["raw"]
opts = srl.default_opts(max_num_residuals=2**31 - 1)
a = srl.Context(0); b = srl.Context(0)
for c in (a, b):
    c.map_insert(pts[:50_000])


def frame(c, i, deferred):
    sw = sweeps[i]
    c.frame_upload(pins[i].array)
    kp = c.frame_select_keypoints(sw["q_pred"], sw["t_pred"], 1.0)
    f = capi.make_frame(sw["q_pred"], sw["t_pred"], sw["t_last"])
    c.build_residuals(f, opts)
    if deferred:
        w, _ = c.frame_commit(sw["q_gt"], sw["t_gt"], want_added=False, world_out=worlds[i].array)
    else:
        w, _ = c.frame_commit(sw["q_gt"], sw["t_gt"])
    return kp, w


ref = [None] * len(sizes)
t0 = time.time()
for k in range(FRAMES):
    i = k % len(sizes)
    kp, w = frame(a, i, True)
    if ref[i] is None:
        ref[i] = (kp.copy(), w.copy())
    elif not (np.array_equal(kp, ref[i][0]) and np.array_equal(w, ref[i][1])):
        raise SystemExit(f"MISMATCH at frame {k} (size {sizes[i]}): keypoints equal {np.array_equal(kp, ref[i][0])}, world equal {np.array_equal(w, ref[i][1])}")
    if k == CHECK - 1:
        for j in range(CHECK):
            kb, wb = frame(b, j % len(sizes), False)
            assert np.array_equal(kb, ref[j % len(sizes)][0]) and np.array_equal(wb, ref[j % len(sizes)][1])
        ma, mb = a.map_download(), b.map_download()
        assert a.map_size() == b.map_size() and all(np.array_equal(x, y) for x, y in zip(ma, mb)), "deferred and synchronous maps differ"
el = time.time() - t0
print(f"pipeline soak ok: {FRAMES} frames of {len(sizes)} sizes in {el:.1f} s ({FRAMES / el:.0f} frames/s incl. the Python loop and the comparisons), every keypoint "
      f"list and world array bitwise equal to the first of its size; map after {CHECK} frames equal to the synchronous form's; epoch wraps crossed: {FRAMES // 65535}; "
      f"final map {a.map_size()}")
for p in pins + worlds:
    p.close()
a.close(); b.close()
