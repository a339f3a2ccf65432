"""Parity report on the GPU box: GPU pass vs live oracle at a given size; prints max relative errors."""
import sys, numpy as np
sys.path.insert(0, '.')
import sr_livo_amd as srl
from sr_livo_amd import capi, synth
from oracle import pyoracle as po
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 0
pts, L = synth.map_candidates(31, 200_000)
m = po.Map('plain'); m.add_points(pts)
sw = synth.make_sweep(32, n, L)
ctx = srl.Context(0); ctx.map_upload(*m.export())
opts = srl.default_opts(max_num_residuals=2**31 - 1, select_mode=mode)
ctx.sweep_upload(sw['raw']); ctx.set_taps(1)
neq, rc = ctx.build_residuals(capi.make_frame(sw['q_pred'], sw['t_pred'], sw['t_last']), opts)
ids, status, ncand = ctx.fetch_neighbors(); res = ctx.fetch_residuals()
o = m.build_plane_residuals(po.default_opts(max_num_residuals=2**31 - 1), sw['raw'], sw['q_pred'], sw['t_pred'], sw['t_last'])
def rel(a, b): return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))
print("n", n, "mode", mode, "status equal", np.array_equal(status, o['status']), "ids equal", np.array_equal(ids, o['ids']), "ties", o['neq'].num_ties, "fallback", neq.num_fallback)
hp = o['status'] >= 1
for k in ('normal', 'a2D', 'weight', 'norm_offset', 'distance'):
    d = np.abs(res[k][hp] - o[k][hp]); print(k, "max abs", d.max(), "rel", rel(res[k][hp], o[k][hp]))
acc = o['status'] == 2
print("J rel", rel(res['jacobian'][acc], o['jacobian'][acc]), "HtH rel", rel(np.array(neq.HtH).reshape(6, 6), o['HtH']), "Hth rel", rel(np.array(neq.Hth), o['Hth']))
d_g = res['distance'][hp]; d_o = o['distance'][hp]
pr = np.abs(d_g - d_o) / np.maximum(np.abs(d_o), 1e-12)
print("worst per-residual rel err of distance", pr.max(), "at |d|=", np.abs(d_o[pr.argmax()]), " frac > 1e-9:", (pr > 1e-9).mean(), " frac > 1e-5:", (pr > 1e-5).mean())
