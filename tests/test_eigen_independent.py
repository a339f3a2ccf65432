"""Independent check of the two pieces of Eigen the oracle and the compiled reference SHARE (oracle/orc_eigen337.h: both libraries
reach Eigen 3.3.7's SelfAdjointEigenSolver<Matrix3d> and Matrix<17,17>::inverse() through the same restatement, so their bitwise
agreement cannot see an error in it) -- on the matrices of the benchmark's own workloads, against code that shares nothing with it:

* src/optimize.cpp:339-346 -- every scatter matrix of the HEADLINE sweep (65 536 keypoints x 20 neighbours, 978 k-point map): eigenvalues,
  the eigenvector of the smallest one wherever it is separated (gap > 1e-6 of the largest), and a2D, against LAPACK (numpy.linalg.eigh);
  and the oracle's end-to-end normal / a2D of the same pass against the LAPACK-derived ones.
* src/optimize.cpp:234-237 -- for every BASELINE configuration the two 17 x 17 matrices the first ESIKF iteration inverts,
  covariance / laser_point_cov and its inverse + H^T H (the projection of :220-232 is the identity in that iteration: the state still
  equals the prediction), against a Gauss-Jordan elimination in 80-bit extended precision.
CPU only.
"""
import numpy as np
import pytest

from oracle import pyoracle as po
from sr_livo_amd import synth

INT_MAX = 2**31 - 1


def _scene(workload):
    n_kp, map_pts, pattern, seed = synth.CONFIGS[workload]
    cands, L = synth.map_candidates(seed, map_pts)
    sweep = synth.make_sweep(seed + 1000, n_kp, L, pattern=pattern)
    omap = po.Map("plain")
    omap.add_points(cands)
    return omap, sweep


def _prior(sweep):
    eo = po.Eskf("plain")
    synth.eskf_prior(eo, sweep["q_pred"], sweep["t_pred"], sweep["vel"])
    return eo


@pytest.fixture(scope="module")
def headline():
    return _scene("HEADLINE")


def test_eigen_solver_on_every_scatter_matrix_of_the_headline_sweep(headline):
    omap, sweep = headline
    opts = po.default_opts(max_num_residuals=INT_MAX)
    with po.threads(8):
        res = omap.build_plane_residuals(opts, sweep["raw"], sweep["q_pred"], sweep["t_pred"], sweep["t_last"])
    keys, counts, xyz = omap.export()
    flat = xyz.reshape(-1, 3).astype(np.float64)                       # point id = voxel * 20 + slot (creation order, like the device map)
    fit = res["status"] >= 1                                           # keypoints that reached computeNeighborhoodDistribution
    assert fit.sum() > 60_000
    nb = flat[res["ids"][fit]]                                         # (M, 20, 3)
    # scatter matrices in extended precision, rounded once: the SAME matrix goes to both solvers
    nbl = nb.astype(np.longdouble)
    d = nbl - nbl.mean(axis=1, keepdims=True)
    Cm = np.einsum("mki,mkj->mij", d, d).astype(np.float64)
    lam_ref, vec_ref = np.linalg.eigh(Cm)                              # LAPACK: ascending eigenvalues, orthonormal columns
    lib = po.load("plain")
    lam = np.empty_like(lam_ref); v0 = np.empty((len(Cm), 3))
    ev = np.empty(3); V = np.empty(9)
    for i in range(len(Cm)):
        rc = lib.orc_eig3_solver(po.EIG_EIGEN_QL, po._dp(np.ascontiguousarray(Cm[i]).reshape(9)), po._dp(ev), po._dp(V))
        assert rc == 0
        lam[i] = ev; v0[i] = V.reshape(3, 3)[:, 0]
    scale = np.abs(lam_ref).max(axis=1)
    assert np.max(np.abs(lam - lam_ref) / scale[:, None]) < 1e-12
    gap = (lam_ref[:, 1] - lam_ref[:, 0]) / scale
    sep = gap > 1e-6
    assert sep.mean() > 0.99                                            # planar neighbourhoods: the normal is determined almost everywhere
    cosang = np.abs(np.einsum("mi,mi->m", v0[sep], vec_ref[sep][:, :, 0]))
    assert np.max(1.0 - cosang) < 1e-12                                 # same direction (sign is arbitrary, fixed by optimize.cpp:49-51)
    sig = np.sqrt(np.abs(lam)); sig_ref = np.sqrt(np.abs(lam_ref))
    a2d = (sig[:, 1] - sig[:, 0]) / sig[:, 2]; a2d_ref = (sig_ref[:, 1] - sig_ref[:, 0]) / sig_ref[:, 2]
    assert np.max(np.abs(a2d - a2d_ref)) < 1e-10
    # the oracle's own pass (its sequential barycentre / scatter sums + the solver + the re-normalisation) against the LAPACK-derived plane
    a2d_pass = res["a2D"][fit]
    assert np.max(np.abs(a2d_pass - a2d_ref)) < 1e-9
    n_pass = res["normal"][fit][sep]
    assert np.max(1.0 - np.abs(np.einsum("mi,mi->m", n_pass, vec_ref[sep][:, :, 0]))) < 1e-12


def test_eigen_solver_on_line_like_neighbourhoods():
    """synth.cable_scene: a fifth of the keypoints have 20 COLLINEAR neighbours (eigenvalues 0.75 / 4e-8 / 3e-8).  The device's closed
    form could not separate the two small ones (round 5: state 1.5e-4 off; it now runs Jacobi sweeps there) -- this checks what it is
    compared WITH: the Eigen restatement shared by oracle and compiled reference must resolve that near-null plane the way LAPACK does,
    i.e. the reference's behaviour on such keypoints is the solver's, not an artefact of the stand-in."""
    pts, sw = synth.cable_scene(77)
    omap = po.Map("plain")
    omap.add_points(pts)
    res = omap.build_plane_residuals(po.default_opts(max_num_residuals=INT_MAX), sw["raw"], sw["q_pred"], sw["t_pred"], sw["t_last"])
    keys, counts, xyz = omap.export()
    flat = xyz.reshape(-1, 3).astype(np.float64)
    line = (res["status"] >= 1) & (res["a2D"] < 0.02)
    assert line.sum() > 1000
    nbl = flat[res["ids"][line]].astype(np.longdouble)
    d = nbl - nbl.mean(axis=1, keepdims=True)
    Cm = np.einsum("mki,mkj->mij", d, d).astype(np.float64)
    lam_ref, vec_ref = np.linalg.eigh(Cm)
    assert np.median(lam_ref[:, 1] / lam_ref[:, 2]) < 1e-6 and np.median((lam_ref[:, 1] - lam_ref[:, 0]) / lam_ref[:, 2]) < 1e-6      # line-like indeed
    lib = po.load("plain")
    ev = np.empty(3); V = np.empty(9)
    worst_vec, worst_a2d = 0.0, 0.0
    for i in range(len(Cm)):
        for solver in (po.EIG_EIGEN_QL, po.EIG_JACOBI):
            rc = lib.orc_eig3_solver(solver, po._dp(np.ascontiguousarray(Cm[i]).reshape(9)), po._dp(ev), po._dp(V))
            assert rc == 0
            rel_gap = (lam_ref[i, 1] - lam_ref[i, 0]) / lam_ref[i, 2]
            # eigenvector of the smallest eigenvalue: determined up to (solver error ~ 1e-16) / (relative gap)
            worst_vec = max(worst_vec, (1.0 - abs(float(V.reshape(3, 3)[:, 0] @ vec_ref[i][:, 0]))) * rel_gap ** 2)
            sig = np.sqrt(np.abs(ev)); sr = np.sqrt(np.abs(lam_ref[i]))
            worst_a2d = max(worst_a2d, abs((sig[1] - sig[0]) / sig[2] - (sr[1] - sr[0]) / sr[2]))
    assert worst_a2d < 1e-9, worst_a2d
    assert worst_vec < 1e-26, worst_vec                 # 1 - cos ~ angle^2 / 2 with angle <~ 1e-13 / gap
    # the oracle's pass itself against the LAPACK-derived a2D
    a2d_ref = (np.sqrt(np.abs(lam_ref[:, 1])) - np.sqrt(np.abs(lam_ref[:, 0]))) / np.sqrt(np.abs(lam_ref[:, 2]))
    assert np.max(np.abs(res["a2D"][line] - a2d_ref)) < 1e-8


def _inv_longdouble(A):
    """Gauss-Jordan with partial pivoting in 80-bit extended precision (numpy.longdouble)"""
    n = len(A)
    M = np.concatenate([A.astype(np.longdouble), np.eye(n, dtype=np.longdouble)], axis=1)
    for c in range(n):
        p = c + int(np.argmax(np.abs(M[c:, c])))
        if p != c:
            M[[c, p]] = M[[p, c]]
        M[c] = M[c] / M[c, c]
        for r in range(n):
            if r != c:
                M[r] = M[r] - M[r, c] * M[c]
    return M[:, n:]


@pytest.mark.skipif(np.finfo(np.longdouble).eps > 1e-18, reason="numpy.longdouble is not extended precision on this platform")
@pytest.mark.parametrize("workload,max_res,frame_id", [("C1", INT_MAX, 100), ("C2", INT_MAX, 100), ("C3", INT_MAX, 100), ("HEADLINE", INT_MAX, 100),
                                                        ("HEADLINE", 600, 100), ("HEADLINE", INT_MAX, 5)])
def test_the_17x17_inverses_of_the_first_esikf_iteration(workload, max_res, frame_id, headline):
    omap, sweep = headline if workload == "HEADLINE" else _scene(workload)
    eo = _prior(sweep)
    P0 = eo.get_cov().copy()
    st = np.concatenate([sweep["q_pred"], sweep["t_pred"], sweep["vel"], np.zeros(6)])
    with po.threads(8):
        u = po.update_iekf(omap, eo, po.default_opts(max_num_residuals=max_res), sweep["raw"], st, sweep["t_last"], frame_id=frame_id, log_iters=2)
    assert u["rc"] >= 1
    HtH = u["log"][0, :36].reshape(6, 6); Hth = u["log"][0, 36:42]
    lib = po.load("plain")

    def inv17(A):
        out = np.empty(289)
        assert lib.orc_inverse17(po._dp(np.ascontiguousarray(A).reshape(289)), po._dp(out)) == 0
        return out.reshape(17, 17)

    A1 = P0 / 0.001                                                     # covariance / laser_point_cov (optimize.cpp:234, lioOptimization.cpp:364)
    T = inv17(A1)
    T_ld = _inv_longdouble(A1)
    cond1 = np.linalg.cond(A1)
    err1 = float(np.max(np.abs(T - T_ld.astype(np.float64))) / np.max(np.abs(T)))
    A2 = T.copy(); A2[:6, :6] += HtH                                    # optimize.cpp:235-236
    S = inv17(A2)
    S_ld = _inv_longdouble(A2)
    cond2 = np.linalg.cond(A2)
    err2 = float(np.max(np.abs(S - S_ld.astype(np.float64))) / np.max(np.abs(S)))
    # what the update reads (optimize.cpp:237-244): the first six columns, and the gain applied to H^T h
    Kh = S[:, :6] @ Hth; Kh_ld = (S_ld[:, :6] @ Hth.astype(np.longdouble)).astype(np.float64)
    errk = float(np.max(np.abs(Kh - Kh_ld)) / max(np.max(np.abs(Kh_ld)), 1e-300))
    print(f"{workload} max_res={max_res} frame_id={frame_id}: cond(P/R) {cond1:.2e} err {err1:.2e}; cond(T + HtH) {cond2:.2e} err {err2:.2e}; K_h err {errk:.2e}")
    # partial-pivot LU in double is backward stable: the error is bounded by the conditioning, cond x 2^-52 with a small constant
    assert err1 < 64 * cond1 * 2.3e-16 and err2 < 64 * cond2 * 2.3e-16
    assert err1 < 1e-10 and err2 < 1e-10 and errk < 1e-10
