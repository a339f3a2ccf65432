"""SelfAdjointEigenSolver<Matrix3d> (optimize.cpp:339) in the oracle: the restatement of Eigen 3.3.7's own algorithm
(tridiagonalisation + implicit symmetric QR; oracle `eig3_eigen_ql`, the default) against LAPACK (numpy.linalg.eigh) and
against the independent FP64 cyclic Jacobi solver, on random and on NEAR-DEGENERATE neighbourhoods
(lambda0 ~ lambda1, exact planes, collinear points, duplicated points) -- the cases where the Eigen boundary matters:
sigma_3 = sqrt(|lambda_0|) carries an absolute error ~ sqrt(eps ||A||) on an exact plane and the normal of a
rotationally symmetric neighbourhood is decided by rounding.  CPU only.
"""
import numpy as np
import pytest

from oracle import pyoracle as po


def scatter(P):
    b = P.sum(0) / len(P)
    E = P - b
    return E.T @ E


def a2d_of(ev):
    s1, s2, s3 = np.sqrt(np.abs(ev[2])), np.sqrt(np.abs(ev[1])), np.sqrt(np.abs(ev[0]))
    return (s2 - s3) / s1


def check_decomposition(A, ev, V, tol=4e-15):
    nrm = max(np.abs(A).max(), 1e-300)
    assert np.all(np.diff(ev) >= 0), "eigenvalues ascending"
    assert np.abs(V.T @ V - np.eye(3)).max() < 1e-14, "orthonormal eigenvectors"
    assert np.abs(A @ V - V * ev).max() <= 32 * tol * nrm, "A V = V diag(ev)"


def test_ql_matches_lapack_on_random_neighbourhoods():
    rng = np.random.default_rng(2)
    for _ in range(2000):
        P = rng.normal(size=(20, 3)) * rng.uniform(0.01, 2.0, 3) + rng.uniform(-50, 50, 3)
        A = scatter(P)
        ev, V, ok = po.eig3(A, po.EIG_EIGEN_QL)
        assert ok
        check_decomposition(A, ev, V)
        w, U = np.linalg.eigh(A)
        assert np.abs(ev - w).max() <= 1e-14 * w[2]
        gap = min(w[1] - w[0], w[2] - w[1])
        for c in range(3):
            assert abs(abs(V[:, c] @ U[:, c]) - 1.0) < 1e-13 * w[2] / gap + 1e-14
        ej, Vj, _ = po.eig3(A, po.EIG_JACOBI)
        assert np.abs(ev - ej).max() <= 1e-14 * w[2]


def test_ql_structural_cases():
    # diagonal input: no rotation at all, sorted ascending with the eigenvectors permuted accordingly
    ev, V, ok = po.eig3(np.diag([3.0, 1.0, 2.0]))
    assert ok and list(ev) == [1.0, 2.0, 3.0]
    assert np.array_equal(np.abs(V), np.array([[0, 0, 1], [1, 0, 0], [0, 1, 0]], float))
    # zero matrix: scale falls back to 1, eigenvalues 0, identity eigenvectors
    ev, V, ok = po.eig3(np.zeros((3, 3)))
    assert ok and np.all(ev == 0) and np.array_equal(V, np.eye(3))
    # a(2,0) == 0: already tridiagonal, the Householder step is skipped (Tridiagonalization.h 3x3 specialisation)
    A = np.array([[2.0, 1.0, 0.0], [1.0, 3.0, 0.5], [0.0, 0.5, 1.0]])
    ev, V, ok = po.eig3(A)
    check_decomposition(A, ev, V)
    assert np.abs(ev - np.linalg.eigvalsh(A)).max() < 1e-15 * 4
    # only the LOWER triangle is read (mat = matrix.triangularView<Lower>())
    B = A.copy(); B[0, 1] = 77.0; B[0, 2] = -5.0; B[1, 2] = 9.0
    ev2, V2, _ = po.eig3(B)
    assert np.array_equal(ev, ev2) and np.array_equal(V, V2)
    # huge and tiny scales go through the max-|coeff| scaling unharmed
    for s in (1e-200, 1e200):
        evs, Vs, ok = po.eig3(A * s)
        assert ok and np.abs(evs / s - ev).max() < 1e-14 * 4


def _stress_sets(rng):
    """(name, points[20,3]) near-degenerate neighbourhoods as the voxel map produces them (f32 coordinates)."""
    out = []
    for i in range(60):
        c = rng.uniform(-80, 80, 3)
        R = np.linalg.qr(rng.normal(size=(3, 3)))[0]
        uv = rng.uniform(-0.5, 0.5, (20, 2))
        # exact plane z = const in world axes: lambda_0 == 0 up to the rounding of the sums
        out.append(("axis_plane", np.c_[uv + c[:2], np.full(20, np.float32(c[2]))].astype(np.float32).astype(np.float64)))
        # tilted plane rounded to f32: lambda_0 ~ (f32 ulp)^2
        out.append(("tilted_plane", (np.c_[uv, np.zeros(20)] @ R.T + c).astype(np.float32).astype(np.float64)))
        # isotropic disc: lambda_1 ~ lambda_2 (in-plane symmetric) -- normal well defined
        ang = np.arange(20) * (2 * np.pi / 20)
        out.append(("ring", (np.c_[np.cos(ang), np.sin(ang), rng.normal(0, 0.02, 20)] * 0.4 @ R.T + c).astype(np.float32).astype(np.float64)))
        # collinear points: lambda_0 ~ lambda_1 ~ 0, the normal is arbitrary in the plane orthogonal to the line
        t = rng.uniform(-0.5, 0.5, 20)
        out.append(("line", (np.outer(t, R[:, 0]) + c).astype(np.float32).astype(np.float64)))
        # rod: thin isotropic cross-section, lambda_0 ~ lambda_1 != 0
        out.append(("rod", (np.outer(t, R[:, 0]) + rng.normal(0, 0.01, (20, 3)) + c).astype(np.float32).astype(np.float64)))
        # isotropic blob: all three eigenvalues close
        out.append(("blob", (rng.normal(0, 0.2, (20, 3)) + c).astype(np.float32).astype(np.float64)))
        # duplicated points (a lattice map queried at a symmetric position)
        base = (rng.uniform(-0.5, 0.5, (5, 3)) + c).astype(np.float32).astype(np.float64)
        out.append(("duplicates", np.repeat(base, 4, axis=0)))
    return out


def test_near_degenerate_stress_set_ql_vs_jacobi_vs_lapack():
    """What the Eigen boundary can cost: both solvers are backward stable (A V = V diag(ev) to a few eps ||A||), so
    eigenvalues agree to eps ||A|| -- which is a2D to ~sqrt(eps) ABSOLUTE on exact planes, i.e. weight to ~1e-8 relative:
    inside the 1e-5 parity budget, far above the 1e-9 the well-conditioned cases reach.  The normal agrees whenever the
    smallest eigenvalue is separated; for lines / rods / blobs it is decided by rounding and no two solvers agree."""
    rng = np.random.default_rng(11)
    worst = {}
    for name, P in _stress_sets(rng):
        A = scatter(P)
        nrm = np.abs(A).max()
        e_q, V_q, ok = po.eig3(A, po.EIG_EIGEN_QL)
        e_j, V_j, _ = po.eig3(A, po.EIG_JACOBI)
        assert ok
        check_decomposition(A, e_q, V_q)
        check_decomposition(A, e_j, V_j)
        w = np.linalg.eigvalsh(A)
        assert np.abs(e_q - w).max() <= 16 * np.finfo(float).eps * nrm, name
        assert np.abs(e_j - w).max() <= 16 * np.finfo(float).eps * nrm, name
        d_a2d = abs(a2d_of(e_q) - a2d_of(e_j))
        gap = (w[1] - w[0]) / w[2]
        d_n = 1.0 - abs(V_q[:, 0] @ V_j[:, 0])
        k = worst.setdefault(name, dict(a2d=0.0, normal=0.0, min_gap=1.0))
        k["a2d"] = max(k["a2d"], d_a2d); k["normal"] = max(k["normal"], d_n); k["min_gap"] = min(k["min_gap"], gap)
        assert d_a2d < 1e-6, (name, d_a2d)                      # sqrt(16 eps) ~ 6e-8 on exact planes
        if gap > 1e-6:
            assert d_n < 1e-9 / gap + 1e-15, (name, d_n, gap)
    assert worst["axis_plane"]["a2d"] < 1e-7
    assert worst["ring"]["normal"] < 1e-12 and worst["tilted_plane"]["normal"] < 1e-10
    print({k: {a: float("%.3g" % b) for a, b in v.items()} for k, v in worst.items()})


def test_full_pass_ql_vs_jacobi_on_the_shared_scene(small_scene, oracle_backend):
    """One buildPlaneResiduals pass with either solver: residual fields and normal equations to 1e-9 relative on the
    synthetic scene (well-separated planar neighbourhoods dominate); reported per field."""
    import sr_livo_amd as srl
    sw = small_scene["sweep"]
    oo = po.opts_from_product(srl.default_opts(max_num_residuals=2**31 - 1))
    r_q = small_scene["map"].build_plane_residuals(oo, sw["raw"], sw["q_pred"], sw["t_pred"], sw["t_last"])
    with po.eig_solver(po.EIG_JACOBI):
        r_j = small_scene["map"].build_plane_residuals(oo, sw["raw"], sw["q_pred"], sw["t_pred"], sw["t_last"])
    assert np.array_equal(r_q["status"], r_j["status"]) and np.array_equal(r_q["ids"], r_j["ids"])
    has = (r_q["status"] == 1) | (r_q["status"] == 2)
    for key in ("normal", "a2D", "weight", "distance", "jacobian"):
        a, b = r_q[key][has], r_j[key][has]
        assert np.abs(a - b).max() / np.abs(b).max() < 1e-9, key
    assert np.abs(r_q["HtH"] - r_j["HtH"]).max() / np.abs(r_j["HtH"]).max() < 1e-10
