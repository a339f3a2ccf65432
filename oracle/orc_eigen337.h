// orc_eigen337.h -- TEST INFRASTRUCTURE (part of the CPU oracle; see srl_oracle.h).
//
// The two Eigen 3.3.7 algorithms the reference path executes and whose sources are NOT under /root/reference
// (Eigen is a system dependency of the reference: CMakeLists.txt:51, README.md:64 "tested with 3.3.7"; it is
// absent from this image), restated from the published sources operation by operation:
//   * Matrix<double,17,17>::inverse()           (src/optimize.cpp:234,237)  -> orc_algos::lu_inverse<N> / lu_inverse_n
//   * SelfAdjointEigenSolver<Matrix3d>::compute (src/optimize.cpp:339)      -> orc_algos::eig3_eigen_ql
// Shared by oracle/srl_oracle.cpp (the restatement) and oracle/ref_shim/Eigen/Core (the stand-in Eigen that
// the reference's own translation units are compiled against in oracle/_ref/libref_path.so).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

namespace orc_algos {

// Matrix<double,17,17>::inverse(): PartialPivLU + solve(I)  
template <int N>
bool lu_inverse(const double *A, double *Ainv) {
    double lu[N][N];
    int perm[N];
    for (int i = 0; i < N; i++) { perm[i] = i; for (int j = 0; j < N; j++) lu[i][j] = A[i * N + j]; }
    for (int k = 0; k < N; k++) {
        int piv = k; double best = std::fabs(lu[k][k]);
        for (int i = k + 1; i < N; i++) { double v = std::fabs(lu[i][k]); if (v > best) { best = v; piv = i; } }
        if (best == 0.0) return false;
        if (piv != k) { for (int j = 0; j < N; j++) std::swap(lu[k][j], lu[piv][j]); std::swap(perm[k], perm[piv]); }
        for (int i = k + 1; i < N; i++) {
            lu[i][k] /= lu[k][k];
            double f = lu[i][k];
            for (int j = k + 1; j < N; j++) lu[i][j] -= f * lu[k][j];
        }
    }
    for (int c = 0; c < N; c++) {
        double y[N];
        for (int i = 0; i < N; i++) {
            double s = (perm[i] == c) ? 1.0 : 0.0;
            for (int j = 0; j < i; j++) s -= lu[i][j] * y[j];
            y[i] = s;
        }
        for (int i = N - 1; i >= 0; i--) {
            double s = y[i];
            for (int j = i + 1; j < N; j++) s -= lu[i][j] * Ainv[j * N + c];
            Ainv[i * N + c] = s / lu[i][i];
        }
    }
    return true;
}

// the same elimination for a run-time size (row-major n x n); used by the stand-in Eigen for every fixed size
inline bool lu_inverse_n(int N, const double *A, double *Ainv) {
    std::vector<double> lu(A, A + (size_t)N * N), y(N);
    std::vector<int> perm(N);
    for (int i = 0; i < N; i++) perm[i] = i;
    for (int k = 0; k < N; k++) {
        int piv = k; double best = std::fabs(lu[k * N + k]);
        for (int i = k + 1; i < N; i++) { double v = std::fabs(lu[i * N + k]); if (v > best) { best = v; piv = i; } }
        if (best == 0.0) return false;
        if (piv != k) { for (int j = 0; j < N; j++) std::swap(lu[k * N + j], lu[piv * N + j]); std::swap(perm[k], perm[piv]); }
        for (int i = k + 1; i < N; i++) {
            lu[i * N + k] /= lu[k * N + k];
            double f = lu[i * N + k];
            for (int j = k + 1; j < N; j++) lu[i * N + j] -= f * lu[k * N + j];
        }
    }
    for (int c = 0; c < N; c++) {
        for (int i = 0; i < N; i++) {
            double s = (perm[i] == c) ? 1.0 : 0.0;
            for (int j = 0; j < i; j++) s -= lu[i * N + j] * y[j];
            y[i] = s;
        }
        for (int i = N - 1; i >= 0; i--) {
            double s = y[i];
            for (int j = i + 1; j < N; j++) s -= lu[i * N + j] * Ainv[j * N + c];
            Ainv[i * N + c] = s / lu[i * N + i];
        }
    }
    return true;
}

// SelfAdjointEigenSolver<Matrix3d>::compute(matrix, ComputeEigenvectors) as the reference instantiates it
// (src/optimize.cpp:339).  Eigen is a system dependency of the reference (CMakeLists.txt:51; README.md:64 tested
// with 3.3.7) and absent from /root/reference and from this image; what follows restates the published 3.3.7
// sources, operation by operation, so that rounding follows the same path:
//   Eigen/src/Eigenvalues/SelfAdjointEigenSolver.h   compute(): lower triangle, scale = max |coeff| (0 -> 1), /= scale,
//                                                    tridiagonalise, computeFromTridiagonal_impl, eigenvalues *= scale
//   Eigen/src/Eigenvalues/Tridiagonalization.h       tridiagonalization_inplace_selector<MatrixType, 3, false>::run
//   SelfAdjointEigenSolver.h                         computeFromTridiagonal_impl (deflation test with precision
//                                                    2 eps and considerAsZero = DBL_MIN, m_maxIterations = 30 per row),
//                                                    tridiagonal_qr_step (Wilkinson shift, chase the bulge)
//   Eigen/src/Jacobi/Jacobi.h                        JacobiRotation::makeGivens (real), applyOnTheRight
//   Eigen/src/Core/MathFunctions.h                   numext::hypot (3.3 hypot_impl)
// Eigenvalues ascending, eigenvectors = columns of V.  Returns false on NoConvergence (Eigen then leaves the
// eigenvalues unsorted; the reference never checks info()).
namespace eigen337 {
struct Givens { double c, s; };
inline Givens make_givens(double p, double q) {       // JacobiRotation<double>::makeGivens(p, q, 0, false_type)
    Givens g;
    if (q == 0.0) {
        g.c = p < 0.0 ? -1.0 : 1.0;
        g.s = 0.0;
    } else if (p == 0.0) {
        g.c = 0.0;
        g.s = q < 0.0 ? 1.0 : -1.0;
    } else if (std::fabs(p) > std::fabs(q)) {
        double t = q / p;
        double u = std::sqrt(1.0 + t * t);
        if (p < 0.0) u = -u;
        g.c = 1.0 / u;
        g.s = -t * g.c;
    } else {
        double t = p / q;
        double u = std::sqrt(1.0 + t * t);
        if (q < 0.0) u = -u;
        g.s = -1.0 / u;
        g.c = -t * g.s;
    }
    return g;
}
inline double hypot_impl(double x, double y) {        // numext::hypot, Eigen 3.3 hypot_impl<double>::run
    double ax = std::fabs(x), ay = std::fabs(y);
    double p, qp;
    if (ax > ay) { p = ax; qp = ay / p; } else { p = ay; qp = ax / p; }
    if (p == 0.0) return 0.0;
    return p * std::sqrt(1.0 + qp * qp);
}
// tridiagonal_qr_step<ColMajor>(diag, subdiag, start, end, matrixQ, n = 3); Q[row][col]
inline void qr_step(double *diag, double *subdiag, int start, int end, double Q[3][3]) {
    double td = (diag[end - 1] - diag[end]) * 0.5;
    double e = subdiag[end - 1];
    double mu = diag[end];
    if (td == 0.0) {
        mu -= std::fabs(e);
    } else {
        double e2 = subdiag[end - 1] * subdiag[end - 1];     // numext::abs2
        double h = hypot_impl(td, e);
        if (e2 == 0.0) mu -= (e / (td + (td > 0.0 ? 1.0 : -1.0))) * (e / h);
        else mu -= e2 / (td + (td > 0.0 ? h : -h));
    }
    double x = diag[start] - mu;
    double z = subdiag[start];
    for (int k = start; k < end; ++k) {
        Givens rot = make_givens(x, z);
        // do T = G' T G
        double sdk = rot.s * diag[k] + rot.c * subdiag[k];
        double dkp1 = rot.s * subdiag[k] + rot.c * diag[k + 1];
        diag[k] = rot.c * (rot.c * diag[k] - rot.s * subdiag[k]) - rot.s * (rot.c * subdiag[k] - rot.s * diag[k + 1]);
        diag[k + 1] = rot.s * sdk + rot.c * dkp1;
        subdiag[k] = rot.c * sdk - rot.s * dkp1;
        if (k > start) subdiag[k - 1] = rot.c * subdiag[k - 1] - rot.s * z;
        x = subdiag[k];
        if (k < end - 1) {
            z = -rot.s * subdiag[k + 1];
            subdiag[k + 1] = rot.c * subdiag[k + 1];
        }
        // Q = Q * G: q.applyOnTheRight(k, k + 1, rot) = apply_rotation_in_the_plane(col k, col k + 1, rot.transpose())
        if (!(rot.c == 1.0 && rot.s == 0.0)) {
            for (int i = 0; i < 3; ++i) {
                double xi = Q[i][k], yi = Q[i][k + 1];
                Q[i][k] = rot.c * xi - rot.s * yi;
                Q[i][k + 1] = rot.s * xi + rot.c * yi;
            }
        }
    }
}
}  // namespace eigen337

inline bool eig3_eigen_ql(const double Ain[3][3], double evals[3], double V[3][3]) {
    using namespace eigen337;
    // mat = matrix.triangularView<Lower>(); scale = mat.cwiseAbs().maxCoeff(); mat.triangularView<Lower>() /= scale
    double m00 = Ain[0][0], m10 = Ain[1][0], m11 = Ain[1][1], m20 = Ain[2][0], m21 = Ain[2][1], m22 = Ain[2][2];
    double scale = 0.0;
    {
        // maxCoeff over the 3x3 with a zero strict upper triangle, column-major visit; NaNs are not propagated specially
        const double col_major[9] = {std::fabs(m00), std::fabs(m10), std::fabs(m20), 0.0, std::fabs(m11), std::fabs(m21), 0.0, 0.0, std::fabs(m22)};
        scale = col_major[0];
        for (int i = 1; i < 9; i++) if (col_major[i] > scale) scale = col_major[i];
    }
    if (scale == 0.0) scale = 1.0;
    m00 /= scale; m10 /= scale; m11 /= scale; m20 /= scale; m21 /= scale; m22 /= scale;

    // tridiagonalization_inplace_selector<Matrix3d, 3, false>::run(mat, diag, subdiag, extractQ = true)
    double diag[3], subdiag[2];
    double Q[3][3];
    const double tol = std::numeric_limits<double>::min();
    diag[0] = m00;
    double v1norm2 = m20 * m20;
    if (v1norm2 <= tol) {
        diag[1] = m11;
        diag[2] = m22;
        subdiag[0] = m10;
        subdiag[1] = m21;
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Q[i][j] = (i == j) ? 1.0 : 0.0;
    } else {
        double beta = std::sqrt(m10 * m10 + v1norm2);
        double invBeta = 1.0 / beta;
        double m01 = m10 * invBeta;
        double m02 = m20 * invBeta;
        double q = 2.0 * m01 * m21 + m02 * (m22 - m11);
        diag[1] = m11 + m02 * q;
        diag[2] = m22 - m02 * q;
        subdiag[0] = beta;
        subdiag[1] = m21 - m01 * q;
        Q[0][0] = 1.0; Q[0][1] = 0.0; Q[0][2] = 0.0;
        Q[1][0] = 0.0; Q[1][1] = m01; Q[1][2] = m02;
        Q[2][0] = 0.0; Q[2][1] = m02; Q[2][2] = -m01;
    }

    // computeFromTridiagonal_impl(diag, subdiag, maxIterations = 30, computeEigenvectors = true, eivec)
    const int n = 3;
    int end = n - 1, start = 0, iter = 0;
    const int maxIterations = 30;
    const double considerAsZero = std::numeric_limits<double>::min();
    const double precision = 2.0 * std::numeric_limits<double>::epsilon();
    while (end > 0) {
        for (int i = start; i < end; ++i)
            // internal::isMuchSmallerThan(|subdiag[i]|, |diag[i]| + |diag[i+1]|, precision)  ==  |x| <= |y| * prec
            if (std::fabs(subdiag[i]) <= (std::fabs(diag[i]) + std::fabs(diag[i + 1])) * precision || std::fabs(subdiag[i]) <= considerAsZero)
                subdiag[i] = 0.0;
        // find the largest unreduced block
        while (end > 0 && subdiag[end - 1] == 0.0) end--;
        if (end <= 0) break;
        // if we spent too many iterations, we give up
        iter++;
        if (iter > maxIterations * n) break;
        start = end - 1;
        while (start > 0 && subdiag[start - 1] != 0.0) start--;
        qr_step(diag, subdiag, start, end, Q);
    }
    const bool ok = iter <= maxIterations * n;
    // Sort eigenvalues and corresponding vectors (selection sort; minCoeff returns the first minimum)
    if (ok) {
        for (int i = 0; i < n - 1; ++i) {
            int k = 0;
            for (int j = 1; j < n - i; ++j) if (diag[i + j] < diag[i + k]) k = j;
            if (k > 0) {
                std::swap(diag[i], diag[k + i]);
                for (int r = 0; r < 3; ++r) std::swap(Q[r][i], Q[r][k + i]);
            }
        }
    }
    // scale back the eigen values
    for (int i = 0; i < 3; i++) evals[i] = diag[i] * scale;
    std::memcpy(V, Q, sizeof(double) * 9);
    return ok;
}

}  // namespace orc_algos
