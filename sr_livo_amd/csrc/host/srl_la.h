// srl_la.h -- fixed-size FP64 linear algebra used by the host mirror of the reference classes.
// Eigen is a system dependency of the reference (CMakeLists.txt:51) that is absent from this image,
// so the product ships the few operations the path needs, following Eigen 3.3 semantics where they
// decide bits or branches: column-wise 3x3 products summed k = 0,1,2,
// Quaternion::toRotationMatrix, Quaternion(Matrix3) (Shepperd), normalized() = v / sqrt(v.v),
// fixed-size inverse() for N > 4 = partial-pivot LU, SelfAdjointEigenSolver<Matrix3d> = tridiagonalisation + implicit
// symmetric QR.  Row-major storage (an ABI detail only).
#pragma once
#include <cmath>
#include <cstring>

// In a HIP translation unit every function below is __host__ __device__ (the kernels use the quaternion / 3 x 3 forms), elsewhere
// the marker is empty.
#if defined(__HIP__)
#include <hip/hip_runtime.h>
#define SRL_HD __host__ __device__
#else
#define SRL_HD
#endif

namespace srl {

template <int R, int C>
struct Mat {
    double a[R * C];
    SRL_HD double &operator()(int i, int j) { return a[i * C + j]; }
    SRL_HD double operator()(int i, int j) const { return a[i * C + j]; }
    SRL_HD double &operator()(int i) { return a[i]; }
    SRL_HD double operator()(int i) const { return a[i]; }
    SRL_HD double &operator[](int i) { return a[i]; }
    SRL_HD double operator[](int i) const { return a[i]; }
    SRL_HD static Mat Zero() { Mat m; for (int i = 0; i < R * C; i++) m.a[i] = 0.0; return m; }
    SRL_HD static Mat Identity() { Mat m = Zero(); for (int i = 0; i < (R < C ? R : C); i++) m(i, i) = 1.0; return m; }
    SRL_HD Mat<C, R> transpose() const { Mat<C, R> t; for (int i = 0; i < R; i++) for (int j = 0; j < C; j++) t(j, i) = (*this)(i, j); return t; }
    SRL_HD double x() const { return a[0]; }
    SRL_HD double y() const { return a[1]; }
    SRL_HD double z() const { return a[2]; }
    SRL_HD double squaredNorm() const { double s = a[0] * a[0]; for (int i = 1; i < R * C; i++) s += a[i] * a[i]; return s; }
    SRL_HD double norm() const { return std::sqrt(squaredNorm()); }
    SRL_HD double dot(const Mat &o) const { double s = a[0] * o.a[0]; for (int i = 1; i < R * C; i++) s += a[i] * o.a[i]; return s; }
    SRL_HD void normalize() { double z = squaredNorm(); if (z > 0.0) { double n = std::sqrt(z); for (int i = 0; i < R * C; i++) a[i] /= n; } }
    SRL_HD Mat normalized() const { Mat m = *this; m.normalize(); return m; }
    template <int BR, int BC> SRL_HD Mat<BR, BC> block(int r0, int c0) const { Mat<BR, BC> b; for (int i = 0; i < BR; i++) for (int j = 0; j < BC; j++) b(i, j) = (*this)(r0 + i, c0 + j); return b; }
    template <int BR, int BC> SRL_HD void setBlock(int r0, int c0, const Mat<BR, BC> &b) { for (int i = 0; i < BR; i++) for (int j = 0; j < BC; j++) (*this)(r0 + i, c0 + j) = b(i, j); }
};

using Vec3 = Mat<3, 1>;
using Vec2 = Mat<2, 1>;
using Mat3 = Mat<3, 3>;
using Mat2 = Mat<2, 2>;
using Mat32 = Mat<3, 2>;
using Vec17 = Mat<17, 1>;
using Mat17 = Mat<17, 17>;

SRL_HD inline Vec3 vec3(double x, double y, double z) { Vec3 v; v.a[0] = x; v.a[1] = y; v.a[2] = z; return v; }

template <int R, int C> SRL_HD Mat<R, C> operator+(const Mat<R, C> &x, const Mat<R, C> &y) { Mat<R, C> r; for (int i = 0; i < R * C; i++) r.a[i] = x.a[i] + y.a[i]; return r; }
template <int R, int C> SRL_HD Mat<R, C> operator-(const Mat<R, C> &x, const Mat<R, C> &y) { Mat<R, C> r; for (int i = 0; i < R * C; i++) r.a[i] = x.a[i] - y.a[i]; return r; }
template <int R, int C> SRL_HD Mat<R, C> operator-(const Mat<R, C> &x) { Mat<R, C> r; for (int i = 0; i < R * C; i++) r.a[i] = -x.a[i]; return r; }
template <int R, int C> SRL_HD Mat<R, C> operator*(const Mat<R, C> &x, double s) { Mat<R, C> r; for (int i = 0; i < R * C; i++) r.a[i] = x.a[i] * s; return r; }
template <int R, int C> SRL_HD Mat<R, C> operator*(double s, const Mat<R, C> &x) { Mat<R, C> r; for (int i = 0; i < R * C; i++) r.a[i] = s * x.a[i]; return r; }
template <int R, int C> SRL_HD Mat<R, C> operator/(const Mat<R, C> &x, double s) { Mat<R, C> r; for (int i = 0; i < R * C; i++) r.a[i] = x.a[i] / s; return r; }
template <int R, int K, int C>
SRL_HD Mat<R, C> operator*(const Mat<R, K> &x, const Mat<K, C> &y) {
    Mat<R, C> r;
    for (int i = 0; i < R; i++)
        for (int j = 0; j < C; j++) {
            double s = x(i, 0) * y(0, j);
            for (int k = 1; k < K; k++) s += x(i, k) * y(k, j);
            r(i, j) = s;
        }
    return r;
}
SRL_HD inline Vec3 cross(const Vec3 &a, const Vec3 &b) {
    return vec3(a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]);
}

// Eigen::Quaterniond restated (w,x,y,z)
struct Quat {
    double w, x, y, z;
    SRL_HD Quat() : w(1), x(0), y(0), z(0) {}
    SRL_HD Quat(double w_, double x_, double y_, double z_) : w(w_), x(x_), y(y_), z(z_) {}
    SRL_HD static Quat Identity() { return Quat(1, 0, 0, 0); }
    SRL_HD double squaredNorm() const { return ((x * x + y * y) + z * z) + w * w; }
    SRL_HD Quat normalized() const {
        double n2 = squaredNorm();
        if (n2 > 0.0) { double n = std::sqrt(n2); return Quat(w / n, x / n, y / n, z / n); }
        return *this;
    }
    SRL_HD void normalize() { *this = normalized(); }
    SRL_HD Quat inverse() const {
        double n2 = squaredNorm();
        if (n2 > 0.0) return Quat(w / n2, -x / n2, -y / n2, -z / n2);
        return Quat(0, 0, 0, 0);
    }
    SRL_HD Quat operator*(const Quat &b) const {
        return Quat(w * b.w - x * b.x - y * b.y - z * b.z, w * b.x + x * b.w + y * b.z - z * b.y,
                    w * b.y + y * b.w + z * b.x - x * b.z, w * b.z + z * b.w + x * b.y - y * b.x);
    }
    // Quaternion * Vector3 (Eigen's _transformVector: two cross products, not a matrix product)
    SRL_HD Vec3 operator*(const Vec3 &v) const;
    SRL_HD Mat3 toRotationMatrix() const {
        const double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
        const double twx = tx * w, twy = ty * w, twz = tz * w;
        const double txx = tx * x, txy = ty * x, txz = tz * x;
        const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
        Mat3 r;
        r(0, 0) = 1.0 - (tyy + tzz); r(0, 1) = txy - twz; r(0, 2) = txz + twy;
        r(1, 0) = txy + twz; r(1, 1) = 1.0 - (txx + tzz); r(1, 2) = tyz - twx;
        r(2, 0) = txz - twy; r(2, 1) = tyz + twx; r(2, 2) = 1.0 - (txx + tyy);
        return r;
    }
    // the branch of Shepperd's method for a non-positive trace, largest diagonal element I (static indices: the same
    // operations as Eigen's run-time i, j, k form, and no indexed local array in device code)
    template <int I>
    SRL_HD static Quat shepperdDiag(const Mat3 &m) {
        constexpr int J = (I + 1) % 3, K = (J + 1) % 3;
        double t = std::sqrt(m(I, I) - m(J, J) - m(K, K) + 1.0);
        const double vi = 0.5 * t;
        t = 0.5 / t;
        const double w = (m(K, J) - m(J, K)) * t;
        const double vj = (m(J, I) + m(I, J)) * t;
        const double vk = (m(K, I) + m(I, K)) * t;
        if (I == 0) return Quat(w, vi, vj, vk);
        if (I == 1) return Quat(w, vk, vi, vj);
        return Quat(w, vj, vk, vi);
    }
    SRL_HD static Quat fromRotationMatrix(const Mat3 &m) {   // Shepperd
        double t = m(0, 0) + m(1, 1) + m(2, 2);
        if (t > 0.0) {
            Quat q;
            t = std::sqrt(t + 1.0);
            q.w = 0.5 * t;
            t = 0.5 / t;
            q.x = (m(2, 1) - m(1, 2)) * t;
            q.y = (m(0, 2) - m(2, 0)) * t;
            q.z = (m(1, 0) - m(0, 1)) * t;
            return q;
        }
        int i = 0;
        if (m(1, 1) > m(0, 0)) i = 1;
        if (m(2, 2) > m(i == 1 ? 1 : 0, i == 1 ? 1 : 0)) i = 2;
        if (i == 0) return shepperdDiag<0>(m);
        if (i == 1) return shepperdDiag<1>(m);
        return shepperdDiag<2>(m);
    }
};

SRL_HD inline Vec3 Quat::operator*(const Vec3 &v) const {
    const Vec3 qv = vec3(x, y, z);
    Vec3 uv = cross(qv, v);
    uv = uv + uv;
    return (v + w * uv) + cross(qv, uv);
}

// Matrix<double,N,N>::inverse() for N > 4: PartialPivLU, then solve against the identity
// inverse_cols<N, M>: the first M columns of that inverse.  Every column of the inverse is the solution of L U x = P e_c, solved
// independently of the others, so the first M columns computed alone carry the same bits as the same columns of the full
// inverse -- updateIEKF only ever reads temp_inv.block<17, 6>(0, 0) of its second inverse (src/optimize.cpp:237-242).
template <int N, int M>
SRL_HD bool inverse_cols(const Mat<N, N> &A, Mat<N, M> &Ainv) {
    double lu[N][N];
    int perm[N];
    for (int i = 0; i < N; i++) { perm[i] = i; for (int j = 0; j < N; j++) lu[i][j] = A(i, j); }
    for (int k = 0; k < N; k++) {
        int piv = k;
        double best = std::fabs(lu[k][k]);
        for (int i = k + 1; i < N; i++) { const double v = std::fabs(lu[i][k]); if (v > best) { best = v; piv = i; } }
        if (best == 0.0) return false;
        if (piv != k) { for (int j = 0; j < N; j++) { const double tmp = lu[k][j]; lu[k][j] = lu[piv][j]; lu[piv][j] = tmp; } const int tp = perm[k]; perm[k] = perm[piv]; perm[piv] = tp; }
        for (int i = k + 1; i < N; i++) {
            lu[i][k] /= lu[k][k];
            const double f = lu[i][k];
            for (int j = k + 1; j < N; j++) lu[i][j] -= f * lu[k][j];
        }
    }
    // Solve L U X = P for all N right-hand sides AT ONCE: every column still accumulates its terms in the order
    // j = 0, 1, ... of the column-by-column loop (same rounding, bit for bit), but the N columns are N independent
    // chains the CPU can overlap -- a single column is one long dependent chain of subtractions (latency-bound: the two
    // 17 x 17 inverses of an ESIKF iteration cost 9 us that way, 2 us this way).
    double Y[N][M];
    for (int i = 0; i < N; i++) {
        for (int c = 0; c < M; c++) Y[i][c] = (perm[i] == c) ? 1.0 : 0.0;
        for (int j = 0; j < i; j++) {
            const double f = lu[i][j];
            for (int c = 0; c < M; c++) Y[i][c] -= f * Y[j][c];
        }
    }
    for (int i = N - 1; i >= 0; i--) {
        for (int j = i + 1; j < N; j++) {
            const double f = lu[i][j];
            for (int c = 0; c < M; c++) Y[i][c] -= f * Y[j][c];
        }
        const double d = lu[i][i];
        for (int c = 0; c < M; c++) Y[i][c] = Y[i][c] / d;
    }
    for (int i = 0; i < N; i++) for (int c = 0; c < M; c++) Ainv(i, c) = Y[i][c];
    return true;
}
template <int N>
SRL_HD bool inverse(const Mat<N, N> &A, Mat<N, N> &Ainv) { return inverse_cols<N, N>(A, Ainv); }

// Eigen::SelfAdjointEigenSolver<Matrix3d> as the reference uses it (src/optimize.cpp:339-346): constructed from a
// symmetric 3x3, then eigenvalues() ascending and eigenvectors().col(i).  Follows Eigen 3.3.7's iterative path
// (SelfAdjointEigenSolver.h compute(): scale the lower triangle by its largest |coefficient|; Tridiagonalization.h:
// the 3x3 real specialisation, one Householder reflector; computeFromTridiagonal_impl: deflate sub-diagonal entries
// below 2 eps (|d_i| + |d_i+1|), implicit symmetric QR steps with Wilkinson shift (tridiagonal_qr_step) built from
// Givens rotations (Jacobi.h makeGivens), at most 30 n iterations; selection sort).  Same operation order, so the result
// rounds the way Eigen's does; the normal of a near-degenerate neighbourhood depends on exactly that.
class SelfAdjointEigenSolver3 {
public:
    enum Info { Success = 0, NoConvergence = 1 };
    explicit SelfAdjointEigenSolver3(const Mat3 &matrix) { compute(matrix); }
    const Vec3 &eigenvalues() const { return m_eivalues; }
    const Mat3 &eigenvectors() const { return m_eivec; }
    Vec3 eigenvector(int c) const { return vec3(m_eivec(0, c), m_eivec(1, c), m_eivec(2, c)); }   // eigenvectors().col(c)
    Info info() const { return m_info; }

private:
    Vec3 m_eivalues;
    Mat3 m_eivec;
    double m_subdiag[2];
    Info m_info = Success;

    struct Rot { double c, s; };
    static Rot givens(double p, double q) {
        Rot r;
        if (q == 0.0) { r.c = p < 0.0 ? -1.0 : 1.0; r.s = 0.0; }
        else if (p == 0.0) { r.c = 0.0; r.s = q < 0.0 ? 1.0 : -1.0; }
        else if (std::fabs(p) > std::fabs(q)) {
            const double t = q / p;
            double u = std::sqrt(1.0 + t * t);
            if (p < 0.0) u = -u;
            r.c = 1.0 / u;
            r.s = -t * r.c;
        } else {
            const double t = p / q;
            double u = std::sqrt(1.0 + t * t);
            if (q < 0.0) u = -u;
            r.s = -1.0 / u;
            r.c = -t * r.s;
        }
        return r;
    }
    static double hypot3(double x, double y) {          // numext::hypot of Eigen 3.3
        const double ax = std::fabs(x), ay = std::fabs(y);
        const double p = ax > ay ? ax : ay;
        if (p == 0.0) return 0.0;
        const double qp = (ax > ay ? ay : ax) / p;
        return p * std::sqrt(1.0 + qp * qp);
    }
    void tridiagonalize(const double l[6]) {            // l = {m00, m10, m11, m20, m21, m22}, already scaled
        const double m00 = l[0], m10 = l[1], m11 = l[2], m20 = l[3], m21 = l[4], m22 = l[5];
        m_eivalues[0] = m00;
        const double v1norm2 = m20 * m20;
        if (v1norm2 <= 2.2250738585072014e-308) {        // std::numeric_limits<double>::min()
            m_eivalues[1] = m11; m_eivalues[2] = m22;
            m_subdiag[0] = m10; m_subdiag[1] = m21;
            m_eivec = Mat3::Identity();
        } else {
            const double beta = std::sqrt(m10 * m10 + v1norm2);
            const double invBeta = 1.0 / beta;
            const double m01 = m10 * invBeta, m02 = m20 * invBeta;
            const double q = 2.0 * m01 * m21 + m02 * (m22 - m11);
            m_eivalues[1] = m11 + m02 * q;
            m_eivalues[2] = m22 - m02 * q;
            m_subdiag[0] = beta;
            m_subdiag[1] = m21 - m01 * q;
            m_eivec = Mat3::Zero();
            m_eivec(0, 0) = 1.0;
            m_eivec(1, 1) = m01; m_eivec(1, 2) = m02;
            m_eivec(2, 1) = m02; m_eivec(2, 2) = -m01;
        }
    }
    void qrStep(int start, int end) {
        double *diag = m_eivalues.a, *sub = m_subdiag;
        const double td = (diag[end - 1] - diag[end]) * 0.5;
        const double e = sub[end - 1];
        double mu = diag[end];
        if (td == 0.0) mu -= std::fabs(e);
        else {
            const double e2 = e * e;
            const double h = hypot3(td, e);
            if (e2 == 0.0) mu -= (e / (td + (td > 0.0 ? 1.0 : -1.0))) * (e / h);
            else mu -= e2 / (td + (td > 0.0 ? h : -h));
        }
        double x = diag[start] - mu, z = sub[start];
        for (int k = start; k < end; ++k) {
            const Rot g = givens(x, z);
            const double sdk = g.s * diag[k] + g.c * sub[k];
            const double dkp1 = g.s * sub[k] + g.c * diag[k + 1];
            diag[k] = g.c * (g.c * diag[k] - g.s * sub[k]) - g.s * (g.c * sub[k] - g.s * diag[k + 1]);
            diag[k + 1] = g.s * sdk + g.c * dkp1;
            sub[k] = g.c * sdk - g.s * dkp1;
            if (k > start) sub[k - 1] = g.c * sub[k - 1] - g.s * z;
            x = sub[k];
            if (k < end - 1) { z = -g.s * sub[k + 1]; sub[k + 1] = g.c * sub[k + 1]; }
            if (!(g.c == 1.0 && g.s == 0.0))
                for (int i = 0; i < 3; ++i) {                 // Q.applyOnTheRight(k, k + 1, g)
                    const double xi = m_eivec(i, k), yi = m_eivec(i, k + 1);
                    m_eivec(i, k) = g.c * xi - g.s * yi;
                    m_eivec(i, k + 1) = g.s * xi + g.c * yi;
                }
        }
    }
    void compute(const Mat3 &m) {
        double l[6] = {m(0, 0), m(1, 0), m(1, 1), m(2, 0), m(2, 1), m(2, 2)};       // triangularView<Lower>()
        double scale = 0.0;
        for (double v : l) { const double av = std::fabs(v); if (av > scale) scale = av; }
        if (scale == 0.0) scale = 1.0;
        for (double &v : l) v /= scale;
        tridiagonalize(l);
        const double tiny = 2.2250738585072014e-308, precision = 2.0 * 2.220446049250313e-16;
        double *diag = m_eivalues.a, *sub = m_subdiag;
        int end = 2, start = 0, iter = 0;
        while (end > 0) {
            for (int i = start; i < end; ++i)
                if (std::fabs(sub[i]) <= (std::fabs(diag[i]) + std::fabs(diag[i + 1])) * precision || std::fabs(sub[i]) <= tiny) sub[i] = 0.0;
            while (end > 0 && sub[end - 1] == 0.0) end--;
            if (end <= 0) break;
            if (++iter > 30 * 3) break;
            start = end - 1;
            while (start > 0 && sub[start - 1] != 0.0) start--;
            qrStep(start, end);
        }
        m_info = iter <= 30 * 3 ? Success : NoConvergence;
        if (m_info == Success)
            for (int i = 0; i < 2; ++i) {
                int k = 0;
                for (int j = 1; j < 3 - i; ++j) if (diag[i + j] < diag[i + k]) k = j;
                if (k > 0) {
                    const double t = diag[i]; diag[i] = diag[i + k]; diag[i + k] = t;
                    for (int r = 0; r < 3; ++r) { const double u = m_eivec(r, i); m_eivec(r, i) = m_eivec(r, i + k); m_eivec(r, i + k) = u; }
                }
            }
        for (int i = 0; i < 3; ++i) diag[i] *= scale;
    }
};

}  // namespace srl
