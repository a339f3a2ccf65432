// srl_la.h -- fixed-size FP64 linear algebra used by the host mirror of the reference classes.
// Eigen is a system dependency of the reference (CMakeLists.txt:51) that is absent from this image,
// so the product ships the few operations the path needs, following Eigen 3.3 semantics where they
// decide bits or branches: column-wise 3x3 products summed k = 0,1,2,
// Quaternion::toRotationMatrix, Quaternion(Matrix3) (Shepperd), normalized() = v / sqrt(v.v),
// fixed-size inverse() for N > 4 = partial-pivot LU.  Row-major storage (an ABI detail only).
#pragma once
#include <cmath>
#include <cstring>

namespace srl {

template <int R, int C>
struct Mat {
    double a[R * C];
    double &operator()(int i, int j) { return a[i * C + j]; }
    double operator()(int i, int j) const { return a[i * C + j]; }
    double &operator()(int i) { return a[i]; }
    double operator()(int i) const { return a[i]; }
    double &operator[](int i) { return a[i]; }
    double operator[](int i) const { return a[i]; }
    static Mat Zero() { Mat m; for (int i = 0; i < R * C; i++) m.a[i] = 0.0; return m; }
    static Mat Identity() { Mat m = Zero(); for (int i = 0; i < (R < C ? R : C); i++) m(i, i) = 1.0; return m; }
    Mat<C, R> transpose() const { Mat<C, R> t; for (int i = 0; i < R; i++) for (int j = 0; j < C; j++) t(j, i) = (*this)(i, j); return t; }
    double x() const { return a[0]; }
    double y() const { return a[1]; }
    double z() const { return a[2]; }
    double squaredNorm() const { double s = a[0] * a[0]; for (int i = 1; i < R * C; i++) s += a[i] * a[i]; return s; }
    double norm() const { return std::sqrt(squaredNorm()); }
    double dot(const Mat &o) const { double s = a[0] * o.a[0]; for (int i = 1; i < R * C; i++) s += a[i] * o.a[i]; return s; }
    void normalize() { double z = squaredNorm(); if (z > 0.0) { double n = std::sqrt(z); for (int i = 0; i < R * C; i++) a[i] /= n; } }
    Mat normalized() const { Mat m = *this; m.normalize(); return m; }
    template <int BR, int BC> Mat<BR, BC> block(int r0, int c0) const { Mat<BR, BC> b; for (int i = 0; i < BR; i++) for (int j = 0; j < BC; j++) b(i, j) = (*this)(r0 + i, c0 + j); return b; }
    template <int BR, int BC> void setBlock(int r0, int c0, const Mat<BR, BC> &b) { for (int i = 0; i < BR; i++) for (int j = 0; j < BC; j++) (*this)(r0 + i, c0 + j) = b(i, j); }
};

using Vec3 = Mat<3, 1>;
using Vec2 = Mat<2, 1>;
using Mat3 = Mat<3, 3>;
using Mat2 = Mat<2, 2>;
using Mat32 = Mat<3, 2>;
using Vec17 = Mat<17, 1>;
using Mat17 = Mat<17, 17>;

inline Vec3 vec3(double x, double y, double z) { Vec3 v; v.a[0] = x; v.a[1] = y; v.a[2] = z; return v; }

template <int R, int C> Mat<R, C> operator+(const Mat<R, C> &x, const Mat<R, C> &y) { Mat<R, C> r; for (int i = 0; i < R * C; i++) r.a[i] = x.a[i] + y.a[i]; return r; }
template <int R, int C> Mat<R, C> operator-(const Mat<R, C> &x, const Mat<R, C> &y) { Mat<R, C> r; for (int i = 0; i < R * C; i++) r.a[i] = x.a[i] - y.a[i]; return r; }
template <int R, int C> Mat<R, C> operator-(const Mat<R, C> &x) { Mat<R, C> r; for (int i = 0; i < R * C; i++) r.a[i] = -x.a[i]; return r; }
template <int R, int C> Mat<R, C> operator*(const Mat<R, C> &x, double s) { Mat<R, C> r; for (int i = 0; i < R * C; i++) r.a[i] = x.a[i] * s; return r; }
template <int R, int C> Mat<R, C> operator*(double s, const Mat<R, C> &x) { Mat<R, C> r; for (int i = 0; i < R * C; i++) r.a[i] = s * x.a[i]; return r; }
template <int R, int C> Mat<R, C> operator/(const Mat<R, C> &x, double s) { Mat<R, C> r; for (int i = 0; i < R * C; i++) r.a[i] = x.a[i] / s; return r; }
template <int R, int K, int C>
Mat<R, C> operator*(const Mat<R, K> &x, const Mat<K, C> &y) {
    Mat<R, C> r;
    for (int i = 0; i < R; i++)
        for (int j = 0; j < C; j++) {
            double s = x(i, 0) * y(0, j);
            for (int k = 1; k < K; k++) s += x(i, k) * y(k, j);
            r(i, j) = s;
        }
    return r;
}
inline Vec3 cross(const Vec3 &a, const Vec3 &b) {
    return vec3(a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]);
}

// Eigen::Quaterniond restated (w,x,y,z)
struct Quat {
    double w, x, y, z;
    Quat() : w(1), x(0), y(0), z(0) {}
    Quat(double w_, double x_, double y_, double z_) : w(w_), x(x_), y(y_), z(z_) {}
    static Quat Identity() { return Quat(1, 0, 0, 0); }
    double squaredNorm() const { return ((x * x + y * y) + z * z) + w * w; }
    Quat normalized() const {
        double n2 = squaredNorm();
        if (n2 > 0.0) { double n = std::sqrt(n2); return Quat(w / n, x / n, y / n, z / n); }
        return *this;
    }
    void normalize() { *this = normalized(); }
    Quat inverse() const {
        double n2 = squaredNorm();
        if (n2 > 0.0) return Quat(w / n2, -x / n2, -y / n2, -z / n2);
        return Quat(0, 0, 0, 0);
    }
    Quat operator*(const Quat &b) const {
        return Quat(w * b.w - x * b.x - y * b.y - z * b.z, w * b.x + x * b.w + y * b.z - z * b.y,
                    w * b.y + y * b.w + z * b.x - x * b.z, w * b.z + z * b.w + x * b.y - y * b.x);
    }
    // Quaternion * Vector3 (Eigen's _transformVector: two cross products, not a matrix product)
    Vec3 operator*(const Vec3 &v) const;
    Mat3 toRotationMatrix() const {
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
    static Quat fromRotationMatrix(const Mat3 &m) {   // Shepperd
        Quat q;
        double t = m(0, 0) + m(1, 1) + m(2, 2);
        if (t > 0.0) {
            t = std::sqrt(t + 1.0);
            q.w = 0.5 * t;
            t = 0.5 / t;
            q.x = (m(2, 1) - m(1, 2)) * t;
            q.y = (m(0, 2) - m(2, 0)) * t;
            q.z = (m(1, 0) - m(0, 1)) * t;
        } else {
            int i = 0;
            if (m(1, 1) > m(0, 0)) i = 1;
            if (m(2, 2) > m(i, i)) i = 2;
            const int j = (i + 1) % 3, k = (j + 1) % 3;
            t = std::sqrt(m(i, i) - m(j, j) - m(k, k) + 1.0);
            double v[3];
            v[i] = 0.5 * t;
            t = 0.5 / t;
            q.w = (m(k, j) - m(j, k)) * t;
            v[j] = (m(j, i) + m(i, j)) * t;
            v[k] = (m(k, i) + m(i, k)) * t;
            q.x = v[0]; q.y = v[1]; q.z = v[2];
        }
        return q;
    }
};

inline Vec3 Quat::operator*(const Vec3 &v) const {
    const Vec3 qv = vec3(x, y, z);
    Vec3 uv = cross(qv, v);
    uv = uv + uv;
    return (v + w * uv) + cross(qv, uv);
}

// Matrix<double,N,N>::inverse() for N > 4: PartialPivLU, then solve against the identity
template <int N>
bool inverse(const Mat<N, N> &A, Mat<N, N> &Ainv) {
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
    for (int c = 0; c < N; c++) {
        double y[N];
        for (int i = 0; i < N; i++) {
            double s = (perm[i] == c) ? 1.0 : 0.0;
            for (int j = 0; j < i; j++) s -= lu[i][j] * y[j];
            y[i] = s;
        }
        for (int i = N - 1; i >= 0; i--) {
            double s = y[i];
            for (int j = i + 1; j < N; j++) s -= lu[i][j] * Ainv(j, c);
            Ainv(i, c) = s / lu[i][i];
        }
    }
    return true;
}

}  // namespace srl
