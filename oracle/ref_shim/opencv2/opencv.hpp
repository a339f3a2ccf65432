// NOT OpenCV: inert stand-ins for the cv names the reference's headers and node file use (include/cloudMap.h:35,80,
// include/lioOptimization.h:76,113-114,217, src/lioOptimization.cpp:41-139,620-760).  cv::Mat is an EMPTY image: the
// vision stage is out of scope and nothing on the scan-matching path touches a pixel.  Test infrastructure.
#pragma once
#include <cstdlib>
namespace cv {
template <class T, int N> struct Vec {
    T v[N];
    Vec() { for (int i = 0; i < N; ++i) v[i] = T(0); }
    Vec(T a, T b, T c) { static_assert(N == 3, "3-channel"); v[0] = a; v[1] = b; v[2] = c; }
    template <class U> Vec(const Vec<U, N> &o) { for (int i = 0; i < N; ++i) v[i] = static_cast<T>(o.v[i]); }
    T &operator[](int i) { return v[i]; }
    const T &operator[](int i) const { return v[i]; }
    T &operator()(int i) { return v[i]; }
    const T &operator()(int i) const { return v[i]; }
    template <class U> Vec &operator+=(const Vec<U, N> &o) { for (int i = 0; i < N; ++i) v[i] = static_cast<T>(v[i] + o.v[i]); return *this; }
};
template <class T, int N> Vec<T, N> operator+(const Vec<T, N> &a, const Vec<T, N> &b) { Vec<T, N> r; for (int i = 0; i < N; ++i) r.v[i] = static_cast<T>(a.v[i] + b.v[i]); return r; }
template <class T, int N> Vec<T, N> operator-(const Vec<T, N> &a, const Vec<T, N> &b) { Vec<T, N> r; for (int i = 0; i < N; ++i) r.v[i] = static_cast<T>(a.v[i] - b.v[i]); return r; }
template <class T, int N> Vec<T, N> operator*(double s, const Vec<T, N> &a) { Vec<T, N> r; for (int i = 0; i < N; ++i) r.v[i] = static_cast<T>(s * a.v[i]); return r; }
typedef Vec<unsigned char, 3> Vec3b;
typedef Vec<float, 3> Vec3f;
class Mat {
public:
    int rows = 0, cols = 0;
    bool empty() const { return true; }
    void release() {}
    Mat clone() const { return Mat(); }
    // an empty image has no pixels: reaching one is a test-harness bug
    template <class T> T &at(int, int) { std::abort(); }
    template <class T> T *ptr(int) { std::abort(); }
};
class RNG { public: RNG() {} explicit RNG(unsigned long long) {} };
struct Scalar { double v[4]; };
}  // namespace cv
