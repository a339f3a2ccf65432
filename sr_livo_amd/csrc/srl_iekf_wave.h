// srl_iekf_wave.h -- updateIEKF's 17-dim algebra (src/optimize.cpp:172-310) as the work of ONE WAVE.
//
// The persistent solve kernel (srl_kernels.hip, PERSIST = 1) keeps the whole ESIKF loop of optimize.cpp:147-312 on the
// device: one wave of the finishing workgroup runs, per iteration,
//     prior()   the part that does not depend on H_x (optimize.cpp:172-234): prior error state, S^2 / SO(3) projection of
//               the error state and of the covariance, temp = (P / R)^-1                    -- while the sweep is associated
//     update()  temp[0:6,0:6] += H^T H, second inverse, gain, d_x, step guard, observe(), convergence rule and -- on the
//               last pass -- the posterior covariance (optimize.cpp:235-310)                 -- behind the reduction
// with the SAME operations in the SAME order as the host mirror (host/lioOptimization.cpp:240-391, host/srl_la.h): every
// matrix element goes through the host's sequence of FP64 operations (the translation unit is compiled with
// -ffp-contract=off), only on another lane.  What can differ from the host in the last bit is sin / cos / acos of the
// device library.  The scalar 3-vector part calls the very functions the host calls (host/srl_la.h, host/utility.h are
// __host__ __device__ in a HIP translation unit).
//
// Layout.  Lane i < 17 holds ROW i of a 17 x 17 matrix in 17 registers (static indices: every loop below is fully
// unrolled).  The LU factorisation with partial pivoting (Matrix<double,17,17>::inverse() = PartialPivLU, srl::inverse_cols)
// never moves a row: a lane keeps its row and a POSITION; the pivot row of a step is broadcast with v_readlane from a
// wave-uniform lane, the forward substitution runs along with the elimination (same multipliers, same order per element),
// the backward substitution runs with one lane per right-hand side after a transposition through LDS.
//
// The code is written against a small wave interface W and instantiated twice: DevWave (a lane is a thread of the wave:
// VD = double, cross-lane operations are v_readlane / DPP / ballot) and HostWave (VD = 64 doubles, loops): the second one
// lets tests/test_iekf_wave.py run this exact source on the CPU against the host mirror, bit for bit
// (srl_debug_iekf_wave_solve, include/srlivo_hip.h).
#pragma once
#include "host/srl_la.h"
#include "host/utility.h"

#include <stdint.h>

namespace srlw {

enum {                       // what update() decides (the loop control of optimize.cpp:147-312)
    IEKF_CONTINUE = 0,       // next iteration with the pose just written
    IEKF_DONE = 1,           // converged, or last iteration: posterior covariance set (optimize.cpp:272-310)
    IEKF_DONE_NO_COV = 2,    // the loop ran out on a guarded step (optimize.cpp:248-251 `continue` on the last pass)
    IEKF_FAIL_RESIDUALS = 3, // summary.success = false (optimize.cpp:110-123,155-156): returned at once
    IEKF_NAN = 4,            // NaN planarity among the visited keypoints (optimize.cpp:348-350)
    IEKF_TIMEOUT = 5,        // a workgroup's row / the verdict did not arrive (the host repeats the solve per iteration)
    IEKF_PREFIX_SHORT = 6,   // finite max_num_residuals: the keypoint prefix held fewer accepted residuals (host repeats)
    IEKF_SINGULAR = 7        // a zero pivot column (srl::inverse_cols returns false; the host decides what that means)
};

struct IekfConsts {          // constant over one solve
    double pred[19];         // state at loop entry: p(3) q(wxyz) v(3) ba(3) bg(3) g(3)  (optimize.cpp:138-143)
    double laser_point_cov;  // lioOptimization.h:221
    double thr_translation;  // icpOptions::threshold_translation_norm
    double thr_orientation;  // icpOptions::threshold_orientation_norm
    int frame_id;            // convergence rule only from the third frame on (optimize.cpp:265)
    int max_num_iter;        // optimize.cpp:135-136
};

// what lives in LDS next to the wave (host emulation: plain memory).  All matrices row-major [i * 17 + j].
struct IekfShared {
    double cov[289];         // the projected covariance of this iteration (optimize.cpp:220-232)
    double temp[289];        // (covariance / laser_point_cov)^-1 (optimize.cpp:234), then scratch
    double scr[2 * 289];     // LU rows / right-hand sides on their way to the backward substitution
    double state[19];        // eskfEstimator's p q v ba bg g (the filter itself lives here during a solve)
    double d_x_new[17];      // projected prior error state (optimize.cpp:213-218)
    double HtH[36], Hth[6];  // normal equations of this iteration (filled by the reduction)
    double d_x[17];          // last step (log / posterior)
    double F[17 * 6];        // fast form: covariance[:, 0:6] * covariance[0:6, 0:6]^-1
    double RG[36];           // fast form: laser_point_cov * covariance[0:6, 0:6]^-1
    int singular;
    int observed;            // observe() calls so far (optimize.cpp:253): the frame's pose is the filter's from the first one on
    int passes;              // passes that delivered normal equations (the kernel's loop bookkeeping)
};

// ---------------------------------------------------------------------------------------------------------------------
// wave backends
// ---------------------------------------------------------------------------------------------------------------------
#if defined(__HIP__)
struct DevWave {
    typedef double VD;
    typedef int VI;
    typedef bool VB;
    static __device__ __forceinline__ VI lane() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0)); }
    static __device__ __forceinline__ VD splat(double x) { return x; }
    static __device__ __forceinline__ VI spl_i(int x) { return x; }
    // value of lane `src` (wave-uniform index) in every lane: two v_readlane_b32
    static __device__ __forceinline__ double bcast(VD v, int src) {
        const long long b = __double_as_longlong(v);
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, src);
        const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), src);
        return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    }
    static __device__ __forceinline__ int bcast_i(VI v, int src) { return __builtin_amdgcn_readlane(v, src); }
    // maximum over lanes 0..16 of non-negative values (lanes >= 17 hold -1): quad butterflies and the two mirrors bring the
    // maximum of lanes 0..15 into every lane of the row, lane 16 is read directly
    template <int CTRL>
    static __device__ __forceinline__ double dpp(double v) {
        const long long b = __double_as_longlong(v);
        const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, 0xF, 0xF, false);
        const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), CTRL, 0xF, 0xF, false);
        return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    }
    static __device__ __forceinline__ double vmax(double a, double b) {      // one v_max_f64 (no NaN canonicalisation around it)
        double r;
        asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
        return r;
    }
    static __device__ __forceinline__ double wave_max17(VD v) {
        v = vmax(v, dpp<0xB1>(v));      // quad_perm [1,0,3,2]
        v = vmax(v, dpp<0x4E>(v));      // quad_perm [2,3,0,1]
        v = vmax(v, dpp<0x141>(v));     // row_half_mirror
        v = vmax(v, dpp<0x140>(v));     // row_mirror
        return vmax(bcast(v, 0), bcast(v, 16));
    }
    // Register pressure: every loop below is fully unrolled and nothing orders the LDS reads / broadcasts of a later row
    // behind the arithmetic of an earlier one -- left alone the scheduler issues all of them first and spills.  `after`
    // makes a pointer / value opaque and dependent on a result of the step before: what is read through it stays behind.
    static __device__ __forceinline__ void fence() { asm volatile("" ::: "memory"); }
    static __device__ __forceinline__ const double *after(const double *p, VD dep) { asm volatile("" : "+v"(p) : "v"(dep)); return p; }
    static __device__ __forceinline__ VD after_v(VD v, VD dep) { asm volatile("" : "+v"(v) : "v"(dep)); return v; }
    static __device__ __forceinline__ unsigned long long ballot(VB p) { return __ballot(p); }
    static __device__ __forceinline__ VD sel(VB c, VD a, VD b) { return c ? a : b; }
    static __device__ __forceinline__ VI sel_i(VB c, VI a, VI b) { return c ? a : b; }
    static __device__ __forceinline__ VD vabs(VD a) { return fabs(a); }
    // body(a, b) for the lanes of m only (v_readlane inside it reads a lane's register whatever the exec mask says)
    template <int NA, int NB, class F>
    static __device__ __forceinline__ void masked(VB m, VD (&a)[NA], VD (&b)[NB], F &&body) { if (m) body(a, b); }
    static __device__ __forceinline__ VD ld(const double *base, VI idx) { return base[idx]; }
    static __device__ __forceinline__ void st(double *base, VI idx, VD v, VB m) { if (m) base[idx] = v; }
    static __device__ __forceinline__ double ldu(const double *p) { return *p; }      // wave-uniform address
    static __device__ __forceinline__ void barrier() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
};
#endif

struct HostWave {
    struct VD { double v[64]; };
    struct VI { int v[64]; };
    struct VB { bool v[64]; };
    static VI lane() { VI r; for (int l = 0; l < 64; l++) r.v[l] = l; return r; }
    static VD splat(double x) { VD r; for (int l = 0; l < 64; l++) r.v[l] = x; return r; }
    static VI spl_i(int x) { VI r; for (int l = 0; l < 64; l++) r.v[l] = x; return r; }
    static double bcast(const VD &v, int src) { return v.v[src]; }
    static int bcast_i(const VI &v, int src) { return v.v[src]; }
    static double wave_max17(const VD &v) { double m = v.v[0]; for (int l = 1; l < 17; l++) m = v.v[l] > m ? v.v[l] : m; return m; }
    static unsigned long long ballot(const VB &p) { unsigned long long m = 0; for (int l = 0; l < 64; l++) if (p.v[l]) m |= 1ull << l; return m; }
    static VD sel(const VB &c, const VD &a, const VD &b) { VD r; for (int l = 0; l < 64; l++) r.v[l] = c.v[l] ? a.v[l] : b.v[l]; return r; }
    static VI sel_i(const VB &c, const VI &a, const VI &b) { VI r; for (int l = 0; l < 64; l++) r.v[l] = c.v[l] ? a.v[l] : b.v[l]; return r; }
    static VD vabs(const VD &a) { VD r; for (int l = 0; l < 64; l++) r.v[l] = std::fabs(a.v[l]); return r; }
    // the emulation runs the body on every lane and keeps the results of the lanes of m
    template <int NA, int NB, class F>
    static void masked(const VB &m, VD (&a)[NA], VD (&b)[NB], F &&body) {
        VD sa[NA], sb[NB];
        for (int i = 0; i < NA; i++) sa[i] = a[i];
        for (int i = 0; i < NB; i++) sb[i] = b[i];
        body(a, b);
        for (int i = 0; i < NA; i++) a[i] = sel(m, a[i], sa[i]);
        for (int i = 0; i < NB; i++) b[i] = sel(m, b[i], sb[i]);
    }
    static VD ld(const double *base, const VI &idx) { VD r; for (int l = 0; l < 64; l++) r.v[l] = base[idx.v[l]]; return r; }
    static void st(double *base, const VI &idx, const VD &v, const VB &m) { for (int l = 0; l < 64; l++) if (m.v[l]) base[idx.v[l]] = v.v[l]; }
    static double ldu(const double *p) { return *p; }
    static void barrier() {}
    static void fence() {}
    static const double *after(const double *p, const VD &) { return p; }
    static VD after_v(const VD &v, const VD &) { return v; }
};
#define SRLW_BIN(op)                                                                                                                   \
    inline HostWave::VD operator op(const HostWave::VD &a, const HostWave::VD &b) { HostWave::VD r; for (int l = 0; l < 64; l++) r.v[l] = a.v[l] op b.v[l]; return r; } \
    inline HostWave::VD operator op(const HostWave::VD &a, double b) { HostWave::VD r; for (int l = 0; l < 64; l++) r.v[l] = a.v[l] op b; return r; }               \
    inline HostWave::VD operator op(double a, const HostWave::VD &b) { HostWave::VD r; for (int l = 0; l < 64; l++) r.v[l] = a op b.v[l]; return r; }
SRLW_BIN(+) SRLW_BIN(-) SRLW_BIN(*) SRLW_BIN(/)
#undef SRLW_BIN
inline HostWave::VD operator-(const HostWave::VD &a) { HostWave::VD r; for (int l = 0; l < 64; l++) r.v[l] = -a.v[l]; return r; }
#define SRLW_CMP(op)                                                                                                                   \
    inline HostWave::VB operator op(const HostWave::VD &a, double b) { HostWave::VB r; for (int l = 0; l < 64; l++) r.v[l] = a.v[l] op b; return r; } \
    inline HostWave::VB operator op(const HostWave::VI &a, int b) { HostWave::VB r; for (int l = 0; l < 64; l++) r.v[l] = a.v[l] op b; return r; }
SRLW_CMP(==) SRLW_CMP(>) SRLW_CMP(<) SRLW_CMP(>=)
#undef SRLW_CMP
inline HostWave::VB operator&&(const HostWave::VB &a, const HostWave::VB &b) { HostWave::VB r; for (int l = 0; l < 64; l++) r.v[l] = a.v[l] && b.v[l]; return r; }
inline HostWave::VI operator*(const HostWave::VI &a, int b) { HostWave::VI r; for (int l = 0; l < 64; l++) r.v[l] = a.v[l] * b; return r; }
inline HostWave::VI operator+(const HostWave::VI &a, int b) { HostWave::VI r; for (int l = 0; l < 64; l++) r.v[l] = a.v[l] + b; return r; }

// ---------------------------------------------------------------------------------------------------------------------
// Matrix<double,17,17>::inverse() restated as srl::inverse_cols<17, M> (host/srl_la.h:139-182), one wave.
//   in : r[j]  lane i < 17 = A(i, j)
//   out: x[i]  lane c < M  = A^-1(i, c)
// Returns false when a pivot column is exactly zero (the host function returns false there too).
// ---------------------------------------------------------------------------------------------------------------------
template <class W, int M>
SRL_HD inline bool wave_inverse_cols(typename W::VD (&r)[17], double *sh_u, double *sh_y, typename W::VD (&x)[17]) {
    typedef typename W::VD VD;
    typedef typename W::VI VI;
    typedef typename W::VB VB;
    const VI lane = W::lane();
    const VB row = lane < 17;
    VI pos = W::sel_i(row, lane, W::spl_i(1000));         // position of this lane's row in the permuted matrix
    VD y[M];                                              // right-hand sides P e_c, rows travel with their lane
#pragma unroll
    for (int c = 0; c < M; c++) y[c] = W::sel(lane == c, W::splat(1.0), W::splat(0.0));
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 17; k++) {
        // pivot: the largest |lu[i][k]| over positions i >= k, the first one in position order on a tie (strict `>` scan)
        const VB elig = row && (pos >= k);
        const VD mine = W::sel(elig, W::vabs(r[k]), W::splat(-1.0));
        const double best = W::wave_max17(mine);
        if (best == 0.0) ok = false;
        unsigned long long cand = W::ballot(elig && (mine == best));
        int p_lane = 0, piv = 1000;
        if ((cand & (cand - 1ull)) == 0ull) {
            p_lane = cand ? (int)__builtin_ctzll(cand) : k;
            piv = W::bcast_i(pos, p_lane);
        } else {
            while (cand) {                                // exact tie of magnitudes: the smallest position wins
                const int l = (int)__builtin_ctzll(cand);
                cand &= cand - 1ull;
                const int pl = W::bcast_i(pos, l);
                if (pl < piv) { piv = pl; p_lane = l; }
            }
        }
        const unsigned long long at_k = W::ballot(row && (pos == k));
        const int q_lane = at_k ? (int)__builtin_ctzll(at_k) : p_lane;
        // rows k and piv change places (host: the rows and perm[] are swapped): here only the positions move
        pos = W::sel_i(lane == p_lane, W::spl_i(k), W::sel_i(lane == q_lane, W::spl_i(piv), pos));
        // rows below the pivot (the pivot lane itself is not among them: its registers are only read)
        const VB upd = row && (pos > k);
        W::masked(upd, r, y, [&](VD (&rr)[17], VD (&yy)[M]) {
            const VD f = rr[k] / W::bcast(rr[k], p_lane);             // lu[i][k] /= lu[k][k]
            rr[k] = f;
#pragma unroll
            for (int j = k + 1; j < 17; j++) rr[j] = rr[j] - f * W::bcast(rr[j], p_lane);      // lu[i][j] -= f * lu[k][j]
#pragma unroll
            for (int c = 0; c < M; c++) yy[c] = yy[c] - f * W::bcast(yy[c], p_lane);          // Y[i][c] -= lu[i][k] * Y[k][c]
        });
        W::fence();
    }
    // U and the forward-substituted right-hand sides by position, then one lane per right-hand side
    const VI base = pos * 17;
#pragma unroll
    for (int j = 0; j < 17; j++) W::st(sh_u, base + j, r[j], row);
#pragma unroll
    for (int c = 0; c < M; c++) W::st(sh_y, base + c, y[c], row);
    W::barrier();
    const VB col = lane < M;
    const VI cl = W::sel_i(col, lane, W::spl_i(0));
#pragma unroll
    for (int i = 0; i < 17; i++) x[i] = W::ld(sh_y, cl + i * 17);
#pragma unroll
    for (int i = 16; i >= 0; i--) {
        const double *urow = W::after(sh_u + i * 17, x[i < 16 ? i + 1 : 16]);      // row i of U is read once row i + 1 is done
#pragma unroll
        for (int j = i + 1; j < 17; j++) x[i] = x[i] - W::ldu(urow + j) * x[j];
        x[i] = x[i] / W::ldu(urow + i);
    }
    W::barrier();
    return ok;
}

// dst rows [row0, row0 + 3) of the first ncols columns = J (3 x 3) * the same rows of src (optimize.cpp:220,296: one
// 3-vector product per column); lane i holds row i.  Other rows of dst are left alone.
template <class W, int NC>
SRL_HD inline void wave_left3(typename W::VD *dst, const srl::Mat3 &J, const typename W::VD *src, int row0) {
    typedef typename W::VD VD;
    const typename W::VI lane = W::lane();
    const typename W::VB m0 = lane == row0, m1 = lane == row0 + 1;
    const VD j0 = W::sel(m0, W::splat(J(0, 0)), W::sel(m1, W::splat(J(1, 0)), W::splat(J(2, 0))));
    const VD j1 = W::sel(m0, W::splat(J(0, 1)), W::sel(m1, W::splat(J(1, 1)), W::splat(J(2, 1))));
    const VD j2 = W::sel(m0, W::splat(J(0, 2)), W::sel(m1, W::splat(J(1, 2)), W::splat(J(2, 2))));
    const typename W::VB mine = (lane >= row0) && (lane < row0 + 3);
    VD prev = j0;
#pragma unroll
    for (int j = 0; j < NC; j++) {
        const VD sj = W::after_v(src[j], prev);
        const double b0 = W::bcast(sj, row0), b1 = W::bcast(sj, row0 + 1), b2 = W::bcast(sj, row0 + 2);
        const VD v = (j0 * b0 + j1 * b1) + j2 * b2;
        dst[j] = W::sel(mine, v, dst[j]);
        prev = v;
    }
}
template <class W, int NC>
SRL_HD inline void wave_left2(typename W::VD *dst, const srl::Mat2 &J, const typename W::VD *src, int row0) {
    typedef typename W::VD VD;
    const typename W::VI lane = W::lane();
    const typename W::VB m0 = lane == row0;
    const VD j0 = W::sel(m0, W::splat(J(0, 0)), W::splat(J(1, 0)));
    const VD j1 = W::sel(m0, W::splat(J(0, 1)), W::splat(J(1, 1)));
    const typename W::VB mine = (lane >= row0) && (lane < row0 + 2);
    VD prev = j0;
#pragma unroll
    for (int j = 0; j < NC; j++) {
        const VD sj = W::after_v(src[j], prev);
        const double b0 = W::bcast(sj, row0), b1 = W::bcast(sj, row0 + 1);
        const VD v = j0 * b0 + j1 * b1;
        dst[j] = W::sel(mine, v, dst[j]);
        prev = v;
    }
}
// dst(i, c0 .. c0+2) = src(i, c0 .. c0+2) * J^T for every row (optimize.cpp:222,300): in-lane
template <class W>
SRL_HD inline void wave_right3(typename W::VD *dst, const srl::Mat3 &J, const typename W::VD *src, int c0) {
    typedef typename W::VD VD;
    const VD s0 = src[c0], s1 = src[c0 + 1], s2 = src[c0 + 2];
#pragma unroll
    for (int c = 0; c < 3; c++) dst[c0 + c] = (s0 * J(c, 0) + s1 * J(c, 1)) + s2 * J(c, 2);
}
template <class W>
SRL_HD inline void wave_right2(typename W::VD *dst, const srl::Mat2 &J, const typename W::VD *src, int c0) {
    typedef typename W::VD VD;
    const VD s0 = src[c0], s1 = src[c0 + 1];
#pragma unroll
    for (int c = 0; c < 2; c++) dst[c0 + c] = s0 * J(c, 0) + s1 * J(c, 1);
}

// ---- the FAST form of the second inverse ------------------------------------------------------------------------------
// updateIEKF only ever reads temp_inv.block<17,6>(0,0) of temp = (C / R)^-1 with temp[0:6,0:6] += H (optimize.cpp:234-242;
// C = projected covariance, R = laser_point_cov, H = H_x^T H_x).  By the block-inverse (Schur complement) identity
//     temp_inv[:, 0:6] = F * (R * G + H)^-1,        G = C66^-1,   F = C[:, 0:6] * G        (C66 = C[0:6, 0:6])
// -- F and R G do not depend on H and are prepared while the sweep is associated; behind the reduction only a symmetric
// positive definite 6 x 6 system is left (Cholesky, no pivoting), ~1/10 of the instructions of the 17 x 17 LU.  Same
// matrix, different operation order: agrees with the LU form to ~1e-10 relative (tests/test_iekf_wave.py bounds it), which
// is why the LU form stays as the pinned reference form (bitwise equal to the host mirror) and this one is a mode.
//
// Cholesky factor of the SPD 6 x 6 matrix whose row i sits in a[0..5] of lane i < 6 (lower triangle read):
// L(c, k), c >= k, packed at c (c + 1) / 2 + k, and 1 / L(c, c) -- wave-uniform values.
template <class W>
SRL_HD inline void wave_chol6(typename W::VD (&a)[6], double (&L)[21], double (&inv_d)[6]) {
    typedef typename W::VD VD;
#pragma unroll
    for (int j = 0; j < 6; j++) {
        const double d = std::sqrt(W::bcast(a[j], j));
        const double id = 1.0 / d;
        L[j * (j + 1) / 2 + j] = d;
        inv_d[j] = id;
        const VD lij = a[j] * id;                          // column j of L (rows i > j)
#pragma unroll
        for (int c = j + 1; c < 6; c++) {
            const double lcj = W::bcast(lij, c);
            L[c * (c + 1) / 2 + j] = lcj;
            a[c] = a[c] - lij * lcj;                       // trailing update of column c (rows i >= c are read later)
        }
    }
}
// z (L L^T) = f for one right-hand side per lane (row vectors of 6)
template <class W>
SRL_HD inline void wave_solve6(const double (&L)[21], const double (&inv_d)[6], const typename W::VD (&f)[6], typename W::VD (&z)[6]) {
    typedef typename W::VD VD;
    VD w[6];
#pragma unroll
    for (int c = 0; c < 6; c++) {
        VD s = f[c];
#pragma unroll
        for (int k = 0; k < c; k++) s = s - w[k] * L[c * (c + 1) / 2 + k];
        w[c] = s * inv_d[c];
    }
#pragma unroll
    for (int c = 5; c >= 0; c--) {
        VD s = w[c];
#pragma unroll
        for (int k = c + 1; k < 6; k++) s = s - z[k] * L[k * (k + 1) / 2 + c];
        z[c] = s * inv_d[c];
    }
}

struct IekfState {
    srl::Vec3 p, v, ba, bg, g;
    srl::Quat q;
    SRL_HD void load(const double *s) {
        p = srl::vec3(s[0], s[1], s[2]); q = srl::Quat(s[3], s[4], s[5], s[6]); v = srl::vec3(s[7], s[8], s[9]);
        ba = srl::vec3(s[10], s[11], s[12]); bg = srl::vec3(s[13], s[14], s[15]); g = srl::vec3(s[16], s[17], s[18]);
    }
    SRL_HD void store(double *s) const {
        for (int a = 0; a < 3; a++) { s[a] = p[a]; s[7 + a] = v[a]; s[10 + a] = ba[a]; s[13 + a] = bg[a]; s[16 + a] = g[a]; }
        s[3] = q.w; s[4] = q.x; s[5] = q.y; s[6] = q.z;
    }
    // eskfEstimator::observe (src/eskfEstimator.cpp:219-230)
    SRL_HD void observe(const double *d) {
        using srlivo::numType;
        p = p + srl::vec3(d[0], d[1], d[2]);
        q = (q * numType::so3ToQuat(srl::vec3(d[3], d[4], d[5]))).normalized();
        v = v + srl::vec3(d[6], d[7], d[8]);
        ba = ba + srl::vec3(d[9], d[10], d[11]);
        bg = bg + srl::vec3(d[12], d[13], d[14]);
        const srl::Mat32 B_x = numType::derivativeS2(g);
        srl::Vec2 dg;
        dg[0] = d[15];
        dg[1] = d[16];
        const srl::Vec3 so3_dg = B_x * dg;
        g = numType::so3ToRotation(so3_dg) * g;
    }
};

// ---------------------------------------------------------------------------------------------------------------------
// prior(): optimize.cpp:172-234 (host/lioOptimization.cpp:251-306).  P = covariance at loop entry, row-major (any memory).
// Leaves sh.cov (projected covariance), sh.temp = (covariance / R)^-1, sh.d_x_new.
// ---------------------------------------------------------------------------------------------------------------------
template <class W, bool FAST = false>
SRL_HD inline void iekf_prior(const IekfConsts &K, const double *P, IekfShared &sh) {
    typedef typename W::VD VD;
    typedef typename W::VI VI;
    using srl::Mat2; using srl::Mat3; using srl::Mat32; using srl::Quat; using srl::Vec2; using srl::Vec3;
    using srlivo::numType;
    IekfState cur, pre;
    cur.load(sh.state);
    pre.load(K.pred);
    // prior error state (optimize.cpp:172-211)
    const Vec3 d_p = cur.p - pre.p;
    const Quat d_q = pre.q.inverse() * cur.q;
    const Vec3 d_so3 = numType::quatToSo3(d_q);
    const Vec3 d_v = cur.v - pre.v;
    const Vec3 d_ba = cur.ba - pre.ba;
    const Vec3 d_bg = cur.bg - pre.bg;
    Vec3 g_predict_normalize = pre.g;
    Vec3 g_normalize = cur.g;
    g_predict_normalize.normalize();
    g_normalize.normalize();
    const Vec3 crs = srl::cross(g_predict_normalize, g_normalize);
    const double dotv = g_predict_normalize.dot(g_normalize);
    Mat3 R_dg;
    if (std::fabs(1.0 - dotv) < 1e-6) R_dg = Mat3::Identity();
    else {
        const Mat3 skew = numType::skewSymmetric(crs);
        R_dg = Mat3::Identity() + skew + ((skew * skew) * (1.0 - dotv)) / (crs[0] * crs[0] + crs[1] * crs[1] + crs[2] * crs[2]);
    }
    const Vec3 so3_dg = numType::rotationToSo3(R_dg);
    const Mat32 B_x_predict = numType::derivativeS2(pre.g);
    const Vec2 d_g = B_x_predict.transpose() * so3_dg;
    double d_x[17];
    for (int a = 0; a < 3; a++) { d_x[a] = d_p[a]; d_x[3 + a] = d_so3[a]; d_x[6 + a] = d_v[a]; d_x[9 + a] = d_ba[a]; d_x[12 + a] = d_bg[a]; }
    d_x[15] = d_g[0];
    d_x[16] = d_g[1];
    const Mat3 J_k_so3 = Mat3::Identity() - 0.5 * numType::skewSymmetric(d_so3);
    const Mat2 J_k_s2 = Mat2::Identity() + ((0.5 * B_x_predict.transpose()) * numType::skewSymmetric(so3_dg)) * B_x_predict;
    {
        const Vec3 t3 = J_k_so3 * d_so3;
        const Vec2 t2 = J_k_s2 * d_g;
        for (int a = 0; a < 17; a++) sh.d_x_new[a] = d_x[a];
        for (int a = 0; a < 3; a++) sh.d_x_new[3 + a] = t3[a];
        sh.d_x_new[15] = t2[0];
        sh.d_x_new[16] = t2[1];
    }
    // covariance projection (optimize.cpp:220-232): rows, then columns, of the so3 / S2 blocks
    const VI lane = W::lane();
    const typename W::VB row = lane < 17;
    const VI rbase = W::sel_i(row, lane, W::spl_i(0)) * 17;
    VD c[17];
#pragma unroll
    for (int j = 0; j < 17; j++) c[j] = W::ld(P, rbase + j);
    wave_left3<W, 17>(c, J_k_so3, c, 3);
    wave_left2<W, 17>(c, J_k_s2, c, 15);
    wave_right3<W>(c, J_k_so3, c, 3);
    wave_right2<W>(c, J_k_s2, c, 15);
#pragma unroll
    for (int j = 0; j < 17; j++) W::st(sh.cov, rbase + j, c[j], row);
    if constexpr (FAST) {
        // F = C[:, 0:6] C66^-1 and R G = R C66^-1 (see wave_chol6)
        VD a[6], f[6], z[6];
        double L[21], inv_d[6];
#pragma unroll
        for (int j = 0; j < 6; j++) { a[j] = c[j]; f[j] = c[j]; }
        wave_chol6<W>(a, L, inv_d);
        wave_solve6<W>(L, inv_d, f, z);
#pragma unroll
        for (int j = 0; j < 6; j++) W::st(sh.F, W::sel_i(row, lane, W::spl_i(0)) * 6 + j, z[j], row);
#pragma unroll
        for (int j = 0; j < 6; j++) f[j] = W::sel(lane == j, W::splat(K.laser_point_cov), W::splat(0.0));
        wave_solve6<W>(L, inv_d, f, z);
        const typename W::VB top = lane < 6;
#pragma unroll
        for (int j = 0; j < 6; j++) W::st(sh.RG, W::sel_i(top, lane, W::spl_i(0)) * 6 + j, z[j], top);
        W::barrier();
        return;
    }
    // temp = (covariance / laser_point_cov).inverse() (optimize.cpp:234)
#pragma unroll
    for (int j = 0; j < 17; j++) c[j] = c[j] / K.laser_point_cov;
    VD x[17];
    const bool ok = wave_inverse_cols<W, 17>(c, sh.scr, sh.scr + 289, x);
    if (!ok) sh.singular = 1;
    const typename W::VB col = lane < 17;
    const VI cl = W::sel_i(col, lane, W::spl_i(0));
#pragma unroll
    for (int i = 0; i < 17; i++) W::st(sh.temp, cl + i * 17, x[i], col);
    W::barrier();
}

// ---------------------------------------------------------------------------------------------------------------------
// update(): optimize.cpp:235-310 (host/lioOptimization.cpp:332-390) with sh.HtH / sh.Hth of this iteration.
// iter = the reference's loop index i + 1 (0 .. max_num_iter).  Returns IEKF_CONTINUE / IEKF_DONE / IEKF_DONE_NO_COV;
// on IEKF_DONE cov_out (row-major, any memory) receives the posterior covariance.  sh.state is the filter.
// ---------------------------------------------------------------------------------------------------------------------
template <class W, bool FAST = false>
SRL_HD inline int iekf_update(const IekfConsts &K, int iter, IekfShared &sh, double *cov_out) {
    typedef typename W::VD VD;
    typedef typename W::VI VI;
    typedef typename W::VB VB;
    using srl::Mat2; using srl::Mat3; using srl::Mat32; using srl::Vec2; using srl::Vec3;
    using srlivo::numType;
    const VI lane = W::lane();
    const VB row = lane < 17;
    const VI rl = W::sel_i(row, lane, W::spl_i(0));
    const VI rbase = rl * 17;
    VD tl[6];                // row `lane` of temp_inv.block<17,6>(0,0)
    if constexpr (FAST) {
        // (R G + H) = L L^T, then temp_inv[i, 0:6] = F[i, :] (R G + H)^-1
        VD a[6], f[6];
        const VB top = lane < 6;
        const VI hb = W::sel_i(top, lane, W::spl_i(0)) * 6;
#pragma unroll
        for (int j = 0; j < 6; j++) { a[j] = W::ld(sh.RG, hb + j) + W::ld(sh.HtH, hb + j); f[j] = W::ld(sh.F, rl * 6 + j); }
        double L[21], inv_d[6];
        wave_chol6<W>(a, L, inv_d);
        wave_solve6<W>(L, inv_d, f, tl);
    } else {
        // temp.block<6,6>(0,0) += H^T H (optimize.cpp:235-236), temp_inv.block<17,6>(0,0) (optimize.cpp:237)
        VD t[17];
    #pragma unroll
        for (int j = 0; j < 17; j++) t[j] = W::ld(sh.temp, rbase + j);
        {
            const VB top = lane < 6;
            const VI hb = W::sel_i(top, lane, W::spl_i(0)) * 6;
    #pragma unroll
            for (int j = 0; j < 6; j++) t[j] = W::sel(top, t[j] + W::ld(sh.HtH, hb + j), t[j]);
        }
        VD x[17];
        const bool ok = wave_inverse_cols<W, 6>(t, sh.scr, sh.scr + 289, x);
        if (!ok) sh.singular = 1;
        {
            const VB col = lane < 6;
            const VI cl = W::sel_i(col, lane, W::spl_i(0));
    #pragma unroll
            for (int i = 0; i < 17; i++) W::st(sh.scr, cl + i * 6, x[i], col);
            W::barrier();
        }
    #pragma unroll
        for (int k = 0; k < 6; k++) tl[k] = W::ld(sh.scr, rl * 6 + k);
        W::barrier();
    }
    // K_h = temp_inv.block<17,6>(0,0) * H^T h ; K_x.block<17,6>(0,0) = temp_inv.block<17,6>(0,0) * H^T H (optimize.cpp:239-242)
    VD kh = tl[0] * W::ldu(sh.Hth + 0);
#pragma unroll
    for (int k = 1; k < 6; k++) kh = kh + tl[k] * W::ldu(sh.Hth + k);
    VD kx[6];
#pragma unroll
    for (int j = 0; j < 6; j++) {
        const double *hcol = W::after(sh.HtH + j, j ? kx[j ? j - 1 : 0] : kh);
        VD s = tl[0] * W::ldu(hcol);
#pragma unroll
        for (int k = 1; k < 6; k++) s = s + tl[k] * W::ldu(hcol + k * 6);
        kx[j] = s;
    }
    // d_x = - K_h + (K_x - I) * d_x_new (optimize.cpp:244)
    VD dxv;
    {
        VD s = (kx[0] - W::sel(lane == 0, W::splat(1.0), W::splat(0.0))) * W::ldu(sh.d_x_new + 0);
#pragma unroll
        for (int j = 1; j < 17; j++) {
            const VD ident = W::sel(lane == j, W::splat(1.0), W::splat(0.0));
            const VD km = (j < 6 ? kx[j < 6 ? j : 0] : W::splat(0.0)) - ident;
            s = s + km * W::ldu(sh.d_x_new + j);
        }
        dxv = (-kh) + s;
    }
    double d_x[17];
#pragma unroll
    for (int a = 0; a < 17; a++) { d_x[a] = W::bcast(dxv, a); sh.d_x[a] = d_x[a]; }
    if constexpr (FAST) {
        // R G + H not positive definite (never for a covariance and a Gram matrix; NaN input): the square roots say so
        bool bad = false;
        for (int a = 0; a < 17; a++) bad = bad || !(d_x[a] == d_x[a]);
        if (bad) sh.singular = 1;
    }

    IekfState st;
    st.load(sh.state);
    const Vec3 g_before = st.g;
    const Vec3 dx_p = srl::vec3(d_x[0], d_x[1], d_x[2]);
    const Vec3 dx_r = srl::vec3(d_x[3], d_x[4], d_x[5]);
    const bool last = iter == K.max_num_iter;                                     // the reference's i == max_num_iter - 1
    if (dx_p.norm() > 100.0 || srlivo::AngularDistance(dx_r) > 100.0)              // optimize.cpp:248-251
        return last ? IEKF_DONE_NO_COV : IEKF_CONTINUE;
    st.observe(d_x);                                                              // optimize.cpp:253
    st.store(sh.state);
    sh.observed = sh.observed + 1;
    bool converage = false;
    if (K.frame_id > 1 && dx_p.norm() < K.thr_translation && srlivo::AngularDistance(dx_r) < K.thr_orientation) converage = true;
    if (!(converage || last)) return IEKF_CONTINUE;

    // posterior covariance (optimize.cpp:272-310)
    const Mat32 B_x_before = numType::derivativeS2(g_before);
    Vec2 dg2;
    dg2[0] = d_x[15];
    dg2[1] = d_x[16];
    const Mat3 J_k_so3 = Mat3::Identity() - 0.5 * numType::skewSymmetric(dx_r);
    const Mat2 J_k_s2 = Mat2::Identity() + ((0.5 * B_x_before.transpose()) * numType::skewSymmetric(B_x_before * dg2)) * B_x_before;
    VD cov[17], cnew[17];
#pragma unroll
    for (int j = 0; j < 17; j++) { cov[j] = W::ld(sh.cov, rbase + j); cnew[j] = cov[j]; }
    wave_left3<W, 17>(cnew, J_k_so3, cov, 3);
    wave_left2<W, 17>(cnew, J_k_s2, cov, 15);
    { VD s[17];
#pragma unroll
      for (int j = 0; j < 17; j++) s[j] = cov[j];
      wave_right3<W>(cnew, J_k_so3, s, 3); wave_right3<W>(cov, J_k_so3, s, 3); }
    { VD s[17];
#pragma unroll
      for (int j = 0; j < 17; j++) s[j] = cov[j];
      wave_right2<W>(cnew, J_k_s2, s, 15); wave_right2<W>(cov, J_k_s2, s, 15); }
    wave_left3<W, 6>(kx, J_k_so3, kx, 3);
    wave_left2<W, 6>(kx, J_k_s2, kx, 15);
    // covariance = covariance_new - K_x.block<17,6>(0,0) * covariance.block<6,17>(0,0)
    VD prev = kx[0];
#pragma unroll
    for (int j = 0; j < 17; j++) {
        const VD cj = W::after_v(cov[j], prev);            // column j's six broadcasts stay behind column j - 1's result
        VD s = kx[0] * W::bcast(cj, 0);
#pragma unroll
        for (int k = 1; k < 6; k++) s = s + kx[k] * W::bcast(cj, k);
        prev = cnew[j] - s;
        W::st(cov_out, rbase + j, prev, row);
    }
    W::barrier();
    return IEKF_DONE;
}

}  // namespace srlw
