// srl_frame_kernels.hip -- frame-resident pipeline (SURVEY 8(f) rows 1+2): the reconstructed sweep stays in HBM
// from keypoint selection to map insertion.
//   srl_frame_upload            p_frame->point_frame raw points -> HBM (once per sweep)
//   srl_frame_select_keypoints  gridSampling / subSampleFrame (src/utility.cpp:167-201): the device transforms the
//                               frame with the prior pose, keys every point at the sampling voxel size, sorts
//                               (key, index) stably and run-length encodes -> first point of every voxel; the
//                               host only replays the ORDER: the reference emits keypoints in
//                               std::tr1::unordered_map iteration order, which depends on the sequence of distinct
//                               keys alone, so the distinct keys are inserted (first-occurrence order) into the same
//                               container type and read back.  The selected raw points are gathered on the device
//                               straight into the resident sweep (no keypoint upload).
//   srl_frame_commit            the re-transform loop of optimize() (optimize.cpp:441-445) + addPointsToMap
//                               (lioOptimization.cpp:520-554) chained on the device.
#include "srl_ctx.h"
#include "srl_frame_scratch.h"
#include "srl_hash.h"
#include "host/srl_la.h"
#include "host/tr1_order.h"


#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <tr1/unordered_map>
#include <vector>

int srl_map_insert_impl(srl_ctx *ctx, const double *world_xyz, bool on_device, int n, double voxel_size,
                        double min_distance_points, int min_num_points, int *num_added, bool defer_counters,
                        const SrlFrameTransform *xf = nullptr, int (*after_first_kernel)(srl_ctx *, void *) = nullptr, void *user = nullptr);

namespace {

using Xf = SrlXf;

// point = R(q) * (R_il * raw + t_il) + t (utility.cpp:314-318), key = short(point / size) (utility.cpp:171-173)
__global__ void k_frame_keys(const double *raw, int n, const Xf X, double size, double *world, unsigned long long *keys, unsigned *idx) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double rx = raw[(size_t)i * 3], ry = raw[(size_t)i * 3 + 1], rz = raw[(size_t)i * 3 + 2];
    const double ix = (X.R_il[0] * rx + X.R_il[1] * ry) + X.R_il[2] * rz + X.t_il[0];
    const double iy = (X.R_il[3] * rx + X.R_il[4] * ry) + X.R_il[5] * rz + X.t_il[1];
    const double iz = (X.R_il[6] * rx + X.R_il[7] * ry) + X.R_il[8] * rz + X.t_il[2];
    const double wx = (X.R[0] * ix + X.R[1] * iy) + X.R[2] * iz + X.t[0];
    const double wy = (X.R[3] * ix + X.R[4] * iy) + X.R[5] * iz + X.t[1];
    const double wz = (X.R[6] * ix + X.R[7] * iy) + X.R[8] * iz + X.t[2];
    if (world) { world[(size_t)i * 3] = wx; world[(size_t)i * 3 + 1] = wy; world[(size_t)i * 3 + 2] = wz; }
    if (keys) {
        keys[i] = srl_pack_key((short)(int)(wx / size), (short)(int)(wy / size), (short)(int)(wz / size));
        idx[i] = (unsigned)i;
    }
}

__global__ void k_first_index(const int *seg_start, const unsigned *sorted_idx, int S, unsigned *first_idx) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < S) first_idx[s] = sorted_idx[seg_start[s]];
}

// run heads of the (key, index)-sorted frame: one {voxel key, index of its first point} pair per occupied voxel, in no
// particular order (the host re-orders by first index anyway); out[0] of `count` = number of pairs
__global__ void k_run_heads(const unsigned long long *keys_sorted, const unsigned *idx_sorted, int n, unsigned long long *out_key,
                            unsigned *out_first, int *count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool head = i < n && (i == 0 || keys_sorted[i] != keys_sorted[i - 1]);
    // wave-aggregated append: one atomic per wave
    const unsigned long long m = __ballot(head);
    if (m == 0ull) return;
    const int lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0));
    const int leader = (int)__builtin_ctzll(m);
    int base = 0;
    if (lane == leader) base = atomicAdd(count, (int)__popcll(m));
    base = __shfl(base, leader);
    if (head) {
        const int p = base + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
        out_key[p] = keys_sorted[i];
        out_first[p] = idx_sorted[i];
    }
}

// gridSampling keeps the FIRST point of every sampling voxel (utility.cpp:175-183): group the points by voxel key in a scratch hash
// table (open addressing, keys claimed by compare-and-swap) and keep the smallest point index per key (atomicMin) -- no sort: the
// order of the voxels is decided on the host anyway (std::tr1::unordered_map iteration order), from the first indices.
__global__ void k_select_group(const double *raw, int n, const Xf X, double size, unsigned long long *keyw, unsigned long long *minw, unsigned mask,
                               unsigned epoch16, unsigned counter32, int *flag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    flag[i] = 0;                           // (k_select_mark, the next kernel, sets the marks: no fill in front of this one)
    const double rx = raw[(size_t)i * 3], ry = raw[(size_t)i * 3 + 1], rz = raw[(size_t)i * 3 + 2];
    const double ix = (X.R_il[0] * rx + X.R_il[1] * ry) + X.R_il[2] * rz + X.t_il[0];
    const double iy = (X.R_il[3] * rx + X.R_il[4] * ry) + X.R_il[5] * rz + X.t_il[1];
    const double iz = (X.R_il[6] * rx + X.R_il[7] * ry) + X.R_il[8] * rz + X.t_il[2];
    const double wx = (X.R[0] * ix + X.R[1] * iy) + X.R[2] * iz + X.t[0];
    const double wy = (X.R[3] * ix + X.R[4] * iy) + X.R[5] * iz + X.t[1];
    const double wz = (X.R[6] * ix + X.R[7] * iy) + X.R[8] * iz + X.t[2];
    const unsigned long long key = srl_pack_key((short)(int)(wx / size), (short)(int)(wy / size), (short)(int)(wz / size));
    const unsigned h = srl_epoch_claim(keyw, mask, epoch16, key, srl_hash_key(key));
    // smallest point index of the voxel: {~frame counter, index}, only ever lowered -- a word of an earlier frame loses against any of this one
    atomicMin(&minw[h], ((unsigned long long)(0xFFFFFFFFu - counter32) << 32) | (unsigned)i);
}
// ... then the voxels in FIRST-OCCURRENCE order (the order subSampleFrame's loop creates them in, utility.cpp:175-183 -- what the
// host's replay of the container's iteration order starts from): every occupied slot marks the index of its first point, an
// exclusive scan over the marks ranks the voxels, and the keys are written out by rank.
__global__ void k_select_mark(const unsigned long long *keyw, const unsigned long long *minw, unsigned cap, unsigned epoch16, int *flag,
                              unsigned long long *key_at, int *bucket_count, unsigned n_buckets) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_buckets) bucket_count[i] = 0;            // (device ordering below: no fill in front of k_tr1_bucket)
    if (i >= cap) return;
    const unsigned long long k = keyw[i];
    if ((unsigned)(k >> 48) != epoch16) return;
    const unsigned f = (unsigned)minw[i];
    flag[f] = 1;
    key_at[f] = k & SRL_KEY48_MASK;
}
// The voxels leave the device HERE: {std::hash<voxel> (cloudMap.h:173-184, what the host's replay of the container needs), index of the
// first point} by rank, stored straight into page-locked host memory; the last block to finish publishes {tag, count} in one 8-byte
// word the host waits on.  No copy command, no stream synchronisation (a D2H copy + hipStreamSynchronize cost ~25 us per frame).

// ... the same hand-over as the per-element work (EmitSink) and the "tile done" step (EmitFin) of the one-launch scan over the marks
// (frames up to SRL_SCAN_SMALL_MAX points): no rank array, no launch of its own
struct EmitSink {
    const unsigned long long *key_at;
    unsigned long long *host_hash;
    unsigned *host_first;
    __device__ void operator()(int i, int is_first, int r) const {
        if (!is_first) return;
        short x, y, z;
        srl_unpack_key(key_at[i], &x, &y, &z);
        const unsigned long long kP1 = 73856093ull, kP2 = 19349669ull, kP3 = 83492791ull;
        host_hash[r] = (unsigned long long)(long long)x * kP1 + (unsigned long long)(long long)y * kP2 + (unsigned long long)(long long)z * kP3;
        host_first[r] = (unsigned)i;
    }
};
struct EmitFin {
    unsigned *sync;                           // [0] workgroups done (reset by the last one), [1] the count (written by the workgroup holding the last tile)
    unsigned long long *host_ctrl;
    unsigned tag;
    __device__ void operator()(int tile_end) const {
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x != 0) return;
        if (blockIdx.x == gridDim.x - 1) __hip_atomic_store(sync + 1, (unsigned)tile_end, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned prev = __hip_atomic_fetch_add(sync, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (prev == gridDim.x - 1) {
            __hip_atomic_store(sync, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned count = __hip_atomic_load(sync + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __threadfence_system();
            __hip_atomic_store(host_ctrl, ((unsigned long long)tag << 32) | count, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
};

// ---------------------------------------------------------------------------- keypoint ORDER on the device
// gridSampling emits the keypoints in the iteration order of a std::tr1::unordered_map (utility.cpp:167-201).  host/tr1_order.h replays
// the container's moves level by level (one counting sort per rehash: ~70 us of host time for the 14k voxels of a 24k-point frame, on the
// critical path between selection and the first pass).  host/tr1_relation.h states the same order as a relation between two voxels, which
// needs no sequential replay: a count per final bucket (k_tr1_bucket), ONE scan over the bucket counts, and per voxel a rank among the
// handful of voxels sharing its bucket (k_tr1_rank, which also gathers the keypoint's raw point into the resident sweep).  One thread
// per VOXEL, not per bucket: a lane that ranked all pairs of a 6-voxel bucket held the kernel for 40 us.  A bucket with more than
// SRL_TR1_BUCKET_SLOTS voxels (adversarial keys) sends the frame to the host replay.
// the ranks stay on the device: {std::hash<voxel>, first point} by first-occurrence rank; the count goes to sync[1]
struct RankSink {
    const unsigned long long *key_at;
    unsigned long long *hash;
    unsigned *first;
    __device__ void operator()(int i, int is_first, int r) const {
        if (!is_first) return;
        short x, y, z;
        srl_unpack_key(key_at[i], &x, &y, &z);
        const unsigned long long kP1 = 73856093ull, kP2 = 19349669ull, kP3 = 83492791ull;       // size_t arithmetic of the reference's hash (cloudMap.h:173-184)
        hash[r] = (unsigned long long)(long long)x * kP1 + (unsigned long long)(long long)y * kP2 + (unsigned long long)(long long)z * kP3;
        first[r] = (unsigned)i;
    }
};
struct CountFin {
    unsigned *sync;
    __device__ void operator()(int tile_end) const {
        if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) sync[1] = (unsigned)tile_end;
    }
};
// per voxel e (first-occurrence rank): bucket at the final level G, and a RECORD in that bucket's member list that carries what nearly
// every comparison of two voxels of one final bucket ends on -- {e, era(e), bucket at level G - 1} -- so that the ranking lane has
// everything after one load of its bucket (the kernels of this chain are a few dependent loads long and nothing else: every level of
// indirection is ~1 us of the frame)
#define SRL_TR1_REC(e, era, bprev) ((unsigned long long)(e) | ((unsigned long long)(era) << 20) | ((unsigned long long)(bprev) << 32))
#define SRL_TR1_REC_E(r) ((unsigned)(r) & 0xFFFFFu)
#define SRL_TR1_REC_ERA(r) (((unsigned)(r) >> 20) & 0x3Fu)
#define SRL_TR1_REC_BPREV(r) ((unsigned)((r) >> 32))
static_assert(SRL_SCAN_SMALL_MAX <= (1 << 20), "a record holds a 20-bit voxel rank");
__global__ void k_tr1_bucket(const SrlTr1Sched *S, unsigned *sync, const unsigned long long *hash, int *bucket_count, unsigned long long *members,
                             unsigned *bucket_of, unsigned long long *host_ctrl, unsigned tag) {
    __shared__ SrlTr1Sched sh;                                   // 50 words: level look-ups walk it per lane
    static_assert(sizeof(SrlTr1Sched) <= 256 * 4, "one word per thread");
    if (threadIdx.x < sizeof(SrlTr1Sched) / 4) reinterpret_cast<unsigned *>(&sh)[threadIdx.x] = reinterpret_cast<const unsigned *>(S)[threadIdx.x];
    __syncthreads();
    const unsigned e = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned total = sync[1];
    if (e < total) {
        const int G = srl_tr1_level(&sh, total);
        if (e == 0) sync[3] = (unsigned)G;
        const unsigned era = (unsigned)srl_tr1_level(&sh, e + 1);
        const unsigned long long h = hash[e];
        const unsigned b = (unsigned)(h % sh.nb[G]);
        const unsigned bprev = G > 0 ? (unsigned)(h % sh.nb[G - 1]) : 0u;
        bucket_of[e] = b;
        const int slot = atomicAdd(&bucket_count[b], 1);
        if (slot < SRL_TR1_BUCKET_SLOTS) members[(size_t)b * SRL_TR1_BUCKET_SLOTS + slot] = SRL_TR1_REC(e, era, bprev);
        else __hip_atomic_store(sync + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // more voxels than k_tr1_rank ranks in place
    }
    // The host needs two things from this chain -- the voxel count and whether a bucket overflowed -- and both are known HERE: the last
    // workgroup publishes {tag, overflow, count}; the scan over the bucket counts and k_tr1_rank run on behind the host's back.
    // (No fence: the overflow marks are agent-scope atomics, acknowledged before the barrier; the host reads nothing but the word itself,
    // so it is stored relaxed -- a system-scope RELEASE writes the whole L2 back: 9 us on this kernel.)
    __syncthreads();
    if (threadIdx.x != 0) return;
    if (__hip_atomic_fetch_add(sync, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) != gridDim.x - 1) return;
    __hip_atomic_store(sync, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned over = __hip_atomic_load(sync + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(sync + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(host_ctrl, ((unsigned long long)tag << 32) | ((unsigned long long)(over ? 1u : 0u) << 31) | total, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
}
// T_G(o, me) from the two records; falls back to the general relation below level G - 1 (two voxels sharing their bucket at two levels)
struct Tr1EraOfSched {
    const SrlTr1Sched *S;
    __device__ int operator()(unsigned e) const { return srl_tr1_level(S, e + 1); }
};
__device__ __forceinline__ bool tr1_rec_before(const SrlTr1Sched *S, const unsigned long long *hash, int G, unsigned long long ro, unsigned long long rme) {
    const unsigned o = SRL_TR1_REC_E(ro), me = SRL_TR1_REC_E(rme);
    if ((int)SRL_TR1_REC_ERA(ro) == G || (int)SRL_TR1_REC_ERA(rme) == G) return me < o;          // T_G(o, me): the larger insertion index first
    // both older: T_G(o, me) = T_{G-1}(me, o)
    const unsigned bm = SRL_TR1_REC_BPREV(rme), bo = SRL_TR1_REC_BPREV(ro);
    if (bm != bo) return bm < bo;
    return srl_tr1_before_t(G - 1, me, o, Tr1EraOfSched{S}, SrlTr1BucketOfHash{S, hash});      // rare (one pair in ~n_{G-1})
}
// one thread per voxel: position = start of its bucket (scan over the bucket counts) + voxels of the bucket that precede it; writes the
// ordered index list and gathers the keypoint's raw point into the resident sweep
__global__ void k_tr1_rank(const SrlTr1Sched *S, unsigned *sync, const unsigned long long *hash, const unsigned *first, const unsigned long long *members,
                           const int *bucket_count, const int *bucket_start, const unsigned *bucket_of, const double *raw,
                           double *x, double *y, double *z, int *sel) {
    const unsigned e = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned total = sync[1];
    if (e < total) {
        const unsigned b = bucket_of[e];
        const unsigned fi = first[e];
        const int G = (int)sync[3];
        // second level of loads, all independent: the bucket's count, start, its first four records, the raw point
        const int c = bucket_count[b];
        const int start = bucket_start[b];
        const unsigned long long *m = members + (size_t)b * SRL_TR1_BUCKET_SLOTS;
        const ulonglong2 r01 = *reinterpret_cast<const ulonglong2 *>(m), r23 = *reinterpret_cast<const ulonglong2 *>(m + 2);
        const double px = raw[(size_t)fi * 3], py = raw[(size_t)fi * 3 + 1], pz = raw[(size_t)fi * 3 + 2];
        if (c <= SRL_TR1_BUCKET_SLOTS) {            // (an overfull bucket: k_tr1_bucket has told the host, which orders the frame itself)
            int rank = 0;
            if (c > 1) {
                // own record: rebuilt from the others' point of view is not possible without its era -> find it among the members
                unsigned long long rme = r01.x;
                if (SRL_TR1_REC_E(r01.y) == e && c > 1) rme = r01.y;
                if (SRL_TR1_REC_E(r23.x) == e && c > 2) rme = r23.x;
                if (SRL_TR1_REC_E(r23.y) == e && c > 3) rme = r23.y;
                for (int i = 4; i < c; ++i) { const unsigned long long r = m[i]; if (SRL_TR1_REC_E(r) == e) rme = r; }
                if (SRL_TR1_REC_E(r01.x) != e && tr1_rec_before(S, hash, G, r01.x, rme)) ++rank;
                if (c > 1 && SRL_TR1_REC_E(r01.y) != e && tr1_rec_before(S, hash, G, r01.y, rme)) ++rank;
                if (c > 2 && SRL_TR1_REC_E(r23.x) != e && tr1_rec_before(S, hash, G, r23.x, rme)) ++rank;
                if (c > 3 && SRL_TR1_REC_E(r23.y) != e && tr1_rec_before(S, hash, G, r23.y, rme)) ++rank;
                for (int i = 4; i < c; ++i) {
                    const unsigned long long r = m[i];
                    if (SRL_TR1_REC_E(r) != e && tr1_rec_before(S, hash, G, r, rme)) ++rank;
                }
            }
            const int k = start + rank;
            sel[k] = (int)fi;
            x[k] = px;
            y[k] = py;
            z[k] = pz;
        }
    }
}

__global__ void k_gather_soa(const double *raw, const int *sel, int m, double *x, double *y, double *z) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= m) return;
    const int i = sel[k];
    x[k] = raw[(size_t)i * 3];
    y[k] = raw[(size_t)i * 3 + 1];
    z[k] = raw[(size_t)i * 3 + 2];
}

struct vkey {
    short x, y, z;
    bool operator==(const vkey &o) const { return x == o.x && y == o.y && z == o.z; }
};
struct vkey_hash {     // std::hash<voxel> (cloudMap.h:173-184)
    std::size_t operator()(const vkey &v) const {
        const size_t kP1 = 73856093, kP2 = 19349669, kP3 = 83492791;
        return v.x * kP1 + v.y * kP2 + v.z * kP3;
    }
};

void fill_xf(Xf &X, const double q[4], const double t[3], const double R_il[9], const double t_il[3]) {
    const srl::Mat3 R = srl::Quat(q[0], q[1], q[2], q[3]).toRotationMatrix();     // q as is (utility.cpp:317)
    std::memcpy(X.R, R.a, sizeof X.R);
    std::memcpy(X.t, t, sizeof X.t);
    std::memcpy(X.R_il, R_il, sizeof X.R_il);
    std::memcpy(X.t_il, t_il, sizeof X.t_il);
}

}  // namespace

int srl_ctx_ensure_work(srl_ctx *ctx, int n);   // srl_capi.cpp
bool srl_ctx_is_pinned(const void *p);          // srl_capi.cpp

namespace {

int ensure_frame(srl_ctx *ctx, int n) {
    if (n > ctx->frame_cap) {
        if (ctx->d_frame_raw) HIPCHK(ctx, hipFree(ctx->d_frame_raw));
        if (ctx->d_frame_world) HIPCHK(ctx, hipFree(ctx->d_frame_world));
        ctx->d_frame_raw = ctx->d_frame_world = nullptr;
        ctx->frame_cap = 0;
        const int cap = std::max(n, 4096);
        HIPCHK(ctx, hipMalloc((void **)&ctx->d_frame_raw, (size_t)cap * 3 * sizeof(double)));
        HIPCHK(ctx, hipMalloc((void **)&ctx->d_frame_world, (size_t)cap * 3 * sizeof(double)));
        ctx->frame_cap = cap;
    }
    return SRL_OK;
}

// ---------------------------------------------------------------------------- sweep reconstruction (row f4)
// Per-point stages of buildFrame (lioOptimization.cpp:833-850): distortFrameByConstant / distortFrameByImu
// (utility.cpp:203-306) then transformAllImuPoint (:320-332), one thread per point, FP64, the reference's
// operation order.  sin / cos / acos come from the device math library, so results agree with the CPU to a few
// ulp, not bit for bit.
struct Q4d { double w, x, y, z; };

__device__ inline void d_quat_to_rot(const Q4d &q, double R[9]) {             // Eigen toRotationMatrix
    const double tx = 2.0 * q.x, ty = 2.0 * q.y, tz = 2.0 * q.z;
    const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    R[0] = 1.0 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1.0 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1.0 - (txx + tyy);
}
__device__ inline Q4d d_q_normalized(const Q4d &q) {
    const double z = ((q.x * q.x + q.y * q.y) + q.z * q.z) + q.w * q.w;
    if (z > 0.0) { const double n = sqrt(z); return Q4d{q.w / n, q.x / n, q.y / n, q.z / n}; }
    return q;
}
__device__ inline Q4d d_q_mul(const Q4d &a, const Q4d &b) {
    return Q4d{a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
               a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z, a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x};
}
__device__ inline Q4d d_so3_to_quat(double x, double y, double z) {            // numType::so3ToQuat, utility.h:299-324
    const double theta = sqrt((x * x + y * y) + z * z);
    if (theta < 0.0001) return d_q_normalized(Q4d{1.0, x / 2.0, y / 2.0, z / 2.0});
    const double ux = x / theta, uy = y / theta, uz = z / theta;
    const double s = sin(0.5 * theta);
    return d_q_normalized(Q4d{cos(0.5 * theta), ux * s, uy * s, uz * s});
}
__device__ inline Q4d d_q_slerp(const Q4d &a, double t, const Q4d &b) {       // Eigen QuaternionBase::slerp
    const double one = 1.0 - 2.220446049250313e-16;
    const double d = ((a.x * b.x + a.y * b.y) + a.z * b.z) + a.w * b.w;
    const double absD = fabs(d);
    double scale0, scale1;
    if (absD >= one) { scale0 = 1.0 - t; scale1 = t; }
    else {
        const double theta = acos(absD), sinTheta = sin(theta);
        scale0 = sin((1.0 - t) * theta) / sinTheta;
        scale1 = sin(t * theta) / sinTheta;
    }
    if (d < 0.0) scale1 = -scale1;
    return Q4d{scale0 * a.w + scale1 * b.w, scale0 * a.x + scale1 * b.x, scale0 * a.y + scale1 * b.y, scale0 * a.z + scale1 * b.z};
}
__device__ inline void d_mv(const double R[9], double x, double y, double z, double &ox, double &oy, double &oz) {
    ox = (R[0] * x + R[1] * y) + R[2] * z;
    oy = (R[3] * x + R[4] * y) + R[5] * z;
    oz = (R[6] * x + R[7] * y) + R[8] * z;
}

struct UndistortArgs {
    const double *raw, *rel, *states;      // n x 3, n, n_states x 17
    const int *seg;                        // IMU mode: interval per point, -1 = never reached
    double *imu, *raw_out;                 // n x 3 each; imu is in/out
    int n, mode;
    double tfb, tfe;
    double q0[4], q1[4], tr0[3], tr1[3];
    double R_il[9], t_il[3];
    double Rinv[9], tinv[3], RilT[9], RilT_til[3];
};

__global__ void k_undistort(const UndistortArgs A) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A.n) return;
    const double rx = A.raw[(size_t)i * 3], ry = A.raw[(size_t)i * 3 + 1], rz = A.raw[(size_t)i * 3 + 2];
    double lx, ly, lz;
    d_mv(A.R_il, rx, ry, rz, lx, ly, lz);
    lx += A.t_il[0]; ly += A.t_il[1]; lz += A.t_il[2];
    double px = A.imu[(size_t)i * 3], py = A.imu[(size_t)i * 3 + 1], pz = A.imu[(size_t)i * 3 + 2];
    if (A.mode == SRL_MC_CONSTANT_VELOCITY) {
        double time_point = A.tfb + A.rel[i] / 1000.0;
        if (fabs(time_point - A.tfb) < 1e-6) time_point = A.tfb + 1e-6;
        if (fabs(time_point - A.tfe) < 1e-6) time_point = A.tfe - 1e-6;
        double alpha = (time_point - A.tfb) / (A.tfe - A.tfb);
        if (alpha > 1) alpha = 1;
        if (alpha < 0) alpha = 0;
        const Q4d qa = d_q_slerp(Q4d{A.q0[0], A.q0[1], A.q0[2], A.q0[3]}, alpha, Q4d{A.q1[0], A.q1[1], A.q1[2], A.q1[3]});
        double R[9];
        d_quat_to_rot(qa, R);
        d_mv(R, lx, ly, lz, px, py, pz);
        px += (1.0 - alpha) * A.tr0[0] + alpha * A.tr1[0];
        py += (1.0 - alpha) * A.tr0[1] + alpha * A.tr1[1];
        pz += (1.0 - alpha) * A.tr0[2] + alpha * A.tr1[2];
    } else if (A.mode == SRL_MC_IMU) {
        const int k = A.seg[i];
        if (k >= 0) {
            const double *a = A.states + 17 * (size_t)k, *b = a + 17;
            const double tb = a[0], te = b[0];
            double time_point = A.tfb + A.rel[i] / 1000.0;
            if (fabs(time_point - tb) < 1e-6) time_point = tb + 1e-6;
            if (fabs(time_point - te) < 1e-6) time_point = te - 1e-6;
            const double dt = time_point - tb;
            const Q4d qp = d_q_normalized(d_q_mul(Q4d{a[10], a[11], a[12], a[13]}, d_so3_to_quat(b[4] * dt, b[5] * dt, b[6] * dt)));
            double R[9];
            d_quat_to_rot(qp, R);
            d_mv(R, lx, ly, lz, px, py, pz);
            px += (a[7] + a[14] * dt) + ((0.5 * b[1]) * dt) * dt;
            py += (a[8] + a[15] * dt) + ((0.5 * b[2]) * dt) * dt;
            pz += (a[9] + a[16] * dt) + ((0.5 * b[3]) * dt) * dt;
        }
    }
    A.imu[(size_t)i * 3] = px; A.imu[(size_t)i * 3 + 1] = py; A.imu[(size_t)i * 3 + 2] = pz;
    // transformAllImuPoint
    double ex, ey, ez, ox, oy, oz;
    d_mv(A.Rinv, px, py, pz, ex, ey, ez);
    ex += A.tinv[0]; ey += A.tinv[1]; ez += A.tinv[2];
    d_mv(A.RilT, ex, ey, ez, ox, oy, oz);
    A.raw_out[(size_t)i * 3] = ox - A.RilT_til[0];
    A.raw_out[(size_t)i * 3 + 1] = oy - A.RilT_til[1];
    A.raw_out[(size_t)i * 3 + 2] = oz - A.RilT_til[2];
}

__global__ void k_gather_aos(const double *src, const int *sel, int m, double *dst) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= m) return;
    const int i = sel[k];
    dst[(size_t)k * 3] = src[(size_t)i * 3];
    dst[(size_t)k * 3 + 1] = src[(size_t)i * 3 + 1];
    dst[(size_t)k * 3 + 2] = src[(size_t)i * 3 + 2];
}

}  // namespace

extern "C" {

int srl_frame_undistort(srl_ctx *ctx, const double *raw_xyz, const double *relative_time_ms, const double *imu_point_in, int n,
                        const srl_imu_state *imu_states, int n_states, double time_frame_begin, int motion_compensation,
                        const double R_il[9], const double t_il[3], double *imu_point_out, double *raw_out) {
    if (!ctx || n < 0 || (n > 0 && (!raw_xyz || !relative_time_ms)) || !imu_states || n_states < 1 || !R_il || !t_il)
        return SRL_ERR_BAD_ARG;
    SRL_DISARM(ctx);
    if (motion_compensation != SRL_MC_IMU && motion_compensation != SRL_MC_CONSTANT_VELOCITY && motion_compensation != SRL_MC_NONE)
        return SRL_ERR_BAD_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    ctx->corr_n = -1;
    if (n > ctx->corr_cap) {
        void *bufs[] = {ctx->d_corr_raw, ctx->d_corr_imu, ctx->d_corr_in, ctx->d_corr_rel, ctx->d_corr_seg};
        for (void *b : bufs) if (b) HIPCHK(ctx, hipFree(b));
        ctx->d_corr_raw = ctx->d_corr_imu = ctx->d_corr_in = ctx->d_corr_rel = nullptr; ctx->d_corr_seg = nullptr;
        ctx->corr_cap = 0;
        const int cap = std::max(n, 4096);
        HIPCHK(ctx, hipMalloc((void **)&ctx->d_corr_raw, (size_t)cap * 24));
        HIPCHK(ctx, hipMalloc((void **)&ctx->d_corr_imu, (size_t)cap * 24));
        HIPCHK(ctx, hipMalloc((void **)&ctx->d_corr_in, (size_t)cap * 24));
        HIPCHK(ctx, hipMalloc((void **)&ctx->d_corr_rel, (size_t)cap * 8));
        HIPCHK(ctx, hipMalloc((void **)&ctx->d_corr_seg, (size_t)cap * 4));
        ctx->corr_cap = cap;
    }
    if (n == 0) { ctx->corr_n = 0; return SRL_OK; }
    hipStream_t st = ctx->stream;
    DevBuf b_states;
    HIPCHK(ctx, b_states.alloc(ctx, (size_t)n_states * sizeof(srl_imu_state)));
    static_assert(sizeof(srl_imu_state) == 17 * sizeof(double), "srl_imu_state is 17 packed doubles");
    HIPCHK(ctx, hipMemcpyAsync(b_states.p, imu_states, (size_t)n_states * sizeof(srl_imu_state), hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_corr_in, raw_xyz, (size_t)n * 24, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_corr_rel, relative_time_ms, (size_t)n * 8, hipMemcpyHostToDevice, st));
    if (imu_point_in) HIPCHK(ctx, hipMemcpyAsync(ctx->d_corr_imu, imu_point_in, (size_t)n * 24, hipMemcpyHostToDevice, st));
    else HIPCHK(ctx, hipMemsetAsync(ctx->d_corr_imu, 0, (size_t)n * 24, st));

    std::vector<int> seg;
    if (motion_compensation == SRL_MC_IMU) {
        // the interval walk of distortFrameByImu (utility.cpp:247-305) is sequential in the point order: a point that
        // fits the current interval advances the point cursor, one that does not advances the interval -- and a point
        // no later interval fits stops everything behind it.  Integer control flow on N timestamps: replayed on the
        // host; the per-point math runs on the device.
        seg.assign((size_t)n, -1);
        int iter = 0;
        for (int k = 0; k + 1 < n_states; k++) {
            const double tb = imu_states[k].timestamp, te = imu_states[k + 1].timestamp;
            while (iter != n) {
                const double time_point = time_frame_begin + relative_time_ms[iter] / 1000.0;
                if (time_point > tb - 1e-6 && time_point < te + 1e-6) seg[iter++] = k;
                else break;
            }
        }
        HIPCHK(ctx, hipMemcpyAsync(ctx->d_corr_seg, seg.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
    }

    UndistortArgs A;
    A.raw = ctx->d_corr_in; A.rel = ctx->d_corr_rel; A.states = b_states.as<double>(); A.seg = ctx->d_corr_seg;
    A.imu = ctx->d_corr_imu; A.raw_out = ctx->d_corr_raw; A.n = n; A.mode = motion_compensation;
    A.tfb = time_frame_begin; A.tfe = imu_states[n_states - 1].timestamp;
    const srl_imu_state &s0 = imu_states[0], &s1 = imu_states[n_states - 1];
    for (int d = 0; d < 4; d++) { A.q0[d] = s0.quat[d]; A.q1[d] = s1.quat[d]; }
    for (int d = 0; d < 3; d++) { A.tr0[d] = s0.trans[d]; A.tr1[d] = s1.trans[d]; A.t_il[d] = t_il[d]; }
    std::memcpy(A.R_il, R_il, sizeof A.R_il);
    const srl::Mat3 Rinv = srl::Quat(s1.quat[0], s1.quat[1], s1.quat[2], s1.quat[3]).inverse().toRotationMatrix();
    srl::Mat3 Ril;
    std::memcpy(Ril.a, R_il, sizeof Ril.a);
    const srl::Mat3 RilT = Ril.transpose();
    const srl::Vec3 tinv = (-1.0 * Rinv) * srl::vec3(s1.trans[0], s1.trans[1], s1.trans[2]);
    const srl::Vec3 rt = RilT * srl::vec3(t_il[0], t_il[1], t_il[2]);
    std::memcpy(A.Rinv, Rinv.a, sizeof A.Rinv);
    std::memcpy(A.RilT, RilT.a, sizeof A.RilT);
    for (int d = 0; d < 3; d++) { A.tinv[d] = tinv[d]; A.RilT_til[d] = rt[d]; }
    hipLaunchKernelGGL(k_undistort, dim3((n + 255) / 256), dim3(256), 0, st, A);
    HIPCHK(ctx, hipGetLastError());
    if (imu_point_out) HIPCHK(ctx, hipMemcpyAsync(imu_point_out, ctx->d_corr_imu, (size_t)n * 24, hipMemcpyDeviceToHost, st));
    if (raw_out) HIPCHK(ctx, hipMemcpyAsync(raw_out, ctx->d_corr_raw, (size_t)n * 24, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    ctx->corr_n = n;
    return SRL_OK;
}

int srl_frame_take(srl_ctx *ctx, const int32_t *index, int m) {
    if (!ctx || m < 0 || (m > 0 && !index)) return SRL_ERR_BAD_ARG;
    SRL_DISARM(ctx);
    if (ctx->corr_n < 0) { ctx->err = "no undistorted sweep (srl_frame_undistort first)"; return SRL_ERR_NO_SWEEP; }
    for (int k = 0; k < m; k++)
        if (index[k] < 0 || index[k] >= ctx->corr_n) { ctx->err = "frame index out of range"; return SRL_ERR_BAD_ARG; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc = ensure_frame(ctx, m);
    if (rc) return rc;
    ctx->frame_n = m;
    ctx->frame_world_n = -1;
    if (m > 0) {
        DevBuf b_sel;
        HIPCHK(ctx, b_sel.alloc(ctx, (size_t)m * 4));
        HIPCHK(ctx, hipMemcpyAsync(b_sel.p, index, (size_t)m * 4, hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(k_gather_aos, dim3((m + 255) / 256), dim3(256), 0, ctx->stream, ctx->d_corr_raw, b_sel.as<int>(), m, ctx->d_frame_raw);
        HIPCHK(ctx, hipGetLastError());
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    return SRL_OK;
}

int srl_frame_upload(srl_ctx *ctx, const double *raw_xyz, int n) {
    if (!ctx || n < 0 || (n > 0 && !raw_xyz)) return SRL_ERR_BAD_ARG;
    SRL_DISARM(ctx);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc0 = ensure_frame(ctx, n);
    if (rc0) return rc0;
    ctx->frame_n = n;
    ctx->frame_world_n = -1;
    srl_stage_begin(ctx);
    if (n > 0) {
        if (!srl_ctx_is_pinned(raw_xyz)) {
            // a pageable source is consumed here
            HIPCHK(ctx, hipMemcpyAsync(ctx->d_frame_raw, raw_xyz, (size_t)n * 3 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        } else {
            // a page-locked source (srl_pinned_alloc) is read by the DMA engine behind this call: the caller may refill it once a later call
            // on this context has returned results (like srl_sweep_upload; srl_sweep_wait waits explicitly).  The copy runs on the COPY
            // stream, behind the last kernel that read the previous frame's raw points and beside whatever the compute stream still has to
            // do for that frame (its deferred map insertion); the compute stream picks up behind the copy.
            int rcc = ensure_copy_stream(ctx);
            if (rcc) return rcc;
            if (ctx->ev_frame_read) HIPCHK(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->ev_frame_read, 0));
            HIPCHK(ctx, hipMemcpyAsync(ctx->d_frame_raw, raw_xyz, (size_t)n * 3 * sizeof(double), hipMemcpyHostToDevice, ctx->copy_stream));
            if (!ctx->upload_ev) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->upload_ev, hipEventDisableTiming));
            HIPCHK(ctx, hipEventRecord(ctx->upload_ev, ctx->copy_stream));
            HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->upload_ev, 0));
            ctx->upload_pending = true;
        }
    }
    srl_stage_end(ctx, 0);
    return SRL_OK;
}

int srl_frame_size(srl_ctx *ctx, int *n) {
    if (!ctx || !n) return SRL_ERR_BAD_ARG;
    *n = ctx->frame_n < 0 ? 0 : ctx->frame_n;
    return ctx->frame_n < 0 ? SRL_ERR_NO_SWEEP : SRL_OK;
}

int srl_frame_select_keypoints(srl_ctx *ctx, const double q[4], const double t[3], const double R_il[9], const double t_il[3],
                               double sample_voxel_size, int32_t *keypoint_index, int *num_keypoints) {
    if (!ctx || !q || !t || !R_il || !t_il || !(sample_voxel_size > 0.0)) return SRL_ERR_BAD_ARG;
    SRL_DISARM(ctx);
    if (ctx->frame_n < 0 || !ctx->d_frame_raw) { ctx->err = "no frame uploaded"; return SRL_ERR_NO_SWEEP; }
    if (ctx->nranks > 1) { ctx->err = "frame pipeline is single-rank (shard with srl_sweep_upload instead)"; return SRL_ERR_UNSUPPORTED; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int n = ctx->frame_n;
    if (num_keypoints) *num_keypoints = 0;
    hipStream_t st = ctx->stream;
    static const bool trace = std::getenv("SRL_FRAME_TIMING") != nullptr;      // stage times on stderr (tools/pipeline_probe.py)
    const auto tp0 = std::chrono::steady_clock::now();
    auto tp1 = tp0, tp2 = tp0, tp3 = tp0;
    srl_stage_begin(ctx);
    ctx->frame_order_used = 0;
    int m = 0;
    if (n > 0) {
        // the selection becomes the resident sweep (SoA, gathered on the device): at most n keypoints
        if (n > ctx->sweep_cap) {
            if (ctx->d_raw) { HIPCHK(ctx, hipFree(ctx->d_raw)); ctx->d_raw = nullptr; }
            const int cap = std::max(n, 1024);
            HIPCHK(ctx, hipMalloc((void **)&ctx->d_raw, (size_t)cap * 3 * sizeof(double)));
            ctx->sweep_cap = cap;
        }
        int rc = srl_ctx_ensure_work(ctx, n);
        if (rc) return rc;
        Xf X;
        fill_xf(X, q, t, R_il, t_il);
        // group by sampling voxel in a scratch table of >= 2 n slots (epoch-tagged: never cleared between frames, srl_frame_scratch.h)
        unsigned cap = 1024;
        while (cap < 2u * (unsigned)n) cap <<= 1;
        // the keypoint ORDER on the device (see "keypoint ORDER on the device" above) when the bucket table of n voxels fits one scan launch
        unsigned nb_max = 0;
        bool dev_order = ctx->frame_order_mode == 0 && n <= SRL_SCAN_MAX;
        if (dev_order) {
            if (ctx->tr1_steps < 0) {
                SrlTr1Sched S;
                std::memset(&S, 0, sizeof S);
                const int steps = srl::Tr1Order::export_schedule(SRL_SCAN_MAX, S.first, S.nb, SRL_TR1_MAX_STEPS);
                if (steps >= 0) {
                    S.steps = steps;
                    HIPCHK(ctx, hipMalloc((void **)&ctx->d_tr1_sched, sizeof S));
                    HIPCHK(ctx, hipMemcpy(ctx->d_tr1_sched, &S, sizeof S, hipMemcpyHostToDevice));
                    std::memcpy(ctx->tr1_first, S.first, sizeof S.first);
                    std::memcpy(ctx->tr1_nb, S.nb, sizeof S.nb);
                    ctx->tr1_steps = steps;
                } else {
                    ctx->tr1_steps = -2;                                      // a growth policy this table cannot hold: host replay from now on
                }
            }
            if (ctx->tr1_steps >= 0) {
                int g = 0;
                while (g < ctx->tr1_steps && ctx->tr1_first[g] <= (unsigned)n) ++g;
                nb_max = ctx->tr1_nb[g];
            }
            dev_order = ctx->tr1_steps >= 0 && nb_max > 0 && nb_max <= (1u << 23);       // (srl_scan takes any size; the member lists are 128 B per bucket)
        }
        DevBuf b_flag, b_keyat, b_hash, b_first, b_bcnt, b_members, b_elem, b_bstart, b_sel, b_sc;
        HIPCHK(ctx, b_flag.alloc(ctx, (size_t)n * 4)); HIPCHK(ctx, b_keyat.alloc(ctx, (size_t)n * 8));
        HIPCHK(ctx, b_sc.alloc(ctx, srl_scan_scratch_ints(std::max(n, (int)nb_max)) * 4));      // tile sums of the scans beyond one launch (srl_scan)
        if (dev_order) {
            HIPCHK(ctx, b_hash.alloc(ctx, (size_t)n * 8)); HIPCHK(ctx, b_first.alloc(ctx, (size_t)n * 4));
            HIPCHK(ctx, b_bcnt.alloc(ctx, (size_t)nb_max * 4)); HIPCHK(ctx, b_members.alloc(ctx, (size_t)nb_max * SRL_TR1_BUCKET_SLOTS * 8));
            HIPCHK(ctx, b_elem.alloc(ctx, (size_t)n * 4)); HIPCHK(ctx, b_bstart.alloc(ctx, (size_t)nb_max * 4));
            HIPCHK(ctx, b_sel.alloc(ctx, (size_t)n * 4));
        }
        int rct = srl_epoch_table_begin(ctx, ctx->sel_table, cap, true);
        if (rct) return rct;
        const SrlEpochTable &T = ctx->sel_table;
        // host exchange block: [0] control word {tag, count} | hashes (8 B) | first indices (4 B) | ordered index list (4 B)
        int rcx = ensure_frame_exchange(ctx, 64 + (size_t)n * 16);
        if (rcx) return rcx;
        unsigned long long *h_ctrl = reinterpret_cast<unsigned long long *>(ctx->h_frame_x);
        unsigned long long *h_hash = reinterpret_cast<unsigned long long *>(ctx->h_frame_x + 64);
        unsigned *first = reinterpret_cast<unsigned *>(ctx->h_frame_x + 64 + (size_t)n * 8);
        int *h_sel = reinterpret_cast<int *>(ctx->h_frame_x + 64 + (size_t)n * 12);
        if (++ctx->frame_tag == 0) ++ctx->frame_tag;
        const unsigned tag = ctx->frame_tag;
        __atomic_store_n(h_ctrl, 0ull, __ATOMIC_RELEASE);
        double *sx = ctx->d_raw, *sy = ctx->d_raw + ctx->sweep_cap, *sz = ctx->d_raw + 2 * (size_t)ctx->sweep_cap;
        hipLaunchKernelGGL(k_select_group, dim3((n + 255) / 256), dim3(256), 0, st, ctx->d_frame_raw, n, X, sample_voxel_size, T.keyw, T.minw, cap - 1, T.epoch16,
                           T.counter32, b_flag.as<int>());
        const unsigned mark_threads = dev_order && nb_max > cap ? nb_max : cap;
        hipLaunchKernelGGL(k_select_mark, dim3((mark_threads + 255) / 256), dim3(256), 0, st, T.keyw, T.minw, cap, T.epoch16, b_flag.as<int>(),
                           b_keyat.as<unsigned long long>(), dev_order ? b_bcnt.as<int>() : (int *)nullptr, dev_order ? nb_max : 0u);
        if (dev_order) {
            // ranks (first-occurrence order) -> bucket of every voxel at the table's final size, {tag, overflow, count} to the host -> scan
            // over the bucket counts -> per voxel: rank inside its bucket, ordered index list, gather of the raw point
            srl_scan(SrlIntArrayIn{b_flag.as<int>()}, RankSink{b_keyat.as<unsigned long long>(), b_hash.as<unsigned long long>(), b_first.as<unsigned>()}, n,
                     b_sc.as<int>(), st, CountFin{ctx->d_frame_sync});
            hipLaunchKernelGGL(k_tr1_bucket, dim3((n + 255) / 256), dim3(256), 0, st, ctx->d_tr1_sched, ctx->d_frame_sync, b_hash.as<unsigned long long>(),
                               b_bcnt.as<int>(), b_members.as<unsigned long long>(), b_elem.as<unsigned>(), h_ctrl, tag);
            srl_scan(SrlIntArrayIn{b_bcnt.as<int>()}, SrlIntArraySink{b_bstart.as<int>()}, (int)nb_max, b_sc.as<int>(), st);
            hipLaunchKernelGGL(k_tr1_rank, dim3((n + 255) / 256), dim3(256), 0, st, ctx->d_tr1_sched, ctx->d_frame_sync, b_hash.as<unsigned long long>(),
                               b_first.as<unsigned>(), b_members.as<unsigned long long>(), b_bcnt.as<int>(), b_bstart.as<int>(), b_elem.as<unsigned>(),
                               ctx->d_frame_raw, sx, sy, sz, b_sel.as<int>());
        } else {
            // ranks, hand-over to the host and the completion word in the scan's own pass
            srl_scan(SrlIntArrayIn{b_flag.as<int>()}, EmitSink{b_keyat.as<unsigned long long>(), h_hash, first}, n, b_sc.as<int>(), st,
                     EmitFin{ctx->d_frame_sync, h_ctrl, tag});
        }
        HIPCHK(ctx, hipGetLastError());
        srl_stage_end(ctx, 1);
        tp1 = std::chrono::steady_clock::now();
        // wait for the control word (the stream is looked at every ~1M polls: a fault must not become a hang)
        unsigned long long ctrl = 0, spins = 0;
        while ((unsigned)((ctrl = __atomic_load_n(h_ctrl, __ATOMIC_ACQUIRE)) >> 32) != tag) {
            if ((++spins & 0xFFFFF) == 0) {
                const hipError_t qe = hipStreamQuery(st);
                if (qe != hipSuccess && qe != hipErrorNotReady) { ctx->err = std::string("keypoint selection: ") + hipGetErrorString(qe); return SRL_ERR_HIP; }
                if (qe == hipSuccess && (unsigned)(__atomic_load_n(h_ctrl, __ATOMIC_ACQUIRE) >> 32) != tag) {
                    ctx->err = "keypoint selection finished without publishing its voxel list";
                    return SRL_ERR_HIP;
                }
            }
        }
        const int S = (int)((unsigned)ctrl & 0x7FFFFFFFu);
        const bool overflow = dev_order && (((unsigned)ctrl >> 31) & 1u);
        srl_stage_end(ctx, 2);
        tp2 = std::chrono::steady_clock::now();
        m = S;
        ctx->frame_order_used = dev_order ? (overflow ? 3 : 1) : 2;
        if (!dev_order || overflow) {
            if (overflow) {
                // a bucket with more voxels than the device ranks in place: fetch {hash, first index} and order on the host like larger frames
                HIPCHK(ctx, hipMemcpyAsync(h_hash, b_hash.p, (size_t)S * 8, hipMemcpyDeviceToHost, st));
                HIPCHK(ctx, hipMemcpyAsync(first, b_first.p, (size_t)S * 4, hipMemcpyDeviceToHost, st));
                HIPCHK(ctx, hipStreamSynchronize(st));
            }
            // the voxels arrive in FIRST-OCCURRENCE order (= the order subSampleFrame's loop creates them): ordered on the device.
            // Iteration order of the std::tr1::unordered_map of subSampleFrame, by replaying its bucket moves on flat arrays
            // (host/tr1_order.h) for the S distinct voxels (not the N points)
            static_assert(sizeof(std::size_t) == sizeof(unsigned long long), "hashes are exchanged as 64-bit words");
            std::vector<int> perm((size_t)S);
            srl::Tr1Order::order(reinterpret_cast<const std::size_t *>(h_hash), S, perm.data());
            // the gather reads the ordered index list straight out of the exchange block (its own region: nothing is written there before
            // the next frame's list, and that is produced behind a wait on a later kernel of this stream)
            for (int r = 0; r < S; r++) h_sel[r] = (int)first[(size_t)perm[(size_t)r]];
            srl_stage_end(ctx, 3);
            tp3 = std::chrono::steady_clock::now();
            if (m > 0) {
                hipLaunchKernelGGL(k_gather_soa, dim3((m + 255) / 256), dim3(256), 0, st, ctx->d_frame_raw, h_sel, m, sx, sy, sz);
                HIPCHK(ctx, hipGetLastError());
            }
        } else {
            srl_stage_end(ctx, 3);
            tp3 = tp2;
            if (keypoint_index && m > 0) {
                // the ordered index list is wanted on the host (tests, tools; the host mirror passes NULL): one copy behind the chain
                HIPCHK(ctx, hipMemcpyAsync(h_sel, b_sel.p, (size_t)m * 4, hipMemcpyDeviceToHost, st));
                HIPCHK(ctx, hipStreamSynchronize(st));
            }
        }
        if (keypoint_index) std::memcpy(keypoint_index, h_sel, (size_t)m * 4);
    } else {
        srl_stage_end(ctx, 3);
    }
    if (num_keypoints) *num_keypoints = m;
    ctx->total_n = m; ctx->shard_begin = 0; ctx->n = m; ctx->sweep_loaded = true; ctx->taps_valid = false; ctx->bound_n = 0; ctx->tail_pending = false;
    ctx->soa_valid_n = m; ctx->passes_in_solve = 0;
    if (n > 0) { const int rcm = srl_mark_frame_read(ctx); if (rcm) return rcm; }
    srl_stage_end(ctx, 4);
    if (trace) {
        const auto tp4 = std::chrono::steady_clock::now();
        auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
        std::fprintf(stderr, "[srl_frame_select_keypoints] n %d -> %d (order %d): enqueue %.0f us, wait %.0f us, host order replay %.0f us, gather + rest %.0f us\n",
                     n, m, ctx->frame_order_used, us(tp0, tp1), us(tp1, tp2), us(tp2, tp3), us(tp3, tp4));
    }
    return SRL_OK;
}

int srl_frame_commit(srl_ctx *ctx, const double q[4], const double t[3], const double R_il[9], const double t_il[3],
                     double voxel_size, int cap, double min_distance_points, int min_num_points, double *world_out, int *num_added) {
    if (!ctx || !q || !t || !R_il || !t_il) return SRL_ERR_BAD_ARG;
    SRL_DISARM(ctx);
    if (cap != SRL_VOXEL_CAP) { ctx->err = "max_num_points_in_voxel must be 20"; return SRL_ERR_UNSUPPORTED; }
    if (ctx->frame_n < 0 || !ctx->d_frame_raw) { ctx->err = "no frame uploaded"; return SRL_ERR_NO_SWEEP; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int n = ctx->frame_n;
    if (num_added) *num_added = 0;
    if (n == 0) return SRL_OK;
    SrlFrameTransform xf;
    xf.raw = ctx->d_frame_raw; xf.world = ctx->d_frame_world;
    fill_xf(xf.X, q, t, R_il, t_il);
    struct Download { double *world_out; int n; } dl = {world_out, n};
    // what has to happen as soon as the world points exist: the frame's raw points are free for the next upload, point3D::point leaves on
    // the copy stream beside the insertion
    auto world_ready = [](srl_ctx *c, void *user) -> int {
        const Download *d = static_cast<const Download *>(user);
        { const int rcm = srl_mark_frame_read(c); if (rcm) return rcm; }       // (= the world points are ready)
        if (d->world_out) {
            int rcc = ensure_copy_stream(c);
            if (rcc) return rcc;
            HIPCHK(c, hipStreamWaitEvent(c->copy_stream, c->ev_frame_read, 0));
            HIPCHK(c, hipMemcpyAsync(d->world_out, c->d_frame_world, (size_t)d->n * 3 * sizeof(double), hipMemcpyDeviceToHost, c->copy_stream));
            if (!c->ev_world) HIPCHK(c, hipEventCreateWithFlags(&c->ev_world, hipEventDisableTiming));
            HIPCHK(c, hipEventRecord(c->ev_world, c->copy_stream));
        }
        return SRL_OK;
    };
    srl_stage_begin(ctx);
    // A frame-sized batch outside the stage-timing mode: the re-transform is the first thing the insertion's first kernel does (one launch
    // less on a chain whose cost is its launches); otherwise a kernel of its own, as stage 5 of srl_debug_frame_timing.
    const bool fused = n <= SRL_SCAN_MAX && !ctx->frame_timing;
    if (!fused) {
        hipLaunchKernelGGL(k_frame_keys, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, ctx->d_frame_raw, n, xf.X, 1.0, ctx->d_frame_world,
                           (unsigned long long *)nullptr, (unsigned *)nullptr);
        HIPCHK(ctx, hipGetLastError());
        srl_stage_end(ctx, 5);
        const int rcw = world_ready(ctx, &dl);
        if (rcw) return rcw;
        srl_stage_end(ctx, 6);
    }
    // num_added == NULL: the insert is only enqueued (its counters are folded in later, srl_map_settle); the caller's world points are
    // waited for on their own event, which fires long before the insert behind them is done
    const int rci = srl_map_insert_impl(ctx, ctx->d_frame_world, true, n, voxel_size, min_distance_points, min_num_points, num_added, num_added == nullptr,
                                        fused ? &xf : nullptr, fused ? +world_ready : nullptr, &dl);
    if (world_out && rci == SRL_OK) HIPCHK(ctx, hipEventSynchronize(ctx->ev_world));
    if (rci == SRL_OK) ctx->frame_world_n = n;            // d_frame_world = the frame as inserted (srl_map_probe_checksum)
    return rci;
}

}  // extern "C"
