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
#include "srl_hash.h"
#include "host/srl_la.h"

#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cstring>
#include <tr1/unordered_map>
#include <vector>

int srl_map_insert_impl(srl_ctx *ctx, const double *world_xyz, bool on_device, int n, double voxel_size,
                        double min_distance_points, int min_num_points, int *num_added);

namespace {

struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
    template <class T> T *as() { return reinterpret_cast<T *>(p); }
};

struct Xf { double R[9], t[3], R_il[9], t_il[3]; };

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

extern "C" {

int srl_frame_upload(srl_ctx *ctx, const double *raw_xyz, int n) {
    if (!ctx || n < 0 || (n > 0 && !raw_xyz)) return SRL_ERR_BAD_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (n > ctx->frame_cap) {
        if (ctx->d_frame_raw) HIPCHK(ctx, hipFree(ctx->d_frame_raw));
        if (ctx->d_frame_world) HIPCHK(ctx, hipFree(ctx->d_frame_world));
        ctx->d_frame_raw = ctx->d_frame_world = nullptr;
        const int cap = std::max(n, 4096);
        HIPCHK(ctx, hipMalloc((void **)&ctx->d_frame_raw, (size_t)cap * 3 * sizeof(double)));
        HIPCHK(ctx, hipMalloc((void **)&ctx->d_frame_world, (size_t)cap * 3 * sizeof(double)));
        ctx->frame_cap = cap;
    }
    ctx->frame_n = n;
    if (n > 0) {
        HIPCHK(ctx, hipMemcpyAsync(ctx->d_frame_raw, raw_xyz, (size_t)n * 3 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    return SRL_OK;
}

int srl_frame_select_keypoints(srl_ctx *ctx, const double q[4], const double t[3], const double R_il[9], const double t_il[3],
                               double sample_voxel_size, int32_t *keypoint_index, int *num_keypoints) {
    if (!ctx || !q || !t || !R_il || !t_il || !(sample_voxel_size > 0.0)) return SRL_ERR_BAD_ARG;
    if (ctx->frame_n < 0 || !ctx->d_frame_raw) { ctx->err = "no frame uploaded"; return SRL_ERR_NO_SWEEP; }
    if (ctx->nranks > 1) { ctx->err = "frame pipeline is single-rank (shard with srl_sweep_upload instead)"; return SRL_ERR_UNSUPPORTED; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int n = ctx->frame_n;
    if (num_keypoints) *num_keypoints = 0;
    hipStream_t st = ctx->stream;
    std::vector<int> order;
    if (n > 0) {
        Xf X;
        fill_xf(X, q, t, R_il, t_il);
        DevBuf b_keys, b_keys2, b_idx, b_idx2, b_ukeys, b_len, b_start, b_nruns, b_first, b_tmp;
        HIPCHK(ctx, b_keys.alloc((size_t)n * 8)); HIPCHK(ctx, b_keys2.alloc((size_t)n * 8));
        HIPCHK(ctx, b_idx.alloc((size_t)n * 4)); HIPCHK(ctx, b_idx2.alloc((size_t)n * 4));
        HIPCHK(ctx, b_ukeys.alloc((size_t)n * 8)); HIPCHK(ctx, b_len.alloc((size_t)n * 4)); HIPCHK(ctx, b_start.alloc((size_t)n * 4));
        HIPCHK(ctx, b_first.alloc((size_t)n * 4)); HIPCHK(ctx, b_nruns.alloc(16));
        hipLaunchKernelGGL(k_frame_keys, dim3((n + 255) / 256), dim3(256), 0, st, ctx->d_frame_raw, n, X, sample_voxel_size,
                           (double *)nullptr, b_keys.as<unsigned long long>(), b_idx.as<unsigned>());
        HIPCHK(ctx, hipGetLastError());
        size_t need = 0, tmp_bytes = 0;
        hipcub::DeviceRadixSort::SortPairs(nullptr, need, b_keys.as<unsigned long long>(), b_keys2.as<unsigned long long>(),
                                           b_idx.as<unsigned>(), b_idx2.as<unsigned>(), n, 0, 48, st);
        tmp_bytes = need;
        hipcub::DeviceRunLengthEncode::Encode(nullptr, need, b_keys2.as<unsigned long long>(), b_ukeys.as<unsigned long long>(),
                                              b_len.as<int>(), b_nruns.as<int>(), n, st);
        tmp_bytes = std::max(tmp_bytes, need);
        hipcub::DeviceScan::ExclusiveSum(nullptr, need, b_len.as<int>(), b_start.as<int>(), n, st);
        tmp_bytes = std::max(tmp_bytes, need) + 4096;
        HIPCHK(ctx, b_tmp.alloc(tmp_bytes));
        size_t tb = tmp_bytes;
        HIPCHK(ctx, hipcub::DeviceRadixSort::SortPairs(b_tmp.p, tb, b_keys.as<unsigned long long>(), b_keys2.as<unsigned long long>(),
                                                       b_idx.as<unsigned>(), b_idx2.as<unsigned>(), n, 0, 48, st));
        tb = tmp_bytes;
        HIPCHK(ctx, hipcub::DeviceRunLengthEncode::Encode(b_tmp.p, tb, b_keys2.as<unsigned long long>(), b_ukeys.as<unsigned long long>(),
                                                          b_len.as<int>(), b_nruns.as<int>(), n, st));
        int S = 0;
        HIPCHK(ctx, hipMemcpyAsync(&S, b_nruns.p, sizeof(int), hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipStreamSynchronize(st));
        tb = tmp_bytes;
        HIPCHK(ctx, hipcub::DeviceScan::ExclusiveSum(b_tmp.p, tb, b_len.as<int>(), b_start.as<int>(), S, st));
        hipLaunchKernelGGL(k_first_index, dim3((S + 255) / 256), dim3(256), 0, st, b_start.as<int>(), b_idx2.as<unsigned>(), S, b_first.as<unsigned>());
        HIPCHK(ctx, hipGetLastError());
        std::vector<unsigned long long> ukeys((size_t)S);
        std::vector<unsigned> first((size_t)S);
        HIPCHK(ctx, hipMemcpyAsync(ukeys.data(), b_ukeys.p, (size_t)S * 8, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipMemcpyAsync(first.data(), b_first.p, (size_t)S * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipStreamSynchronize(st));

        // order replay (host, V distinct voxels instead of N points): first-occurrence order into the same container
        std::vector<int> byfirst((size_t)S);
        for (int i = 0; i < S; i++) byfirst[i] = i;
        std::sort(byfirst.begin(), byfirst.end(), [&](int a, int b) { return first[a] < first[b]; });
        std::tr1::unordered_map<vkey, int, vkey_hash> grid;
        for (int i : byfirst) {
            vkey k;
            srl_unpack_key(ukeys[i], &k.x, &k.y, &k.z);
            grid[k] = (int)first[i];
        }
        order.reserve(S);
        for (const auto &kv : grid) order.push_back(kv.second);
    }
    const int m = (int)order.size();
    if (num_keypoints) *num_keypoints = m;
    if (keypoint_index) for (int k = 0; k < m; k++) keypoint_index[k] = order[k];

    // the selection becomes the resident sweep: gather raw points on the device (SoA)
    ctx->total_n = m; ctx->shard_begin = 0; ctx->n = m; ctx->taps_valid = false;
    if (m > ctx->sweep_cap) {
        if (ctx->d_raw) { HIPCHK(ctx, hipFree(ctx->d_raw)); ctx->d_raw = nullptr; }
        const int cap = std::max(m, 1024);
        HIPCHK(ctx, hipMalloc((void **)&ctx->d_raw, (size_t)cap * 3 * sizeof(double)));
        ctx->sweep_cap = cap;
    }
    int rc = srl_ctx_ensure_work(ctx, m);
    if (rc) return rc;
    if (m > 0) {
        DevBuf b_sel;
        HIPCHK(ctx, b_sel.alloc((size_t)m * 4));
        HIPCHK(ctx, hipMemcpyAsync(b_sel.p, order.data(), (size_t)m * 4, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_gather_soa, dim3((m + 255) / 256), dim3(256), 0, st, ctx->d_frame_raw, b_sel.as<int>(), m,
                           ctx->d_raw, ctx->d_raw + ctx->sweep_cap, ctx->d_raw + 2 * (size_t)ctx->sweep_cap);
        HIPCHK(ctx, hipGetLastError());
        HIPCHK(ctx, hipStreamSynchronize(st));
    }
    return SRL_OK;
}

int srl_frame_commit(srl_ctx *ctx, const double q[4], const double t[3], const double R_il[9], const double t_il[3],
                     double voxel_size, int cap, double min_distance_points, int min_num_points, double *world_out, int *num_added) {
    if (!ctx || !q || !t || !R_il || !t_il) return SRL_ERR_BAD_ARG;
    if (cap != SRL_VOXEL_CAP) { ctx->err = "max_num_points_in_voxel must be 20"; return SRL_ERR_UNSUPPORTED; }
    if (ctx->frame_n < 0 || !ctx->d_frame_raw) { ctx->err = "no frame uploaded"; return SRL_ERR_NO_SWEEP; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int n = ctx->frame_n;
    if (num_added) *num_added = 0;
    if (n == 0) return SRL_OK;
    Xf X;
    fill_xf(X, q, t, R_il, t_il);
    hipLaunchKernelGGL(k_frame_keys, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, ctx->d_frame_raw, n, X, 1.0, ctx->d_frame_world,
                       (unsigned long long *)nullptr, (unsigned *)nullptr);
    HIPCHK(ctx, hipGetLastError());
    if (world_out) HIPCHK(ctx, hipMemcpyAsync(world_out, ctx->d_frame_world, (size_t)n * 3 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    return srl_map_insert_impl(ctx, ctx->d_frame_world, true, n, voxel_size, min_distance_points, min_num_points, num_added);
}

}  // extern "C"
