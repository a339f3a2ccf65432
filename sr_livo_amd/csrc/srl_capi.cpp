// srl_capi.cpp -- implementation of the C-ABI declared in include/srlivo_hip.h.
// Host plumbing only (buffers, stream, hash-table build, RCCL); all arithmetic of the hot path runs
// in the HIP kernels of srl_kernels.hip / srl_map_kernels.hip.  No CPU fallback exists.
#include "../../include/srlivo_hip.h"
#include "../../include/srlivo_hip_debug.h"
#include "host/srl_la.h"
#include "srl_device.h"
#include "srl_hash.h"
#include "srl_heap.h"

#include <hip/hip_runtime.h>
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

// map kernels (srl_map_kernels.hip)
struct SrlMapState;
int srl_map_insert_device(struct srl_ctx *ctx, const double *world_xyz, int n, double voxel_size, int cap,
                          double min_distance_points, int min_num_points, int *num_added);

#include "srl_ctx.h"

namespace {

unsigned next_pow2(unsigned v) { unsigned p = 1; while (p < v) p <<= 1; return p; }
inline long long steady_ns() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// Live contexts per device.  An armed launch keeps one workgroup per CU resident until its pose arrives: with a second context on the
// same device each side would stall the other for the full linger, so launches are only armed while the context has its device to
// itself inside this process (arm_mode 1; srl_set_armed_launch(ctx, 2) overrides).
constexpr int SRL_MAX_DEVICES = 64;
std::atomic<int> g_live_ctx[SRL_MAX_DEVICES];
inline int live_contexts(int device) { return (device >= 0 && device < SRL_MAX_DEVICES) ? g_live_ctx[device].load(std::memory_order_relaxed) : 1; }

int ensure_work(srl_ctx *ctx, int n) {
    if (n <= ctx->work_cap) return SRL_OK;
    const int cap = std::max(n, 1024);
    // small passes use 16 or 32 keypoints per workgroup (srl_keypoints_per_block): size for whichever gives more workgroups
    const int nblocks = std::max((cap + SRL_KPB - 1) / SRL_KPB, 1024 + 8);
    int rc;
    if ((rc = ensure(ctx, ctx->d_rec, (size_t)cap * 8))) return rc;
    if ((rc = ensure(ctx, ctx->d_status, (size_t)cap))) return rc;
    if ((rc = ensure(ctx, ctx->d_bound, (size_t)cap * 4))) return rc;
    ctx->bound_n = 0;
    if ((rc = ensure(ctx, ctx->d_partials, (size_t)nblocks * SRL_PART_STRIDE))) return rc;
    if ((rc = ensure(ctx, ctx->d_binfo, (size_t)nblocks))) return rc;
    ctx->work_cap = cap;
    ctx->block_cap = nblocks;
    return SRL_OK;
}

// is the host pointer page-locked (hipHostMalloc / hipHostRegister)?  Pageable memory makes the query fail.
bool srl_is_pinned(const void *p) {
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return attr.type == hipMemoryTypeHost;
}

int srl_ring_init(srl_ctx *ctx) {
    if (ctx->h_ring) return SRL_OK;
    HIPCHK(ctx, hipHostMalloc((void **)&ctx->h_ring, (size_t)srl_ctx::RING_SLOTS * srl_ctx::RING_SLOT_BYTES, hipHostMallocDefault));
    for (int i = 0; i < srl_ctx::RING_SLOTS; i++) { HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ring_ev[i], hipEventDisableTiming)); ctx->ring_busy[i] = false; }
    return SRL_OK;
}

int ensure_taps(srl_ctx *ctx, int n, int K) {
    if (n <= ctx->tap_cap && K <= ctx->tap_K) return SRL_OK;
    const int cap = std::max(n, ctx->tap_cap), kk = std::max(K, ctx->tap_K);
    int rc;
    if ((rc = ensure(ctx, ctx->d_tap_ids, (size_t)cap * kk))) return rc;
    if ((rc = ensure(ctx, ctx->d_tap_ncand, (size_t)cap))) return rc;
    if ((rc = ensure(ctx, ctx->d_tap_normal, (size_t)cap * 3))) return rc;
    if ((rc = ensure(ctx, ctx->d_tap_a2d, (size_t)cap))) return rc;
    if ((rc = ensure(ctx, ctx->d_tap_offset, (size_t)cap))) return rc;
    ctx->tap_cap = cap;
    ctx->tap_K = kk;
    return SRL_OK;
}

}  // namespace

int srl_ctx_ensure_work(srl_ctx *ctx, int n) { return ensure_work(ctx, n); }   // used by srl_frame_kernels.hip
bool srl_ctx_is_pinned(const void *p) { return srl_is_pinned(p); }             // used by srl_frame_kernels.hip

// shared with srl_map_kernels.hip
int srl_ctx_grow_map(srl_ctx *ctx, unsigned need_slabs, unsigned need_slots);

extern "C" {

int srl_device_count(int *count) {
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) { if (count) *count = 0; return SRL_ERR_NO_DEVICE; }
    if (count) *count = c;
    return SRL_OK;
}

const char *srl_status_str(int s) {
    switch (s) {
        case SRL_OK: return "ok";
        case SRL_ERR_NO_DEVICE: return "no HIP device (the product has no CPU fallback)";
        case SRL_ERR_HIP: return "HIP runtime error";
        case SRL_ERR_BAD_ARG: return "bad argument";
        case SRL_ERR_UNSUPPORTED: return "option outside the supported envelope";
        case SRL_ERR_NO_MAP: return "no map uploaded";
        case SRL_ERR_NO_SWEEP: return "no sweep uploaded";
        case SRL_ERR_COMM: return "RCCL error";
        case SRL_ERR_NAN_PLANARITY: return "NaN planarity (optimize.cpp:348-350 throws)";
        case SRL_ERR_NOT_ENOUGH_RESIDUALS: return "not enough residuals (optimize.cpp:110)";
        default: return "unknown status";
    }
}

void srl_icp_opts_default(srl_icp_opts *o) {
    o->threshold_voxel_occupancy = 1;
    o->init_num_frames = 20;
    o->size_voxel_map = 1.0;
    o->num_iters_icp = 5;
    o->min_number_neighbors = 20;
    o->voxel_neighborhood = 1;
    o->power_planarity = 2.0;
    o->max_number_neighbors = 20;
    o->max_dist_to_plane_icp = 0.3;
    o->threshold_orientation_norm = 0.1;
    o->threshold_translation_norm = 0.01;
    o->max_num_residuals = 600;
    o->weight_alpha = 0.9;
    o->weight_neighborhood = 0.1;
}

int srl_ctx_create(int device, srl_ctx **out) {
    if (!out) return SRL_ERR_BAD_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) return SRL_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return SRL_ERR_NO_DEVICE;
    srl_ctx *ctx = new srl_ctx();
    ctx->device = device;
    { int cu = 0; if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cu > 0) ctx->num_cu = cu; }
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return SRL_ERR_NO_DEVICE; }
    if (hipMalloc((void **)&ctx->d_out, sizeof(SrlDevOut)) != hipSuccess ||
        hipHostMalloc((void **)&ctx->h_out, sizeof(SrlDevOut)) != hipSuccess ||
        hipMalloc((void **)&ctx->d_count, sizeof(long long)) != hipSuccess ||
        hipHostMalloc((void **)&ctx->h_count, sizeof(long long)) != hipSuccess) {
        delete ctx;
        return SRL_ERR_HIP;
    }
    {
        const size_t row_bytes = (size_t)(SRL_FUSED_MAX_BLOCKS + SRL_FUSED_MAX_GROUPS) * SRL_ROW_GRANULES * 8;     // rows + super rows
        if (hipMalloc((void **)&ctx->d_granules, row_bytes) != hipSuccess || hipMemset(ctx->d_granules, 0, row_bytes) != hipSuccess) { delete ctx; return SRL_ERR_HIP; }
    }
    if (hipHostMalloc((void **)&ctx->h_mail, sizeof(SrlMailbox), hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess) {
        delete ctx;
        return SRL_ERR_HIP;
    }
    std::memset(ctx->h_mail, 0, sizeof(SrlMailbox));
    for (int i = 0; i < 4; i++) hipEventCreate(&ctx->ev[i]);
    if (device < SRL_MAX_DEVICES) g_live_ctx[device].fetch_add(1, std::memory_order_relaxed);
    ctx->counted = true;
    *out = ctx;
    return SRL_OK;
}

int srl_ctx_destroy(srl_ctx *ctx) {
    if (!ctx) return SRL_OK;
    hipSetDevice(ctx->device);
    if (ctx->counted && ctx->device < SRL_MAX_DEVICES) g_live_ctx[ctx->device].fetch_sub(1, std::memory_order_relaxed);
    if (ctx->armed) srl_ctx_disarm(ctx);
    if (ctx->stream) hipStreamSynchronize(ctx->stream);
    if (ctx->copy_stream) hipStreamSynchronize(ctx->copy_stream);
    if (ctx->pose_box_pinned) hipHostFree(ctx->pose_box_pinned);
    if (ctx->pose_box_dev) hipFree(ctx->pose_box_dev);
    if (ctx->d_pose_relay) hipFree(ctx->d_pose_relay);
    if (ctx->comm && srl_rccl()) srl_rccl()->CommDestroy(ctx->comm);
    if (ctx->parked_comm && srl_rccl()) srl_rccl()->CommDestroy(ctx->parked_comm);
    void *bufs[] = {ctx->d_corr_in, ctx->d_corr_rel, ctx->d_corr_imu, ctx->d_corr_raw, ctx->d_corr_seg, ctx->d_frame_raw, ctx->d_frame_world, ctx->d_table, ctx->d_slabs, ctx->d_raw, ctx->d_rec, ctx->d_status, ctx->d_partials, ctx->d_binfo,
                    ctx->d_out, ctx->d_count, ctx->d_bound, ctx->d_granules, ctx->d_rec_granules, ctx->d_raw_next, ctx->d_stage_next, ctx->d_stage_cur, ctx->d_tap_ids, ctx->d_tap_ncand, ctx->d_tap_normal, ctx->d_tap_a2d,
                    ctx->d_tap_offset, ctx->d_gather, ctx->d_peer, ctx->d_mail};
    for (int r = 0; r < SRL_MAX_PEERS; r++) if (ctx->peer_mapped[r]) hipIpcCloseMemHandle(ctx->peer_mapped[r]);
    if (ctx->d_inbox) hipFree(ctx->d_inbox);
    for (void *b : bufs) if (b) hipFree(b);
    if (ctx->h_out) hipHostFree(ctx->h_out);
    if (ctx->h_count) hipHostFree(ctx->h_count);
    if (ctx->h_mail) hipHostFree(ctx->h_mail);
    if (ctx->h_arm_stamps) hipHostFree(ctx->h_arm_stamps);
    if (ctx->h_scratch) hipHostFree(ctx->h_scratch);
    if (ctx->h_frame_x) hipHostFree(ctx->h_frame_x);
    for (SrlEpochTable *t : {&ctx->sel_table, &ctx->ins_table}) { if (t->keyw) hipFree(t->keyw); if (t->minw) hipFree(t->minw); }
    if (ctx->d_frame_sync) hipFree(ctx->d_frame_sync);
    if (ctx->d_tr1_sched) hipFree(ctx->d_tr1_sched);
    if (ctx->h_insert_cnt) hipHostFree(ctx->h_insert_cnt);
    if (ctx->ev_insert) hipEventDestroy(ctx->ev_insert);
    if (ctx->ev_world) hipEventDestroy(ctx->ev_world);
    if (ctx->ev_frame_read) hipEventDestroy(ctx->ev_frame_read);
    if (ctx->h_ring) { hipHostFree(ctx->h_ring); for (int i = 0; i < srl_ctx::RING_SLOTS; i++) if (ctx->ring_ev[i]) hipEventDestroy(ctx->ring_ev[i]); }
    for (auto &b : ctx->pool_free) hipFree(b.p);
    ctx->pool_free.clear();
    for (int i = 0; i < 4; i++) if (ctx->ev[i]) hipEventDestroy(ctx->ev[i]);
    for (int i = 0; i < srl_ctx::PROF_RING; i++) for (int k = 0; k < 2; k++) if (ctx->ring[i][k]) hipEventDestroy(ctx->ring[i][k]);
    for (int sl = 0; sl < 2; sl++) for (int k = 0; k < 2; k++) if (ctx->up_ev[sl][k]) hipEventDestroy(ctx->up_ev[sl][k]);
    if (ctx->upload_ev) hipEventDestroy(ctx->upload_ev);
    if (ctx->copy_stream) { hipStreamSynchronize(ctx->copy_stream); hipStreamDestroy(ctx->copy_stream); }
    if (ctx->prefix_stream) { hipStreamSynchronize(ctx->prefix_stream); hipStreamDestroy(ctx->prefix_stream); }
    if (ctx->stream) hipStreamDestroy(ctx->stream);
    delete ctx;
    return SRL_OK;
}

const char *srl_last_error(const srl_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

// ------------------------------------------------------------------------------------------ map
int srl_map_upload(srl_ctx *ctx, const int16_t *keys_xyz, const int32_t *counts, const float *xyz, int V, int cap) {
    if (!ctx || V < 0 || (V > 0 && (!keys_xyz || !counts || !xyz))) return SRL_ERR_BAD_ARG;
    SRL_DISARM(ctx);
    if (cap != SRL_VOXEL_CAP) { ctx->err = "max_num_points_in_voxel must be 20"; return SRL_ERR_UNSUPPORTED; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    { const int rcs = srl_map_settle(ctx); if (rcs) return rcs; }
    ctx->bound_n = 0;
    // capacity with headroom so that srl_map_insert can add voxels without an immediate rebuild
    const unsigned slab_cap = std::max<unsigned>(1024u, (unsigned)V + (unsigned)V / 2u + 4096u);
    if (slab_cap > SRL_MAX_SLABS) { ctx->err = "map too large: slab byte offsets are 32-bit (16.7 M voxels)"; return SRL_ERR_UNSUPPORTED; }
    const unsigned table_cap = next_pow2(std::max<unsigned>(2048u, SRL_TABLE_FACTOR * slab_cap));
    std::vector<SrlSlab> slabs((size_t)V);
    std::vector<SrlMapSlot> table((size_t)table_cap);
    for (auto &s : table) { s.key = SRL_EMPTY_KEY; s.slab = 0; s.count = 0; }
    long long npts = 0;
    const unsigned mask = table_cap - 1;
    for (int v = 0; v < V; v++) {
        SrlSlab &s = slabs[v];
        std::memset(&s, 0, sizeof s);
        const int c = counts[v];
        if (c < 0 || c > cap) { ctx->err = "voxel count out of range"; return SRL_ERR_BAD_ARG; }
        for (int i = 0; i < c; i++)
            for (int d = 0; d < 3; d++) s.xyz[i][d] = xyz[((size_t)v * cap + i) * 3 + d];
        s.count = (unsigned)c;
        s.key = srl_pack_key(keys_xyz[3 * v], keys_xyz[3 * v + 1], keys_xyz[3 * v + 2]);
        npts += c;
        unsigned h = srl_hash_key(s.key) & mask;
        while (table[h].key != SRL_EMPTY_KEY) {
            if (table[h].key == s.key) { ctx->err = "duplicate voxel key in upload"; return SRL_ERR_BAD_ARG; }
            h = (h + 1) & mask;
        }
        table[h].key = s.key;
        table[h].slab = (unsigned)v;
        table[h].count = (unsigned)c;
    }
    if (ctx->d_slabs) { HIPCHK(ctx, hipFree(ctx->d_slabs)); ctx->d_slabs = nullptr; }
    if (ctx->d_table) { HIPCHK(ctx, hipFree(ctx->d_table)); ctx->d_table = nullptr; }
    HIPCHK(ctx, hipMalloc((void **)&ctx->d_slabs, ((size_t)slab_cap + 1) * SRL_SLAB_BYTES));       // + the all-inf slab
    HIPCHK(ctx, hipMemsetD32Async((hipDeviceptr_t)(ctx->d_slabs + (size_t)slab_cap * SRL_SLAB_BYTES), 0x7f800000, SRL_SLAB_BYTES / 4, ctx->stream));
    HIPCHK(ctx, hipMalloc((void **)&ctx->d_table, (size_t)table_cap * sizeof(SrlMapSlot)));
    HIPCHK(ctx, hipMemsetAsync(ctx->d_slabs, 0, (size_t)slab_cap * SRL_SLAB_BYTES, ctx->stream));
    if (V > 0) HIPCHK(ctx, hipMemcpyAsync(ctx->d_slabs, slabs.data(), (size_t)V * SRL_SLAB_BYTES, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_table, table.data(), (size_t)table_cap * sizeof(SrlMapSlot), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->slab_cap = slab_cap;
    ctx->table_cap = table_cap;
    ctx->num_voxels = V;
    ctx->num_points = npts;
    return SRL_OK;
}

int srl_map_size(srl_ctx *ctx, int64_t *num_points, int32_t *num_voxels) {
    if (!ctx) return SRL_ERR_BAD_ARG;
    { const int rcs = srl_map_settle(ctx); if (rcs) return rcs; }      // (a deferred insert's counters: srl_frame_commit without num_added)
    if (num_points) *num_points = ctx->num_points;
    if (num_voxels) *num_voxels = ctx->num_voxels;
    return SRL_OK;
}

int srl_map_download(srl_ctx *ctx, int16_t *keys_xyz, int32_t *counts, float *xyz, int max_voxels) {
    if (!ctx) return SRL_ERR_BAD_ARG;
    SRL_DISARM(ctx);
    if (!ctx->d_slabs) return SRL_ERR_NO_MAP;
    { const int rcs = srl_map_settle(ctx); if (rcs) return rcs; }
    const int V = ctx->num_voxels;
    if (max_voxels < V) return SRL_ERR_BAD_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    std::vector<SrlSlab> slabs((size_t)V);
    if (V > 0) HIPCHK(ctx, hipMemcpyAsync(slabs.data(), ctx->d_slabs, (size_t)V * SRL_SLAB_BYTES, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (int v = 0; v < V; v++) {
        short x, y, z;
        srl_unpack_key(slabs[v].key, &x, &y, &z);
        if (keys_xyz) { keys_xyz[3 * v] = x; keys_xyz[3 * v + 1] = y; keys_xyz[3 * v + 2] = z; }
        if (counts) counts[v] = (int32_t)slabs[v].count;
        if (xyz)
            for (int i = 0; i < SRL_CAP; i++)
                for (int d = 0; d < 3; d++)
                    xyz[((size_t)v * SRL_CAP + i) * 3 + d] = (i < (int)slabs[v].count) ? slabs[v].xyz[i][d] : 0.0f;
    }
    return SRL_OK;
}

int srl_map_insert(srl_ctx *ctx, const double *world_xyz, int n, double voxel_size, int cap,
                   double min_distance_points, int min_num_points, int *num_added) {
    if (!ctx || n < 0 || (n > 0 && !world_xyz)) return SRL_ERR_BAD_ARG;
    SRL_DISARM(ctx);
    if (cap != SRL_VOXEL_CAP) { ctx->err = "max_num_points_in_voxel must be 20"; return SRL_ERR_UNSUPPORTED; }
    return srl_map_insert_device(ctx, world_xyz, n, voxel_size, cap, min_distance_points, min_num_points, num_added);
}

// ------------------------------------------------------------------------------------------ sweep
namespace {
// AoS keypoints host -> device staging buffer `d_stage` on stream `st` (no synchronisation): one DMA from page-locked memory,
// else through the pinned ring (CPU copy of chunk i + 1 overlaps the DMA of chunk i; the caller's buffer is consumed on return)
int upload_aos(srl_ctx *ctx, const char *src, size_t bytes, double *d_stage, hipStream_t st, bool last = true, int pinned = -1) {
    if (pinned < 0 ? srl_is_pinned(src) : pinned != 0) {
        HIPCHK(ctx, hipMemcpyAsync(d_stage, src, bytes, hipMemcpyHostToDevice, st));
        // the caller's buffer is being read by the DMA engine: srl_sweep_wait() returns once it is free again
        // (`last`: the final piece of an upload that goes out in several DMAs -- one event behind all of them)
        if (last) {
            if (!ctx->upload_ev) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->upload_ev, hipEventDisableTiming));
            HIPCHK(ctx, hipEventRecord(ctx->upload_ev, st));
            ctx->upload_pending = true;
        }
        return SRL_OK;
    }
    int rc2 = srl_ring_init(ctx);
    if (rc2) return rc2;
    size_t off = 0;
    while (off < bytes) {
        const size_t len = std::min(bytes - off, (size_t)srl_ctx::RING_SLOT_BYTES);
        const int slot = ctx->ring_next++ % srl_ctx::RING_SLOTS;
        if (ctx->ring_busy[slot]) { HIPCHK(ctx, hipEventSynchronize(ctx->ring_ev[slot])); ctx->ring_busy[slot] = false; }
        std::memcpy(ctx->h_ring + (size_t)slot * srl_ctx::RING_SLOT_BYTES, src + off, len);
        {
            hipError_t e__ = hipMemcpyAsync(reinterpret_cast<char *>(d_stage) + off, ctx->h_ring + (size_t)slot * srl_ctx::RING_SLOT_BYTES, len,
                                            hipMemcpyHostToDevice, st);
            if (e__ != hipSuccess) {
                char buf[256];
                std::snprintf(buf, sizeof buf, "ring upload: %s (dst %p + %zu, ring %p slot %d, len %zu, stream %p, main stream %p)", hipGetErrorString(e__),
                              (void *)d_stage, off, (void *)ctx->h_ring, slot, len, (void *)st, (void *)ctx->stream);
                ctx->err = buf;
                return SRL_ERR_HIP;
            }
        }
        HIPCHK(ctx, hipEventRecord(ctx->ring_ev[slot], st));
        ctx->ring_busy[slot] = true;
        off += len;
    }
    return SRL_OK;
}
}  // namespace

int srl_sweep_upload(srl_ctx *ctx, const double *raw_xyz, int n) {
    if (!ctx || n < 0 || (n > 0 && !raw_xyz)) return SRL_ERR_BAD_ARG;
    SRL_DISARM(ctx);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int b = 0, cnt = 0;
    srl_shard_range(n, ctx->nranks, ctx->rank, &b, &cnt);
    ctx->total_n = n;
    ctx->shard_begin = b;
    ctx->n = cnt;
    ctx->sweep_loaded = true;
    ctx->taps_valid = false;
    ctx->passes_in_solve = 0;
    ctx->tail_pending = false;
    ctx->bound_n = 0;
    if (cnt > ctx->sweep_cap) {
        const int cap = std::max(cnt, 1024);
        int rc = ensure(ctx, ctx->d_raw, (size_t)cap * 3);
        if (rc) return rc;
        ctx->sweep_cap = cap;
    }
    int rc = ensure_work(ctx, cnt);
    if (rc) return rc;
    if (cnt > 0) {
        // stage AoS in the rec buffer (>= 8 doubles per keypoint), transpose to SoA on the device.  Everything is
        // stream-ordered with the solve that follows: no synchronisation here.
        // (caller's buffer page-locked -- srl_pinned_alloc / srl_host_register: one DMA straight from it, and the caller
        //  must leave it untouched until the next call that returns results; pageable: through the context's pinned ring)
        int rcu = upload_aos(ctx, reinterpret_cast<const char *>(raw_xyz + (size_t)b * 3), (size_t)cnt * 3 * sizeof(double), ctx->d_rec, ctx->stream);
        if (rcu) return rcu;
        HIPCHK(ctx, srl_launch_aos_to_soa(ctx->d_rec, cnt, ctx->d_raw, ctx->d_raw + ctx->sweep_cap, ctx->d_raw + 2 * (size_t)ctx->sweep_cap, ctx->stream));
    }
    ctx->soa_valid_n = cnt;
    return SRL_OK;
}

// The NEXT sweep, uploaded while the current one is being solved: DMA + SoA transpose run on the context's copy stream
// into a second sweep buffer; srl_sweep_swap makes it current (the compute stream waits on the upload's event -- the
// host does not).  With a node that receives sweep k + 1 while it solves sweep k, the H2D hop leaves the critical path.
int srl_sweep_wait(srl_ctx *ctx) {
    if (!ctx) return SRL_ERR_BAD_ARG;
    SRL_DISARM(ctx);
    if (ctx->upload_pending) {
        HIPCHK(ctx, hipSetDevice(ctx->device));
        HIPCHK(ctx, hipEventSynchronize(ctx->upload_ev));
        if (ctx->prefix_stream) HIPCHK(ctx, hipStreamSynchronize(ctx->prefix_stream));      // (a prefix-first upload: its first piece travelled there)
        ctx->upload_pending = false;
    }
    return SRL_OK;
}

int srl_sweep_prefetch(srl_ctx *ctx, const double *raw_xyz, int n) {
    if (!ctx || n < 0 || (n > 0 && !raw_xyz)) return SRL_ERR_BAD_ARG;
    // An armed launch stays: the upload runs on the copy stream into the context's OTHER sweep buffer, which the launch only reads when
    // it is fired for that sweep (srl_sweep_swap).  (Its prologue may read the buffer it was armed on while this upload rewrites it:
    // those values are discarded -- a launch fired for another sweep recomputes them, assoc_body's prologue.)
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (!ctx->copy_stream) SRL_DISARM(ctx);       // first use: stream / event creation may synchronise the device
    { const int rcc = ensure_copy_stream(ctx); if (rcc) return rcc; }
    if (!ctx->up_ev[0][0]) {
        SRL_DISARM(ctx);
        for (int sl = 0; sl < 2; sl++) for (int k = 0; k < 2; k++) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->up_ev[sl][k], hipEventDisableTiming));
    }
    int b = 0, cnt = 0;
    srl_shard_range(n, ctx->nranks, ctx->rank, &b, &cnt);
    if (cnt > ctx->next_cap || cnt > ctx->stage_next_cap) {
        const int cap = std::max(cnt, 1024);
        SRL_DISARM(ctx);                                                // the buffers move (hipFree waits for the whole device)
        HIPCHK(ctx, hipStreamSynchronize(ctx->copy_stream));
        int rc;
        if (cnt > ctx->next_cap) { if ((rc = ensure(ctx, ctx->d_raw_next, (size_t)cap * 3))) return rc; ctx->next_cap = cap; HIPCHK(ctx, hipMemsetAsync(ctx->d_raw_next, 0, (size_t)cap * 3 * sizeof(double), ctx->copy_stream)); }
        if (cnt > ctx->stage_next_cap) { if ((rc = ensure(ctx, ctx->d_stage_next, (size_t)cap * 3))) return rc; ctx->stage_next_cap = cap; }
    }
    // DMA only: the points stay AoS in the staging buffer and are transposed by the first pass that reads them (SrlAssocArgs::aos) --
    // a transpose kernel on the copy stream would wait for compute units the association kernels (resident back to back, armed
    // launches included) do not release before the solve it is meant to overlap has ended.
    // Prefix first: what the first pass over this sweep will visit (the prefix the running solve's passes visit) gets its own DMA and event
    const int slot = ctx->next_slot ^ 1;
    const int pre = (ctx->prefix_hint > 0 && ctx->prefix_hint < cnt) ? ctx->prefix_hint : cnt;
    if (cnt > 0) {
        const char *src = reinterpret_cast<const char *>(raw_xyz + (size_t)b * 3);
        const int pinned = srl_is_pinned(src) ? 1 : 0;                  // (asked of the runtime once per sweep, not per piece)
        if (pre < cnt && pinned) {
            // two DMAs side by side: the prefix on its own stream (its event: up_ev[slot][0]), the rest on the copy stream
            if (!ctx->prefix_stream) { SRL_DISARM(ctx); HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->prefix_stream, hipStreamNonBlocking)); }
            int rcu = upload_aos(ctx, src, (size_t)pre * 3 * sizeof(double), ctx->d_stage_next, ctx->prefix_stream, false, 1);
            if (rcu) return rcu;
            HIPCHK(ctx, hipEventRecord(ctx->up_ev[slot][0], ctx->prefix_stream));
            rcu = upload_aos(ctx, src + (size_t)pre * 3 * sizeof(double), (size_t)(cnt - pre) * 3 * sizeof(double), ctx->d_stage_next + (size_t)pre * 3, ctx->copy_stream, true, 1);
            if (rcu) return rcu;
            ctx->up_one_dma[slot] = false;
        } else {
            // (pageable memory goes through the one pinned ring, chunk by chunk, on one stream)
            int rcu = upload_aos(ctx, src, (size_t)cnt * 3 * sizeof(double), ctx->d_stage_next, ctx->copy_stream, true, pinned);
            if (rcu) return rcu;
            ctx->up_one_dma[slot] = true;
        }
    } else {
        ctx->up_one_dma[slot] = true;
    }
    // (one DMA: its event is both the prefix's and the tail's -- every event record is ~1.5 us of the host's time beside the first kernel)
    HIPCHK(ctx, hipEventRecord(ctx->up_ev[slot][1], ctx->copy_stream));
    ctx->next_ready = ctx->up_ev[slot][1];
    ctx->next_slot = slot; ctx->next_prefix_n = ctx->up_one_dma[slot] ? cnt : pre;
    ctx->next_n = cnt; ctx->next_begin = b; ctx->next_total = n;
    return SRL_OK;
}

int srl_sweep_swap(srl_ctx *ctx) {
    if (!ctx) return SRL_ERR_BAD_ARG;
    if (ctx->next_n < 0) { SRL_DISARM(ctx); ctx->err = "srl_sweep_swap: nothing prefetched"; return SRL_ERR_NO_SWEEP; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    // A launch armed behind the last pass of the sweep that ends here carries BOTH sweep buffers: if the upload has already landed (the
    // usual case: it was issued a whole solve ago) the launch stays and becomes the first pass of the sweep swapped in -- fired through
    // the pose box with SRL_ARM_ALT, no launch on the critical path of the new solve.  An upload still in flight is awaited by the
    // compute stream as before, and a launch already waiting in front of that dependency is cancelled (it could start too early).
    // A waiting launch only needs the PREFIX of the new sweep (it serves a pass of at most as many keypoints as the pass it was armed
    // behind -- with a finite max_num_residuals the first few thousand): the prefix went out first with an event of its own.  The rest may
    // still be crossing PCIe; `tail_pending` makes the first pass that needs it order the stream behind the full event (build_residuals_pass).
    const bool one_dma = ctx->up_one_dma[ctx->next_slot];
    hipEvent_t ev_full = ctx->up_ev[ctx->next_slot][1], ev_pre = ctx->up_ev[ctx->next_slot][one_dma ? 1 : 0];
    hipError_t up_pre = hipEventQuery(ev_pre);                           // (the prefix travels on a stream of its own: asked first)
    hipError_t up = (one_dma || up_pre != hipSuccess) ? up_pre : hipEventQuery(ev_full);
    if (up_pre == hipErrorNotReady && ctx->armed && ctx->next_n <= ctx->work_cap) {
        // the prefix itself is still on its way (a 100 KB DMA behind the tail of the sweep before it) and a launch is waiting that could
        // serve the new sweep: cancelling it and launching afresh costs ~8 us -- give the DMA up to 25 us first
        const long long t_give_up = steady_ns() + 25000;
        while ((up_pre = hipEventQuery(ev_pre)) == hipErrorNotReady && steady_ns() < t_give_up) { }
    }
    for (hipError_t e : {up, up_pre})
        if (e != hipSuccess && e != hipErrorNotReady) { ctx->err = std::string("hipEventQuery: ") + hipGetErrorString(e); return SRL_ERR_HIP; }
    if (up_pre != hipSuccess || ctx->next_n > ctx->work_cap) SRL_DISARM(ctx);   // (growing the work buffers frees them: never under a waiting launch)
    if (up_pre != hipSuccess) {
        HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ev_full, 0));       // compute waits for the upload; the host does not
        if (!one_dma) HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ev_pre, 0));
        up = hipSuccess;                                               // (the stream is ordered behind everything)
    }
    ctx->tail_pending = up != hipSuccess;
    ctx->bound_n = 0;                                       // another sweep: the bounds of the old one say nothing about it
    ctx->cur_slot = ctx->next_slot;
    ctx->cur_prefix_n = up == hipSuccess ? ctx->next_n : ctx->next_prefix_n;
    ctx->passes_in_solve = 0;
    std::swap(ctx->d_raw, ctx->d_raw_next);
    std::swap(ctx->sweep_cap, ctx->next_cap);
    std::swap(ctx->d_stage_cur, ctx->d_stage_next);
    std::swap(ctx->stage_cur_cap, ctx->stage_next_cap);
    ctx->soa_valid_n = 0;                                   // the planes of d_raw are filled by the first pass over the sweep
    ctx->n = ctx->next_n; ctx->shard_begin = ctx->next_begin; ctx->total_n = ctx->next_total;
    ctx->next_n = -1;
    ctx->sweep_loaded = true;
    ctx->taps_valid = false;
    return ensure_work(ctx, ctx->n);
}

int srl_thread_pin_to_gpu_numa(srl_ctx *ctx, int *numa_node) {
    if (!ctx) return SRL_ERR_BAD_ARG;
    if (numa_node) *numa_node = -1;
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, ctx->device) != hipSuccess) { (void)hipGetLastError(); return SRL_ERR_UNSUPPORTED; }
    for (char *c = bus; *c; ++c) if (*c >= 'A' && *c <= 'F') *c = (char)(*c - 'A' + 'a');     // sysfs spells the address in lower case
    const std::string base = std::string("/sys/bus/pci/devices/") + bus;
    int node = -1;
    if (FILE *f = std::fopen((base + "/numa_node").c_str(), "r")) { if (std::fscanf(f, "%d", &node) != 1) node = -1; std::fclose(f); }
    char list[1024] = {0};
    if (FILE *f = std::fopen((base + "/local_cpulist").c_str(), "r")) { if (!std::fgets(list, (int)sizeof list, f)) list[0] = 0; std::fclose(f); }
    if (node < 0 || !list[0]) return SRL_ERR_UNSUPPORTED;
    cpu_set_t allowed, want;
    CPU_ZERO(&want);
    if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return SRL_ERR_UNSUPPORTED;
    int any = 0;
    for (const char *p = list; *p && *p != '\n';) {           // "0-63,128-191"
        char *end = nullptr;
        const long a = std::strtol(p, &end, 10);
        long b = a;
        if (end == p) break;
        p = end;
        if (*p == '-') { b = std::strtol(p + 1, &end, 10); p = end; }
        for (long c = a; c <= b && c < CPU_SETSIZE; ++c) if (CPU_ISSET((int)c, &allowed)) { CPU_SET((int)c, &want); any = 1; }
        if (*p == ',') ++p;
    }
    if (!any) return SRL_ERR_UNSUPPORTED;                     // the caller's own affinity excludes that node: leave it alone
    if (sched_setaffinity(0, sizeof want, &want) != 0) return SRL_ERR_UNSUPPORTED;
    if (numa_node) *numa_node = node;
    return SRL_OK;
}

int srl_pinned_alloc(size_t bytes, void **out) {
    if (!out) return SRL_ERR_BAD_ARG;
    *out = nullptr;
    if (hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return SRL_ERR_HIP; }
    return SRL_OK;
}
int srl_pinned_free(void *p) {
    if (!p) return SRL_OK;
    return hipHostFree(p) == hipSuccess ? SRL_OK : SRL_ERR_HIP;
}
int srl_host_register(void *p, size_t bytes) {
    if (!p || !bytes) return SRL_ERR_BAD_ARG;
    if (hipHostRegister(p, bytes, hipHostRegisterDefault) != hipSuccess) { (void)hipGetLastError(); return SRL_ERR_HIP; }
    return SRL_OK;
}
int srl_host_unregister(void *p) {
    if (!p) return SRL_ERR_BAD_ARG;
    if (hipHostUnregister(p) != hipSuccess) { (void)hipGetLastError(); return SRL_ERR_HIP; }
    return SRL_OK;
}

int srl_sweep_shard(srl_ctx *ctx, int *begin, int *count, int *total) {
    if (!ctx) return SRL_ERR_BAD_ARG;
    if (begin) *begin = ctx->shard_begin;
    if (count) *count = ctx->n;
    if (total) *total = ctx->total_n;
    return SRL_OK;
}

int srl_set_taps(srl_ctx *ctx, int enable) {
    if (!ctx) return SRL_ERR_BAD_ARG;
    SRL_DISARM(ctx);
    ctx->taps = enable != 0;
    if (!ctx->taps) ctx->taps_valid = false;
    return SRL_OK;
}

namespace {
// light profiling: read the (start, end) event pairs of association launches that have completed by now
int drain_ring(srl_ctx *ctx, bool all) {
    while (ctx->ring_tail != ctx->ring_head) {
        hipEvent_t *e = ctx->ring[ctx->ring_tail % srl_ctx::PROF_RING];
        if (!all && hipEventQuery(e[1]) != hipSuccess) break;
        if (all) HIPCHK(ctx, hipEventSynchronize(e[1]));
        const unsigned slot = ctx->ring_tail % srl_ctx::PROF_RING;
        if (ctx->ring_void[slot] || ctx->ring_marker[slot]) {      // a cancelled armed launch / an entry that was only the next launch's start
            ctx->ring_void[slot] = false; ctx->ring_marker[slot] = false; ctx->ring_tail++;
            continue;
        }
        if (ctx->ring_gen[slot] != ctx->timing_gen) { ctx->ring_tail++; continue; }      // enqueued before the last srl_timing_mark
        float ms = 0.f;
        // an armed launch has no start event of its own: it is enqueued behind the launch of the pass before it and the stream turns
        // to it the moment that one completes -- its start IS the end event of its predecessor in the ring
        hipEvent_t start = ctx->ring_prev[slot] >= 0 ? ctx->ring[ctx->ring_prev[slot]][1] : e[0];
        HIPCHK(ctx, hipEventElapsedTime(&ms, start, e[1]));
        ctx->timing.assoc_ms = ms;
        ctx->timing.sum_assoc_ms += ms;
        ctx->timing.calls += 1;
        ctx->ring_tail++;
    }
    return SRL_OK;
}
}  // namespace

int srl_set_profiling(srl_ctx *ctx, int enable) {
    if (!ctx || enable < 0 || enable > 3) return SRL_ERR_BAD_ARG;
    SRL_DISARM(ctx);
    if (ctx->profiling == 2) { int rc = drain_ring(ctx, true); if (rc) return rc; }
    ctx->profiling = enable;
    if (ctx->profiling) std::memset(&ctx->timing, 0, sizeof ctx->timing);
    if (ctx->profiling == 2) {
        HIPCHK(ctx, hipSetDevice(ctx->device));
        for (int i = 0; i < srl_ctx::PROF_RING; i++)
            for (int k = 0; k < 2; k++) if (!ctx->ring[i][k]) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ring[i][k], hipEventReleaseToDevice));   // (device-scope release: a timing marker between two kernels must not write the L2 back)
        ctx->ring_head = ctx->ring_tail = 0;
        std::memset(ctx->ring_void, 0, sizeof ctx->ring_void);
        std::memset(ctx->ring_marker, 0, sizeof ctx->ring_marker);
        ctx->prof_count = 0; ctx->ring_last_count = -2; ctx->armed_measured = false; ctx->cur_measured = true;
    }
    return SRL_OK;
}

int srl_timing_mark(srl_ctx *ctx) {
    if (!ctx) return SRL_ERR_BAD_ARG;
    ctx->timing_gen++;
    std::memset(&ctx->timing, 0, sizeof ctx->timing);
    // with a period, the first timed launch behind the mark is the FOURTH (then every period-th): the launches of the first step behind a
    // barrier -- an un-armed launch into an idle GPU and the pass behind it -- would otherwise be one sample in eight of a 20-step region
    ctx->prof_count = ctx->prof_period > 1 ? (unsigned long long)(ctx->prof_period - 2) : 0ull;
    ctx->ring_last_count = -2;
    return SRL_OK;
}

int srl_set_profiling_period(srl_ctx *ctx, int period) {
    if (!ctx || period < 1 || period > 64) return SRL_ERR_BAD_ARG;
    SRL_DISARM(ctx);
    if (ctx->profiling == 2) { int rc = drain_ring(ctx, true); if (rc) return rc; }
    ctx->prof_period = period;
    ctx->prof_count = 0; ctx->ring_last_count = -2; ctx->armed_measured = false; ctx->cur_measured = true;
    return SRL_OK;
}

// debug (SRL_ABLATE=128): per-workgroup {start, end, xcc} stamps of the last association launch, 100 MHz ticks
int srl_debug_block_times(srl_ctx *ctx, double *out, int max_blocks, int *nblocks) {
    if (!ctx || !out || !nblocks) return SRL_ERR_BAD_ARG;
    SRL_DISARM(ctx);
    const int nb = std::min(max_blocks, ctx->last_nblocks);
    std::vector<double> tmp((size_t)nb * SRL_PART_STRIDE);
    HIPCHK(ctx, hipMemcpy(tmp.data(), ctx->d_partials, tmp.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (int b = 0; b < nb; b++) for (int k = 0; k < 3; k++) out[(size_t)b * 3 + k] = tmp[(size_t)b * SRL_PART_STRIDE + 28 + k];
    *nblocks = nb;
    return SRL_OK;
}

int srl_get_timing(srl_ctx *ctx, srl_timing *t) {
    if (!ctx || !t) return SRL_ERR_BAD_ARG;
    SRL_DISARM(ctx);
    if (ctx->profiling == 2) { int rc = drain_ring(ctx, true); if (rc) return rc; }
    *t = ctx->timing;
    return SRL_OK;
}

// ------------------------------------------------------------------------------------------ armed launches
// The per-iteration loop of updateIEKF (optimize.cpp:147-312) is: kernel -> normal equations -> 17-dim update on the host -> next pose ->
// kernel.  The launch call (~3 us), the dispatch (~1.5 us) and the ramp of the next kernel used to sit on that critical path.  An ARMED
// launch is the next pass's kernel enqueued while the current pass is still running (the host has nothing else to do then): its
// arguments are the current pass's, except the pose, which does not exist yet and arrives later through the POSE BOX -- 43 tagged
// 8-byte granules the host writes when the update is done (pose_box_write) and wave 0 of every workgroup polls (assoc_body's
// prologue in srl_kernels.hip).  A pass whose arguments equal the armed launch's FIRES it (one 384-byte write instead of a launch);
// anything else -- other options, another sweep, another entry point -- cancels it (control granule) and launches normally.
namespace {

// Can the CPU store into device memory (pose box kind 1)?  Asked of the runtime, never probed by faulting: the device must report a
// large PCIe BAR (hipDeviceAttributeIsLargeBar: the whole of its memory is CPU-addressable) and the runtime must know the
// allocation as device memory with a host-usable address.  One answer per device, computed once under a lock.
bool device_memory_is_host_writable(int device, const void *p) {
    static std::mutex mu;
    static int cached[SRL_MAX_DEVICES];            // 0 unknown, 1 yes, -1 no
    std::lock_guard<std::mutex> lk(mu);
    if (device >= 0 && device < SRL_MAX_DEVICES && cached[device] != 0) return cached[device] > 0;
    int large_bar = 0;
    bool ok = hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, device) == hipSuccess && large_bar != 0;
    if (ok) {
        hipPointerAttribute_t at;
        ok = hipPointerGetAttributes(&at, p) == hipSuccess && at.type == hipMemoryTypeDevice && at.devicePointer == p;
    }
    (void)hipGetLastError();
    if (device >= 0 && device < SRL_MAX_DEVICES) cached[device] = ok ? 1 : -1;
    return ok;
}

// a word that names the device within this node: FNV-1a of its PCI bus id (never 0)
unsigned long long device_identity(int device) {
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, device) != hipSuccess) { (void)hipGetLastError(); std::snprintf(bus, sizeof bus, "device-%d", device); }
    unsigned long long h = 1469598103934665603ull;
    for (const char *c = bus; *c; ++c) { h ^= (unsigned char)(*c >= 'A' && *c <= 'Z' ? *c - 'A' + 'a' : *c); h *= 1099511628211ull; }
    return h ? h : 1ull;
}

int ensure_pose_box(srl_ctx *ctx) {
    if (ctx->h_pose_box) return SRL_OK;
    if (!ctx->d_pose_relay) {
        HIPCHK(ctx, hipMalloc((void **)&ctx->d_pose_relay, SRL_POSE_BOX_GRANULES * 8));
        HIPCHK(ctx, hipMemset(ctx->d_pose_relay, 0, SRL_POSE_BOX_GRANULES * 8));
    }
    // kind -1 (default): device memory when the CPU can write it (large BAR: every workgroup polls locally, ~0.7 us less per pass
    // than a box in host memory that workgroup 0 polls across PCIe and relays), else pinned host memory
    if (ctx->pose_box_kind != 0) {
        if (!ctx->pose_box_dev) {
            HIPCHK(ctx, hipExtMallocWithFlags((void **)&ctx->pose_box_dev, 4096, hipDeviceMallocFinegrained));
            HIPCHK(ctx, hipMemset(ctx->pose_box_dev, 0, 4096));
            HIPCHK(ctx, hipDeviceSynchronize());
            ctx->pose_box_dev_visible = device_memory_is_host_writable(ctx->device, ctx->pose_box_dev);
        }
        if (ctx->pose_box_dev_visible) { ctx->pose_box_kind = 1; ctx->h_pose_box = ctx->pose_box_dev; return SRL_OK; }
        if (ctx->pose_box_kind == 1) { ctx->err = "pose box: device memory is not CPU-visible on this system (no large BAR)"; return SRL_ERR_UNSUPPORTED; }
        ctx->pose_box_kind = 0;
    }
    if (!ctx->pose_box_pinned) {
        HIPCHK(ctx, hipHostMalloc((void **)&ctx->pose_box_pinned, 4096, hipHostMallocCoherent | hipHostMallocMapped));
        std::memset(ctx->pose_box_pinned, 0, 4096);
    }
    ctx->h_pose_box = ctx->pose_box_pinned;
    return SRL_OK;
}

// the pose of launch `epoch` (or its cancellation: code = SRL_ARM_CANCEL, pose ignored): 48 tagged granules = six 64-byte lines
inline void pose_box_write(srl_ctx *ctx, const double *Rn, const double *R, const double *t, unsigned epoch, unsigned code, unsigned n = 0u,
                           const double *t_last = nullptr) {
    unsigned long long line[SRL_POSE_BOX_WRITTEN];
    const unsigned long long tag = (unsigned long long)epoch << 32;
    auto put = [&](int d, double v) {
        unsigned long long bits;
        std::memcpy(&bits, &v, 8);
        line[2 * d] = tag | (bits & 0xFFFFFFFFull);
        line[2 * d + 1] = tag | (bits >> 32);
    };
    for (int i = 0; i < 9; i++) { put(i, Rn ? Rn[i] : 0.0); put(9 + i, R ? R[i] : 0.0); }
    for (int i = 0; i < 3; i++) put(18 + i, t ? t[i] : 0.0);
    line[SRL_POSE_BOX_CTRL] = tag | code;
    line[SRL_POSE_BOX_N] = tag | n;                          // keypoints of the pass (the launch may serve another sweep than it was armed on)
    for (int i = 0; i < 3; i++) {                            // t_last (optimize.cpp:25) belongs to the sweep, like the pose
        unsigned long long bits = 0ull;
        if (t_last) std::memcpy(&bits, &t_last[i], 8);
        line[SRL_POSE_BOX_TLAST + 2 * i] = tag | (bits & 0xFFFFFFFFull);
        line[SRL_POSE_BOX_TLAST + 2 * i + 1] = tag | (bits >> 32);
    }
    for (int i = SRL_POSE_BOX_USED; i < SRL_POSE_BOX_WRITTEN; i++) line[i] = tag;
    volatile unsigned long long *box = ctx->h_pose_box;
    for (int i = 0; i < SRL_POSE_BOX_WRITTEN; i++) box[i] = line[i];          // every granule validates itself: no ordering between the stores is needed
    __atomic_thread_fence(__ATOMIC_SEQ_CST);                // ... only that they leave the core now (mfence: drains write-combining buffers too)
}
}  // namespace

int srl_ctx_disarm(srl_ctx *ctx) {
    if (!ctx || !ctx->armed) return SRL_OK;
    pose_box_write(ctx, nullptr, nullptr, nullptr, (unsigned)ctx->armed_sig.seq, SRL_ARM_CANCEL);
    if (ctx->seq < ctx->armed_sig.seq) ctx->seq = ctx->armed_sig.seq;      // the cancelled launch's sequence number is spent
    ctx->armed = false;
    if (ctx->armed_ring >= 0) ctx->ring_void[ctx->armed_ring] = true;
    ctx->armed_ring = -1;
    ctx->arm_stats[2]++;
    return SRL_OK;
}

int srl_disarm(srl_ctx *ctx) {
    if (!ctx) return SRL_ERR_BAD_ARG;
    return srl_ctx_disarm(ctx);
}

int srl_set_armed_launch(srl_ctx *ctx, int mode) {
    if (!ctx || mode < 0 || mode > 2) return SRL_ERR_BAD_ARG;
    SRL_DISARM(ctx);
    ctx->arm_mode = mode;
    return SRL_OK;
}

int srl_debug_set_pose_box(srl_ctx *ctx, int kind) {
    if (!ctx || kind < -1 || kind > 1) return SRL_ERR_BAD_ARG;
    SRL_DISARM(ctx);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    const int old = ctx->pose_box_kind;
    ctx->pose_box_kind = kind;
    ctx->h_pose_box = nullptr;
    const int rc = ensure_pose_box(ctx);
    if (rc) { ctx->pose_box_kind = old; ctx->h_pose_box = nullptr; }
    return rc;
}

int srl_debug_set_arm_linger(srl_ctx *ctx, double host_linger_us, double kernel_linger_us) {
    if (!ctx || !(host_linger_us >= 0.0) || !(kernel_linger_us > 0.0) || kernel_linger_us > 4.0e7) return SRL_ERR_BAD_ARG;
    SRL_DISARM(ctx);
    ctx->arm_host_linger_us = host_linger_us;
    ctx->arm_linger_ticks = (unsigned)(kernel_linger_us * 100.0);      // 100 MHz clock
    return SRL_OK;
}

int srl_debug_pass_stamps(srl_ctx *ctx, int enable, long long *gpu_out, long long *host_out) {
    if (!ctx) return SRL_ERR_BAD_ARG;
    SRL_DISARM(ctx);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (enable && !ctx->h_arm_stamps) {
        HIPCHK(ctx, hipHostMalloc((void **)&ctx->h_arm_stamps, 64 * 32 * sizeof(long long), hipHostMallocCoherent | hipHostMallocMapped));
        std::memset(ctx->h_arm_stamps, 0, 64 * 32 * sizeof(long long));
    }
    if (gpu_out && ctx->h_arm_stamps) std::memcpy(gpu_out, ctx->h_arm_stamps, 64 * 32 * sizeof(long long));
    if (host_out) std::memcpy(host_out, ctx->arm_host_stamps, sizeof ctx->arm_host_stamps);
    if (!enable && ctx->h_arm_stamps) { hipHostFree(ctx->h_arm_stamps); ctx->h_arm_stamps = nullptr; }
    return SRL_OK;
}

int srl_debug_set_frame_epoch(srl_ctx *ctx, int frames_to_wrap) {
    if (!ctx || frames_to_wrap < 0 || frames_to_wrap > 0xFFFF) return SRL_ERR_BAD_ARG;
    SRL_DISARM(ctx);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (SrlEpochTable *t : {&ctx->sel_table, &ctx->ins_table}) {
        // entries written so far carry epochs <= the old counter: moving the counter forward keeps all of them stale
        const unsigned target = 0xFFFFu - (unsigned)frames_to_wrap;
        if (t->epoch16 > target) return SRL_ERR_BAD_ARG;
        t->epoch16 = target;
    }
    return SRL_OK;
}

int srl_debug_frame_timing(srl_ctx *ctx, int enable, double out16[16]) {
    if (!ctx) return SRL_ERR_BAD_ARG;
    SRL_DISARM(ctx);
    if (out16) std::memcpy(out16, ctx->frame_stage_us, sizeof ctx->frame_stage_us);
    std::memset(ctx->frame_stage_us, 0, sizeof ctx->frame_stage_us);
    ctx->frame_timing = enable != 0;
    return SRL_OK;
}

int srl_get_arm_stats(srl_ctx *ctx, uint64_t out[4]) {
    if (!ctx || !out) return SRL_ERR_BAD_ARG;
    for (int i = 0; i < 4; i++) out[i] = ctx->arm_stats[i];
    return SRL_OK;
}

// ------------------------------------------------------------------------------------------ hot path
// one association + reduction pass over the first n_eff keypoints of this rank's shard (n_eff == ctx->n: all of them)
#define SRL_INTERNAL_FUSED_TIMEOUT 1      // build_residuals_pass only: never leaves srl_build_residuals
#define SRL_INTERNAL_PEER_TIMEOUT 3       // ... a peer's row had not arrived within the kernel's bounded spin: the HOST decides what follows
#define SRL_INTERNAL_ARM_EXPIRED 2        // the armed launch this pass fired had given up waiting: the pass is repeated with a normal launch

// the kernel arguments of one pass
static int prepare_assoc_args(srl_ctx *ctx, const srl_frame *f, const srl_icp_opts *o, int n_eff, SrlAssocArgs &a, int &nb_out) {
    // init-mode switches (optimize.cpp:21-23)
    const bool init_mode = f->frame_id < o->init_num_frames;
    const int nb = init_mode ? 2 : o->voxel_neighborhood;
    const int thr = init_mode ? 1 : o->threshold_voxel_occupancy;
    const int K = o->max_number_neighbors;
    if (nb < 1 || nb > 2) { ctx->err = "voxel_neighborhood must be 1 or 2"; return SRL_ERR_UNSUPPORTED; }
    if (K < 1 || K > SRL_MAX_NEIGHBORS) { ctx->err = "max_number_neighbors must be in [1,32]"; return SRL_ERR_UNSUPPORTED; }
    if (!(o->size_voxel_map > 0.0)) return SRL_ERR_BAD_ARG;

    std::memset(&a, 0, sizeof a);
    a.raw_x = ctx->d_raw;
    a.raw_y = ctx->d_raw + ctx->sweep_cap;
    a.raw_z = ctx->d_raw + 2 * (size_t)ctx->sweep_cap;
    a.n = n_eff;
    a.aos = (n_eff > ctx->soa_valid_n) ? ctx->d_stage_cur : nullptr;     // the pass reads (all of) its points AoS and files the SoA planes
    a.table = ctx->d_table;
    a.table_mask = ctx->table_cap - 1;
    a.slabs = ctx->d_slabs;
    a.inf_off = ctx->slab_cap * (unsigned)SRL_SLAB_BYTES;
    // records {J, d, w} + status feed the ordered cut-off (optimize.cpp:107) and the parity taps only: 4 MB of stores per
    // 64k sweep that the throughput configuration (max_num_residuals > number of keypoints) never reads
    const bool cut_possible_here = o->max_num_residuals <= 0 || (long long)o->max_num_residuals <= (long long)ctx->total_n;
    a.write_rec = (ctx->taps || cut_possible_here) ? 1 : 0;
    {
        const srl::Quat q(f->q[0], f->q[1], f->q[2], f->q[3]);
        const srl::Mat3 Rn = q.normalized().toRotationMatrix();    // optimize.cpp:35
        const srl::Mat3 R = q.toRotationMatrix();                  // optimize.cpp:95,101
        std::memcpy(a.Rn, Rn.a, sizeof a.Rn);
        std::memcpy(a.R, R.a, sizeof a.R);
    }
    std::memcpy(a.t, f->t, sizeof a.t);
    std::memcpy(a.t_last, f->t_last, sizeof a.t_last);
    std::memcpy(a.R_il, f->R_il, sizeof a.R_il);
    std::memcpy(a.t_il, f->t_il, sizeof a.t_il);
    a.size_voxel = o->size_voxel_map;
    a.max_dist = o->max_dist_to_plane_icp;
    {
        double lw = std::abs(o->weight_alpha), ln = std::abs(o->weight_neighborhood);   // optimize.cpp:55-61
        const double sum = lw + ln;
        lw /= sum;
        ln /= sum;
        a.lambda_w = lw;
        a.lambda_n = ln;
    }
    a.power_planarity = o->power_planarity;
    a.nbr_scale = o->max_dist_to_plane_icp * o->min_number_neighbors;   // kMaxPointToPlane * kMinNumNeighbors
    a.K = K;
    a.min_nb = o->min_number_neighbors;
    a.thr_cap = thr;
    a.select_mode = ctx->select_mode;                           // 0 unless srl_debug_set_select_mode was called (tests)
    a.ablate = ctx->ablate;                                     // 0 unless srl_debug_set_ablate was called (profiling tools only)
    a.stamps = ctx->h_arm_stamps;                               // null unless srl_debug_pass_stamps is on
    a.rec = ctx->d_rec;
    a.status = ctx->d_status;
    a.partials = ctx->d_partials;
    a.binfo = ctx->d_binfo;
    {
        // bounds of the previous pass: usable when they were written with these options (sweep and map changes reset bound_n themselves)
        float sv = (float)o->size_voxel_map;
        int sv_bits; std::memcpy(&sv_bits, &sv, 4);
        const int sig[4] = {K, nb, thr, sv_bits};
        if (std::memcmp(sig, ctx->bound_sig, sizeof sig) != 0) { ctx->bound_n = 0; std::memcpy(ctx->bound_sig, sig, sizeof sig); }
        a.bound_out = ctx->d_bound;
        a.bound_in = ctx->bound_mode != 0 ? ctx->d_bound : nullptr;
        a.bound_use = a.bound_in ? std::min(ctx->bound_n, n_eff) : 0;
    }
    nb_out = nb;
    return SRL_OK;
}

static int build_residuals_pass(srl_ctx *ctx, const srl_frame *f, const srl_icp_opts *o, srl_normal_eq *out, int n_eff) {
    const auto t_entry = std::chrono::steady_clock::now();
    HIPCHK(ctx, hipSetDevice(ctx->device));
    SrlAssocArgs a;
    int nb = 1;
    { const int rca = prepare_assoc_args(ctx, f, o, n_eff, a, nb); if (rca) return rca; }
    const int K = o->max_number_neighbors;
    const bool cut_possible_here = o->max_num_residuals <= 0 || (long long)o->max_num_residuals <= (long long)ctx->total_n;
    ctx->taps_valid = false;
    if (ctx->taps) {
        int rc = ensure_taps(ctx, ctx->n, K);
        if (rc) return rc;
        a.tap_ids = ctx->d_tap_ids;
        a.tap_ncand = ctx->d_tap_ncand;
        a.tap_normal = ctx->d_tap_normal;
        a.tap_a2d = ctx->d_tap_a2d;
        a.tap_offset = ctx->d_tap_offset;
        if (ctx->n > 0) {
            HIPCHK(ctx, hipMemsetAsync(ctx->d_tap_ids, 0xFF, (size_t)ctx->n * K * sizeof(int), ctx->stream));
            HIPCHK(ctx, hipMemsetAsync(ctx->d_tap_normal, 0, (size_t)ctx->n * 3 * sizeof(double), ctx->stream));
            HIPCHK(ctx, hipMemsetAsync(ctx->d_tap_a2d, 0, (size_t)ctx->n * sizeof(double), ctx->stream));
            HIPCHK(ctx, hipMemsetAsync(ctx->d_tap_offset, 0, (size_t)ctx->n * sizeof(double), ctx->stream));
        }
    }
    // launch shape: keypoints per wave by sweep size; 16-wave workgroups (one per CU) whenever their LDS footprint fits
    int kpw = srl_keypoints_per_wave(n_eff);
    // Fused final reduction (the last workgroup sums the block partials and publishes the result itself: one kernel per ESIKF
    // iteration) needs: single rank, every accepted residual counts (no ordered cut can trigger), nobody reads per-keypoint
    // records, and few, large workgroups -- every workgroup publishes a row through memory and the last one re-reads them
    // all.  Measured per srl_build_residuals call (tools/shape_sweep.py; kernel us / wall us):
    //   16k keypoints  4-wave workgroups, 2 kernels 23.4 / 41.1    16-wave fused 27.0 / 32.4
    //   24k keypoints                               30.4 / 47.2                  35.7 / 41.3
    //   64k keypoints  16-wave, 2 kernels           52   / 78                    58   / 72     (1 000 small workgroups fused: +14 us)
    const bool single_rank = ctx->nranks == 1 && !(ctx->comm && ctx->force_coll);
    // A sharded sweep fuses too when no ordered cut can trigger (every rank's finishing workgroup then sums its own shard):
    // the rank's totals go  - straight into the peers' inboxes (direct peer exchange: still ONE kernel per pass),
    //                       - into a device-side mailbox the RCCL all-reduce works on,  or
    //                       - into the host mailbox, for the host all-reduce callback.
    // With the ordered cut a rank's budget depends on the counts of the ranks before it: count kernel, exchange, reduce kernel.
    const bool fuse_any = !ctx->taps && a.ablate == 0 && o->max_num_residuals > 0 && ctx->fuse_reduce && !ctx->dbg_gather;
    const bool fuse_base = single_rank && fuse_any;
    const bool can_fuse = fuse_any && !cut_possible_here;
    // ... and WITH the ordered cut (the shipped max_num_residuals = 600): the finisher also locates the workgroup that holds the
    // max-th accepted residual and re-accumulates that workgroup's records, which travel as tagged granules like the rows
    // (small workgroups only: one record granule per finisher thread)
    bool can_fuse_cut = fuse_base && cut_possible_here;
    // Launch shape.  Sweeps of >= 2 048 keypoints: 16-wave workgroups (one per CU) with the smallest instantiated
    // keypoints-per-wave count that still places the sweep in ONE round of workgroups -- a wave's serial chain is as short as
    // the sweep allows and no CU idles while another runs a second workgroup.  Per call, kernel / wall us (tools/shape_sweep.py):
    //    4 096 keypoints  fused: 4 per wave 25.5 / 31.0 -> 2 per wave 21.5 / 26.8
    //   24 576 keypoints  fused: 8 per wave (192 workgroups) 36.7 / 45.9 -> 6 per wave (256) 33.4 / 42.4;  two kernels: 4-wave
    //                     workgroups x 8 per wave 29.2 / 49.6 -> 27.8 / 47.2
    //   16 384 keypoints  fused: 2 per wave (512 workgroups, two rounds) 37.0 / 42.2 against 4 per wave 28.3 / 33.7
    // Smaller sweeps, or K too large for the 16-wave LDS footprint: 4-wave workgroups.
    int wpb = 4;
    {
        const int k1 = srl_keypoints_per_wave_one_round(n_eff, ctx->num_cu);
        if (n_eff >= 2048 && srl_assoc_lds_bytes(K, nb, k1, 16) <= SRL_LDS_LIMIT) { kpw = k1; wpb = 16; }
    }
    if (ctx->force_kpw) {            // srl_debug_set_launch_shape: tuning experiments only
        kpw = ctx->force_kpw;
        const bool only16 = kpw != 4 && kpw != 8 && kpw != 16;
        wpb = ((ctx->force_wpb == 16 || only16) && srl_assoc_lds_bytes(K, nb, kpw, 16) <= SRL_LDS_LIMIT) ? 16 : 4;
        if (only16 && wpb != 16) kpw = 4;
    }
    if (ctx->ablate != 0 && srl_assoc_lds_bytes(K, nb, 16, 16) <= SRL_LDS_LIMIT) { kpw = 16; wpb = 16; }   // the one shape that carries the debug switches
    const int kpb = kpw * wpb;
    const int nblocks = (n_eff + kpb - 1) / kpb;
    ctx->last_nblocks = nblocks;
    if (nblocks > ctx->block_cap) {  // a forced launch shape can need more workgroups than ensure_work provided for
        int rcb;
        if ((rcb = ensure(ctx, ctx->d_partials, (size_t)nblocks * SRL_PART_STRIDE))) return rcb;
        if ((rcb = ensure(ctx, ctx->d_binfo, (size_t)nblocks))) return rcb;
        ctx->block_cap = nblocks;
        a.partials = ctx->d_partials;
        a.binfo = ctx->d_binfo;
    }
    const bool prof = ctx->profiling == 1;
    const bool prof_light = ctx->profiling == 2;
    hipEvent_t *ring_ev = nullptr;
    if (prof_light) {
        if (ctx->ring_head - ctx->ring_tail >= (unsigned)srl_ctx::PROF_RING) { int rc = drain_ring(ctx, true); if (rc) return rc; }
        ring_ev = ctx->ring[ctx->ring_head % srl_ctx::PROF_RING];
    }

    can_fuse_cut = can_fuse_cut && wpb == 16 && kpb <= SRL_FUSED_CUT_MAX_KPB && nblocks <= 512;      // (the finisher reads one row's counters per thread)
    const bool fused = (can_fuse && wpb == 16 && nblocks <= SRL_FUSED_MAX_BLOCKS) || can_fuse_cut;
    unsigned long long seq_now = ++ctx->seq;          // (a cancelled armed launch below takes this number with it: see there)
    const bool coll = ctx->comm && (ctx->nranks > 1 || ctx->force_coll) && !ctx->peer_on;
    const bool peer = ctx->peer_on && ctx->nranks > 1;
    if (peer && ctx->peer_failed) {
        ctx->err = "direct peer exchange: this session has failed (a rank's row never arrived); srl_peer_detach, srl_peer_export and srl_peer_attach on every rank start a new one";
        return SRL_ERR_COMM;
    }
    unsigned peer_epoch = 0;
    int peer_slot = 0;
    auto next_exchange = [&]() {           // one tag per exchange, the same on every rank (never 0: an untouched inbox holds zeros)
        ++ctx->peer_seq;
        peer_epoch = (unsigned)(ctx->peer_seq & 0xFFFFFFFFull);
        if (peer_epoch == 0) { ++ctx->peer_seq; peer_epoch = 1; }
        peer_slot = (int)(ctx->peer_seq & 1ull);
    };
    const bool tagged_mail = fused && !(ctx->comm && (ctx->nranks > 1 || ctx->force_coll) && !ctx->peer_on);   // fused and not the RCCL form: the finisher reports to the host
    if (fused) {
        a.granules = ctx->d_granules;
        a.mailbox = ctx->h_mail;
        a.mail_tagged = tagged_mail ? 1 : 0;
        a.seq = seq_now;
        if (peer) {
            next_exchange();
            a.peer = ctx->d_peer; a.peer_epoch = peer_epoch; a.peer_slot = peer_slot;
        } else if (coll) {
            if (!ctx->d_mail) { int rcm = ensure(ctx, ctx->d_mail, 1); if (rcm) return rcm; }
            a.mailbox = ctx->d_mail;
        }
    }
    if (can_fuse_cut) {
        const size_t need = (size_t)nblocks * kpb * 16;
        if (need > ctx->rec_granule_cap) {
            int rcg;
            if ((rcg = ensure(ctx, ctx->d_rec_granules, need))) return rcg;
            HIPCHK(ctx, hipMemsetAsync(ctx->d_rec_granules, 0, need * sizeof(unsigned long long), ctx->stream));   // no stale tag can match
            ctx->rec_granule_cap = need;
        }
        a.rec_granules = ctx->d_rec_granules;
        a.cut_max = o->max_num_residuals;
        a.write_rec = 0;                                    // nobody reads the global records: the finisher has the granules
    }

    // ---- armed launch: fire the kernel that is already waiting for this pass, if its arguments are this pass's
    const bool fast_sel = ctx->select_mode == 0 || ctx->select_mode == 4;
    // (a grid larger than the chip is armed too: its first round waits resident, the later rounds find the pose in the box when they start)
    // SHARDED passes are armed too where the rank's pass is ONE kernel that reports to this host: the direct peer exchange (the rows of the
    // ranks meet inside the finishing workgroups; the armed launch carries the tag of the NEXT exchange -- every fused pass is exactly one)
    // and the host-callback transport.  Every rank arms and fires its own launch from its own copy of the 17-dim update; a rank whose
    // launch was cancelled or gave up simply launches the pass -- the exchange does not care how a kernel got there.  Not with RCCL
    // (the all-reduce and the publish kernel sit between two passes on the stream) and not with the ordered cut (three exchanges per pass).
    const bool host_cb = ctx->nranks > 1 && !coll && !peer && ctx->cb_ar != nullptr && !ctx->dbg_gather;
    // RCCL: armed as well when the pass is fused -- the launch is enqueued BEHIND this pass's all-reduce and publish kernel (stream order),
    // so it is resident about when the host reads the result; a launch that gave up contributes a time-out flag to the reduced range
    // (assoc_body's prologue) so that every rank repeats the pass.
    const bool arm_ok = ctx->arm_mode != 0 && fused && (single_rank || peer || host_cb || coll) && wpb == 16 && fast_sel &&
                        a.ablate == 0 && !prof && !ctx->taps;
    auto signature = [](const SrlAssocArgs &src) {
        SrlAssocArgs sg = src;
        std::memset(sg.Rn, 0, sizeof sg.Rn); std::memset(sg.R, 0, sizeof sg.R); std::memset(sg.t, 0, sizeof sg.t);
        std::memset(sg.t_last, 0, sizeof sg.t_last);       // (optimize.cpp:25: per sweep -- it travels through the pose box with the pose)
        sg.pose_box = nullptr; sg.pose_relay = nullptr; sg.pose_relayed = 0; sg.pose_epoch = 0; sg.arm_linger_ticks = 0;
        // the sweep (either buffer of the context) and its keypoint count travel with the pose: compared separately below
        sg.raw_x = sg.raw_y = sg.raw_z = nullptr; sg.alt_x = sg.alt_y = sg.alt_z = nullptr; sg.n = 0; sg.aos = nullptr; sg.alt_aos = nullptr;
        sg.bound_use = 0;        // (an armed launch carries its own: the keypoints of the pass it was armed behind)
        return sg;
    };
    // The current sweep was swapped in when only its PREFIX had landed (srl_sweep_swap): a pass inside the prefix may fire a waiting launch;
    // anything else is launched normally BEHIND the full upload (a resident kernel cannot be made to wait for an event)
    bool needs_tail = false;
    if (ctx->tail_pending) {
        if (hipEventQuery(ctx->up_ev[ctx->cur_slot][1]) == hipSuccess) { ctx->tail_pending = false; ctx->cur_prefix_n = ctx->n; }
        else needs_tail = n_eff > ctx->cur_prefix_n;
        (void)hipGetLastError();
    }
    bool fired = false;
    if (ctx->armed) {
        const SrlAssocArgs sg = signature(a);
        const double age_us = (double)(steady_ns() - ctx->armed_at_ns) * 1e-3;
        // which of the launch's two sweep buffers holds this pass's sweep (the one it was armed on, or -- after srl_sweep_swap -- the other)
        // (on the buffer it was armed on the launch reads the SoA planes: they must be valid; on the other one it reads the staging buffer)
        const bool on_raw = a.raw_x == ctx->armed_raw && ctx->sweep_cap == ctx->armed_raw_cap && a.aos == nullptr;
        const bool on_alt = !on_raw && ctx->armed_alt != nullptr && a.raw_x == ctx->armed_alt && ctx->sweep_cap == ctx->armed_alt_cap &&
                            a.aos != nullptr && a.aos == ctx->armed_alt_aos;
        if (arm_ok && !needs_tail && age_us < ctx->arm_host_linger_us && nb == ctx->armed_nb && kpw == ctx->armed_kpw && nblocks <= ctx->armed_nblocks && (on_raw || on_alt) &&
            std::memcmp(&sg, &ctx->armed_sig, sizeof sg) == 0) {
            pose_box_write(ctx, a.Rn, a.R, a.t, (unsigned)seq_now, SRL_ARM_GO | (on_alt ? SRL_ARM_ALT : 0u), (unsigned)a.n, a.t_last);
            ctx->armed = false;
            ctx->armed_ring = -1;
            ctx->arm_stats[1]++;
            fired = true;
        } else {
            // not this pass's launch: cancel it.  It carried THIS pass's sequence number, and if it has already given up on its own it
            // has left its "expired" report under that number in the mailbox -- so the pass takes the next one (a launch that
            // reuses the number would read that report as its own result)
            const int rcd = srl_ctx_disarm(ctx);
            if (rcd) return rcd;
            seq_now = ++ctx->seq;
            if (fused) a.seq = seq_now;
        }
    }
    const auto t_prep = std::chrono::steady_clock::now();
    // Light profiling with a period P > 1 (srl_set_profiling_period): of every P association launches the second is TIMED and the first only
    // leaves its end event as the second's start; the others carry no event at all (an event record costs ~2.5 us of the loop).
    auto prof_role = [&](unsigned long long count) -> int {        // 2 = timed, 1 = start marker of the next launch, 0 = no events
        if (ctx->prof_period <= 1) return 2;
        const unsigned long long r = count % (unsigned long long)ctx->prof_period;
        return r == 1 ? 2 : (r == 0 ? 1 : 0);
    };
    if (fired) {
        ctx->cur_measured = ctx->armed_measured;
        ctx->cur_gen = ctx->armed_gen;
    } else {
        if (ctx->tail_pending) {             // a normal launch reads whatever it likes of the sweep: behind the whole upload
            HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->up_ev[ctx->cur_slot][1], 0));
            ctx->tail_pending = false; ctx->cur_prefix_n = ctx->n;
        }
        ctx->cur_gen = ctx->timing_gen;
        const int role = prof_light ? prof_role(ctx->prof_count) : 0;
        if (prof) HIPCHK(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
        if (role == 2) HIPCHK(ctx, hipEventRecord(ring_ev[0], ctx->stream));
        HIPCHK(ctx, srl_launch_assoc(a, nb, kpw, wpb, ctx->stream));
        if (prof) HIPCHK(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
        if (role != 0) {
            const unsigned slot = ctx->ring_head % srl_ctx::PROF_RING;
            HIPCHK(ctx, hipEventRecord(ring_ev[1], ctx->stream));
            ctx->ring_prev[slot] = -1; ctx->ring_marker[slot] = role == 1; ctx->ring_void[slot] = false; ctx->ring_gen[slot] = ctx->timing_gen;
            ctx->ring_last_count = (long long)ctx->prof_count;
            ctx->ring_head++;
        }
        if (prof_light) ctx->prof_count++;
        ctx->cur_measured = !prof_light || role == 2;
    }
    const auto t_launched = std::chrono::steady_clock::now();
    if (ctx->h_arm_stamps) {
        long long *hs = ctx->arm_host_stamps[seq_now & 63ull];
        hs[0] = std::chrono::duration_cast<std::chrono::nanoseconds>(t_entry.time_since_epoch()).count();
        hs[1] = std::chrono::duration_cast<std::chrono::nanoseconds>(t_launched.time_since_epoch()).count();
        hs[3] = fired ? 1 : 0;
    }
    // Arm only where the launch is likely to be fired (an armed launch nobody fires holds one workgroup per CU until it is cancelled or
    // leaves by itself): not behind the pass that is expected to be the last of this solve -- the previous solve's pass count,
    // srl_solve_end -- unless a prefetched sweep is waiting, in which case the launch becomes the first pass of THAT sweep
    // (srl_sweep_swap); and not while another context of this process lives on the same device.  arm_mode 2: always.
    const bool arm_wanted = ctx->arm_mode == 2 ||
                            (live_contexts(ctx->device) <= 1 && !ctx->peer_shares_device && !ctx->comm_shares_device &&
                             (ctx->next_n >= 0 || ctx->expected_passes == 0 || ctx->passes_in_solve + 1 < ctx->expected_passes));
    auto arm_next = [&]() -> int {
        // ... and arm the next pass now, while this one runs: same arguments, the sequence number this context hands out next
        int rcp = ensure_pose_box(ctx);
        if (rcp) return rcp;
        SrlAssocArgs nx = a;
        nx.seq = ctx->seq + 1;
        nx.bound_use = a.bound_in ? a.n : 0;              // this pass writes the bounds of its a.n keypoints; the kernel drops them if fired for another sweep
        // the context's other sweep buffer, if it exists: the launch can then be fired for the sweep srl_sweep_swap makes current
        // (sharded ranks too: every rank prefetches and swaps its own point range, the launch learns its count with the pose)
        const bool has_alt = ctx->d_raw_next != nullptr && ctx->next_cap > 0 && ctx->d_stage_next != nullptr;
        nx.aos = nullptr;                                  // (its own buffer's planes are valid once this pass has run)
        nx.alt_aos = has_alt ? ctx->d_stage_next : nullptr;
        nx.alt_x = has_alt ? ctx->d_raw_next : nullptr;
        nx.alt_y = has_alt ? ctx->d_raw_next + ctx->next_cap : nullptr;
        nx.alt_z = has_alt ? ctx->d_raw_next + 2 * (size_t)ctx->next_cap : nullptr;
        if (peer) {                                        // the exchange this launch will take part in: the one after this pass's
            unsigned long long s2 = ctx->peer_seq + 1;
            unsigned e2 = (unsigned)(s2 & 0xFFFFFFFFull);
            if (e2 == 0) { ++s2; e2 = 1; }
            nx.peer_epoch = e2;
            nx.peer_slot = (int)(s2 & 1ull);
        }
        nx.pose_box = ctx->h_pose_box;                    // (host-mapped pinned memory and CPU-visible device memory: one address for both sides)
        nx.pose_relay = ctx->d_pose_relay;
        nx.pose_relayed = ctx->pose_box_kind == 1 ? 0 : 1;
        nx.pose_epoch = (unsigned)nx.seq;
        nx.arm_linger_ticks = ctx->arm_linger_ticks;
        hipEvent_t *nev = nullptr;
        bool own_start = true;
        const int arole = prof_light ? prof_role(ctx->prof_count) : 0;
        if (arole != 0) {
            if (ctx->ring_head - ctx->ring_tail >= (unsigned)srl_ctx::PROF_RING - 2) { int rc = drain_ring(ctx, true); if (rc) return rc; }
            nev = ctx->ring[ctx->ring_head % srl_ctx::PROF_RING];
            // the launch before this one is the ring entry before (if it left one): its end event is the armed launch's start (an event
            // record between the two kernels would sit on the path the armed launch is there to shorten: measured +5 us per iteration)
            own_start = ctx->ring_head == 0 || ctx->ring_void[(ctx->ring_head - 1) % srl_ctx::PROF_RING] ||
                        ctx->ring_last_count != (long long)ctx->prof_count - 1;
            if (own_start && arole == 2) HIPCHK(ctx, hipEventRecord(nev[0], ctx->stream));
        }
        HIPCHK(ctx, srl_launch_assoc(nx, nb, kpw, wpb, ctx->stream));
        ctx->armed_ring = -1;
        ctx->armed_measured = !prof_light;
        ctx->armed_gen = ctx->timing_gen;
        if (arole != 0) {
            HIPCHK(ctx, hipEventRecord(nev[1], ctx->stream));
            ctx->armed_ring = (int)(ctx->ring_head % srl_ctx::PROF_RING);
            ctx->ring_void[ctx->armed_ring] = false;
            ctx->ring_marker[ctx->armed_ring] = arole == 1;
            ctx->ring_gen[ctx->armed_ring] = ctx->timing_gen;
            ctx->ring_prev[ctx->armed_ring] = own_start ? -1 : (int)((ctx->ring_head - 1) % srl_ctx::PROF_RING);
            ctx->ring_last_count = (long long)ctx->prof_count;
            ctx->ring_head++;
            ctx->armed_measured = arole == 2;
        }
        if (prof_light) ctx->prof_count++;
        ctx->armed_sig = signature(nx);
        ctx->armed_nb = nb; ctx->armed_kpw = kpw; ctx->armed_nblocks = nblocks;
        ctx->armed_raw = nx.raw_x; ctx->armed_raw_cap = ctx->sweep_cap;
        ctx->armed_alt = nx.alt_x; ctx->armed_alt_cap = has_alt ? ctx->next_cap : 0; ctx->armed_alt_aos = nx.alt_aos;
        ctx->armed_at_ns = steady_ns();
        ctx->armed = true;
        ctx->arm_stats[0]++;
        return SRL_OK;
    };
    if (arm_ok && arm_wanted && !coll) { const int rca = arm_next(); if (rca) return rca; }

    // residual budget of this rank (sequential early exit, optimize.cpp:107, across ordered shards)
    int64_t budget = o->max_num_residuals;
    int mode = 0;
    const bool multi = ctx->nranks > 1 || coll;
    // the ordered cut can only trigger when max_num_residuals <= number of keypoints: otherwise no exchange of counts.
    // max_num_residuals <= 0: the loop stops at the first keypoint with a plane, wherever (in whichever shard) that is.
    const bool cut_possible = o->max_num_residuals <= 0 || (long long)o->max_num_residuals <= (long long)ctx->total_n;
    const long long *gather_dev = nullptr;
    if (!multi || !cut_possible) {
        srl_shard_budget(o->max_num_residuals, nullptr, ctx->nranks, ctx->rank, &budget, &mode);
    } else {
        // per-rank counts (accepted residuals; keypoints with a plane when max_num_residuals <= 0)
        HIPCHK(ctx, srl_launch_count(ctx->d_binfo, nblocks, o->max_num_residuals <= 0 ? 1 : 0, ctx->d_count, ctx->stream));
        if (peer) {
            // the counts travel like the rows: every rank stores its word into every inbox, one wave collects them
            next_exchange();
            HIPCHK(ctx, srl_launch_peer_counts(ctx->d_peer, peer_epoch, peer_slot, ctx->d_count, ctx->d_gather, ctx->stream));
            gather_dev = ctx->d_gather;
        } else if (coll) {
            // gathered on the stream; the reduce kernel derives its budget and mode from the counts of earlier ranks
            // itself -- no D2H copy, no host synchronisation in the loop
            NCCLCHK(ctx, AllGather(ctx->d_count, ctx->d_gather, 1, ncclInt64, ctx->comm, ctx->stream));
            gather_dev = ctx->d_gather;
        } else if (ctx->dbg_gather) {
            // srl_debug_set_gather_counts: the counts of the other ranks are already in d_gather -- the reduce kernel derives
            // budget and mode from them exactly as behind a real all-gather (test hook for rank > 0 on a single GPU)
            gather_dev = ctx->d_gather;
        } else {
            HIPCHK(ctx, hipMemcpyAsync(ctx->h_count, ctx->d_count, sizeof(long long), hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
            if (!ctx->cb_ag) { ctx->err = "nranks > 1 without communicator"; return SRL_ERR_COMM; }
            int64_t mine = *ctx->h_count;
            std::vector<int64_t> all64((size_t)ctx->nranks, 0);
            if (ctx->cb_ag(&mine, all64.data(), ctx->cb_user) != 0) { ctx->err = "allgather callback failed"; return SRL_ERR_COMM; }
            srl_shard_budget(o->max_num_residuals, all64.data(), ctx->nranks, ctx->rank, &budget, &mode);
        }
    }

    SrlReduceArgs ra;
    ra.rec = ctx->d_rec;
    ra.status = ctx->d_status;
    ra.partials = ctx->d_partials;
    ra.binfo = ctx->d_binfo;
    ra.n = n_eff;
    ra.kpb = kpb;
    ra.nblocks = nblocks;
    ra.max_res = budget;
    ra.gather = gather_dev;
    ra.rank = ctx->rank;
    ra.max_num_residuals = o->max_num_residuals;
    ra.out = ctx->d_out;
    // single rank: the reduce kernel publishes straight into host-mapped memory and the host spins on the
    // sequence word -- no D2H copy, no stream synchronisation on the per-iteration critical path
    // (a fused pass of a sharded sweep without RCCL / peers also ends in the host mailbox: the callback all-reduce follows it)
    const bool mailbox = ((ctx->nranks == 1) && !coll) || (fused && !coll);
    ra.mailbox = (mailbox && !fused) ? ctx->h_mail : nullptr;
    ra.seq = seq_now;
    if (!fused) {
        HIPCHK(ctx, srl_launch_reduce(ra, mode, ctx->stream));
        if (peer) {
            next_exchange();
            HIPCHK(ctx, srl_launch_peer_rows(ctx->d_peer, peer_epoch, peer_slot, ctx->d_out, ctx->h_mail, seq_now, gather_dev, ctx->stream));
        }
    }
    if (prof) HIPCHK(ctx, hipEventRecord(ctx->ev[2], ctx->stream));
    const auto t_enq = std::chrono::steady_clock::now();
    // everything is enqueued: the caller's H-independent host work runs now, beside the kernels (srl_build_residuals_overlap)
    if (ctx->overlap_fn) {
        void (*fn)(void *) = ctx->overlap_fn;
        ctx->overlap_fn = nullptr;                 // once per call, in the first pass
        fn(ctx->overlap_user);
    }

    // the one exchange step: sum of the normal equations over the point-range shards
    const int n_red = SRL_REDUCED_DOUBLES;   // HtH, Hth, loss, 6 counters carried as doubles (incl. visited keypoints), fused time-out flag
    long long visited_local = 0;
    if (coll) {
        // one ncclAllReduce of 50 doubles on the context's stream; last_visited sits behind the reduced range
        SrlDevOut *mine = fused ? &ctx->d_mail->out : ctx->d_out;
        NCCLCHK(ctx, AllReduce(mine, mine, n_red, ncclDouble, ncclSum, ctx->comm, ctx->stream));
        HIPCHK(ctx, srl_launch_publish(mine, ctx->h_mail, ra.seq, ctx->stream));
        if (arm_ok && arm_wanted) { const int rca = arm_next(); if (rca) return rca; }       // (behind the collective: see arm_ok)
    }
    const bool host_reduce = ctx->nranks > 1 && !coll && !peer;      // the caller's all-reduce callback (CPU / gloo tests, foreign transports)
    if (coll || mailbox || peer) {
        constexpr int NW = (int)(sizeof(SrlDevOut) / 8);
        const unsigned tag32 = (unsigned)ra.seq;
        unsigned long long *gran = ctx->h_mail->g;
        // arrived: the plain form's sequence word; the tagged form: every granule carries this pass's tag (the last one is looked at
        // first: while it is stale nothing else is read)
        auto arrived = [&]() -> bool {
            if (!tagged_mail) return __atomic_load_n(&ctx->h_mail->seq, __ATOMIC_ACQUIRE) == ra.seq;
            if ((unsigned)(__atomic_load_n(&gran[2 * NW - 1], __ATOMIC_RELAXED) >> 32) != tag32) return false;
            for (int i = 0; i < 2 * NW - 1; i++) if ((unsigned)(__atomic_load_n(&gran[i], __ATOMIC_RELAXED) >> 32) != tag32) return false;
            return true;
        };
        unsigned long long spins = 0;
        bool arm_expired = false;
        while (!arrived()) {
            if (fired && __atomic_load_n(&ctx->h_mail->expired, __ATOMIC_ACQUIRE) == ra.seq) { arm_expired = true; break; }   // the launch this pass fired had left
            if ((++spins & 0xFFFFF) == 0) {           // every ~1M polls: make sure the stream has not faulted
                const hipError_t qe = hipStreamQuery(ctx->stream);
                if (qe != hipSuccess && qe != hipErrorNotReady) { ctx->err = std::string("reduce kernel: ") + hipGetErrorString(qe); return SRL_ERR_HIP; }
                if (qe == hipSuccess && !arrived()) {
                    char dbg[256];
                    int bad = -1; unsigned bad_tag = 0;
                    if (tagged_mail) for (int i = 0; i < 2 * NW; i++) if ((unsigned)(gran[i] >> 32) != tag32) { bad = i; bad_tag = (unsigned)(gran[i] >> 32); break; }
                    std::snprintf(dbg, sizeof dbg, "reduce kernel finished without publishing (seq %llu, tagged %d, fired %d, fused %d, blocks %d, first stale granule %d tag %u, plain seq %llu)",
                                  (unsigned long long)ra.seq, (int)tagged_mail, (int)fired, (int)fused, nblocks, bad, bad_tag, (unsigned long long)ctx->h_mail->seq);
                    ctx->err = dbg;
                    return SRL_ERR_HIP;
                }
            }
        }
        if (arm_expired) {
            ctx->arm_stats[3]++;
            ctx->err = "armed launch expired";
            return SRL_INTERNAL_ARM_EXPIRED;
        }
        if (tagged_mail) {
            unsigned long long *w = reinterpret_cast<unsigned long long *>(ctx->h_out);
            for (int i = 0; i < NW; i++)
                w[i] = (__atomic_load_n(&gran[2 * i], __ATOMIC_RELAXED) & 0xFFFFFFFFull) | (__atomic_load_n(&gran[2 * i + 1], __ATOMIC_RELAXED) << 32);
        } else {
            std::memcpy(ctx->h_out, &ctx->h_mail->out, sizeof(SrlDevOut));
        }
        visited_local = ctx->h_out->last_visited + 1;
        if (host_reduce) {
            if (!ctx->cb_ar) { ctx->err = "nranks > 1 without communicator"; return SRL_ERR_COMM; }
            if (ctx->cb_ar(reinterpret_cast<double *>(ctx->h_out), n_red, ctx->cb_user) != 0) { ctx->err = "allreduce callback failed"; return SRL_ERR_COMM; }
        }
        if (ctx->h_out->pad == SRL_PEER_TIMEOUT_MARK) {
            // A rank's row had not arrived when the kernel's bounded spin ran out (~0.3-1 s: a kernel must not hold the GPU for ever).  Whether
            // that rank is late or gone is not decided in the kernel: srl_build_residuals repeats the pass with the SAME exchange tags until
            // the wall-clock deadline of srl_peer_set_deadline_ms, then gives the session up for every rank (peer_pass_timed_out below).
            ctx->err = "direct peer exchange: a rank's row has not arrived yet";
            return SRL_INTERNAL_PEER_TIMEOUT;
        }
        if (ctx->h_out->pad != 0 || ctx->h_out->d_timeout > 0.5) {     // (summed over the ranks: all of them repeat the pass together)
            // the finishing workgroup gave up waiting for a row (bounded spin: another process holding compute units back, a
            // preempted queue): not an error of the data -- the caller repeats the pass with the reduction in its own kernel
            ctx->err = "fused final reduction timed out waiting for a workgroup's row";
            if (fired && coll) ctx->arm_stats[3]++;      // (RCCL form: this is how a fired launch that had given up shows -- its record carries the flag)
            return SRL_INTERNAL_FUSED_TIMEOUT;
        }
    } else {
        HIPCHK(ctx, hipMemcpyAsync(ctx->h_out, ctx->d_out, sizeof(SrlDevOut), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        visited_local = ctx->h_out->last_visited + 1;
        if (ctx->nranks > 1) {
            if (!ctx->cb_ar) { ctx->err = "nranks > 1 without communicator"; return SRL_ERR_COMM; }
            if (ctx->cb_ar(reinterpret_cast<double *>(ctx->h_out), n_red, ctx->cb_user) != 0) { ctx->err = "allreduce callback failed"; return SRL_ERR_COMM; }
            // another rank's fused pass timed out (its flag is part of the sum): this rank repeats the pass with it
            if (ctx->h_out->d_timeout > 0.5) { ctx->err = "fused final reduction timed out on another rank"; return SRL_INTERNAL_FUSED_TIMEOUT; }
        }
    }
    if (prof) HIPCHK(ctx, hipEventRecord(ctx->ev[3], ctx->stream));
    const auto t_res = std::chrono::steady_clock::now();
    if (ctx->armed) ctx->armed_at_ns = steady_ns();      // the armed launch starts waiting about now (when this pass ends): its age counts from here
    if (ctx->h_arm_stamps) ctx->arm_host_stamps[seq_now & 63ull][2] = std::chrono::duration_cast<std::chrono::nanoseconds>(t_res.time_since_epoch()).count();

    // total visited keypoints over all shards (part of the reduced range) -> global index of the last visited one
    const long long visited_total = (long long)(ctx->h_out->d_visited + 0.5);
    if (fused && !can_fuse_cut) {
        // a fused pass without a cut visits every keypoint it was given, on every rank: anything else means workgroups ran on another
        // keypoint count than the host fired them with (a relayed pose box once lost the count of a swapped-in sweep)
        const long long expect = ctx->nranks > 1 ? (long long)ctx->total_n : (long long)n_eff;
        if (visited_total != expect) {
            char buf[256];
            std::snprintf(buf, sizeof buf, "fused pass visited %lld of %lld keypoints (fired %d, seq %llu; residuals %.0f, time-out flag %.0f, marker %lld)", visited_total, expect,
                          (int)fired, (unsigned long long)seq_now, ctx->h_out->d_num_res, ctx->h_out->d_timeout, (long long)ctx->h_out->pad);
            ctx->err = buf;
            return SRL_ERR_HIP;
        }
    }

    const SrlDevOut &r = *ctx->h_out;
    std::memcpy(out->HtH, r.HtH, sizeof out->HtH);
    std::memcpy(out->Hth, r.Hth, sizeof out->Hth);
    out->loss_sum = r.loss;
    out->num_residuals = (int32_t)(r.d_num_res + 0.5);
    out->success = out->num_residuals >= o->min_number_neighbors ? 1 : 0;     // optimize.cpp:110
    out->sum_candidates = (int64_t)(r.d_sum_pk + 0.5);
    out->last_visited = visited_total - 1;
    out->nan_error = r.d_nan > 0.5 ? 1 : 0;
    out->num_fallback = (int32_t)(r.d_fallback + 0.5);

    if (a.aos != nullptr && n_eff > ctx->soa_valid_n) ctx->soa_valid_n = n_eff;      // this pass filed the SoA planes of its keypoints
    ctx->bound_n = n_eff;                                                            // ... and the bounds of its keypoints (entries behind them: an earlier, shorter or longer pass -- never read)
    ctx->last_K = K;
    ctx->last_nb = nb;
    ctx->last_visited_local = visited_local - 1;
    ctx->taps_valid = ctx->taps;

    if (prof) {
        HIPCHK(ctx, hipEventSynchronize(ctx->ev[3]));
        hipEventElapsedTime(&ctx->timing.assoc_ms, ctx->ev[0], ctx->ev[1]);
        hipEventElapsedTime(&ctx->timing.reduce_ms, ctx->ev[1], ctx->ev[2]);
        hipEventElapsedTime(&ctx->timing.total_ms, ctx->ev[0], ctx->ev[3]);
        ctx->timing.calls += 1;
        ctx->timing.sum_assoc_ms += ctx->timing.assoc_ms;
        ctx->timing.sum_reduce_ms += ctx->timing.reduce_ms;
        ctx->timing.sum_total_ms += ctx->timing.total_ms;
        ctx->timing.sum_keypoints += n_eff;
        const auto t_end = std::chrono::steady_clock::now();
        ctx->timing.sum_host_launch_us += std::chrono::duration<double, std::micro>(t_enq - t_entry).count();
        ctx->timing.sum_host_wait_us += std::chrono::duration<double, std::micro>(t_res - t_enq).count();
        ctx->timing.sum_host_total_us += std::chrono::duration<double, std::micro>(t_end - t_entry).count();
    }
    {
        // algorithmic bytes of this rank's association pass: 24 + 12*(2r+1)^3 + 12*P_k per keypoint.
        // sum_pk in h_out is the all-reduced value; the per-rank value is recomputed from the ratio when sharded.
        const long long side = 2 * nb + 1;
        const long long per_kp = 24 + 12 * side * side * side;
        const double pk_share = (ctx->nranks > 1 && ctx->total_n > 0) ? r.d_sum_pk * ((double)ctx->n / (double)ctx->total_n) : r.d_sum_pk;
        ctx->timing.algorithmic_bytes = per_kp * (long long)n_eff + (long long)(12.0 * pk_share);
        // (light profiling with a period: only the passes whose launch is one of the timed ones count, so bytes and durations pair up)
        if (prof || (prof_light && ctx->cur_measured && ctx->cur_gen == ctx->timing_gen)) {
            ctx->timing.sum_algorithmic_bytes += ctx->timing.algorithmic_bytes; ctx->timing.sum_keypoints += prof_light ? n_eff : 0; ctx->timing.sum_passes += 1;
        }
    }
    if (ctx->profiling == 3) {
        // host stamps only (no events): argument preparation, the launch call itself, the wait for the result
        ctx->timing.calls += 1;
        ctx->timing.sum_host_launch_us += std::chrono::duration<double, std::micro>(t_launched - t_prep).count();
        ctx->timing.sum_host_wait_us += std::chrono::duration<double, std::micro>(t_res - t_enq).count();
        ctx->timing.sum_host_total_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_entry).count();
        ctx->timing.sum_assoc_ms += std::chrono::duration<double, std::milli>(t_prep - t_entry).count();      // argument preparation (ms)
        ctx->timing.sum_reduce_ms += std::chrono::duration<double, std::milli>(t_enq - t_launched).count();  // launch returned -> everything enqueued + overlap callback (ms)
    }
    if (out->nan_error) { ctx->err = "NaN planarity"; return SRL_ERR_NAN_PLANARITY; }
    return SRL_OK;
}

int srl_fetch_neighbors(srl_ctx *ctx, int32_t *ids, uint8_t *status, int32_t *num_candidates) {
    if (!ctx) return SRL_ERR_BAD_ARG;
    SRL_DISARM(ctx);
    if (!ctx->taps_valid) { ctx->err = "taps were not enabled for the last srl_build_residuals"; return SRL_ERR_BAD_ARG; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int n = ctx->n, K = ctx->last_K;
    if (n == 0) return SRL_OK;
    if (ids) HIPCHK(ctx, hipMemcpyAsync(ids, ctx->d_tap_ids, (size_t)n * K * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    if (status) HIPCHK(ctx, hipMemcpyAsync(status, ctx->d_status, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    if (num_candidates) HIPCHK(ctx, hipMemcpyAsync(num_candidates, ctx->d_tap_ncand, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (status) for (long long k = ctx->last_visited_local + 1; k < n; k++) status[k] = 3;   // not visited (optimize.cpp:107)
    return SRL_OK;
}

int srl_fetch_residuals(srl_ctx *ctx, double *normal, double *a2D, double *weight, double *norm_offset,
                        double *distance, double *jacobian) {
    if (!ctx) return SRL_ERR_BAD_ARG;
    SRL_DISARM(ctx);
    if (!ctx->taps_valid) { ctx->err = "taps were not enabled for the last srl_build_residuals"; return SRL_ERR_BAD_ARG; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int n = ctx->n;
    if (n == 0) return SRL_OK;
    std::vector<double> rec((size_t)n * 8);
    HIPCHK(ctx, hipMemcpyAsync(rec.data(), ctx->d_rec, rec.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (normal) HIPCHK(ctx, hipMemcpyAsync(normal, ctx->d_tap_normal, (size_t)n * 3 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (a2D) HIPCHK(ctx, hipMemcpyAsync(a2D, ctx->d_tap_a2d, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (norm_offset) HIPCHK(ctx, hipMemcpyAsync(norm_offset, ctx->d_tap_offset, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < n; k++) {
        if (jacobian) for (int c = 0; c < 6; c++) jacobian[(size_t)k * 6 + c] = rec[(size_t)k * 8 + c];
        if (distance) distance[k] = rec[(size_t)k * 8 + 6];
        if (weight) weight[k] = rec[(size_t)k * 8 + 7];
    }
    return SRL_OK;
}

int srl_search_neighbors(srl_ctx *ctx, const double *world_xyz, int n, int nb_voxels_visited, double size_voxel_map,
                         int max_num_neighbors, int threshold_voxel_capacity, int32_t *ids, float *nb_xyz,
                         int32_t *num_found) {
    if (!ctx || n < 0 || (n > 0 && (!world_xyz || !ids || !num_found))) return SRL_ERR_BAD_ARG;
    SRL_DISARM(ctx);
    if (!ctx->d_table) return SRL_ERR_NO_MAP;
    if (nb_voxels_visited < 1 || nb_voxels_visited > 2 || max_num_neighbors < 1 || max_num_neighbors > SRL_MAX_NEIGHBORS)
        return SRL_ERR_UNSUPPORTED;
    if (n == 0) return SRL_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int K = max_num_neighbors;
    // scratch from the context's pool (RAII: returned on every exit path, no hipMalloc / hipFree per call)
    DevBuf b_q, b_ids, b_nf, b_xyz;
    HIPCHK(ctx, b_q.alloc(ctx, (size_t)n * 3 * sizeof(double)));
    HIPCHK(ctx, b_ids.alloc(ctx, (size_t)n * K * sizeof(int)));
    HIPCHK(ctx, b_nf.alloc(ctx, (size_t)n * sizeof(int)));
    if (nb_xyz) HIPCHK(ctx, b_xyz.alloc(ctx, (size_t)n * K * 3 * sizeof(float)));
    double *d_q = b_q.as<double>();
    int *d_ids = b_ids.as<int>(), *d_nf = b_nf.as<int>();
    float *d_xyz = nb_xyz ? b_xyz.as<float>() : nullptr;
    HIPCHK(ctx, hipMemcpyAsync(d_q, world_xyz, (size_t)n * 3 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(d_ids, 0xFF, (size_t)n * K * sizeof(int), ctx->stream));
    if (d_xyz) HIPCHK(ctx, hipMemsetAsync(d_xyz, 0, (size_t)n * K * 3 * sizeof(float), ctx->stream));
    SrlSearchArgs a;
    a.q = d_q; a.n = n; a.table = ctx->d_table; a.table_mask = ctx->table_cap - 1; a.slabs = ctx->d_slabs;
    a.size_voxel = size_voxel_map; a.K = K; a.thr_cap = threshold_voxel_capacity; a.select_mode = ctx->search_select_mode;
    a.ids = d_ids; a.nb_xyz = d_xyz; a.num_found = d_nf;
    HIPCHK(ctx, srl_launch_search(a, nb_voxels_visited, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ids, d_ids, (size_t)n * K * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(num_found, d_nf, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    if (nb_xyz) HIPCHK(ctx, hipMemcpyAsync(nb_xyz, d_xyz, (size_t)n * K * 3 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return SRL_OK;
}

int srl_transform_points(srl_ctx *ctx, const double *raw_xyz, int n, const double q[4], const double t[3],
                         const double R_il[9], const double t_il[3], double *out_xyz) {
    if (!ctx || n < 0 || (n > 0 && (!raw_xyz || !out_xyz)) || !q || !t || !R_il || !t_il) return SRL_ERR_BAD_ARG;
    SRL_DISARM(ctx);
    if (n == 0) return SRL_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    DevBuf b_in, b_o;
    HIPCHK(ctx, b_in.alloc(ctx, (size_t)n * 3 * sizeof(double)));
    HIPCHK(ctx, b_o.alloc(ctx, (size_t)n * 3 * sizeof(double)));
    double *d_in = b_in.as<double>(), *d_o = b_o.as<double>();
    HIPCHK(ctx, hipMemcpyAsync(d_in, raw_xyz, (size_t)n * 3 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    SrlXform X;
    const srl::Mat3 R = srl::Quat(q[0], q[1], q[2], q[3]).toRotationMatrix();   // utility.cpp:317: q_end.toRotationMatrix()
    std::memcpy(X.R, R.a, sizeof X.R);
    std::memcpy(X.t, t, sizeof X.t);
    std::memcpy(X.R_il, R_il, sizeof X.R_il);
    std::memcpy(X.t_il, t_il, sizeof X.t_il);
    HIPCHK(ctx, srl_launch_transform(d_in, n, X, d_o, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(out_xyz, d_o, (size_t)n * 3 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return SRL_OK;
}

// debug / parity hooks (never used by the product path)
int srl_debug_set_ablate(srl_ctx *ctx, int bits) {
    if (!ctx) return SRL_ERR_BAD_ARG;
    SRL_DISARM(ctx);
    ctx->ablate = bits;
    return SRL_OK;
}
int srl_debug_set_launch_shape(srl_ctx *ctx, int keypoints_per_wave, int waves_per_workgroup) {
    if (!ctx) return SRL_ERR_BAD_ARG;
    SRL_DISARM(ctx);
    if (keypoints_per_wave != 0 && keypoints_per_wave != 2 && keypoints_per_wave != 3 && keypoints_per_wave != 6 && keypoints_per_wave != 12 && keypoints_per_wave != 4 && keypoints_per_wave != 8 && keypoints_per_wave != 16) return SRL_ERR_BAD_ARG;
    if (waves_per_workgroup != 0 && waves_per_workgroup != 4 && waves_per_workgroup != 16) return SRL_ERR_BAD_ARG;
    ctx->force_kpw = keypoints_per_wave;
    ctx->force_wpb = waves_per_workgroup;
    return SRL_OK;
}
int srl_debug_set_bound_culling(srl_ctx *ctx, int enable) {
    if (!ctx) return SRL_ERR_BAD_ARG;
    SRL_DISARM(ctx);
    ctx->bound_mode = enable ? 1 : 0;
    ctx->bound_n = 0;
    return SRL_OK;
}
int srl_debug_set_fused_reduce(srl_ctx *ctx, int enable) {
    if (!ctx) return SRL_ERR_BAD_ARG;
    SRL_DISARM(ctx);
    ctx->fuse_reduce = enable != 0;
    return SRL_OK;
}
int srl_debug_set_select_mode(srl_ctx *ctx, int select_mode) {
    if (!ctx || select_mode < 0 || select_mode > 5) return SRL_ERR_BAD_ARG;
    SRL_DISARM(ctx);
    ctx->select_mode = select_mode;
    return SRL_OK;
}
int srl_debug_set_frame_order_mode(srl_ctx *ctx, int mode) {
    if (!ctx || mode < 0 || mode > 1) return SRL_ERR_BAD_ARG;
    ctx->frame_order_mode = mode;
    return SRL_OK;
}
int srl_debug_frame_order_used(srl_ctx *ctx, int *used) {
    if (!ctx || !used) return SRL_ERR_BAD_ARG;
    *used = ctx->frame_order_used;
    return SRL_OK;
}
int srl_debug_set_search_select_mode(srl_ctx *ctx, int select_mode) {
    if (!ctx || select_mode < 0 || select_mode > 5) return SRL_ERR_BAD_ARG;
    ctx->search_select_mode = select_mode;
    return SRL_OK;
}
int srl_debug_heap_topk(const double *distances, int n, int K, int32_t *out_index) {
    // the device's heap routines (srl_heap.h) on the host: candidates offered in index order, result = read-out order
    if (n < 0 || K < 1 || K > SRL_MAX_NEIGHBORS || (n > 0 && !distances) || !out_index) return SRL_ERR_BAD_ARG;
    double hd[SRL_MAX_NEIGHBORS];
    int he[SRL_MAX_NEIGHBORS];
    int size = 0;
    for (int i = 0; i < n; i++) size = srl_heap_offer(hd, he, size, K, distances[i], i);
    int out[SRL_MAX_NEIGHBORS];
    srl_heap_drain(hd, he, size, out);
    for (int i = 0; i < size; i++) out_index[i] = out[i];
    return size;
}
int srl_debug_device_sqrt(srl_ctx *ctx, const double *in, int n, double *out) {
    if (!ctx || n < 0 || (n > 0 && (!in || !out))) return SRL_ERR_BAD_ARG;
    SRL_DISARM(ctx);
    if (n == 0) return SRL_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    DevBuf b;
    HIPCHK(ctx, b.alloc(ctx, (size_t)n * sizeof(double)));
    HIPCHK(ctx, hipMemcpyAsync(b.as<double>(), in, (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, srl_launch_sqrt(b.as<double>(), n, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(out, b.as<double>(), (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return SRL_OK;
}

void srl_shard_range(int n, int nranks, int rank, int *begin, int *count) {
    const long long b = (long long)rank * n / nranks;
    const long long e = (long long)(rank + 1) * n / nranks;
    if (begin) *begin = (int)b;
    if (count) *count = (int)(e - b);
}

void srl_shard_budget(int max_num_residuals, const int64_t *accepted_per_rank, int nranks, int rank,
                      int64_t *budget, int *mode) {
    (void)nranks;
    int64_t b = max_num_residuals;
    int m = 0;
    if (max_num_residuals <= 0) {
        // optimize.cpp:107 with the class default -1: the loop is left at the first keypoint that reaches the break test,
        // i.e. the first one with >= min_number_neighbors neighbours (:78-79 `continue`s past the others).  Counts given:
        // keypoints with a plane per rank -- this rank searches only if no earlier rank holds one.
        int64_t prior = 0;
        for (int r = 0; r < rank; r++) prior += accepted_per_rank ? accepted_per_rank[r] : (int64_t)1;
        m = (prior > 0) ? 2 : 1;
    } else {
        int64_t prior = 0;
        for (int r = 0; r < rank; r++) prior += accepted_per_rank ? accepted_per_rank[r] : 0;
        b = (int64_t)max_num_residuals - prior;
        if (b <= 0) m = 2;      // an earlier shard already reached max_num_residuals
    }
    if (budget) *budget = b;
    if (mode) *mode = m;
}

int srl_build_residuals(srl_ctx *ctx, const srl_frame *f, const srl_icp_opts *o, srl_normal_eq *out) {
    if (!ctx || !f || !o || !out) return SRL_ERR_BAD_ARG;
    if (!ctx->d_table) return SRL_ERR_NO_MAP;
    if (!ctx->sweep_loaded) return SRL_ERR_NO_SWEEP;
    if (ctx->total_n == 0) {
        // an empty keypoint set is a valid pass: the loop of optimize.cpp:68 does not run, num_residuals = 0 fails the
        // test at :110 and the caller gets summary.success = false (no exception, nothing visited)
        std::memset(out, 0, sizeof *out);
        out->last_visited = -1;
        ctx->taps_valid = false;
        return SRL_OK;
    }
    // Finite max_num_residuals (600 in the shipped yaml files): the sequential loop of optimize.cpp:68-107 stops at the
    // max-th accepted keypoint and never looks at the rest, so a single rank first runs only a prefix that almost surely
    // contains it (4 x max + 2048 keypoints; ~95 % of visited keypoints are accepted).  If the prefix holds fewer accepted
    // keypoints than max, nothing can be concluded and the pass is repeated over the whole shard.  Same accepted set and
    // same cut as the full pass; the sums agree up to FP64 summation order (a short pass uses fewer keypoints per
    // workgroup, srl_keypoints_per_block).  Taps keep the full pass.
    int n_eff = ctx->n;
    const bool single = ctx->nranks == 1 && !(ctx->comm && ctx->force_coll);
    if (single && !ctx->taps && o->max_num_residuals > 0) {
        const long long pre = ((4LL * o->max_num_residuals + 2048 + 63) / 64) * 64;
        if (pre < (long long)ctx->n) n_eff = (int)pre;
    }
    ctx->prefix_hint = n_eff < ctx->n ? n_eff : 0;      // what the next prefetch sends first (srl_sweep_prefetch)
    // One attempt at a pass = build_residuals_pass, repeated while a PEER's row has not arrived (direct peer exchange only).
    auto attempt = [&](int n_pass) -> int {
        const unsigned long long peer_seq0 = ctx->peer_seq;              // (a repeated pass re-polls the SAME exchanges)
        const long long t_first = steady_ns();
        int r = build_residuals_pass(ctx, f, o, out, n_pass);
        while (r == SRL_INTERNAL_PEER_TIMEOUT) {
            // Direct peer exchange, a row missing.  Skew between ranks (a first launch, a map insertion, a host stall on the other side) is
            // not a failure: this rank's own rows are in every inbox already (stores of one exchange are idempotent), so the pass is simply
            // run again with the same tags until the late rank's row is there -- the late rank finds every row and returns SRL_OK as well.
            // No rank runs ahead meanwhile: the next exchange needs a row of THIS rank.  Past the deadline the session is given up for
            // everybody: the poison word of every inbox is set, and a rank that times out looks at its own word first.
            const int rcd = ctx->armed ? srl_ctx_disarm(ctx) : SRL_OK;
            if (rcd) return rcd;
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
            unsigned long long poisoned = 0;
            HIPCHK(ctx, hipMemcpy(&poisoned, ctx->d_inbox + SRL_PEER_POISON_WORD, sizeof poisoned, hipMemcpyDeviceToHost));
            const bool expired = (steady_ns() - t_first) / 1000000ll >= (long long)ctx->peer_deadline_ms;
            if (poisoned || expired) {
                if (!poisoned) {
                    const unsigned long long mark = 1ull + (unsigned long long)ctx->rank;
                    for (int rk = 0; rk < ctx->nranks; rk++)
                        if (ctx->peer_inbox[rk] && hipMemcpy(ctx->peer_inbox[rk] + SRL_PEER_POISON_WORD, &mark, sizeof mark, hipMemcpyHostToDevice) != hipSuccess)
                            (void)hipGetLastError();                      // (a dead peer's mapping: nothing left to tell it)
                }
                ctx->peer_failed = true;
                ctx->err = poisoned ? "direct peer exchange: another rank has given this session up (its deadline for a missing row expired); "
                                      "srl_peer_detach, srl_peer_export and srl_peer_attach on every rank start a new one"
                                    : "direct peer exchange: a rank's row did not arrive within the deadline (srl_peer_set_deadline_ms); the session "
                                      "is given up on every rank -- srl_peer_detach, srl_peer_export and srl_peer_attach start a new one";
                return SRL_ERR_COMM;
            }
            ctx->peer_retries++;
            ctx->peer_seq = peer_seq0;
            r = build_residuals_pass(ctx, f, o, out, n_pass);
        }
        return r;
    };
    // a pass whose fused reduction timed out is repeated once with the separate reduce kernel (stream-ordered: it cannot time out)
    auto pass = [&](int n_pass) -> int {
        const unsigned long long peer_seq_in = ctx->peer_seq;
        int r = attempt(n_pass);
        if (r == SRL_INTERNAL_ARM_EXPIRED) {                 // nobody was listening: the same pass with a normal launch
            SRL_DISARM(ctx);
            // ... and with the SAME exchange tag: the expired launch left in its prologue, it never stored this rank's row for the exchange the
            // attempt above counted -- the peers are still polling for that one (a relaunch one exchange ahead of its peers
            // would have made both sides spin until the deadline)
            ctx->peer_seq = peer_seq_in;
            r = attempt(n_pass);
            if (r == SRL_INTERNAL_ARM_EXPIRED) r = SRL_ERR_HIP;
        }
        if (r == SRL_INTERNAL_FUSED_TIMEOUT) {
            const bool fuse = ctx->fuse_reduce;
            ctx->fuse_reduce = false;
            const int rcd = ctx->armed ? srl_ctx_disarm(ctx) : SRL_OK;      // (before waiting for the stream: nobody would fire it)
            const hipError_t es = rcd ? hipSuccess : hipStreamSynchronize(ctx->stream);
            if (rcd || es != hipSuccess) {
                ctx->fuse_reduce = fuse;                                     // never leave the context un-fused behind an error
                if (rcd) return rcd;
                ctx->err = std::string("hipStreamSynchronize: ") + hipGetErrorString(es);
                return SRL_ERR_HIP;
            }
            r = attempt(n_pass);
            ctx->fuse_reduce = fuse;
            if (r == SRL_INTERNAL_FUSED_TIMEOUT || r == SRL_INTERNAL_ARM_EXPIRED) r = SRL_ERR_HIP;
        }
        return r;
    };
    int rc = pass(n_eff);
    if ((rc == SRL_OK || rc == SRL_ERR_NAN_PLANARITY) && n_eff < ctx->n && out->num_residuals < o->max_num_residuals)
        rc = pass(ctx->n);
    ctx->passes_in_solve++;
    return rc;
}

int srl_solve_end(srl_ctx *ctx) {
    if (!ctx) return SRL_ERR_BAD_ARG;
    if (ctx->passes_in_solve > 0) ctx->expected_passes = ctx->passes_in_solve;
    ctx->passes_in_solve = 0;
    // a launch armed behind the last pass is only worth keeping when the next sweep is already on its way (it will be that sweep's first pass)
    if (ctx->armed && ctx->next_n < 0 && ctx->arm_mode != 2) return srl_ctx_disarm(ctx);
    return SRL_OK;
}

int srl_build_residuals_overlap(srl_ctx *ctx, const srl_frame *f, const srl_icp_opts *o, srl_normal_eq *out, srl_overlap_fn fn, void *user) {
    if (!ctx) return SRL_ERR_BAD_ARG;
    ctx->overlap_fn = fn;
    ctx->overlap_user = user;
    const int rc = srl_build_residuals(ctx, f, o, out);
    ctx->overlap_fn = nullptr;                     // an early return (no sweep, empty sweep, bad argument) never ran it
    ctx->overlap_user = nullptr;
    return rc;
}

// ------------------------------------------------------------------------------------------ comm
int srl_comm_unique_id(void *id) {
    if (!id) return SRL_ERR_BAD_ARG;
    static_assert(sizeof(ncclUniqueId) <= SRL_COMM_ID_BYTES, "unique id does not fit");
    ncclUniqueId u;
    const SrlRccl *rc = srl_rccl();
    if (!rc || rc->GetUniqueId(&u) != ncclSuccess) return SRL_ERR_COMM;
    std::memset(id, 0, SRL_COMM_ID_BYTES);
    std::memcpy(id, &u, sizeof u);
    return SRL_OK;
}

int srl_comm_set_library(const char *path) {
    return srl_rccl_set_library(path) ? SRL_OK : SRL_ERR_BAD_ARG;
}

int srl_comm_init_rank(srl_ctx *ctx, int nranks, int rank, const void *id) {
    if (!ctx || !id || nranks < 1 || rank < 0 || rank >= nranks) return SRL_ERR_BAD_ARG;
    SRL_DISARM(ctx);
    if (ctx->peer_on) { ctx->err = "srl_comm_init_rank: direct peer exchange is attached (srl_peer_detach first)"; return SRL_ERR_BAD_ARG; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (ctx->comm && srl_rccl()) { srl_rccl()->CommDestroy(ctx->comm); ctx->comm = nullptr; }
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof u);
    NCCLCHK(ctx, CommInitRank(&ctx->comm, nranks, u, rank));
    ctx->nranks = nranks;
    ctx->rank = rank;
    ctx->cb_ar = nullptr; ctx->cb_ag = nullptr; ctx->cb_user = nullptr;
    { const char *fc = std::getenv("SRL_FORCE_COLLECTIVES"); ctx->force_coll = fc && std::atoi(fc) != 0; }
    int rc = ensure_gather(ctx, (size_t)nranks);
    if (rc) return rc;
    // which devices do the ranks sit on?  One all-gather of the device identities (as for the peer exchange: ranks that share a device
    // -- a test arrangement -- do not arm launches)
    ctx->comm_shares_device = false;
    if (nranks > 1) {
        const long long mine = (long long)device_identity(ctx->device);
        HIPCHK(ctx, hipMemcpyAsync(ctx->d_count, &mine, sizeof mine, hipMemcpyHostToDevice, ctx->stream));
        NCCLCHK(ctx, AllGather(ctx->d_count, ctx->d_gather, 1, ncclInt64, ctx->comm, ctx->stream));
        std::vector<long long> all((size_t)nranks, 0);
        HIPCHK(ctx, hipMemcpyAsync(all.data(), ctx->d_gather, (size_t)nranks * sizeof(long long), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        for (int r = 0; r < nranks; r++) if (r != rank && all[(size_t)r] == mine) ctx->comm_shares_device = true;
    }
    return SRL_OK;
}

int srl_comm_backend_info(char *origin, int origin_len, int *version, int *preloaded) {
    const SrlRccl *rc = srl_rccl();
    if (!rc) { if (origin && origin_len > 0) std::snprintf(origin, (size_t)origin_len, "%s", srl_rccl_error()); return SRL_ERR_COMM; }
    if (origin && origin_len > 0) std::snprintf(origin, (size_t)origin_len, "%s", rc->origin);
    if (version) *version = rc->version;
    if (preloaded) *preloaded = rc->preloaded ? 1 : 0;
    return SRL_OK;
}

int srl_comm_info(srl_ctx *ctx, int *transport, int *nranks, int *rank, int *ranks_seen, int64_t *passes_armed) {
    if (!ctx) return SRL_ERR_BAD_ARG;
    int tr = 0, seen = 1;
    if (ctx->peer_on) {
        tr = 2;
        seen = ctx->peer_seen;
    } else if (ctx->comm) {
        tr = 1;
        seen = ctx->nranks;
        const SrlRccl *rc = srl_rccl();
        int c = 0;
        if (rc && rc->CommCount && rc->CommCount(ctx->comm, &c) == ncclSuccess) seen = c;
    } else if (ctx->cb_ar && ctx->nranks > 1) {
        tr = 3;
        seen = ctx->nranks;
    }
    if (transport) *transport = tr;
    if (nranks) *nranks = ctx->nranks;
    if (rank) *rank = ctx->rank;
    if (ranks_seen) *ranks_seen = seen;
    if (passes_armed) *passes_armed = (int64_t)ctx->arm_stats[1];
    return SRL_OK;
}

int srl_comm_suspend(srl_ctx *ctx, int suspend) {
    if (!ctx) return SRL_ERR_BAD_ARG;
    SRL_DISARM(ctx);
    if (suspend) {
        if (ctx->parked_nranks == 0) {
            ctx->parked_comm = ctx->comm; ctx->parked_nranks = ctx->nranks; ctx->parked_rank = ctx->rank;
            ctx->comm = nullptr; ctx->nranks = 1; ctx->rank = 0;
        }
    } else if (ctx->parked_nranks != 0) {
        ctx->comm = ctx->parked_comm; ctx->nranks = ctx->parked_nranks; ctx->rank = ctx->parked_rank;
        ctx->parked_comm = nullptr; ctx->parked_nranks = 0; ctx->parked_rank = 0;
    }
    return SRL_OK;
}

int srl_comm_destroy(srl_ctx *ctx) {
    if (!ctx) return SRL_ERR_BAD_ARG;
    SRL_DISARM(ctx);
    if (ctx->parked_comm && srl_rccl()) { srl_rccl()->CommDestroy(ctx->parked_comm); ctx->parked_comm = nullptr; }
    ctx->parked_nranks = 0; ctx->parked_rank = 0;
    if (ctx->comm && srl_rccl()) { srl_rccl()->CommDestroy(ctx->comm); ctx->comm = nullptr; }
    ctx->nranks = 1; ctx->rank = 0;
    ctx->cb_ar = nullptr; ctx->cb_ag = nullptr; ctx->cb_user = nullptr;
    ctx->comm_shares_device = false;
    return SRL_OK;
}

// ---- direct peer exchange: the sharded sum through stores into the peers' inboxes (no RCCL call on the data path) ----
int srl_peer_export(srl_ctx *ctx, void *ipc_handle, void **local_ptr) {
    if (!ctx) return SRL_ERR_BAD_ARG;
    SRL_DISARM(ctx);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (ctx->peer_on) { ctx->err = "srl_peer_export: peers are attached (srl_peer_detach first)"; return SRL_ERR_BAD_ARG; }
    const size_t bytes = (size_t)SRL_PEER_INBOX_ALLOC_GRANULES * sizeof(unsigned long long);      // rows + the session's poison word
    if (!ctx->d_inbox)       // fine-grained: a peer's store is visible to this device's polling loads without cache maintenance
        HIPCHK(ctx, hipExtMallocWithFlags((void **)&ctx->d_inbox, bytes, hipDeviceMallocFinegrained));
    // Every export starts a new session: tag 0 = "nothing here", exchange tags start at 1 again at srl_peer_attach.  The reset
    // happens HERE -- before the handle leaves this rank, hence before any peer can store into the inbox -- never at attach
    // time, when a faster peer may already have delivered the first row of the session.
    if (ctx->stream) HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipMemset(ctx->d_inbox, 0, bytes));
    { const unsigned long long id = device_identity(ctx->device); HIPCHK(ctx, hipMemcpy(ctx->d_inbox + SRL_PEER_DEVICE_WORD, &id, sizeof id, hipMemcpyHostToDevice)); }
    if (ipc_handle) {
        static_assert(sizeof(hipIpcMemHandle_t) <= SRL_PEER_HANDLE_BYTES, "IPC handle does not fit");
        hipIpcMemHandle_t h;
        HIPCHK(ctx, hipIpcGetMemHandle(&h, ctx->d_inbox));
        std::memset(ipc_handle, 0, SRL_PEER_HANDLE_BYTES);
        std::memcpy(ipc_handle, &h, sizeof h);
    }
    if (local_ptr) *local_ptr = ctx->d_inbox;
    return SRL_OK;
}

int srl_peer_set_deadline_ms(srl_ctx *ctx, int deadline_ms) {
    if (!ctx || deadline_ms < 0) return SRL_ERR_BAD_ARG;
    ctx->peer_deadline_ms = deadline_ms;
    return SRL_OK;
}

int srl_peer_stats(srl_ctx *ctx, int64_t *passes_repeated, int *session_failed) {
    if (!ctx) return SRL_ERR_BAD_ARG;
    if (passes_repeated) *passes_repeated = (int64_t)ctx->peer_retries;
    if (session_failed) *session_failed = ctx->peer_failed ? 1 : 0;
    return SRL_OK;
}

int srl_peer_detach(srl_ctx *ctx) {
    if (!ctx) return SRL_ERR_BAD_ARG;
    SRL_DISARM(ctx);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (ctx->stream) HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (int r = 0; r < SRL_MAX_PEERS; r++) if (ctx->peer_mapped[r]) { hipIpcCloseMemHandle(ctx->peer_mapped[r]); ctx->peer_mapped[r] = nullptr; }
    for (int r = 0; r < SRL_MAX_PEERS; r++) ctx->peer_inbox[r] = nullptr;
    if (ctx->peer_on) { ctx->peer_on = false; ctx->nranks = 1; ctx->rank = 0; }
    ctx->peer_shares_device = false;
    return SRL_OK;
}

int srl_peer_attach(srl_ctx *ctx, int nranks, int rank, const void *ipc_handles, void *const *local_ptrs) {
    if (!ctx || nranks < 1 || nranks > SRL_MAX_PEERS || rank < 0 || rank >= nranks || (nranks > 1 && !ipc_handles && !local_ptrs)) return SRL_ERR_BAD_ARG;
    SRL_DISARM(ctx);
    if (ctx->comm || ctx->cb_ar) { ctx->err = "srl_peer_attach: another transport is attached (srl_comm_destroy first)"; return SRL_ERR_BAD_ARG; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (!ctx->d_inbox) { ctx->err = "srl_peer_attach: srl_peer_export first (it creates and resets this rank's inbox)"; return SRL_ERR_BAD_ARG; }
    { const int rcd = srl_peer_detach(ctx); if (rcd) return rcd; }
    SrlPeerTable t;
    std::memset(&t, 0, sizeof t);
    t.nranks = nranks; t.rank = rank;
    for (int r = 0; r < nranks; r++) {
        if (r == rank) { t.inbox[r] = ctx->d_inbox; continue; }
        if (local_ptrs && local_ptrs[r]) {
            // same process: the peer's allocation itself (another device: peer access is switched on once)
            hipPointerAttribute_t at;
            HIPCHK(ctx, hipPointerGetAttributes(&at, local_ptrs[r]));
            if (at.device != ctx->device) {
                const hipError_t pe = hipDeviceEnablePeerAccess(at.device, 0);
                if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) { ctx->err = std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(pe); return SRL_ERR_HIP; }
                (void)hipGetLastError();
            }
            t.inbox[r] = (unsigned long long *)local_ptrs[r];
        } else {
            if (!ipc_handles) { ctx->err = "srl_peer_attach: no handle for a peer"; return SRL_ERR_BAD_ARG; }
            hipIpcMemHandle_t h;
            std::memcpy(&h, (const char *)ipc_handles + (size_t)r * SRL_PEER_HANDLE_BYTES, sizeof h);
            void *p = nullptr;
            HIPCHK(ctx, hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
            ctx->peer_mapped[r] = p;
            t.inbox[r] = (unsigned long long *)p;
        }
    }
    if (!ctx->d_peer) { int rcp = ensure(ctx, ctx->d_peer, 1); if (rcp) return rcp; }
    HIPCHK(ctx, hipMemcpy(ctx->d_peer, &t, sizeof t, hipMemcpyHostToDevice));
    // does a peer's inbox live on this very device?  (its export left the device's identity behind the rows)
    ctx->peer_shares_device = false;
    {
        const unsigned long long mine = device_identity(ctx->device);
        for (int r = 0; r < nranks; r++) {
            if (r == rank || !t.inbox[r]) continue;
            unsigned long long theirs = 0;
            HIPCHK(ctx, hipMemcpy(&theirs, t.inbox[r] + SRL_PEER_DEVICE_WORD, sizeof theirs, hipMemcpyDeviceToHost));
            if (theirs == mine) ctx->peer_shares_device = true;
        }
    }
    for (int r = 0; r < SRL_MAX_PEERS; r++) ctx->peer_inbox[r] = r < nranks ? t.inbox[r] : nullptr;
    ctx->peer_retries = 0;
    ctx->peer_seen = 0;
    for (int r = 0; r < nranks; r++) ctx->peer_seen += t.inbox[r] != nullptr ? 1 : 0;       // inboxes actually mapped (srl_comm_info)
    { int rcg = ensure_gather(ctx, (size_t)nranks); if (rcg) return rcg; }
    ctx->nranks = nranks; ctx->rank = rank;
    ctx->peer_on = nranks > 1;
    ctx->peer_seq = 0;                       // every rank starts counting from the same attach
    ctx->peer_failed = false;
    ctx->dbg_gather = false;
    return SRL_OK;
}

int srl_debug_set_gather_counts(srl_ctx *ctx, int nranks, int rank, const int64_t *counts) {
    if (ctx) SRL_DISARM(ctx);
    // counts == NULL: back to an unsharded context.  Otherwise the context behaves as rank `rank` of `nranks` whose
    // all-gather of per-rank counts has already delivered `counts` (on-device budget derivation), with an identity all-reduce.
    if (!ctx || (counts && (nranks < 1 || rank < 0 || rank >= nranks))) return SRL_ERR_BAD_ARG;
    if (ctx->comm) { ctx->err = "srl_debug_set_gather_counts: a communicator is attached"; return SRL_ERR_BAD_ARG; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (!counts) { ctx->dbg_gather = false; ctx->nranks = 1; ctx->rank = 0; ctx->cb_ar = nullptr; ctx->cb_ag = nullptr; return SRL_OK; }
    int rc = ensure_gather(ctx, (size_t)nranks);
    if (rc) return rc;
    std::vector<long long> c64(counts, counts + nranks);
    HIPCHK(ctx, hipMemcpy(ctx->d_gather, c64.data(), (size_t)nranks * sizeof(long long), hipMemcpyHostToDevice));
    ctx->nranks = nranks; ctx->rank = rank;
    ctx->cb_ar = [](double *, int, void *) { return 0; };
    ctx->cb_ag = nullptr;
    ctx->dbg_gather = true;
    return SRL_OK;
}

int srl_comm_set_host_callbacks(srl_ctx *ctx, int nranks, int rank, srl_allreduce_fn ar, srl_allgather_i64_fn ag, void *user) {
    if (!ctx || nranks < 1 || rank < 0 || rank >= nranks || (nranks > 1 && (!ar || !ag))) return SRL_ERR_BAD_ARG;
    SRL_DISARM(ctx);
    if (ctx->peer_on) { ctx->err = "srl_comm_set_host_callbacks: direct peer exchange is attached (srl_peer_detach first)"; return SRL_ERR_BAD_ARG; }
    if (ctx->comm && srl_rccl()) { srl_rccl()->CommDestroy(ctx->comm); ctx->comm = nullptr; }
    ctx->nranks = nranks; ctx->rank = rank;
    ctx->cb_ar = ar; ctx->cb_ag = ag; ctx->cb_user = user;
    ctx->dbg_gather = false;
    return SRL_OK;
}

}  // extern "C"
