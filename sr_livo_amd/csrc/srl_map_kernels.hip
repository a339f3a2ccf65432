// srl_map_kernels.hip -- device-resident voxel map mutation for gfx950:
// lioOptimization::addPointsToMap / addPointToMap (src/lioOptimization.cpp:520-554, 400-446).
//
// The reference inserts frame points one by one; the result depends on the order only WITHIN a voxel
// (first come, first kept; min-distance test against residents incl. earlier points of the same
// batch) and on the order in which new voxels are created (it defines slab ids = point ids).
// Device formulation that reproduces both bit-for-bit:
//   1. key[i]  = short(float(p_i) / voxel_size) per axis (insert keys come from the FP32-rounded
//      position, lioOptimization.cpp:403-405 via cloudMap.cpp:7,28)
//   2. stable radix sort of (key, i)  -> per-voxel segments in original point order
//   3. run-length encode -> touched voxels; hash lookup; new voxels ranked by the index of their
//      first point (= creation order) -> slab ids; CAS-insert into the open-addressing table
//   4. one thread per touched voxel replays its segment sequentially (IsFull, min-distance, cap).
#include "srl_ctx.h"
#include "srl_frame_scratch.h"
#include "srl_hash.h"


#include <vector>

namespace {


// Frame-sized batches: the sort only has to bring the points of a voxel TOGETHER, in their original order -- not the voxels into
// key order.  So the 48-bit key is first replaced by the slot the voxel claims in a scratch table of >= 2 n slots (open
// addressing, compare-and-swap; one slot per distinct key), and the stable radix sort runs over log2(slots) bits: 2 passes for a
// 24k-point frame instead of 6.  Which segment comes first is irrelevant downstream (creation order = first-point rank, replay =
// per segment).
// XF: the points are first re-transformed (optimize.cpp:441-445: point = R(q) * (R_il * raw + t_il) + t, the operation order of
// k_frame_keys / transformPoint) and stored where the rest of the insertion -- and the download of point3D::point -- reads them
template <bool XF>
__global__ void k_point_slots(const double *xyz, int n, double voxel_size, unsigned long long *keyw, unsigned mask, unsigned epoch16, unsigned *slot_out,
                              unsigned *idx, int *new_flag, const SrlFrameTransform T) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double wx, wy, wz;
    if (XF) {
        const SrlXf &X = T.X;
        const double rx = T.raw[(size_t)i * 3], ry = T.raw[(size_t)i * 3 + 1], rz = T.raw[(size_t)i * 3 + 2];
        const double ix = (X.R_il[0] * rx + X.R_il[1] * ry) + X.R_il[2] * rz + X.t_il[0];
        const double iy = (X.R_il[3] * rx + X.R_il[4] * ry) + X.R_il[5] * rz + X.t_il[1];
        const double iz = (X.R_il[6] * rx + X.R_il[7] * ry) + X.R_il[8] * rz + X.t_il[2];
        wx = (X.R[0] * ix + X.R[1] * iy) + X.R[2] * iz + X.t[0];
        wy = (X.R[3] * ix + X.R[4] * iy) + X.R[5] * iz + X.t[1];
        wz = (X.R[6] * ix + X.R[7] * iy) + X.R[8] * iz + X.t[2];
        T.world[(size_t)i * 3] = wx; T.world[(size_t)i * 3 + 1] = wy; T.world[(size_t)i * 3 + 2] = wz;
    } else {
        wx = xyz[(size_t)i * 3]; wy = xyz[(size_t)i * 3 + 1]; wz = xyz[(size_t)i * 3 + 2];
    }
    const float fx = (float)wx, fy = (float)wy, fz = (float)wz;
    const short kx = (short)(int)((double)fx / voxel_size);
    const short ky = (short)(int)((double)fy / voxel_size);
    const short kz = (short)(int)((double)fz / voxel_size);
    const unsigned long long key = srl_pack_key(kx, ky, kz);
    slot_out[i] = srl_epoch_claim(keyw, mask, epoch16, key, srl_hash_key(key));     // (epoch-tagged scratch table: no fill per frame)
    if (idx) idx[i] = (unsigned)i;         // (the frame path sorts positions: srl_radix_sort_pairs needs no value array)
    new_flag[i] = 0;                       // (the segment scan, behind the sort, sets the marks: no fill in front of this one)
}
// head flag of sorted position i / the per-element work of the scan over them: segment starts, and the voxel key of every sorted
// position restored from the scratch table (in the pass of the scan itself: k_scan_small)
struct HeadFlag32 {
    const unsigned *slots;
    __host__ __device__ int operator()(int i) const { return (i == 0 || slots[i] != slots[i - 1]) ? 1 : 0; }
};
struct SegmentSink {
    const unsigned *slots_sorted;
    const unsigned *idx_sorted;
    const unsigned long long *keyw;
    int *seg_start;
    unsigned long long *keys_sorted;          // written at segment heads only: the voxel key of the segment
    const SrlMapSlot *table;                  // the map's table: the table lookup happens here, at the head of every segment
    unsigned mask;
    int *seg_slot;
    unsigned char *is_new;
    int *new_flag;                            // or nullptr (min_num_points > 0: nothing is created)
    int *seg_of_first;                        // [first point index of a NEW voxel] -> its segment (for CreateSink)
    int *counters;
    int n;
    __device__ void operator()(int i, int head, int excl) const {
        if (head) {
            const unsigned long long key = keyw[slots_sorted[i]] & SRL_KEY48_MASK;
            keys_sorted[i] = key;
            seg_start[excl] = i;
            unsigned h = srl_hash_key(key) & mask;
            int slot = -1;
            for (unsigned probe = 0; probe <= mask; ++probe) {
                const unsigned long long k = table[h].key;
                if (k == key) { slot = (int)h; break; }
                if (k == SRL_EMPTY_KEY) break;
                h = (h + 1) & mask;
            }
            seg_slot[excl] = slot;
            is_new[excl] = slot < 0 ? 1 : 0;
            if (slot < 0 && new_flag) {
                const unsigned first = idx_sorted[i];              // stable sort: the segment's first element is its earliest point
                new_flag[first] = 1;
                seg_of_first[first] = excl;
            }
        }
        if (i == n - 1) { counters[0] = excl + head; counters[1] = 0; counters[2] = 0; }      // segments | new voxels (CreateSink) | points added (k_replay)
    }
};
// ... and the per-element work of the scan over the new-voxel marks (point-index space): the exclusive prefix IS the creation rank
// (the sequential loop of lioOptimization.cpp:520-554 creates voxels in the order their first points arrive)
struct CreateSink {
    const int *seg_of_first;
    const int *seg_start;
    const unsigned long long *keys_sorted;
    int V;
    SrlMapSlot *table;
    unsigned mask;
    unsigned char *slabs;
    int *seg_slot;
    int *counters;
    int n;
    __device__ void operator()(int i, int is_first_of_new, int rank) const {
        if (is_first_of_new) {
            const int s = seg_of_first[i];
            const unsigned long long key = keys_sorted[seg_start[s]];
            const unsigned slab = (unsigned)(V + rank);
            SrlSlab *sl = reinterpret_cast<SrlSlab *>(slabs + (size_t)slab * SRL_SLAB_BYTES);
            sl->count = 0;
            sl->pad = 0;
            sl->key = key;
            unsigned h = srl_hash_key(key) & mask;
            for (unsigned probe = 0; probe <= mask; ++probe) {
                const unsigned long long prev = atomicCAS(&table[h].key, SRL_EMPTY_KEY, key);
                if (prev == SRL_EMPTY_KEY) {
                    table[h].slab = slab;
                    table[h].count = 0;
                    seg_slot[s] = (int)h;
                    break;
                }
                h = (h + 1) & mask;
            }
        }
        if (i == n - 1) counters[1] = rank + is_first_of_new;
    }
};


// sequential replay of one voxel's segment (lioOptimization.cpp:409-445), one thread per voxel.  The kernel is latency bound,
// not work bound: a 1 M-point batch touches 67 k voxels = ONE wave per SIMD, and every point costs its thread two dependent
// memory round trips (sorted index -> coordinates, ~2 us together from HBM) before 20 compares.  So the incoming points are
// fetched SRL_REPLAY_BATCH at a time -- all indices, then all coordinates in flight together -- and the voxel's stored points
// are kept in registers, so that the 20-wide compare is arithmetic only (read from the slab it cost ~20 dependent L1 round
// trips per point: that, not the work, was the kernel).  Same decisions in the same order, the same FP64 operations.
// Measured on the 1 M-point / 10 M-point builds (67 k / 500 k voxels; tools/map_build_probe.py under rocprofv3): round 2's
// form (one load per iteration, compares against the slab) 161 us / 3.03 ms; batched loads alone 146 us / 2.08 ms; batched
// loads + stored points in registers 88 us / 0.70 ms (this kernel).  Rejected: a 32-lane group per voxel 321 us (32x the
// waves for a <= 20-wide compare); a hybrid -- this kernel for segments <= 48 points plus one WAVE per longer segment (chunks
// of 64 points resolved in order with ballots) -- 157 + 131 us and 2.5 + 1.2 ms: the long segments were never the tail (a
// dense voxel fills to 20 and stops), the per-point latency of the many short ones was.
#define SRL_REPLAY_BATCH 8
__global__ void __launch_bounds__(128) k_replay(const int *seg_start, const unsigned *sorted_idx, const int *counters, int n, const double *xyz,
                         const int *seg_slot, const unsigned char *is_new, SrlMapSlot *table, unsigned char *slabs,
                         double voxel_size, double min_distance_points, int min_num_points, int *added_total) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    const int S = counters[0];
    if (s >= S) return;
    const int slot = seg_slot[s];
    if (slot < 0) return;                       // voxel absent and min_num_points > 0: nothing is created
    const unsigned slab = table[slot].slab;
    SrlSlab *sl = reinterpret_cast<SrlSlab *>(slabs + (size_t)slab * SRL_SLAB_BYTES);
    int count = (int)sl->count;
    const bool fresh = is_new[s] != 0;
    const int j0 = seg_start[s], j1 = s + 1 < S ? seg_start[s + 1] : n;
    int added = 0;
    const double min_d2 = min_distance_points * min_distance_points;
    // the voxel's stored points live in REGISTERS for the whole replay (statically indexed: every loop over them is unrolled
    // and predicated on k < count); the slab is written through as points are added
    float sx[SRL_CAP], sy[SRL_CAP], sz[SRL_CAP];
#pragma unroll
    for (int k = 0; k < SRL_CAP; ++k) {
        const bool h = k < count;
        sx[k] = h ? sl->xyz[k][0] : 0.0f; sy[k] = h ? sl->xyz[k][1] : 0.0f; sz[k] = h ? sl->xyz[k][2] : 0.0f;
    }
    for (int jb = j0; jb < j1 && count < SRL_CAP; jb += SRL_REPLAY_BATCH) {
        unsigned idx[SRL_REPLAY_BATCH];
        float px[SRL_REPLAY_BATCH], py[SRL_REPLAY_BATCH], pz[SRL_REPLAY_BATCH];
#pragma unroll
        for (int b = 0; b < SRL_REPLAY_BATCH; ++b) idx[b] = sorted_idx[jb + b < j1 ? jb + b : j1 - 1];
#pragma unroll
        for (int b = 0; b < SRL_REPLAY_BATCH; ++b) {
            px[b] = (float)xyz[(size_t)idx[b] * 3]; py[b] = (float)xyz[(size_t)idx[b] * 3 + 1]; pz[b] = (float)xyz[(size_t)idx[b] * 3 + 2];
        }
#pragma unroll
        for (int b = 0; b < SRL_REPLAY_BATCH; ++b) {
            if (jb + b >= j1 || count == SRL_CAP) break;            // IsFull(): every later point of the batch is dropped too
            const float fx = px[b], fy = py[b], fz = pz[b];
            bool add;
            if (fresh && count == 0) {
                add = true;                         // new voxel: first point is stored unconditionally (:437-443)
            } else {
                double sq_dist_min = 10 * voxel_size * voxel_size;
#pragma unroll
                for (int k = 0; k < SRL_CAP; ++k) {
                    if (k < count) {
                        const double dx = (double)sx[k] - (double)fx;
                        const double dy = (double)sy[k] - (double)fy;
                        const double dz = (double)sz[k] - (double)fz;
                        const double sq = (dx * dx + dy * dy) + dz * dz;
                        if (sq < sq_dist_min) sq_dist_min = sq;
                    }
                }
                add = (sq_dist_min > min_d2) && (min_num_points <= 0 || count >= min_num_points);
            }
            if (add) {
                sl->xyz[count][0] = fx; sl->xyz[count][1] = fy; sl->xyz[count][2] = fz;
#pragma unroll
                for (int k = 0; k < SRL_CAP; ++k) if (k == count) { sx[k] = fx; sy[k] = fy; sz[k] = fz; }
                ++count;
                ++added;
            }
        }
    }
    sl->count = (unsigned)count;
    table[slot].count = (unsigned)count;
    if (added) atomicAdd(added_total, added);
}

// srl_map_probe_checksum: every point looks its voxel up and adds srl_probe_mix(key, count, last stored point) to one 64-bit sum
__global__ void k_probe_checksum(const double *xyz, int n, int stride, double voxel_size, const SrlMapSlot *table, unsigned mask, const unsigned char *slabs,
                                 unsigned long long *sum) {
    const long long i = (long long)(blockIdx.x * blockDim.x + threadIdx.x) * stride;
    unsigned long long mine = 0ull;
    if (i < n) {
        const float fx = (float)xyz[(size_t)i * 3], fy = (float)xyz[(size_t)i * 3 + 1], fz = (float)xyz[(size_t)i * 3 + 2];
        const short kx = (short)(int)((double)fx / voxel_size), ky = (short)(int)((double)fy / voxel_size), kz = (short)(int)((double)fz / voxel_size);
        const unsigned long long key = srl_pack_key(kx, ky, kz);
        unsigned h = srl_hash_key(key) & mask;
        for (unsigned probe = 0; probe <= mask; ++probe) {
            const SrlMapSlot sl = table[h];
            if (sl.key == key) {
                const SrlSlab *sb = reinterpret_cast<const SrlSlab *>(slabs + (size_t)sl.slab * SRL_SLAB_BYTES);
                const int c = (int)sb->count;
                if (c > 0) mine = srl_probe_mix(kx, ky, kz, c, sb->xyz[c - 1][0], sb->xyz[c - 1][1], sb->xyz[c - 1][2]);
                break;
            }
            if (sl.key == SRL_EMPTY_KEY) break;
            h = (h + 1) & mask;
        }
    }
    // wave sum first (integers: order free), one atomic per wave
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mine += __shfl_xor(mine, off);
    if ((threadIdx.x & 63) == 0 && mine != 0ull) atomicAdd(sum, mine);
}

__global__ void k_rebuild_table(const unsigned char *slabs, int V, SrlMapSlot *table, unsigned mask) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    const SrlSlab *sl = reinterpret_cast<const SrlSlab *>(slabs + (size_t)v * SRL_SLAB_BYTES);
    const unsigned long long key = sl->key;
    unsigned h = srl_hash_key(key) & mask;
    for (unsigned probe = 0; probe <= mask; ++probe) {
        const unsigned long long prev = atomicCAS(&table[h].key, SRL_EMPTY_KEY, key);
        if (prev == SRL_EMPTY_KEY) { table[h].slab = (unsigned)v; table[h].count = sl->count; return; }
        h = (h + 1) & mask;
    }
}

__global__ void k_fill_empty(SrlMapSlot *table, unsigned cap) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cap) return;
    SrlMapSlot s; s.key = SRL_EMPTY_KEY; s.slab = 0; s.count = 0;
    table[i] = s;
}

unsigned next_pow2u(unsigned v) { unsigned p = 1; while (p < v) p <<= 1; return p; }

}  // namespace

// grow slab storage / hash table so that `need_slabs` voxels fit with load <= 0.5
extern "C" int srl_map_probe_checksum(srl_ctx *ctx, const double *world_xyz, int n, int stride, double voxel_size, uint64_t *checksum) {
    if (!ctx || !checksum || !(voxel_size > 0.0) || (world_xyz && n < 0) || stride < 1) return SRL_ERR_BAD_ARG;
    SRL_DISARM(ctx);
    *checksum = 0;
    if (!ctx->d_table) return SRL_ERR_NO_MAP;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    { const int rcs = srl_map_settle(ctx); if (rcs) return rcs; }
    const double *d_pts = nullptr;
    DevBuf b_pts, b_sum;
    if (world_xyz) {
        if (n == 0) return SRL_OK;
        HIPCHK(ctx, b_pts.alloc(ctx, (size_t)n * 3 * sizeof(double)));
        HIPCHK(ctx, hipMemcpyAsync(b_pts.p, world_xyz, (size_t)n * 3 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        d_pts = b_pts.as<double>();
    } else {
        if (ctx->frame_world_n <= 0 || !ctx->d_frame_world) return SRL_OK;     // nothing committed (or a newer frame uploaded since): the empty sum
        n = ctx->frame_world_n;
        d_pts = ctx->d_frame_world;
    }
    HIPCHK(ctx, b_sum.alloc(ctx, 256));
    HIPCHK(ctx, hipMemsetAsync(b_sum.p, 0, 8, ctx->stream));
    const int probes = (n + stride - 1) / stride;
    hipLaunchKernelGGL(k_probe_checksum, dim3((probes + 255) / 256), dim3(256), 0, ctx->stream, d_pts, n, stride, voxel_size, ctx->d_table, ctx->table_cap - 1,
                       ctx->d_slabs, b_sum.as<unsigned long long>());
    HIPCHK(ctx, hipGetLastError());
    unsigned long long out = 0ull;
    HIPCHK(ctx, hipMemcpyAsync(&out, b_sum.p, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    *checksum = (uint64_t)out;
    return SRL_OK;
}

int srl_ctx_grow_map(srl_ctx *ctx, unsigned need_slabs, unsigned need_slots) {
    const unsigned new_slab_cap = ctx->slab_cap >= need_slabs ? ctx->slab_cap : std::max(need_slabs + need_slabs / 2u, 1024u);
    if (new_slab_cap > SRL_MAX_SLABS) { ctx->err = "map too large: slab byte offsets are 32-bit (16.7 M voxels)"; return SRL_ERR_UNSUPPORTED; }
    unsigned new_table_cap = ctx->table_cap;
    while (new_table_cap < need_slots || new_table_cap < SRL_TABLE_FACTOR * new_slab_cap) new_table_cap = next_pow2u(new_table_cap ? new_table_cap * 2u : 2048u);
    if (new_slab_cap != ctx->slab_cap) {
        unsigned char *ns = nullptr;
        HIPCHK(ctx, hipMalloc((void **)&ns, ((size_t)new_slab_cap + 1) * SRL_SLAB_BYTES));          // + the all-inf slab
        HIPCHK(ctx, hipMemsetD32Async((hipDeviceptr_t)(ns + (size_t)new_slab_cap * SRL_SLAB_BYTES), 0x7f800000, SRL_SLAB_BYTES / 4, ctx->stream));
        HIPCHK(ctx, hipMemsetAsync(ns, 0, (size_t)new_slab_cap * SRL_SLAB_BYTES, ctx->stream));
        if (ctx->d_slabs && ctx->num_voxels > 0)
            HIPCHK(ctx, hipMemcpyAsync(ns, ctx->d_slabs, (size_t)ctx->num_voxels * SRL_SLAB_BYTES, hipMemcpyDeviceToDevice, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->d_slabs) HIPCHK(ctx, hipFree(ctx->d_slabs));
        ctx->d_slabs = ns;
        ctx->slab_cap = new_slab_cap;
    }
    if (new_table_cap != ctx->table_cap) {
        SrlMapSlot *nt = nullptr;
        HIPCHK(ctx, hipMalloc((void **)&nt, (size_t)new_table_cap * sizeof(SrlMapSlot)));
        hipLaunchKernelGGL(k_fill_empty, dim3((new_table_cap + 255) / 256), dim3(256), 0, ctx->stream, nt, new_table_cap);
        if (ctx->num_voxels > 0)
            hipLaunchKernelGGL(k_rebuild_table, dim3((ctx->num_voxels + 255) / 256), dim3(256), 0, ctx->stream,
                               ctx->d_slabs, ctx->num_voxels, nt, new_table_cap - 1);
        HIPCHK(ctx, hipGetLastError());
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->d_table) HIPCHK(ctx, hipFree(ctx->d_table));
        ctx->d_table = nt;
        ctx->table_cap = new_table_cap;
    }
    return SRL_OK;
}

int srl_map_insert_impl(srl_ctx *ctx, const double *world_xyz, bool on_device, int n, double voxel_size,
                        double min_distance_points, int min_num_points, int *num_added, bool defer_counters,
                        const SrlFrameTransform *xf = nullptr, int (*after_first_kernel)(srl_ctx *, void *) = nullptr, void *user = nullptr);

// debug / parity hook: the frame path's own stable sort (srl_frame_scratch.h) on caller data
extern "C" int srl_debug_radix_sort_pairs(srl_ctx *ctx, const uint32_t *keys, int n, int bits, uint32_t *keys_sorted, uint32_t *positions_sorted) {
    if (!ctx || n < 0 || n > (1 << 25) || bits < 1 || bits > 3 * SRL_RADIX_MAX_BITS || (n > 0 && (!keys || !keys_sorted || !positions_sorted)))
        return SRL_ERR_BAD_ARG;
    if (n == 0) return SRL_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    DevBuf b_in, b_k, b_v, b_tk, b_tv;
    HIPCHK(ctx, b_in.alloc(ctx, (size_t)n * 4)); HIPCHK(ctx, b_k.alloc(ctx, (size_t)n * 4)); HIPCHK(ctx, b_v.alloc(ctx, (size_t)n * 4));
    HIPCHK(ctx, b_tk.alloc(ctx, (size_t)n * 4)); HIPCHK(ctx, b_tv.alloc(ctx, (size_t)n * 4));
    DevBuf b_sc;
    HIPCHK(ctx, b_sc.alloc(ctx, srl_radix_scratch_ints(n) * 4));
    HIPCHK(ctx, hipMemcpyAsync(b_in.p, keys, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    srl_radix_sort_pairs(b_in.as<unsigned>(), nullptr, b_k.as<unsigned>(), b_v.as<unsigned>(), b_tk.as<unsigned>(), b_tv.as<unsigned>(), n, (unsigned)bits, ctx->stream,
                         b_sc.as<int>());
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(keys_sorted, b_k.p, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(positions_sorted, b_v.p, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return SRL_OK;
}

int srl_map_insert_device(srl_ctx *ctx, const double *world_xyz, int n, double voxel_size, int cap,
                          double min_distance_points, int min_num_points, int *num_added) {
    (void)cap;
    return srl_map_insert_impl(ctx, world_xyz, false, n, voxel_size, min_distance_points, min_num_points, num_added, false);
}

// world_xyz: host pointer, or (on_device) a device pointer that stays valid for the duration of the call.
//
// addPointsToMap's order semantics (lioOptimization.cpp:400-446,520-554) on the device, ONE host synchronisation per call (round 3:
// four, two of them through pageable copies):
//   keys -> stable radix sort (key, index) -> head flags scanned into segment ids -> segment starts -> table lookup per segment ->
//   new voxels ranked by the index of their first point (a scan over marks in point-index space = the sequential loop's creation
//   order) -> created (slab = V + rank, key CAS-inserted) -> one thread per touched voxel replays its segment sequentially.
// Nothing is read back before the end: the kernels take the segment count from device memory, storage is grown beforehand for the
// worst case (every point a new voxel) when the batch is a frame (n <= 131072); a bulk load reads the segment count once to size the map.
// defer_counters (frame-sized batches, num_added == NULL): return once everything is enqueued; the counters are folded into the
// map's totals by srl_map_settle() -- at the next insert, srl_map_size, srl_map_download.  The solve that follows needs none of them
// and is ordered behind the insert on the stream.
int srl_map_insert_impl(srl_ctx *ctx, const double *world_xyz, bool on_device, int n, double voxel_size,
                        double min_distance_points, int min_num_points, int *num_added, bool defer_counters,
                        const SrlFrameTransform *xf, int (*after_first_kernel)(srl_ctx *, void *), void *user) {
    if (num_added) *num_added = 0;
    if (n == 0) return SRL_OK;
    if (!(voxel_size > 0.0)) return SRL_ERR_BAD_ARG;
    ctx->bound_n = 0;                // the map changes: what the last pass learnt about its keypoints' neighbourhoods is void
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    { const int rcs = srl_map_settle(ctx); if (rcs) return rcs; }
    if (!ctx->h_insert_cnt) {
        HIPCHK(ctx, hipHostMalloc((void **)&ctx->h_insert_cnt, 64, hipHostMallocDefault));
        HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_insert, hipEventDisableTiming));
    }
    if (!ctx->d_table) {            // empty map: create storage
        ctx->slab_cap = 0; ctx->table_cap = 0; ctx->num_voxels = 0; ctx->num_points = 0;
        int rc = srl_ctx_grow_map(ctx, 4096u, 8192u);
        if (rc) return rc;
    }
    // frame_sized: storage is grown beforehand for the worst case (every point a new voxel) and nothing is read back before the end; a bulk
    // load (> 1 M points) reads the segment count once to size the map.  The KERNELS are the same for both (round 6: frames beyond 131 072
    // points and bulk loads used to leave for hipcub's radix sort and scans): own radix passes and scans of any size (srl_frame_scratch.h).
    const bool frame_sized = n <= SRL_SCAN_MAX;
    if (frame_sized) {
        const unsigned need_slabs = (unsigned)ctx->num_voxels + (unsigned)n;
        if (need_slabs > ctx->slab_cap || SRL_TABLE_FACTOR * need_slabs > ctx->table_cap) {
            int rc = srl_ctx_grow_map(ctx, need_slabs, SRL_TABLE_FACTOR * need_slabs);
            if (rc) return rc;
        }
    }

    DevBuf b_xyz, b_keys2, b_idx, b_idx2, b_prefix, b_start, b_cnt, b_slot, b_isnew, b_first, b_newflag;
    const double *d_xyz = world_xyz;
    if (!on_device) {
        HIPCHK(ctx, b_xyz.alloc(ctx, (size_t)n * 3 * sizeof(double)));
        HIPCHK(ctx, hipMemcpyAsync(b_xyz.p, world_xyz, (size_t)n * 3 * sizeof(double), hipMemcpyHostToDevice, st));
        d_xyz = b_xyz.as<double>();
    }
    HIPCHK(ctx, b_keys2.alloc(ctx, (size_t)n * 8));
    HIPCHK(ctx, b_idx.alloc(ctx, (size_t)n * 4));
    HIPCHK(ctx, b_idx2.alloc(ctx, (size_t)n * 4));
    HIPCHK(ctx, b_prefix.alloc(ctx, (size_t)n * 4));
    HIPCHK(ctx, b_start.alloc(ctx, (size_t)n * 4));
    HIPCHK(ctx, b_cnt.alloc(ctx, 64));                       // counters: [0] segments, [1] new voxels, [2] points added
    HIPCHK(ctx, b_slot.alloc(ctx, (size_t)n * 4));
    HIPCHK(ctx, b_isnew.alloc(ctx, (size_t)n));
    HIPCHK(ctx, b_first.alloc(ctx, (size_t)n * 4));
    HIPCHK(ctx, b_newflag.alloc(ctx, (size_t)n * 4));
    int *cnt = b_cnt.as<int>();
    srl_stage_begin(ctx);
    DevBuf b_sc;
    HIPCHK(ctx, b_sc.alloc(ctx, (srl_radix_scratch_ints(n) + srl_scan_scratch_ints(n)) * 4));
    int *sc_radix = b_sc.as<int>(), *sc_scan = b_sc.as<int>() + srl_radix_scratch_ints(n);
    {
        // group by scratch-table slot, stable sort over log2(slots) bits (k_point_slots); the scan over the head flags writes
        // the segment starts in the same pass.  (The kernels initialise what they accumulate into themselves: k_point_slots, SegmentSink.)
        unsigned cap2 = 1024, bits = 10;
        while (cap2 < 2u * (unsigned)n) { cap2 <<= 1; ++bits; }
        int rct = srl_epoch_table_begin(ctx, ctx->ins_table, cap2, false);
        if (rct) return rct;
        const SrlEpochTable &T = ctx->ins_table;
        DevBuf b_slot_in, b_slot_sorted;
        HIPCHK(ctx, b_slot_in.alloc(ctx, (size_t)n * 4)); HIPCHK(ctx, b_slot_sorted.alloc(ctx, (size_t)n * 4));
        if (xf) {
            hipLaunchKernelGGL(k_point_slots<true>, dim3((n + 255) / 256), dim3(256), 0, st, d_xyz, n, voxel_size, T.keyw, cap2 - 1, T.epoch16,
                               b_slot_in.as<unsigned>(), (unsigned *)nullptr, b_newflag.as<int>(), *xf);
        } else {
            hipLaunchKernelGGL(k_point_slots<false>, dim3((n + 255) / 256), dim3(256), 0, st, d_xyz, n, voxel_size, T.keyw, cap2 - 1, T.epoch16,
                               b_slot_in.as<unsigned>(), (unsigned *)nullptr, b_newflag.as<int>(), SrlFrameTransform());
        }
        HIPCHK(ctx, hipGetLastError());
        if (after_first_kernel) { const int rca = after_first_kernel(ctx, user); if (rca) return rca; }
        // (slot, position) sorted stably over log2(slots) bits: passes of our own (srl_frame_scratch.h: one launch each up to 131 072 points,
        // two up to 1 M, five for bulk loads) instead of the library's sort
        srl_radix_sort_pairs(b_slot_in.as<unsigned>(), nullptr, b_slot_sorted.as<unsigned>(), b_idx2.as<unsigned>(), b_prefix.as<unsigned>(), b_idx.as<unsigned>(), n,
                             bits, st, sc_radix);
        HIPCHK(ctx, hipGetLastError());
        srl_stage_end(ctx, 7);                                    // slots + sort
        srl_scan(HeadFlag32{b_slot_sorted.as<unsigned>()},
                 SegmentSink{b_slot_sorted.as<unsigned>(), b_idx2.as<unsigned>(), T.keyw, b_start.as<int>(), b_keys2.as<unsigned long long>(), ctx->d_table,
                             ctx->table_cap - 1, b_slot.as<int>(), b_isnew.as<unsigned char>(), min_num_points <= 0 ? b_newflag.as<int>() : (int *)nullptr,
                             b_first.as<int>(), cnt, n}, n, sc_scan, st);
        HIPCHK(ctx, hipGetLastError());
    }

    srl_stage_end(ctx, 8);                                    // segments
    int rcs = ensure_host_scratch(ctx, 64);
    if (rcs) return rcs;
    int *h_cnt = reinterpret_cast<int *>(ctx->h_scratch);
    if (!frame_sized) {
        // a bulk load: the number of touched voxels sizes the map (worst-case sizing would reserve a slab per point)
        HIPCHK(ctx, hipMemcpyAsync(h_cnt, cnt, sizeof(int), hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipStreamSynchronize(st));
        const unsigned need_slabs = (unsigned)ctx->num_voxels + (unsigned)h_cnt[0];
        if (need_slabs > ctx->slab_cap || SRL_TABLE_FACTOR * need_slabs > ctx->table_cap) {
            int rc = srl_ctx_grow_map(ctx, need_slabs, SRL_TABLE_FACTOR * need_slabs);
            if (rc) return rc;
        }
    }
    const unsigned mask = ctx->table_cap - 1;
    const bool create_new = min_num_points <= 0;             // min_num_points > 0: a point never opens a voxel (lioOptimization.cpp:437)
    {
        // the lookup has happened at the segment heads of the scan above; creation = the per-element work of the scan over the new-voxel marks
        if (create_new) {
            srl_scan(SrlIntArrayIn{b_newflag.as<int>()},
                     CreateSink{b_first.as<int>(), b_start.as<int>(), b_keys2.as<unsigned long long>(), ctx->num_voxels, ctx->d_table, mask, ctx->d_slabs,
                                b_slot.as<int>(), cnt, n}, n, sc_scan, st);
            HIPCHK(ctx, hipGetLastError());
        }
    }
    srl_stage_end(ctx, 9);                                    // lookup + creation
    hipLaunchKernelGGL(k_replay, dim3((n + 127) / 128), dim3(128), 0, st, b_start.as<int>(), b_idx2.as<unsigned>(), cnt, n,
                       d_xyz, b_slot.as<int>(), b_isnew.as<unsigned char>(), ctx->d_table, ctx->d_slabs, voxel_size,
                       min_distance_points, min_num_points, cnt + 2);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_insert_cnt, cnt, 3 * sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipEventRecord(ctx->ev_insert, st));
    ctx->insert_pending = true;
    if (defer_counters && frame_sized && !num_added && !ctx->frame_timing) return SRL_OK;
    { const int rcs = srl_map_settle(ctx); if (rcs) return rcs; }
    srl_stage_end(ctx, 10);                                   // replay + counters
    if (num_added) *num_added = ctx->h_insert_cnt[2];
    return SRL_OK;
}
