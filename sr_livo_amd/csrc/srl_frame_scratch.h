// srl_frame_scratch.h -- two building blocks of the per-frame kernels (keypoint selection, addPointsToMap on a frame), both there to take
// launches off a chain that is latency bound: a 24k-point frame keeps the device busy for ~250 us spread over ~40 launches of a few
// microseconds each, so every fill, every scan-state initialisation and every gap between two of them is a measurable part of the frame.
//
// 1. EpochTable: an open-addressing scratch table (voxel key -> slot) that is never cleared between frames.  Every entry carries the
//    16-bit epoch of the frame that wrote it in the bits the 48-bit voxel key leaves free; an entry of another epoch IS an empty slot and
//    is claimed by compare-and-swap from the stale value.  The table is filled with zeros when it is allocated and when the epoch wraps
//    (epoch 0 is never used), not per frame (a 64k-slot table is a 512 KB fill = one more launch of ~5 us in front of every use).
//    The companion word of the selection ("smallest point index of this voxel") needs no reset either: it holds
//    {0xFFFFFFFF - frame counter, index} and is only ever lowered by atomicMin, so any value of an earlier frame loses against the first
//    write of the current one.
// 2. k_scan_small / srl_scan: exclusive prefix sum in one launch (<= 131 072 ints), two (<= 1 M) or four (bulk), with a per-element sink that
//    can do the consumer's work in the same pass (segment starts of the sorted frame).  hipcub's decoupled look-back scan is built for millions of items: at 24k it
//    costs a state-initialisation kernel + the scan kernel, 10-16 us and two launches, three times per frame.
#pragma once
#include <hip/hip_runtime.h>

#include "srl_ctx.h"

#define SRL_KEY48_MASK 0xFFFFFFFFFFFFull

// transformPoint's operands (utility.cpp:314-318): point = R(q) * (R_il * raw + t_il) + t
struct SrlXf { double R[9], t[3], R_il[9], t_il[3]; };
// srl_frame_commit hands the re-transform of the frame (optimize.cpp:441-445) to the first kernel of the insertion behind it
struct SrlFrameTransform {
    const double *raw;       // the resident frame's raw points (AoS)
    double *world;           // where point3D::point goes (AoS); the insertion reads it from there
    SrlXf X;
};

// (struct SrlEpochTable: srl_ctx.h -- the context owns one for the selection and one for the insertion)

// make the table ready for a frame that needs `want_cap` slots (power of two); returns the epoch to tag this frame's entries with
inline int srl_epoch_table_begin(srl_ctx *ctx, SrlEpochTable &t, unsigned want_cap, bool with_min) {
    bool clear = false;
    if (want_cap > t.cap || (with_min && !t.minw)) {
        // grow: the old block may still be read by a kernel of the previous frame
        if (t.keyw || t.minw) HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        if (t.keyw) { HIPCHK(ctx, hipFree(t.keyw)); t.keyw = nullptr; }
        if (t.minw) { HIPCHK(ctx, hipFree(t.minw)); t.minw = nullptr; }
        const unsigned cap = want_cap > t.cap ? want_cap : t.cap;
        HIPCHK(ctx, hipMalloc((void **)&t.keyw, (size_t)cap * 8));
        if (with_min) HIPCHK(ctx, hipMalloc((void **)&t.minw, (size_t)cap * 8));
        t.cap = cap;
        clear = true;
    }
    if (++t.epoch16 > 0xFFFFu) { t.epoch16 = 1; clear = true; }
    if (++t.counter32 == 0xFFFFFFFFu) { t.counter32 = 1; clear = true; }
    if (clear) {
        HIPCHK(ctx, hipMemsetAsync(t.keyw, 0, (size_t)t.cap * 8, ctx->stream));
        if (t.minw) HIPCHK(ctx, hipMemsetAsync(t.minw, 0xFF, (size_t)t.cap * 8, ctx->stream));
    }
    return SRL_OK;
}
inline void srl_epoch_table_free(SrlEpochTable &t) {
    if (t.keyw) (void)hipFree(t.keyw);
    if (t.minw) (void)hipFree(t.minw);
    t = SrlEpochTable();
}

#if defined(__HIPCC__)
// slot of `key` in the table (claimed if absent); mask = slots in use for this frame - 1 (<= allocated - 1)
__device__ __forceinline__ unsigned srl_epoch_claim(unsigned long long *keyw, unsigned mask, unsigned epoch16, unsigned long long key, unsigned hash) {
    const unsigned long long want = ((unsigned long long)epoch16 << 48) | key;
    unsigned h = hash & mask;
    for (unsigned probe = 0; probe <= mask; ++probe) {
        unsigned long long k = keyw[h];
        if ((unsigned)(k >> 48) != epoch16) {                          // an entry of an earlier frame: free
            const unsigned long long prev = atomicCAS(&keyw[h], k, want);
            if (prev == k) return h;                                   // claimed
            k = prev;                                                  // somebody of THIS frame was faster (a stale entry only ever becomes a
        }                                                              // current one): look at what they wrote
        if (k == want) return h;
        h = (h + 1) & mask;
    }
    return h;
}

// Exclusive scan of in(0..n-1) with a per-element sink.
//   n <= SRL_SCAN_SMALL_MAX: ONE launch without any dependency between workgroups: workgroup b owns elements [1024 b, 1024 b + 1024) and
//   first adds up everything in front of them itself (b independent, coalesced loads per thread).  Redundant work n^2 / 2048 loads --
//   0.3 M for a 24k-point frame, spread over 24 compute units -- against a second launch (hipcub's look-back scan initialises its tile
//   states in a kernel of its own) or a chain of waits (one workgroup walking the array measured 15 us at 24k, this form ~4 us).
//   n <= SRL_SCAN_MAX (round 6: frames beyond 131 072 points -- BASELINE config 4's sweep has 262 144 -- used to leave for hipcub and the
//   host): TWO launches -- k_scan_tile_sums leaves one sum per 1024-element tile, and the same scan kernel adds up the <= 1024 tile sums in
//   front of its tile (one load per thread) instead of the elements: O(n) work, still no dependency between workgroups.
//   larger n (bulk map loads): the tile sums are scanned themselves and the kernel takes its tile's offset from the result (srl_scan).
// sink(i, value, exclusive prefix) is called once per element.
#define SRL_SCAN_SMALL_MAX 131072
#define SRL_SCAN_MAX (1 << 20)
struct SrlNoFin {
    __device__ void operator()(int) const {}
};
template <class In>
__global__ void __launch_bounds__(1024) k_scan_tile_sums(In in, int n, int *sums) {
    __shared__ int wave_part[16];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int i = blockIdx.x * 1024 + t;
    int v = i < n ? in(i) : 0;
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    if (lane == 0) wave_part[w] = v;
    __syncthreads();
    if (t == 0) { int s = 0; for (int k = 0; k < 16; k++) s += wave_part[k]; sums[blockIdx.x] = s; }
}
// fin(inclusive total up to the end of this workgroup's tile) is called by every thread of the workgroup after its sink calls (the last
// workgroup's value is the grand total): a place for "this tile is done" protocols.
// tile_aux / aux_mode: 0 = none (the elements in front are summed), 1 = tile_aux[k] = sum of tile k (k_scan_tile_sums), 2 = tile_aux[k] =
// exclusive prefix of the tile sums (already scanned)
template <class In, class Sink, class Fin = SrlNoFin>
__global__ void __launch_bounds__(1024) k_scan_small(In in, Sink sink, int n, Fin fin = Fin(), const int *tile_aux = nullptr, int aux_mode = 0) {
    __shared__ int wave_part[16], wave_front[16];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int base = blockIdx.x * 1024;
    // everything in front of this workgroup
    int front = 0;
    if (aux_mode == 2) {
        front = t == 0 ? tile_aux[blockIdx.x] : 0;
    } else if (aux_mode == 1) {
        for (int j = t; j < (int)blockIdx.x; j += 1024) front += tile_aux[j];
    } else {
        // eight elements in flight per thread: a trip per element is a chain of base / 1024 load latencies (~1 us each at frame size)
        constexpr int U = 8;
        for (int j0 = t; j0 < base; j0 += U * 1024) {
            int v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { const int j = j0 + u * 1024; v[u] = j < base ? in(j) : 0; }
#pragma unroll
            for (int u = 0; u < U; ++u) front += v[u];
        }
    }
    for (int d = 32; d >= 1; d >>= 1) front += __shfl_xor(front, d);
    // own tile
    const int i = base + t;
    const int v = i < n ? in(i) : 0;
    int incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int up = __shfl_up(incl, d);
        if (lane >= d) incl += up;
    }
    if (lane == 63) wave_part[w] = incl;
    if (lane == 0) wave_front[w] = front;
    __syncthreads();
    int run = 0;
    for (int k = 0; k < 16; k++) run += wave_front[k];
    for (int k = 0; k < w; k++) run += wave_part[k];
    if (i < n) sink(i, v, run + incl - v);
    int tile_end = 0;
    for (int k = 0; k < 16; k++) tile_end += wave_front[k] + wave_part[k];
    fin(tile_end);
}
inline int srl_scan_small_grid(int n) { return (n + 1023) / 1024; }
struct SrlIntArrayIn {
    const int *p;
    __device__ int operator()(int i) const { return p[i]; }
};
struct SrlIntArraySink {
    int *p;
    __device__ void operator()(int i, int, int excl) const { p[i] = excl; }
};
// ints of scratch srl_scan needs for n elements (tile sums, and for n > SRL_SCAN_MAX their scan)
inline size_t srl_scan_scratch_ints(int n) { return n <= SRL_SCAN_SMALL_MAX ? 1 : 2 * (size_t)srl_scan_small_grid(n) + 2; }
// the scan of any size: 1 launch (n <= 131 072), 2 (n <= 1 M), else tile sums -> their scan (recursively) -> the scan with tile offsets
template <class In, class Sink, class Fin = SrlNoFin>
inline void srl_scan(In in, Sink sink, int n, int *scratch, hipStream_t st, Fin fin = Fin()) {
    const int tiles = srl_scan_small_grid(n);
    if (n <= SRL_SCAN_SMALL_MAX) {
        hipLaunchKernelGGL((k_scan_small<In, Sink, Fin>), dim3(tiles), dim3(1024), 0, st, in, sink, n, fin, (const int *)nullptr, 0);
        return;
    }
    hipLaunchKernelGGL((k_scan_tile_sums<In>), dim3(tiles), dim3(1024), 0, st, in, n, scratch);
    if (n <= SRL_SCAN_MAX) {
        hipLaunchKernelGGL((k_scan_small<In, Sink, Fin>), dim3(tiles), dim3(1024), 0, st, in, sink, n, fin, (const int *)scratch, 1);
        return;
    }
    int *offsets = scratch + tiles + 1;
    // (tiles <= 131 072 for n <= 2^27: the sums' own scan is one launch)
    hipLaunchKernelGGL((k_scan_small<SrlIntArrayIn, SrlIntArraySink, SrlNoFin>), dim3(srl_scan_small_grid(tiles)), dim3(1024), 0, st, SrlIntArrayIn{scratch},
                       SrlIntArraySink{offsets}, tiles, SrlNoFin(), (const int *)nullptr, 0);
    hipLaunchKernelGGL((k_scan_small<In, Sink, Fin>), dim3(tiles), dim3(1024), 0, st, in, sink, n, fin, (const int *)offsets, 2);
}

// One STABLE least-significant-digit radix pass over `bits` (<= 9) key bits at `shift`, n <= SRL_SCAN_SMALL_MAX pairs, in ONE launch and
// again without any dependency between workgroups: workgroup b (tile [1024 b, 1024 b + 1024)) histograms ALL keys itself -- what lies in
// front of its tile and the rest, separately -- so it knows where every digit starts and how many equal digits precede its tile; inside
// the tile the rank is (equal digits in earlier waves) + (equal digits in lower lanes), by ballot matching.  n / 1024 LDS atomics per
// thread (24 at a 24k-point frame).  The library's sort of a frame is a block sort + 5 merge launches + 2 helper kernels (35 us of
// device time and 8 launches on a chain whose cost is its launches); two of these passes sort (slot, index) over <= 18 bits.
#define SRL_RADIX_MAX_BITS 9
// per-tile digit histograms for the passes over more than SRL_SCAN_SMALL_MAX keys: tile-major hist[tile][D] (digit_major = 0: a pass's
// workgroup sums columns) or digit-major hist[d][tiles] (digit_major = 1: one scan over the whole matrix gives every (digit, tile) its
// global start)
static __global__ void __launch_bounds__(1024) k_radix_hist(const unsigned *keys_in, int n, unsigned shift, unsigned bits, int *hist, int digit_major) {
    constexpr int DMAX = 1 << SRL_RADIX_MAX_BITS;
    __shared__ int s_h[DMAX];
    const int t = threadIdx.x;
    const int D = 1 << bits, tiles = (int)gridDim.x;
    for (int d = t; d < D; d += 1024) s_h[d] = 0;
    __syncthreads();
    const int i = blockIdx.x * 1024 + t;
    if (i < n) atomicAdd(&s_h[(keys_in[i] >> shift) & ((unsigned)D - 1u)], 1);
    __syncthreads();
    for (int d = t; d < D; d += 1024) hist[digit_major ? (size_t)d * tiles + blockIdx.x : (size_t)blockIdx.x * D + d] = s_h[d];
}
// aux_mode 0: the workgroup histograms all keys itself (n <= SRL_SCAN_SMALL_MAX); 1: aux = hist[tile][D] (k_radix_hist): it sums the columns;
// 2: aux = exclusive scan of the digit-major histogram matrix: aux[d * tiles + b] is where digit d of tile b starts
static __global__ void __launch_bounds__(1024) k_radix_pass(const unsigned *keys_in, const unsigned *vals_in, unsigned *keys_out, unsigned *vals_out, int n,
                                                      unsigned shift, unsigned bits, const int *aux = nullptr, int aux_mode = 0) {
    constexpr int DMAX = 1 << SRL_RADIX_MAX_BITS;
    __shared__ int s_front[DMAX], s_rest[DMAX], s_base[DMAX], s_wsum[16];
    __shared__ int s_wave[16][DMAX];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int D = 1 << bits;
    const unsigned dmask = (unsigned)D - 1u;
    for (int d = t; d < D; d += 1024) { s_front[d] = 0; s_rest[d] = 0; }
    for (int d = t; d < 16 * DMAX; d += 1024) (&s_wave[0][0])[d] = 0;
    __syncthreads();
    const int tiles = (n + 1023) / 1024;
    if (aux_mode == 2) {
        for (int d = t; d < D; d += 1024) s_base[d] = aux[(size_t)d * tiles + blockIdx.x];
    } else {
        if (aux_mode == 1) {
            // column sums of the tile histograms: 1024 / D threads per digit, each over an interleaved share of the tiles, eight loads in flight
            const int parts = 1024 / D, d = t % D, part = t / D;
            if (part < parts) {
                int fr = 0, re = 0;
                constexpr int U = 8;
                for (int k0 = part; k0 < tiles; k0 += U * parts) {
                    int v[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) { const int k = k0 + u * parts; v[u] = k < tiles ? aux[(size_t)k * D + d] : 0; }
#pragma unroll
                    for (int u = 0; u < U; ++u) { const int k = k0 + u * parts; if (k < (int)blockIdx.x) fr += v[u]; else re += v[u]; }
                }
                if (fr) atomicAdd(&s_front[d], fr);
                if (re) atomicAdd(&s_rest[d], re);
            }
        } else {
            // histogram of everything: the tiles in front of this one, then this one and the tiles behind it
            // (eight loads in flight per thread before their atomics: one key per trip is a chain of n / 1024 load latencies, 1 us each)
            constexpr int U = 8;
            for (int k0 = 0; k0 < tiles; k0 += U) {
                unsigned kk[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int j = (k0 + u) * 1024 + t;
                    kk[u] = j < n ? keys_in[j] : 0u;
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int k = k0 + u;
                    if (k * 1024 + t < n) atomicAdd(k < (int)blockIdx.x ? &s_front[(kk[u] >> shift) & dmask] : &s_rest[(kk[u] >> shift) & dmask], 1);
                }
            }
        }
        __syncthreads();
        // s_base[d] = keys with a smaller digit anywhere + keys with digit d in front of this tile
        const int v = t < D ? s_front[t] + s_rest[t] : 0;
        int incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int up = __shfl_up(incl, d);
            if (lane >= d) incl += up;
        }
        if (lane == 63) s_wsum[w] = incl;
        __syncthreads();
        int run = 0;
        for (int k = 0; k < w; k++) run += s_wsum[k];
        if (t < D) s_base[t] = run + incl - v + s_front[t];
    }
    // own tile: stable rank of every key among the tile's keys with the same digit
    const int i = blockIdx.x * 1024 + t;
    const bool valid = i < n;
    const unsigned key = valid ? keys_in[i] : 0u;
    const unsigned dg = (key >> shift) & dmask;
    unsigned long long same = __ballot(valid);
    for (unsigned b = 0; b < bits; ++b) {
        const bool bit = (dg >> b) & 1u;
        const unsigned long long bal = __ballot(bit);
        same &= bit ? bal : ~bal;
    }
    const int below = __popcll(same & ((1ull << lane) - 1ull));
    if (valid && below == 0) s_wave[w][dg] = __popcll(same);
    __syncthreads();
    if (!valid) return;
    int pos = s_base[dg] + below;
    for (int k = 0; k < w; k++) pos += s_wave[k][dg];
    keys_out[pos] = key;
    vals_out[pos] = vals_in ? vals_in[i] : (unsigned)i;
}
// ints of scratch srl_radix_sort_pairs needs for n pairs (the tile histograms of one pass, and for bulk sizes their scan and its scratch)
inline size_t srl_radix_scratch_ints(int n) {
    if (n <= SRL_SCAN_SMALL_MAX) return 1;
    const size_t cells = (size_t)((n + 1023) / 1024) << SRL_RADIX_MAX_BITS;
    return n <= SRL_SCAN_MAX ? cells : 2 * cells + srl_scan_scratch_ints((int)cells);
}
// one stable pass of any size
inline void srl_radix_pass(const unsigned *keys, const unsigned *vals, unsigned *keys_out, unsigned *vals_out, int n, unsigned shift, unsigned bits, int *scratch,
                           hipStream_t st) {
    const dim3 grid((n + 1023) / 1024), block(1024);
    if (n <= SRL_SCAN_SMALL_MAX) {
        hipLaunchKernelGGL(k_radix_pass, grid, block, 0, st, keys, vals, keys_out, vals_out, n, shift, bits, (const int *)nullptr, 0);
    } else if (n <= SRL_SCAN_MAX) {
        hipLaunchKernelGGL(k_radix_hist, grid, block, 0, st, keys, n, shift, bits, scratch, 0);
        hipLaunchKernelGGL(k_radix_pass, grid, block, 0, st, keys, vals, keys_out, vals_out, n, shift, bits, (const int *)scratch, 1);
    } else {
        const int cells = (int)grid.x << bits;
        int *hist = scratch, *start = scratch + ((size_t)grid.x << SRL_RADIX_MAX_BITS), *sc = start + ((size_t)grid.x << SRL_RADIX_MAX_BITS);
        hipLaunchKernelGGL(k_radix_hist, grid, block, 0, st, keys, n, shift, bits, hist, 1);
        srl_scan(SrlIntArrayIn{hist}, SrlIntArraySink{start}, cells, sc, st);
        hipLaunchKernelGGL(k_radix_pass, grid, block, 0, st, keys, vals, keys_out, vals_out, n, shift, bits, (const int *)start, 2);
    }
}
// stable sort of (key, value) pairs by the low `bits` (<= 27) bits of the key: ceil(bits / 9) passes, each one launch (n <= 131 072), two
// (n <= 1 M) or five (bulk).  vals == nullptr: the values are the positions 0..n-1.  tmp_keys / tmp_vals: n words each (untouched by a
// one-pass sort); scratch: srl_radix_scratch_ints(n) ints.
inline void srl_radix_sort_pairs(const unsigned *keys, const unsigned *vals, unsigned *keys_sorted, unsigned *vals_sorted, unsigned *tmp_keys,
                                 unsigned *tmp_vals, int n, unsigned bits, hipStream_t st, int *scratch = nullptr) {
    const unsigned passes = (bits + SRL_RADIX_MAX_BITS - 1) / SRL_RADIX_MAX_BITS;
    const unsigned per = (bits + passes - 1) / passes;
    // ping-pong so that the LAST pass writes the caller's output
    const unsigned *ki = keys, *vi = vals;
    unsigned shift = 0;
    for (unsigned p = 0; p < passes; ++p) {
        const bool to_out = ((passes - 1 - p) % 2) == 0;
        unsigned *ko = to_out ? keys_sorted : tmp_keys, *vo = to_out ? vals_sorted : tmp_vals;
        const unsigned b = p + 1 == passes ? bits - shift : per;
        srl_radix_pass(ki, vi, ko, vo, n, shift, b, scratch, st);
        ki = ko; vi = vo; shift += b;
    }
}
#endif
