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
// 2. k_scan_small: exclusive prefix sum of up to 131072 ints in one launch, with a per-element sink that can do the consumer's work in
//    the same pass (segment starts of the sorted frame).  hipcub's decoupled look-back scan is built for millions of items: at 24k it
//    costs a state-initialisation kernel + the scan kernel, 10-16 us and two launches, three times per frame.
#pragma once
#include <hip/hip_runtime.h>

#include "srl_ctx.h"

#define SRL_KEY48_MASK 0xFFFFFFFFFFFFull

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

// Exclusive scan of in(0..n-1), n <= SRL_SCAN_SMALL_MAX, in ONE launch without any dependency between workgroups: workgroup b owns
// elements [1024 b, 1024 b + 1024) and first adds up everything in front of them itself (b independent, coalesced loads per thread).
// Redundant work n^2 / 2048 loads -- 0.3 M for a 24k-point frame, spread over 24 compute units -- against a second launch (hipcub's
// look-back scan initialises its tile states in a kernel of its own) or a chain of waits (one workgroup walking the array measured
// 15 us at 24k, this form ~4 us).  sink(i, value, exclusive prefix) is called once per element.
#define SRL_SCAN_SMALL_MAX 131072
struct SrlNoFin {
    __device__ void operator()(int) const {}
};
// fin(inclusive total up to the end of this workgroup's tile) is called by every thread of the workgroup after its sink calls (the last
// workgroup's value is the grand total): a place for "this tile is done" protocols
template <class In, class Sink, class Fin = SrlNoFin>
__global__ void __launch_bounds__(1024) k_scan_small(In in, Sink sink, int n, Fin fin = Fin()) {
    __shared__ int wave_part[16], wave_front[16];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int base = blockIdx.x * 1024;
    // everything in front of this workgroup
    int front = 0;
    for (int j = t; j < base; j += 1024) front += in(j);
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
#endif
