// srl_kernels.hip -- hand-written HIP kernels for gfx950 (MI355X, wave64) implementing the
// SR-LIVO LIO scan-matching hot path.  Compiled with -ffp-contract=off: the FP64 distance /
// transform arithmetic must round exactly like the reference's x86-64 (no FMA) build so that the
// neighbour sets are bit-identical.
//
// Kernel map:
//   srl_assoc_kernel<NB, FAST, KPW, WPB>   buildPlaneResiduals (optimize.cpp:18-131), KPW x WPB keypoints per workgroup:
//        phase 0  thread/keypoint : transformKeypoints (optimize.cpp:30-40, :83), voxel key, FP32 prefilter constants
//        phase 1  wave/keypoint pair : searchNeighbors (optimize.cpp:365-426) -- (2r+1)^3 hash probes by lanes (one
//                                   keypoint per half-wave, issued one pair ahead), occupied voxels compacted by
//                                   ballot/mbcnt into LDS, candidates 60 per round (coalesced 12-B loads from the 256-B
//                                   slabs, both keypoints' rounds in flight together), exact top-K by {FP32 distances ->
//                                   13-step ballot bisection on the per-lane minima -> conservative threshold -> ballot
//                                   compaction of the ~25 survivors into LDS -> FP64 rank-by-counting, one finish per
//                                   pair -> tie check -> libstdc++ heap replay when distances tie}
//        phase 2  lane/keypoint   : computeNeighborhoodDistribution (optimize.cpp:316-353), weights,
//                                   plane, signed distance gate, Jacobian (optimize.cpp:42-53,85-105),
//                                   then an in-order H^T H / H^T h partial per workgroup;
//        fused final reduction (last workgroup of the grid), with or without the ordered cut (optimize.cpp:107).
//   srl_reduce_kernel      the sequential early-exit (optimize.cpp:107) as an ordered prefix cut +
//                          the deterministic final reduction of the partials (optimize.cpp:235,239).
//   srl_search_kernel<NB>  searchNeighbors for a batch of world points (parity / API surface).
#include "srl_device.h"
#include "srl_hash.h"
#include "srl_heap.h"

#include <math.h>
#include <type_traits>

namespace {

struct alignas(8) VoxEnt { unsigned slab; unsigned count; };   // 8-B aligned: one ds_read_b64 with an immediate offset per entry
struct Surv { double d2; float x, y, z; unsigned id; };
static_assert(sizeof(Surv) == 24, "survivor record is 24 bytes");
static_assert(SRL_SURV_CAP * 24 + (SRL_MAXK + 1) * 8 <= SRL_WAVE_SCRATCH, "general-path scratch (survivors + sorted d2) must fit");
static_assert(64 * 8 + SRL_MAXK * 16 + SRL_MAXK * 4 <= SRL_WAVE_SCRATCH, "heap replay scratch must fit");

__device__ __forceinline__ int lane_id() {
    return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0));
}
__device__ __forceinline__ int lanes_below(unsigned long long m) {
    return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
}

struct Cand { bool valid; double d2; float x, y, z; unsigned id; };

// candidate e = (compacted voxel e/20, slot e%20); d2 = ((dx*dx + dy*dy) + dz*dz) in FP64, no FMA
__device__ __forceinline__ Cand eval_cand(int e, int nv, const VoxEnt *vox, const unsigned char *slabs,
                                          double qx, double qy, double qz) {
    Cand c;
    c.valid = false;
    c.d2 = __builtin_huge_val();
    c.x = c.y = c.z = 0.0f;
    c.id = 0;
    const unsigned cv = (unsigned)e / SRL_CAP;
    const unsigned slot = (unsigned)e - cv * SRL_CAP;
    if ((int)cv < nv) {
        const VoxEnt ve = vox[cv];
        if (slot < ve.count) {
            const float *p = reinterpret_cast<const float *>(slabs + (size_t)ve.slab * SRL_SLAB_BYTES + slot * 12u);
            c.x = p[0]; c.y = p[1]; c.z = p[2];
            const double dx = (double)c.x - qx;
            const double dy = (double)c.y - qy;
            const double dz = (double)c.z - qz;
            c.d2 = (dx * dx + dy * dy) + dz * dz;
            c.valid = true;
            c.id = ve.slab * SRL_CAP + slot;
        }
    }
    return c;
}

// ---- (2r+1)^3 hash probes: lanes probe, occupied voxels are compacted (visit order kept) ----
// qf_cull (the keypoint's FP32 query, [5] = squared cull radius; null = visit everything) and total_all (the candidates of ALL found
// voxels, P_k of the reference's loop): the fast path of the init mode starts from the previous pass's bounds like the r = 1 path
// (SrlAssocArgs::bound_in, probe_finish) -- 125 probes, ~45 occupied voxels, ~6 of them within reach of the K nearest
template <int NB>
__device__ __forceinline__ int probe_voxels(double qx, double qy, double qz, double size_voxel, int thr_cap,
                                            const SrlMapSlot *table, unsigned mask, VoxEnt *vox, int lane, const float *qf_cull = nullptr,
                                            int *total_all = nullptr) {
    constexpr int SIDE = 2 * NB + 1;
    constexpr int NV = SIDE * SIDE * SIDE;
    asm volatile("" : "+v"(lane));   // keep lane-derived constants of this rarely taken path out of the caller's loop preheader
    // static_cast<short>(point / size_voxel_map): truncation toward zero (optimize.cpp:372-374)
    const short kx = (short)(int)(qx / size_voxel);
    const short ky = (short)(int)(qy / size_voxel);
    const short kz = (short)(int)(qz / size_voxel);
    int nv = 0, tot = 0;
#pragma unroll
    for (int base = 0; base < NV; base += 64) {
        const int i = base + lane;
        bool found = false;
        unsigned slab = 0, cnt = 0;
        float bd2 = 0.0f;
        if (i < NV) {
            // visit order: x outer, y, z inner (optimize.cpp:379-381)
            const int ix = i / (SIDE * SIDE);
            const int iy = (i / SIDE) % SIDE;
            const int iz = i % SIDE;
            const short vx = (short)(kx + (ix - NB));
            const short vy = (short)(ky + (iy - NB));
            const short vz = (short)(kz + (iz - NB));
            const unsigned long long key = srl_pack_key(vx, vy, vz);
            unsigned h = srl_hash_key(key) & mask;
            for (unsigned probe = 0; probe <= mask; ++probe) {
                const SrlMapSlot s = table[h];
                if (s.key == key) {
                    // NumPoints() < threshold_voxel_capacity -> skipped (optimize.cpp:389)
                    found = (int)s.count >= thr_cap && s.count > 0;
                    slab = s.slab;
                    cnt = s.count;
                    break;
                }
                if (s.key == SRL_EMPTY_KEY) break;
                h = (h + 1) & mask;
            }
            if (qf_cull) {
                // distance of the FP32 query to the voxel's box (probe_finish has the derivation), squared
                const int vv[3] = {(int)vx, (int)vy, (int)vz};
                const float sz = (float)size_voxel;
#pragma unroll
                for (int ax = 0; ax < 3; ++ax) {
                    const int v = vv[ax];
                    const float lo = (float)(v > 0 ? v : v - 1) * sz, hi = (float)(v < 0 ? v : v + 1) * sz;
                    const float d = fmaxf(fmaxf(lo - qf_cull[ax], qf_cull[ax] - hi), 0.0f);
                    bd2 += d * d;
                }
            }
        }
        tot += found ? (int)cnt : 0;
        if (qf_cull) found = found && bd2 <= qf_cull[5];
        const unsigned long long m = __ballot(found);
        if (found) {
            VoxEnt ve; ve.slab = slab; ve.count = cnt;
            vox[nv + lanes_below(m)] = ve;
        }
        nv += __popcll(m);
    }
    if (total_all) {
        for (int d = 32; d >= 1; d >>= 1) tot += __shfl_xor(tot, d);
        *total_all = tot;
    }
    __builtin_amdgcn_wave_barrier();
    return nv;
}

// Selection results: the fast paths hand anything they cannot finish to the general path.
enum { SEL_OVERFLOW = 0, SEL_DONE = 1, SEL_TIE = 2 };
// Two squared distances a <= b can only round to the same sqrt (the reference compares norm() = sqrt(d2),
// optimize.cpp:395,398) when b <= a (1 + 2^-50): sqrt(b) - sqrt(a) >= (b - a) / (2 sqrt(b)) and one ulp of sqrt(a) is
// at most 2^-52 sqrt(a).  Anything closer than that is treated as a tie and replayed literally.
#define SRL_NEAR_TIE 0x1.0000000000004p+0

// ---- the reference's literal heap sequence (optimize.cpp:394-404, 411-422; srl_heap.h) for ONE keypoint ----
// All lanes stage 64 candidate distances per round (visit order = candidate index), lane 0 offers them to the
// heap one by one; lane i < size then emits rank i.  Slow (one lane) but exact for every input, ties included.
template <class Sink>
__device__ __forceinline__ void select_topk_replay(double qx, double qy, double qz, int nv, const VoxEnt *vox,
                                                const unsigned char *slabs, int K, void *scratch, int lane, Sink &sink,
                                                int &total_out) {
    double *stage = reinterpret_cast<double *>(scratch);          // [64]
    double *hd = stage + 64;                                      // [32]
    int *he = reinterpret_cast<int *>(hd + SRL_MAXK);             // [32]
    int *out = he + SRL_MAXK;                                     // [32]
    const int rounds = (nv * SRL_CAP + 63) >> 6;
    int size = 0, total = 0;
    __builtin_amdgcn_wave_barrier();
    for (int j = 0; j < rounds; ++j) {
        const Cand c = eval_cand(lane + 64 * j, nv, vox, slabs, qx, qy, qz);
        stage[lane] = c.valid ? sqrt(c.d2) : -1.0;                // distance = (neighbor_point - point).norm()
        total += __popcll(__ballot(c.valid));
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) {
            for (int i = 0; i < 64; ++i) {
                const double d = stage[i];
                if (d >= 0.0) size = srl_heap_offer(hd, he, size, K, d, 64 * j + i);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0) srl_heap_drain(hd, he, size, out);
    __builtin_amdgcn_wave_barrier();
    size = __shfl(size, 0);
    if (lane < size) {
        const Cand c = eval_cand(out[lane], nv, vox, slabs, qx, qy, qz);
        sink.put(lane, c.x, c.y, c.z, c.id);
    }
    total_out = total;
    __builtin_amdgcn_wave_barrier();
}

// ---- exact top-K selection, general path (any r, any number of survivors) ----
// Sink::put(rank, x, y, z, id) is called by exactly one lane per selected rank.  A (near-)tie among the K+1 smallest
// distances hands the keypoint to select_topk_replay.  select_mode: 1 forces the streaming extraction, 5 the replay.
template <class Sink>
__device__ __forceinline__ void select_topk(double qx, double qy, double qz, int nv, const VoxEnt *vox,
                                            const unsigned char *slabs, int K, int select_mode, Surv *surv,
                                            int lane, Sink &sink, int &total_out, int &fallback_out) {
    const int rounds = (nv * SRL_CAP + 63) >> 6;
    const float kInfF = __builtin_huge_valf();
    asm volatile("" : "+v"(lane));   // see probe_voxels: no hoisting of the sort network's lane constants
    if (select_mode == 5) {
        select_topk_replay(qx, qy, qz, nv, vox, slabs, K, surv, lane, sink, total_out);
        fallback_out = 1;
        return;
    }

    // pass 1: stream all candidates, per-lane minimum of (float)d2, count P_k
    float lmin = kInfF;
    int total = 0;
    for (int j = 0; j < rounds; ++j) {
        const Cand c = eval_cand(lane + 64 * j, nv, vox, slabs, qx, qy, qz);
        const float key = c.valid ? (float)c.d2 : kInfF;
        lmin = fminf(lmin, key);
        total += __popcll(__ballot(c.valid));
    }
    total_out = total;
    const int nsel = total < K ? total : K;
    bool use_fallback = (select_mode == 1);
    bool tie = false;
    int c_surv = 0;
    double *sorted = reinterpret_cast<double *>(reinterpret_cast<unsigned char *>(surv) + SRL_SURV_CAP * sizeof(Surv));   // [K + 1]

    if (!use_fallback) {
        // K-th smallest of the 64 per-lane minima = an upper bound of the K-th smallest distance
        // (every lane's minimum is a distinct candidate): 64-lane bitonic sort of the f32 bit patterns.
        unsigned v = __float_as_uint(lmin);
#pragma unroll
        for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
            for (int j = k >> 1; j > 0; j >>= 1) {
                const unsigned o = __shfl_xor(v, j);
                const bool up = (lane & k) == 0;
                const bool lo = (lane & j) == 0;
                const unsigned mn = v < o ? v : o;
                const unsigned mx = v < o ? o : v;
                v = (lo == up) ? mn : mx;
            }
        }
        // (float)d2 rounds to nearest: a candidate just above the K-th distance may round onto tau, one just below the
        // (K+1)-th may round up past it -- the next float up keeps every candidate that can tie with the K-th.
        const unsigned tbits = __shfl(v, K - 1);
        const float tau = __uint_as_float(tbits < 0x7f800000u ? tbits + 1u : tbits);

        // pass 2: survivors {(float)d2 <= tau} are a prefix of the true order that contains the top-K;
        // compact them (visit order kept) into this wave's LDS scratch.
        for (int j = 0; j < rounds; ++j) {
            const Cand c = eval_cand(lane + 64 * j, nv, vox, slabs, qx, qy, qz);
            const bool s = c.valid && ((float)c.d2 <= tau);
            const unsigned long long m = __ballot(s);
            const int pos = c_surv + lanes_below(m);
            if (s && pos < SRL_SURV_CAP) {
                Surv r; r.d2 = c.d2; r.x = c.x; r.y = c.y; r.z = c.z; r.id = c.id;
                surv[pos] = r;
            }
            c_surv += __popcll(m);
        }
        __builtin_amdgcn_wave_barrier();
        if (c_surv > SRL_SURV_CAP) use_fallback = true;
    }

    if (!use_fallback) {
        // exact FP64 rank by counting (a permutation: equal d2 are separated by survivor index), then the tie check on
        // neighbours in sorted order; c_surv <= SRL_SURV_CAP = 64: one survivor per lane
        const int i = lane;
        const bool act = i < c_surv;
        Surv me;
        if (act) me = surv[i];
        else { me.d2 = __builtin_huge_val(); me.x = me.y = me.z = 0.0f; me.id = 0; }
        int rank = 0;
        for (int j = 0; j < c_surv; ++j) {
            const double dj = surv[j].d2;
            rank += ((dj < me.d2) || (dj == me.d2 && j < i)) ? 1 : 0;
        }
        if (act && rank <= K) sorted[rank] = me.d2;
        __builtin_amdgcn_wave_barrier();
        bool bad = false;
        if (act && rank < K && rank + 1 < c_surv) bad = !(sorted[rank + 1] > me.d2 * SRL_NEAR_TIE);
        tie = __ballot(bad) != 0ull;
        if (!tie && act && rank < K) sink.put(rank, me.x, me.y, me.z, me.id);
        fallback_out = 0;
    } else {
        // streaming extraction: nsel (+1 for the tie check at the cut-off) passes, each taking the lexicographic
        // successor of (d2, e)
        double last_d2 = -1.0;
        int last_e = -1;
        const int next = total < K + 1 ? total : K + 1;
        for (int r = 0; r < next; ++r) {
            double bd2 = __builtin_huge_val();
            int be = 0x7fffffff;
            for (int j = 0; j < rounds; ++j) {
                const int e = lane + 64 * j;
                const Cand c = eval_cand(e, nv, vox, slabs, qx, qy, qz);
                const bool gt = c.valid && (c.d2 > last_d2 || (c.d2 == last_d2 && e > last_e));
                if (gt && (c.d2 < bd2 || (c.d2 == bd2 && e < be))) { bd2 = c.d2; be = e; }
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                const double od = __shfl_xor(bd2, off);
                const int oe = __shfl_xor(be, off);
                if (od < bd2 || (od == bd2 && oe < be)) { bd2 = od; be = oe; }
            }
            if (r > 0 && !(bd2 > last_d2 * SRL_NEAR_TIE)) { tie = true; break; }
            if (r < nsel && (be & 63) == lane) {
                const Cand c = eval_cand(be, nv, vox, slabs, qx, qy, qz);
                sink.put(r, c.x, c.y, c.z, c.id);
            }
            last_d2 = bd2;
            last_e = be;
        }
        fallback_out = 1;
    }
    __builtin_amdgcn_wave_barrier();
    if (tie) {
        select_topk_replay(qx, qy, qz, nv, vox, slabs, K, surv, lane, sink, total_out);
        fallback_out = 1;
    }
}

// ---- fast path for r = 1 (<= 27 voxels) -------------------------------------------------------------
// Lane roles are fixed per wave (no integer division in the loop): lane l < 27 of each half-wave probes voxel offset
// (l/9-1, l/3%3-1, l%3-1) [visit order x,y,z]; in candidate round j lane l evaluates slot l%20 of
// compacted voxel 3j + l/20 (60 of 64 lanes busy, <= 9 rounds), so visit order = (round, lane).
struct LaneRole {
    int pdx, pdy, pdz;     // probe offset (half-wave lane < 27)
    int c0, slot;          // candidate role: voxel-in-round (0..2, 3 = idle) and slot (0..19)
};
__device__ __forceinline__ LaneRole lane_role(int lane) {
    LaneRole r;
    const int pl = lane & 31;            // the two half-waves probe for two consecutive keypoints
    const int ix = pl / 9, iy = (pl / 3) % 3, iz = pl % 3;
    r.pdx = ix - 1; r.pdy = iy - 1; r.pdz = iz - 1;
    r.c0 = lane / SRL_CAP;
    r.slot = lane - r.c0 * SRL_CAP;
    return r;
}

// Probe state of one keypoint: key, home slot and the first two table slots (a 2-slot window resolves
// > 99 % of the probes at load <= 0.25).  Issued one keypoint ahead so the L2 round trip of the hash
// lookup overlaps the previous keypoint's selection.
struct ProbeReq {
    unsigned long long key;
    unsigned h;
    SrlMapSlot s0, s1;
};
// kv: the wave's voxel keys (4 ints per keypoint); half-wave h = lane >> 5 probes for keypoint kl + h
__device__ __forceinline__ ProbeReq probe_issue(const int *kv, int kl, const LaneRole &role, const SrlMapSlot *table,
                                                unsigned mask, int lane) {
    ProbeReq r;
    r.key = SRL_EMPTY_KEY; r.h = 0;
    r.s0.key = SRL_EMPTY_KEY; r.s0.slab = 0; r.s0.count = 0;
    r.s1 = r.s0;
    if ((lane & 31) < 27) {
        const int *k = kv + (kl + (lane >> 5)) * 4;
        r.key = srl_pack_key((short)(k[0] + role.pdx), (short)(k[1] + role.pdy), (short)(k[2] + role.pdz));
        r.h = srl_hash_key(r.key) & mask;
        r.s0 = table[r.h];
        r.s1 = table[(r.h + 1) & mask];
    }
    return r;
}
// Compacts the occupied voxels of both keypoints into their lists (list h at vox + 32 h, visit order kept,
// zero-filled up to 27 entries so the candidate rounds need no bounds branch).  Returns nv_A | nv_B << 8.
// cull (wave-uniform): qf_pair / kv_pair = the pair's FP32 queries (8 floats per keypoint, [5] = squared cull radius) and voxel keys (4
// ints per keypoint); a found voxel whose box lies further from the query than the radius keeps its count in P_k (the reference's loop
// visits it) but is left out of the candidate list -- none of its points can be among the K nearest (SrlAssocArgs::bound_in)
__device__ __forceinline__ int probe_finish(const ProbeReq &r, int thr_cap, const SrlMapSlot *table, unsigned mask,
                                            VoxEnt *vox, int lane, int *ncand_pair, bool cull = false, const float *qf_pair = nullptr,
                                            const int *kv_pair = nullptr, const LaneRole *role = nullptr, float size_voxel = 1.0f) {
    const bool prober = (lane & 31) < 27;
    // the 2-slot window by selects (non-probing lanes carry EMPTY keys and match nothing) ...
    const bool m0 = prober && r.s0.key == r.key;
    const bool m1 = prober && !m0 && r.s0.key != SRL_EMPTY_KEY && r.s1.key == r.key;
    bool found = m0 || m1;
    unsigned slab = m0 ? r.s0.slab : r.s1.slab, cnt = m0 ? r.s0.count : r.s1.count;
    // ... and ONE uniform branch for the rare case (< 1 % of the probes at load <= 0.25) that both slots hold other keys
    const bool more = prober && !found && r.s0.key != SRL_EMPTY_KEY && r.s1.key != SRL_EMPTY_KEY;
    if (__ballot(more) != 0ull) {
        if (more) {
            unsigned h = (r.h + 2) & mask;
            for (unsigned probe = 2; probe <= mask; ++probe) {
                const SrlMapSlot sl = table[h];
                if (sl.key == r.key) { slab = sl.slab; cnt = sl.count; found = true; break; }
                if (sl.key == SRL_EMPTY_KEY) break;
                h = (h + 1) & mask;
            }
        }
    }
    found = found && (int)cnt >= thr_cap && cnt > 0;            // NumPoints() < threshold -> skipped (optimize.cpp:389)
    // P_k of the two keypoints (the candidates the reference's loop visits, optimize.cpp:391-404) = the resident points of the
    // voxels found: added up here, once per voxel, instead of by ballots over every candidate round.  Summed across the lanes of
    // each half-wave in registers (DPP row shifts + row broadcasts: the standard wave reduction; lane 31 ends with the low half's
    // sum, lane 63 with the wave's) and filed by ONE store per keypoint -- as an LDS atomic per found voxel, ~12 lanes on one
    // address, it was 80 % of the kernel's LDS bank-conflict cycles (SQ_LDS_ADDR_CONFLICT 11.0 / SQ_LDS_BANK_CONFLICT 12.9 of 16.2
    // cycles per keypoint, tools/ablate_pmc.py).
    {
#if defined(__HIP_DEVICE_COMPILE__)
        int v = found ? (int)cnt : 0;
        v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);      // row_shr:1
        v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);      // row_shr:2
        v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);      // row_shr:4   (inclusive prefix over 1 + 2 + 4 ... lanes of a row)
        v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);      // row_shr:8   -> lane 15 of every row: the row's sum
        v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);      // row_bcast:15 into rows 1 and 3 -> lane 31: rows 0 + 1, lane 63: rows 2 + 3
        if ((lane & 31) == 31) ncand_pair[lane >> 5] = v;
#endif
    }
    if (cull) {
        // box of voxel key v along one axis (keys by truncation toward zero, optimize.cpp:372-374): v > 0: [v, v + 1), v < 0: (v - 1, v],
        // v = 0: (-1, 1), times the voxel size; distance of the FP32 query to it, squared and summed.  The radius carries the slack for
        // everything rounded here (SrlAssocArgs::bound_in, phase 0).
        const int hh = lane >> 5;
        const float *qf = qf_pair + 8 * hh;
        const int *kv = kv_pair + 4 * hh;
        const float r2 = qf[5];
        float bd2 = 0.0f;
        const int pd[3] = {role->pdx, role->pdy, role->pdz};
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            const int v = kv[ax] + pd[ax];
            const float lo = (float)(v > 0 ? v : v - 1) * size_voxel, hi = (float)(v < 0 ? v : v + 1) * size_voxel;
            const float d = fmaxf(fmaxf(lo - qf[ax], qf[ax] - hi), 0.0f);
            bd2 += d * d;
        }
        found = found && bd2 <= r2;
    }
    const unsigned long long m = __ballot(found);
    const unsigned m_lo = (unsigned)m, m_hi = (unsigned)(m >> 32);
    const int nv_a = __popc(m_lo), nv_b = __popc(m_hi);
    if (prober) {
        const bool upper = lane >= 32;
        // rank inside the own half: mbcnt_lo counts the low-half bits below a low lane, mbcnt_hi the high-half bits
        // below a high lane (and nothing for low lanes)
        const unsigned f_lo = found ? m_lo : (~m_lo & 0x7FFFFFFu), f_hi = found ? m_hi : (~m_hi & 0x7FFFFFFu);
        const int below = upper ? (int)__builtin_amdgcn_mbcnt_hi(f_hi, 0u) : (int)__builtin_amdgcn_mbcnt_lo(f_lo, 0u);
        const int nv_own = upper ? nv_b : nv_a;
        VoxEnt ve; ve.slab = found ? slab : 0u; ve.count = found ? cnt : 0u;
        vox[(upper ? 32 : 0) + (found ? below : nv_own + below)] = ve;
    }
    __builtin_amdgcn_wave_barrier();
    return nv_a | (nv_b << 8);
}

// where the selected neighbours go: the workgroup's neighbour planes in LDS (and the parity tap)
struct LdsSink {
    float *col;         // &nbx[0][kl]; planes x | y | z, each K rows of `row` floats
    int row;            // KPW + 1
    int plane;          // K * row
    int *tap_ids;       // global row or null
    __device__ __forceinline__ void put(int rank, float x, float y, float z, unsigned id) {
        float *p = col + rank * row;
        p[0] = x;
        p[plane] = y;
        p[2 * plane] = z;
        if (tap_ids) tap_ids[rank] = (int)id;
    }
};

// ---- fast exact top-K, FP32 prefilter variant (default for r = 1) -----------------------------------
// Only the ~25 survivors of a conservative FP32 threshold are ever evaluated in FP64:
//   pass 1  : coalesced 12-B loads (all rounds in flight), d2f = |p - fl32(q)|^2 in FP32 (FMA allowed: it
//             is a filter, never a result), per-lane minimum
//   tau_f   : bisection (ballot/popcount) -> upper bound of the K-th smallest per-lane minimum
//   margin  : |d2f - d2| <= M for every candidate with d2 <= 2 tau_f (M from the FP32 rounding model below),
//             so the true K nearest all satisfy d2f <= tau_f + 2M: these survivors are compacted (visit
//             order) into LDS as {x, y, z, voxel/slot code}
//   exact   : lane i < c evaluates survivor i's d2 in FP64 with the reference's operation order
//             ((dx*dx + dy*dy) + dz*dz, no FMA), strict rank by counting, clash check, emit from registers.
struct alignas(16) SurvRec { float x, y, z; int code; };
#define SRL_PAIR_MAX_ROUNDS 4          // keypoint pairs with at most this many candidate rounds each are selected with both in flight

__device__ __forceinline__ float d2_f32(float px, float py, float pz, float qx, float qy, float qz) {
#pragma clang fp contract(fast)
    const float dx = px - qx, dy = py - qy, dz = pz - qz;
    return dx * dx + dy * dy + dz * dz;
}

// The FP64 finish shared by the fast paths: the c <= 64 survivors of the FP32 prefilter sit in LDS as {x, y, z, code};
// lane i evaluates survivor i exactly, strict rank by counting, tie check, winners emit.
template <class Sink>
__device__ __forceinline__ int finish_survivors(double qx, double qy, double qz, int c, const VoxEnt *vox, int K, void *scratch,
                                                int lane, Sink &sink, double *tau_out = nullptr) {
    SurvRec *recs = reinterpret_cast<SurvRec *>(__builtin_assume_aligned(scratch, 16));            // [64], 16-B aligned
    double *keys = reinterpret_cast<double *>(reinterpret_cast<unsigned char *>(scratch) + 1024);  // [66]
    double *sorted = keys + 66;                                                                    // [K + 1] d2 by rank, behind the key pads
    __builtin_amdgcn_wave_barrier();
    const bool act = lane < c;
    SurvRec me;
    me.x = me.y = me.z = 0.0f; me.code = 0;
    double my = __builtin_huge_val();
    if (act) {
        me = recs[lane];
        const double dx = (double)me.x - qx;
        const double dy = (double)me.y - qy;
        const double dz = (double)me.z - qz;
        my = (dx * dx + dy * dy) + dz * dz;          // the reference's evaluation order (optimize.cpp:394-395)
    }
    keys[lane] = my;                                  // lanes >= c write +inf: keys[c], keys[c+1] pad an odd count
    // the two +inf key pads and, right behind them, sorted[0..K] = NaN ("nobody holds this rank") in one store
    if (lane < K + 3) keys[64 + lane] = lane < 2 ? __builtin_huge_val() : __builtin_nan("");
    __builtin_amdgcn_wave_barrier();
    int rank = 0;
    if (c <= 32) {
        // the usual case: both half-waves work on the same <= 32 keys -- half h counts the keys [16 h, 16 h + 16) below
        // key (lane & 31), one cross-half add finishes the rank.  Fixed trip count: slots >= c hold +inf and never count.
        const double mine = keys[lane & 31];
        const double2 *kp = reinterpret_cast<const double2 *>(keys + (lane >> 5) * 16);
        int r = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const double2 kk = kp[j];
            r += (kk.x < mine) ? 1 : 0;
            r += (kk.y < mine) ? 1 : 0;
        }
        rank = r + __shfl_xor(r, 32);
    } else {
#pragma unroll 4
        for (int j = 0; j < c; j += 2) {
            const double2 kk = *reinterpret_cast<const double2 *>(keys + j);
            rank += (kk.x < my) ? 1 : 0;
            rank += (kk.y < my) ? 1 : 0;
        }
    }
    // Tie check over the K+1 smallest (ties further out cannot change the selected set or its order).  The reference
    // compares sqrt(d2) (optimize.cpp:395,398), so neighbours in sorted order closer than SRL_NEAR_TIE count as tied.
    // Winners file their d2 under their rank; strict ranks are a permutation unless survivors are exactly equal -- then the
    // rank behind them stays NaN.  Lane i compares ranks i - 1 and i: a NaN on either side fails the comparison too.
    const bool win = act && rank < K;
    if (act && rank <= K) sorted[rank] = my;
    __builtin_amdgcn_wave_barrier();
    bool bad = false;
    if (lane >= 1 && lane <= K && lane < c) bad = !(sorted[lane] > sorted[lane - 1] * SRL_NEAR_TIE);
    if (__ballot(bad)) return SEL_TIE;             // the reference's literal heap sequence decides (select_topk_replay)
    if (tau_out && act && rank == K - 1) *tau_out = my;      // the K-th nearest: the next pass's bound (SrlAssocArgs::bound_out)
    if (win) {
        const VoxEnt ve = vox[me.code >> 5];
        sink.put(rank, me.x, me.y, me.z, ve.slab * SRL_CAP + ((unsigned)me.code & 31u));
    }
    __builtin_amdgcn_wave_barrier();
    return SEL_DONE;
}

// R = compile-time number of candidate rounds (3 voxels each): straight-line code, arrays stay in registers.
// cand_issue puts every round's coalesced 12-B load in flight; cand_select consumes them.  They are separate so that a keypoint
// PAIR can have both keypoints' loads in flight before the first one is worked on (select_pair_f32_r): the second keypoint's
// L2 / MALL round trip then hides behind the first one's bisection and FP64 finish.
template <int R>
__device__ __forceinline__ void cand_issue(const VoxEnt *vox, const unsigned char *slabs, unsigned inf_off, const LaneRole &role,
                                           float (&px)[R], float (&py)[R], float (&pz)[R], int &total_out) {
    // all voxel entries (branch-free LDS reads; the list is zero-filled up to 27 entries), then every round's
    // coalesced 12-B load, all in flight before the first use.
    VoxEnt ve[R];
    const int cbase = role.c0 < 3 ? role.c0 : 0;
#pragma unroll
    for (int j = 0; j < R; ++j) ve[j] = vox[3 * j + cbase];
    // Branch-free loads: a lane without a candidate (slot >= count, zero-filled list tail, idle lanes 60..63) reads the
    // all-inf slab instead, so its FP32 distance is +inf by arithmetic -- no predicate to keep, no exec juggling.
    // Slab byte offsets stay in 32 bits (map < 16.7 M voxels, checked at map growth): scalar base + lane offset.
    const unsigned slot_eff = role.c0 < 3 ? (unsigned)role.slot : 31u;
    const unsigned slot_off = (unsigned)role.slot * 12u;
    unsigned off[R];
    int total = 0;
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const bool h = slot_eff < ve[j].count;
        total += __popcll(__ballot(h));
        off[j] = h ? ve[j].slab * (unsigned)SRL_SLAB_BYTES + slot_off : inf_off;
    }
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const float *p = reinterpret_cast<const float *>(slabs + off[j]);
        px[j] = p[0]; py[j] = p[1]; pz[j] = p[2];
    }
    total_out = total;
}

// FP32 distances of one keypoint's rounds and their per-lane minimum
template <int R>
__device__ __forceinline__ float cand_d2f(const float (&px)[R], const float (&py)[R], const float (&pz)[R], const float *qf, float (&d2f)[R]) {
    const float qxf = qf[0], qyf = qf[1], qzf = qf[2];      // FP32 query, prepared in phase 0
    float lmin = __builtin_huge_valf();
#pragma unroll
    for (int j = 0; j < R; ++j) {
        d2f[j] = d2_f32(px[j], py[j], pz[j], qxf, qyf, qzf);
        lmin = fminf(lmin, d2f[j]);
    }
    return lmin;
}
// FP32 error model: a = abs error of fl32(q) per axis; per-axis difference error <= a + u*|d|; sum of
// squares (3 terms, FMA or not) adds <= 4u relative.  For d2, d2f <= T:  |d2f - d2| <= m(T) with
//   m(T) = 4 a sqrt(T) + 8 u T + 4 a^2   (u = 2^-24), evaluated at T = 2 tau_f + 1e-6 with sqrt(T) replaced by
//   its upper bound (T + 1) / 2 (AM-GM; no transcendental), and doubled: linear in tau_f, coefficients from phase 0.
// Fewer than K candidates: every candidate survives (FLT_MAX; empty lanes hold +inf and never pass).
__device__ __forceinline__ float thr_from_bisection(unsigned lo, const float *qf) {
    const float tau_f = __uint_as_float(lo | 0x3FFFFu);   // >= K candidates have d2f <= tau_f (if that many exist)
    float thr = 3.4028235e38f;
    if (tau_f < __builtin_huge_valf()) thr = qf[3] + qf[4] * tau_f;      // (tau + 2 m(2 tau + 1e-6)) (1 + eps), linear in tau
    return thr;
}
// survivors of all rounds, compacted in visit order into recs[0 ..)
template <int R>
__device__ __forceinline__ void cand_compact(const float (&px)[R], const float (&py)[R], const float (&pz)[R], const float (&d2f)[R], float thr,
                                             const unsigned long long (&svm)[R], const LaneRole &role, SurvRec *recs) {
    const int code0 = (role.c0 << 5) | role.slot;
    int base = 0;
    (void)d2f; (void)thr;
#pragma unroll
    for (int j = 0; j < R; ++j) {
        {
            if (__builtin_amdgcn_inverse_ballot_w64(svm[j])) {   // (a round without survivors runs the store with an empty exec mask)   // this lane's bit of the ballot: the mask goes straight into exec, no second compare
                SurvRec r; r.x = px[j]; r.y = py[j]; r.z = pz[j]; r.code = code0 + ((3 * j) << 5);
                recs[base + lanes_below(svm[j])] = r;
            }
            base += __popcll(svm[j]);
        }
    }
}

template <int R, class Sink>
__device__ __forceinline__ int cand_select(const float (&px)[R], const float (&py)[R], const float (&pz)[R], double qx, double qy, double qz,
                                           const float *qf, const VoxEnt *vox, int K, void *scratch, int lane, const LaneRole &role, Sink &sink,
                                           int ablate) {
    float d2f[R];
    const unsigned v = __float_as_uint(cand_d2f<R>(px, py, pz, qf, d2f));
    unsigned lo = 0;
#pragma unroll
    for (int bit = 30; bit >= 18; --bit) {
        const unsigned trial = lo | (1u << bit);
        const int cnt = __popcll(__ballot(v < trial));
        lo = (cnt < K) ? trial : lo;
    }
    const float thr = thr_from_bisection(lo, qf);
    SurvRec *recs = reinterpret_cast<SurvRec *>(__builtin_assume_aligned(scratch, 16));            // [64], 16-B aligned
    unsigned long long svm[R];
    int c = 0;
#pragma unroll
    for (int j = 0; j < R; ++j) {
        svm[j] = __ballot(d2f[j] <= thr);
        c += __popcll(svm[j]);
    }
    if (c > 64) return SEL_OVERFLOW;       // checked before anything is written: the stores below need no clamp
    cand_compact<R>(px, py, pz, d2f, thr, svm, role, recs);
    if (ablate & 2) return SEL_DONE;
    return finish_survivors(qx, qy, qz, c, vox, K, scratch, lane, sink, reinterpret_cast<double *>(const_cast<float *>(qf) + 6));
}

template <int R, class Sink>
__device__ __forceinline__ int select_topk_f32_r(double qx, double qy, double qz, const float *qf, int nv, const VoxEnt *vox,
                                                  const unsigned char *slabs, unsigned inf_off, int K, void *scratch, int lane,
                                                  const LaneRole &role, Sink &sink, int &total_out, int ablate) {
    (void)nv;
    float px[R], py[R], pz[R];
    cand_issue<R>(vox, slabs, inf_off, role, px, py, pz, total_out);
    return cand_select<R>(px, py, pz, qx, qy, qz, qf, vox, K, scratch, lane, role, sink, ablate);
}

// FP64 finish of a keypoint PAIR with at most 32 survivors each (the usual case: ~25): half-wave h works on keypoint h --
// lane (h, i) evaluates survivor i of its keypoint exactly and counts the 32 keys of its own keypoint below it.  Same
// arithmetic, ranks and tie rule as finish_survivors; the fixed part (LDS round trips, rank filing, tie check, emission)
// is paid once per pair instead of once per keypoint.  Scratch: recs[64] (A: 0..31, B: 32..63) | keys[64] | sorted[64].
// Needs K <= 31 (rank K is filed too).  Returns done_a | done_b << 8.
__device__ __forceinline__ int finish_pair(const double *qa, const double *qb, int ca, int cb, const VoxEnt *vox, int K, void *scratch, int lane,
                                           const LdsSink &sink_a, const LdsSink &sink_b, const float *qfa, const float *qfb) {
    SurvRec *recs = reinterpret_cast<SurvRec *>(__builtin_assume_aligned(scratch, 16));
    double *keys = reinterpret_cast<double *>(reinterpret_cast<unsigned char *>(scratch) + 1024);
    double *sorted = keys + 64;
    __builtin_amdgcn_wave_barrier();
    const int h = lane >> 5, i = lane & 31;
    const int c = h ? cb : ca;
    const bool act = i < c;
    const double *q = h ? qb : qa;
    SurvRec me;
    me.x = me.y = me.z = 0.0f; me.code = 0;
    double my = __builtin_huge_val();
    if (act) {
        me = recs[lane];
        const double dx = (double)me.x - q[0];
        const double dy = (double)me.y - q[1];
        const double dz = (double)me.z - q[2];
        my = (dx * dx + dy * dy) + dz * dz;          // the reference's evaluation order (optimize.cpp:394-395)
    }
    keys[lane] = my;                                  // slots >= c hold +inf and never count
    sorted[lane] = __builtin_nan("");                 // "nobody holds this rank", both keypoints
    __builtin_amdgcn_wave_barrier();
    const double2 *kp = reinterpret_cast<const double2 *>(keys + 32 * h);
    int rank = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const double2 kk = kp[j];
        rank += (kk.x < my) ? 1 : 0;
        rank += (kk.y < my) ? 1 : 0;
    }
    const bool win = act && rank < K;
    if (act && rank <= K) sorted[32 * h + rank] = my;
    __builtin_amdgcn_wave_barrier();
    bool bad = false;
    if (i >= 1 && i <= K && i < c) bad = !(sorted[32 * h + i] > sorted[32 * h + i - 1] * SRL_NEAR_TIE);
    const unsigned long long bm = __ballot(bad);
    const bool tie_a = (unsigned)bm != 0u, tie_b = (unsigned)(bm >> 32) != 0u;
    if (win && !(h ? tie_b : tie_a)) {
        const VoxEnt ve = vox[32 * h + (me.code >> 5)];
        LdsSink s = sink_a;
        if (h) { s.col = sink_b.col; s.tap_ids = sink_b.tap_ids; }
        s.put(rank, me.x, me.y, me.z, ve.slab * SRL_CAP + ((unsigned)me.code & 31u));
        // the K-th nearest: the next pass's bound (SrlAssocArgs::bound_out); qf[6..7] of the keypoint, a double
        if (rank == K - 1) *reinterpret_cast<double *>(const_cast<float *>(h ? qfb : qfa) + 6) = my;
    }
    __builtin_amdgcn_wave_barrier();
    return (tie_a ? SEL_TIE : SEL_DONE) | ((tie_b ? SEL_TIE : SEL_DONE) << 8);
}

// Both keypoints of a pair with R rounds each (R = the larger of the two round counts: the shorter list is zero-filled, its
// extra rounds read the all-inf slab).  The two keypoints' loads are in flight together, their bisections run as two
// independent dependency chains in the same instruction stream, and -- when both have at most 32 survivors -- one FP64
// finish serves both (finish_pair).  Returns done_a | done_b << 8.
template <int R>
__device__ __forceinline__ int select_pair_f32_r(const double *qa, const double *qb, const float *qfa, const float *qfb, const VoxEnt *vox,
                                                 const unsigned char *slabs, unsigned inf_off, int K, void *scratch, int lane,
                                                 const LaneRole &role, LdsSink &sink_a, LdsSink &sink_b, int &total_a, int &total_b, int ablate) {
    float ax[R], ay[R], az[R], bx[R], by[R], bz[R];
    cand_issue<R>(vox, slabs, inf_off, role, ax, ay, az, total_a);
    cand_issue<R>(vox + 32, slabs, inf_off, role, bx, by, bz, total_b);
    float da[R], db[R];
    const unsigned va = __float_as_uint(cand_d2f<R>(ax, ay, az, qfa, da));
    const unsigned vb = __float_as_uint(cand_d2f<R>(bx, by, bz, qfb, db));
    // Bisection on the bit pattern of the per-lane minima.  Every pattern in [2^-15, 2) m^2 starts 0111 in bits 30..27, so with
    // the minima clamped into that band nine steps (bits 26..18) do what thirteen did -- the same lo whenever the K-th smallest
    // minimum lies inside the band (a minimum below it only ever makes the bound looser, never wrong).  If the K-th smallest
    // is not below 2 m^2 (lo ends on the band's last pattern: sparse voxels, fewer than K candidates) the full search runs.
    constexpr unsigned BAND_LO = 0x38000000u, BAND_HI = 0x40000000u, BAND_LAST = 0x3FFC0000u;
    const unsigned ca_v = va < BAND_LO ? BAND_LO : (va > BAND_HI ? BAND_HI : va);
    const unsigned cb_v = vb < BAND_LO ? BAND_LO : (vb > BAND_HI ? BAND_HI : vb);
    unsigned lo_a = BAND_LO, lo_b = BAND_LO;
#pragma unroll
    for (int bit = 26; bit >= 18; --bit) {
        const unsigned ta = lo_a | (1u << bit), tb = lo_b | (1u << bit);
        const int cnt_a = __popcll(__ballot(ca_v < ta));
        const int cnt_b = __popcll(__ballot(cb_v < tb));
        lo_a = (cnt_a < K) ? ta : lo_a;
        lo_b = (cnt_b < K) ? tb : lo_b;
    }
    if (lo_a == BAND_LAST || lo_b == BAND_LAST) {            // rare: the original 13 steps, rolled
        lo_a = 0; lo_b = 0;
#pragma nounroll
        for (int bit = 30; bit >= 18; --bit) {
            const unsigned ta = lo_a | (1u << bit), tb = lo_b | (1u << bit);
            const int cnt_a = __popcll(__ballot(va < ta));
            const int cnt_b = __popcll(__ballot(vb < tb));
            lo_a = (cnt_a < K) ? ta : lo_a;
            lo_b = (cnt_b < K) ? tb : lo_b;
        }
    }
    const float thr_a = thr_from_bisection(lo_a, qfa), thr_b = thr_from_bisection(lo_b, qfb);
    unsigned long long sa[R], sb[R];
    int ca = 0, cb = 0;
#pragma unroll
    for (int j = 0; j < R; ++j) {
        sa[j] = __ballot(da[j] <= thr_a);
        sb[j] = __ballot(db[j] <= thr_b);
        ca += __popcll(sa[j]);
        cb += __popcll(sb[j]);
    }
    SurvRec *recs = reinterpret_cast<SurvRec *>(__builtin_assume_aligned(scratch, 16));
    if (ca <= 32 && cb <= 32 && K <= 31 && !(ablate & (2 | 512))) {
        cand_compact<R>(ax, ay, az, da, thr_a, sa, role, recs);
        cand_compact<R>(bx, by, bz, db, thr_b, sb, role, recs + 32);
        return finish_pair(qa, qb, ca, cb, vox, K, scratch, lane, sink_a, sink_b, qfa, qfb);
    }
    // one after the other through the single-keypoint finish (more than 32 survivors, K = 32, or a debug switch)
    int done_a = SEL_OVERFLOW, done_b = SEL_OVERFLOW;
    if (ca <= 64) {
        cand_compact<R>(ax, ay, az, da, thr_a, sa, role, recs);
        done_a = (ablate & 2) ? SEL_DONE : finish_survivors(qa[0], qa[1], qa[2], ca, vox, K, scratch, lane, sink_a, reinterpret_cast<double *>(const_cast<float *>(qfa) + 6));
    }
    __builtin_amdgcn_wave_barrier();
    if (cb <= 64) {
        cand_compact<R>(bx, by, bz, db, thr_b, sb, role, recs);
        done_b = (ablate & 2) ? SEL_DONE : finish_survivors(qb[0], qb[1], qb[2], cb, vox + 32, K, scratch, lane, sink_b, reinterpret_cast<double *>(const_cast<float *>(qfb) + 6));
    }
    return done_a | (done_b << 8);
}

template <class Sink>
__device__ __forceinline__ int select_topk_f32(double qx, double qy, double qz, const float *qf, int nv, const VoxEnt *vox,
                                                const unsigned char *slabs, unsigned inf_off, int K, void *scratch, int lane,
                                                const LaneRole &role, Sink &sink, int &total_out, int ablate) {
    // straight-line instantiation per number of candidate rounds (3 voxels per round)
    switch ((nv + 2) / 3) {
        case 0: case 1: case 2: case 3:
            return select_topk_f32_r<3>(qx, qy, qz, qf, nv, vox, slabs, inf_off, K, scratch, lane, role, sink, total_out, ablate);
        case 4: return select_topk_f32_r<4>(qx, qy, qz, qf, nv, vox, slabs, inf_off, K, scratch, lane, role, sink, total_out, ablate);
        case 5: return select_topk_f32_r<5>(qx, qy, qz, qf, nv, vox, slabs, inf_off, K, scratch, lane, role, sink, total_out, ablate);
        case 6: return select_topk_f32_r<6>(qx, qy, qz, qf, nv, vox, slabs, inf_off, K, scratch, lane, role, sink, total_out, ablate);
        case 7: return select_topk_f32_r<7>(qx, qy, qz, qf, nv, vox, slabs, inf_off, K, scratch, lane, role, sink, total_out, ablate);
        default: return select_topk_f32_r<9>(qx, qy, qz, qf, nv, vox, slabs, inf_off, K, scratch, lane, role, sink, total_out, ablate);
    }
}

// r = 2 (init mode, 125 voxels): the same FP32 prefilter + FP64 finish with the candidate rounds in a loop.  Up to 42
// rounds do not fit in registers, so the FP32 distances are evaluated twice (threshold pass, survivor pass); the second
// pass hits L1/L2.  The voxel list is the visit-order list of probe_voxels<2>, zero-filled to a multiple of 3 entries.
template <class Sink>
__device__ __forceinline__ int select_topk_f32_loop(double qx, double qy, double qz, const float *qf, int nv, const VoxEnt *vox,
                                                    const unsigned char *slabs, unsigned inf_off, int K, void *scratch, int lane,
                                                    const LaneRole &role, Sink &sink, int &total_out) {
    const float kInfF = __builtin_huge_valf();
    const float qxf = qf[0], qyf = qf[1], qzf = qf[2];
    const int rounds = (nv + 2) / 3;
    const int cbase = role.c0 < 3 ? role.c0 : 0;
    const unsigned slot_eff = role.c0 < 3 ? (unsigned)role.slot : 31u;
    const unsigned slot_off = (unsigned)role.slot * 12u;
    float lmin = kInfF;
    int total = 0;
    constexpr int CH = 4;                 // rounds per chunk: their loads are all in flight before the first use
    // byte offset of this lane's candidate in round j (the all-inf slab when it has none, or when j is past the list)
    auto cand_off = [&](int j, bool &has) {
        const int jj = j < rounds ? j : rounds - 1;                       // wave-uniform clamp: never read past the zero fill
        const VoxEnt ve = vox[3 * jj + cbase];
        has = (j < rounds) && (slot_eff < ve.count);
        return has ? ve.slab * (unsigned)SRL_SLAB_BYTES + slot_off : inf_off;
    };
    for (int j0 = 0; j0 < rounds; j0 += CH) {
        unsigned off[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) { bool h; off[u] = cand_off(j0 + u, h); total += __popcll(__ballot(h)); }
        float x[CH], y[CH], z[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) { const float *p = reinterpret_cast<const float *>(slabs + off[u]); x[u] = p[0]; y[u] = p[1]; z[u] = p[2]; }
#pragma unroll
        for (int u = 0; u < CH; ++u) lmin = fminf(lmin, d2_f32(x[u], y[u], z[u], qxf, qyf, qzf));
    }
    total_out = total;
    const unsigned v = __float_as_uint(lmin);
    unsigned lo = 0;
#pragma unroll
    for (int bit = 30; bit >= 18; --bit) {
        const unsigned trial = lo | (1u << bit);
        const int cnt = __popcll(__ballot(v < trial));
        lo = (cnt < K) ? trial : lo;
    }
    const float tau_f = __uint_as_float(lo | 0x3FFFFu);
    float thr = 3.4028235e38f;
    if (tau_f < kInfF) thr = qf[3] + qf[4] * tau_f;      // same margin as select_topk_f32_r

    SurvRec *recs = reinterpret_cast<SurvRec *>(__builtin_assume_aligned(scratch, 16));
    const int code0 = (role.c0 << 5) | role.slot;
    int c = 0;
    for (int j0 = 0; j0 < rounds; j0 += CH) {
        unsigned off[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) { bool h; off[u] = cand_off(j0 + u, h); }
        float x[CH], y[CH], z[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) { const float *p = reinterpret_cast<const float *>(slabs + off[u]); x[u] = p[0]; y[u] = p[1]; z[u] = p[2]; }
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const bool sv = d2_f32(x[u], y[u], z[u], qxf, qyf, qzf) <= thr;
            const unsigned long long m = __ballot(sv);
            if (m != 0ull) {
                const int pos = c + lanes_below(m);
                if (sv && pos < 64) { SurvRec r; r.x = x[u]; r.y = y[u]; r.z = z[u]; r.code = code0 + ((3 * (j0 + u)) << 5); recs[pos] = r; }
                c += __popcll(m);
            }
        }
    }
    if (c > 64) return SEL_OVERFLOW;
    return finish_survivors(qx, qy, qz, c, vox, K, scratch, lane, sink, reinterpret_cast<double *>(const_cast<float *>(qf) + 6));
}

// ---------------------------------------------------------------------------------------------
// small FP64 helpers (fixed evaluation order, mirrors the oracle / Eigen semantics)
// ---------------------------------------------------------------------------------------------
struct D3 { double x, y, z; };
__device__ __forceinline__ D3 d3(double x, double y, double z) { D3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ D3 sub(D3 a, D3 b) { return d3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ double dot3(D3 a, D3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ D3 matvec(const double *M, D3 v) {
    return d3((M[0] * v.x + M[1] * v.y) + M[2] * v.z, (M[3] * v.x + M[4] * v.y) + M[5] * v.z,
              (M[6] * v.x + M[7] * v.y) + M[8] * v.z);
}
__device__ __forceinline__ D3 add(D3 a, D3 b) { return d3(a.x + b.x, a.y + b.y, a.z + b.z); }
// 1 / b for the tolerance-bound arithmetic of phase 2 (plane fit, weights): hardware reciprocal + two Newton steps
// (error below one ulp of the result) instead of the ~12-instruction IEEE division; never used where bits decide
// (voxel keys, candidate distances).
__device__ __forceinline__ double rcp_nr(double b) {
    double r = __builtin_amdgcn_rcp(b);
    r = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ D3 normalized3_fast(D3 a) {
    const double z = dot3(a, a);
    if (z > 0.0) { const double inv = rcp_nr(sqrt(z)); return d3(a.x * inv, a.y * inv, a.z * inv); }
    return a;
}
__device__ __forceinline__ D3 normalized3(D3 a) {
    const double z = dot3(a, a);
    if (z > 0.0) { const double n = sqrt(z); return d3(a.x / n, a.y / n, a.z / n); }
    return a;
}

// FP64 cyclic Jacobi for a symmetric 3x3 (SelfAdjointEigenSolver<Matrix3d> restated, optimize.cpp:339):
// eigenvalues ascending in ev[], eigenvector of the smallest eigenvalue in n0.
__device__ void eig3_jacobi(const double Ain[3][3], double ev[3], D3 &n0) {
    double a[3][3], V[3][3];
    double scale = 0.0;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) scale = fmax(scale, fabs(Ain[i][j]));
    if (!(scale > 0.0)) scale = 1.0;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) { a[i][j] = Ain[i][j] / scale; V[i][j] = (i == j) ? 1.0 : 0.0; }
    for (int sweep = 0; sweep < 32; sweep++) {
        const double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
        if (off < 1e-40) break;
#pragma unroll
        for (int p = 0; p < 2; p++) {
#pragma unroll
            for (int q = p + 1; q < 3; q++) {
                const double apq = a[p][q];
                if (apq == 0.0) continue;
                const double theta = (a[q][q] - a[p][p]) / (2.0 * apq);
                double t = 1.0 / (fabs(theta) + sqrt(theta * theta + 1.0));
                if (theta < 0.0) t = -t;
                const double c = 1.0 / sqrt(t * t + 1.0);
                const double s = t * c;
                const int r = 3 - p - q;
                const double app = a[p][p], aqq = a[q][q];
                a[p][p] = app - t * apq;
                a[q][q] = aqq + t * apq;
                a[p][q] = a[q][p] = 0.0;
                const double arp = a[r][p], arq = a[r][q];
                a[r][p] = a[p][r] = c * arp - s * arq;
                a[r][q] = a[q][r] = s * arp + c * arq;
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - s * vkq;
                    V[k][q] = s * vkp + c * vkq;
                }
            }
        }
    }
    const double d0 = a[0][0] * scale, d1 = a[1][1] * scale, d2 = a[2][2] * scale;
    // stable ascending order of three values (same comparison sequence as the oracle)
    int i0 = 0, i1 = 1, i2 = 2;
    double e0 = d0, e1 = d1, e2 = d2;
    if (e1 < e0) { double tdv = e0; e0 = e1; e1 = tdv; int ti = i0; i0 = i1; i1 = ti; }
    if (e2 < e1) { double tdv = e1; e1 = e2; e2 = tdv; int ti = i1; i1 = i2; i2 = ti; }
    if (e1 < e0) { double tdv = e0; e0 = e1; e1 = tdv; int ti = i0; i0 = i1; i1 = ti; }
    ev[0] = e0; ev[1] = e1; ev[2] = e2;
    // column i0 of V without runtime register indexing
    n0.x = (i0 == 0) ? V[0][0] : ((i0 == 1) ? V[0][1] : V[0][2]);
    n0.y = (i0 == 0) ? V[1][0] : ((i0 == 1) ? V[1][1] : V[1][2]);
    n0.z = (i0 == 0) ? V[2][0] : ((i0 == 1) ? V[2][1] : V[2][2]);
    (void)i2;
}

// Closed-form symmetric 3x3 eigen-decomposition (eigenvalues from the depressed cubic + cross-product eigenvector
// of the smallest one): ~5x fewer FP64 instructions than the Jacobi sweeps above and no data-dependent
// loop.  Same outputs (eigenvalues ascending, unit eigenvector of the smallest) to ~1e-13 relative for
// the well-separated, near-planar neighbourhoods the path weights up; used by the fused kernel, the
// Jacobi version stays as the reference form (selected with select_mode 4, and used by the oracle).
__device__ __forceinline__ void eig3_closed(const double A[3][3], double ev[3], D3 &n0) {
#pragma clang fp contract(fast)      // tolerance-bound arithmetic: let the products fuse
    const double a00 = A[0][0], a11 = A[1][1], a22 = A[2][2], a01 = A[0][1], a02 = A[0][2], a12 = A[1][2];
    const double p1 = a01 * a01 + a02 * a02 + a12 * a12;
    const double q = (a00 + a11 + a22) * 0.33333333333333333;
    const double b00 = a00 - q, b11 = a11 - q, b22 = a22 - q;
    const double p2 = b00 * b00 + b11 * b11 + b22 * b22 + 2.0 * p1;
    if (!(p2 > 0.0)) {               // multiple of the identity (or zero)
        ev[0] = ev[1] = ev[2] = q;
        n0 = d3(1.0, 0.0, 0.0);
        return;
    }
    const double p = sqrt(p2 * 0.16666666666666667);
    const double ip = rcp_nr(p);
    const double c00 = b00 * ip, c11 = b11 * ip, c22 = b22 * ip, c01 = a01 * ip, c02 = a02 * ip, c12 = a12 * ip;
    double r = 0.5 * (c00 * (c11 * c22 - c12 * c12) - c01 * (c01 * c22 - c12 * c02) + c02 * (c01 * c12 - c11 * c02));
    r = fmin(1.0, fmax(-1.0, r));
    // Roots of 4 x^3 - 3 x = r (x = cos of the three angles of the trigonometric form) without acos / cos:
    // Newton from -1 on the root y in [-1, -sqrt(3)/2] of 4 y^3 - 3 y = -|r| -- a simple root with f' >= 6 there,
    // monotone convergence (f concave increasing), error 0.13 -> 4e-2 -> 3e-3 -> 1e-5 -> 4e-10 -> < 1e-16; the
    // hardware reciprocal is enough for the step.  By symmetry y is the smallest root for r <= 0 and minus the largest
    // root for r > 0; the quadratic gives the other two.
    const double tneg = -fabs(r);
    double y = -1.0;
#pragma unroll
    for (int it = 0; it < 6; ++it) {
        const double y2 = y * y;
        const double f = __builtin_fma(y, __builtin_fma(4.0, y2, -3.0), -tneg);
        const double fp = __builtin_fma(12.0, y2, -3.0);
        y = __builtin_fma(-f, __builtin_amdgcn_rcp(fp), y);
    }
    const double rho = (r <= 0.0) ? y : -y;                                  // a root of the cubic in x
    const double sq = sqrt(fmax(0.0, 3.0 * __builtin_fma(-rho, rho, 1.0)));  // the other two: (-rho +- sq) / 2
    const double xa = 0.5 * (-rho - sq), xb = 0.5 * (-rho + sq);
    const double x_small = (r <= 0.0) ? rho : xa;
    const double x_large = (r <= 0.0) ? xb : rho;
    const double e2 = q + 2.0 * p * x_large;                                 // largest
    const double e0 = q + 2.0 * p * x_small;                                 // smallest
    const double e1 = 3.0 * q - e0 - e2;
    ev[0] = e0; ev[1] = e1; ev[2] = e2;
    // null vector of (A - e0 I): the largest of the three row cross products
    const D3 r0 = d3(a00 - e0, a01, a02), r1 = d3(a01, a11 - e0, a12), r2 = d3(a02, a12, a22 - e0);
    const D3 x01 = d3(r0.y * r1.z - r0.z * r1.y, r0.z * r1.x - r0.x * r1.z, r0.x * r1.y - r0.y * r1.x);
    const D3 x02 = d3(r0.y * r2.z - r0.z * r2.y, r0.z * r2.x - r0.x * r2.z, r0.x * r2.y - r0.y * r2.x);
    const D3 x12 = d3(r1.y * r2.z - r1.z * r2.y, r1.z * r2.x - r1.x * r2.z, r1.x * r2.y - r1.y * r2.x);
    const double n01 = dot3(x01, x01), n02 = dot3(x02, x02), n12 = dot3(x12, x12);
    D3 best = x01;
    double nb = n01;
    if (n02 > nb) { best = x02; nb = n02; }
    if (n12 > nb) { best = x12; nb = n12; }
    if (!(nb > 0.0)) {
        // A - e0 I has rank <= 1 (collinear neighbours, or a multiple of the identity): the null space is a plane and the
        // reference's normal is whatever Eigen's rotations leave in column 0.  Return a unit vector of that plane: orthogonal
        // to the largest row, built with the axis that row is least aligned with ((1,0,0) when every row vanishes).
        const double q0 = dot3(r0, r0), q1 = dot3(r1, r1), q2 = dot3(r2, r2);
        D3 r = r0;
        double qr = q0;
        if (q1 > qr) { r = r1; qr = q1; }
        if (q2 > qr) { r = r2; qr = q2; }
        if (!(qr > 0.0)) { n0 = d3(1.0, 0.0, 0.0); return; }
        const double ax = fabs(r.x), ay = fabs(r.y), az = fabs(r.z);
        D3 c;
        if (ax <= ay && ax <= az) c = d3(0.0, r.z, -r.y);            // r x e_x
        else if (ay <= az) c = d3(-r.z, 0.0, r.x);                   // r x e_y
        else c = d3(r.y, -r.x, 0.0);                                 // r x e_z
        const double ic = rcp_nr(sqrt(dot3(c, c)));
        n0 = d3(c.x * ic, c.y * ic, c.z * ic);
        return;
    }
    const double inv = rcp_nr(sqrt(nb));
    n0 = d3(best.x * inv, best.y * inv, best.z * inv);
}

// dynamic LDS carve (all offsets multiples of 16; guide G17).  Every wave owns its 16 keypoints from
// transform to partial normal equations, so all areas below are PER WAVE (no block barrier until the end).
// keypoints per wave KPW in {4, 8, 16} (template parameter of the kernel): 16 amortises the per-wave phases best and is
// used for large sweeps; small sweeps take fewer keypoints per wave so that the grid still fills the chip and the serial
// chain of a wave is short (latency, not throughput, is what a 3k-keypoint sweep pays for).
struct LdsLayout {
    // workgroup-level keypoint arrays (KPB = WPB x KPW keypoints), then one region per wave, then the block tail
    int nb_row;                    // row stride (floats) of the neighbour planes
    int off_nb, off_pw, off_pimu, off_qf, off_kv, off_nfound, off_ncand, off_next, off_defer;
    int off_wave, wave_bytes, off_vox, off_scratch;     // per-wave: off_wave + w * wave_bytes + {off_vox, off_scratch}
    int off_wpart, off_winfo, total;
    int off_pose;                  // armed launches only: the pose of this pass + the control word
};
__host__ __device__ inline LdsLayout lds_layout(int K, int nb_voxels, int kpw, int wpb, int armed = 0) {
    const int kpb = wpb * kpw;
    LdsLayout L;
    // stride = 17 (mod 32): the K winner lanes of one keypoint and the (keypoint, sub-lane) readers of phase 2 spread over the banks
    L.nb_row = ((kpb + 31) & ~31) + 17;
    auto up16 = [](int x) { return (x + 15) & ~15; };
    int o = 0;
    L.off_nb = o;      o += up16(3 * K * L.nb_row * 4);        // planes x | y | z, K rows of nb_row floats
    L.off_pw = o;      o += up16(kpb * 3 * 8);
    L.off_pimu = o;    o += up16(kpb * 3 * 8);
    L.off_qf = o;      o += up16(kpb * 8 * 4);                 // per keypoint: FP32 query (3), prefilter margin coefficients (2), pad
    L.off_kv = o;      o += up16((kpb + 2) * 4 * 4);           // + zero entries behind the last pair
    L.off_nfound = o;  o += up16(kpb * 4);
    L.off_ncand = o;   o += up16(kpb * 4);
    L.off_next = o;    o += 16;                                // workgroup counters: next pair, deferred keypoints, next deferred
    L.off_defer = o;   o += up16(kpb * 2);                     // keypoints the fast path handed on (index | tie flag << 15)
    L.off_wave = o;
    int w = 0;
    L.off_vox = w;     w += (nb_voxels == 1 ? 64 : 128) * 8;   // r = 1: two 32-entry lists (a keypoint pair is probed at once)
    L.off_scratch = w; w += SRL_WAVE_SCRATCH;
    L.wave_bytes = w;
    o += wpb * w;
    L.off_wpart = o;   o += wpb * 32 * 8;
    L.off_winfo = o;   o += wpb * 8 * 4;
    L.off_pose = o;
    if (armed) o += up16((SRL_POSE_DOUBLES + 2) * 8);          // Rn[9] R[9] t[3] as they arrive through the pose box, + the control word
    L.total = o;
    return L;
}


// FAST: 0 = general path only, 1 = FP32-prefilter fast path (r = 1: keypoint pairs, rounds in registers; r = 2: looped)
// WPB = waves per workgroup: 16 (one workgroup per CU; every wave of a SIMD belongs to it) or 4 (when the 16-wave LDS
// footprint does not fit: K > 24).  With four independent 4-wave workgroups per CU the SIMD arbiter's age priority let
// the oldest wave of every SIMD finish its 16 keypoints in 34 us and the youngest in 47 us; with 16-wave workgroups the
// pair counter of phase 1 evens that out (workgroups 45..48 us).  The kernel itself is not shorter for it -- a CU is busy
// (instruction issue + LDS) for the same ~47 us either way, tools/block_times.py -- but the reduce kernel sums 256
// partials instead of 1 024 and a short sweep needs fewer workgroups to fill the chip.
// karg = the kernarg segment (the kernel's __builtin_amdgcn_kernarg_segment_ptr(): inside a called function the builtin
// yields a null pointer, so the persistent solve hands it down).
#if defined(__HIP_DEVICE_COMPILE__)
typedef const __attribute__((address_space(4))) char *KargBytes;
#else
typedef const char *KargBytes;
#endif
// one 64-bit word as two tagged granules {tag, low half} {tag, high half} in ONE 16-byte system-scope store: the lanes of a wave
// write consecutive 16-byte chunks, i.e. whole 64-byte lines (52 lanes = 13 lines) -- as 8-byte stores at stride 16 the same bytes
// crossed PCIe as 104 partial-line writes and reached the host 10-16 us later (measured, tools/arm_timeline.py)
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ void store_granule_pair(unsigned long long *dst, unsigned tag, unsigned long long w) {
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    v4u v;
    v.x = (unsigned)w; v.y = tag; v.z = (unsigned)(w >> 32); v.w = tag;
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(dst), "v"(v) : "memory");
}
#else
__device__ inline void store_granule_pair(unsigned long long *, unsigned, unsigned long long) {}
#endif

// The published rows of the fused final reduction: component c of a row = the 16 bytes {low half, epoch, high half, epoch} at granules
// 2c, 2c + 1 -- ONE write-through (sc1) 16-byte store per lane, whole lines per workgroup, and ONE 16-byte sc1 load per component on
// the finisher's side (half the load instructions of two 8-byte granules; 8-byte accesses run at 0.54-0.70x the 16-byte rate, guide).
// Buffer addressing: the loads sit in a polling loop -- bit 31 of aux marks them volatile for the compiler, bit 4 is sc1.
#define SRL_OWN_ROW_OFFSET 13312              // LDS (behind everything the finisher stages, inside the phase-1 areas): the finishing workgroup's own row (32 doubles), handed over without a trip through memory
#if defined(__HIP_DEVICE_COMPILE__)
typedef unsigned v4u32 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rows_rsrc(unsigned long long *granules) {
    return __builtin_amdgcn_make_buffer_rsrc((void *)granules, 0, (SRL_FUSED_MAX_BLOCKS + SRL_FUSED_MAX_GROUPS) * SRL_ROW_GRANULES * 8, 0x00020000);
}
__device__ __forceinline__ void row_store(__amdgpu_buffer_rsrc_t rs, int block, int comp, unsigned epoch, double v) {
    const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
    v4u32 x;
    x.x = (unsigned)bits; x.y = epoch; x.z = (unsigned)(bits >> 32); x.w = epoch;
    __builtin_amdgcn_raw_buffer_store_b128(x, rs, (block * SRL_ROW_GRANULES + 2 * comp) * 8, 0, 16);
}
__device__ __forceinline__ v4u32 row_load(__amdgpu_buffer_rsrc_t rs, int block, int comp) {
    return __builtin_amdgcn_raw_buffer_load_b128(rs, (block * SRL_ROW_GRANULES + 2 * comp) * 8, 0, (int)0x80000010);
}
#endif

// debug time line of an ARMED pass (srl_debug_pass_stamps, tools/arm_timeline.py): workgroup 0 files slots 0..7, the finishing
// workgroup slots 8..15 of row (seq & 63); the pointer is re-read from the kernarg segment at every site (nothing stays live)
// Compiled in only with -DSRL_ARM_STAMPS (tools/build_variant.sh stamps "-DSRL_ARM_STAMPS -DSRL_STAMP_DETAIL"): the ten sites cost the
// armed pass 0.2-0.4 us even with a null buffer (measured A/B), so the product build carries none.  Slots >= 16 (inside the finisher
// and inside phase 2) additionally need -DSRL_STAMP_DETAIL.
#if defined(__HIP_DEVICE_COMPILE__) && defined(SRL_ARM_STAMPS)
__device__ __forceinline__ void arm_stamp(KargBytes karg, int slot) {
#if !defined(SRL_STAMP_DETAIL)
    if (slot >= 16) return;
#endif
    typedef const __attribute__((address_space(4))) SrlAssocArgs *KP;
    KP q = (KP)karg;
    asm volatile("" : "+s"(q));
    long long *st = q->stamps;
    if (st != nullptr && threadIdx.x == 0) {
        const bool first = blockIdx.x == 0, last = blockIdx.x == gridDim.x - 1;
        if (slot >= 16) { if (last) __hip_atomic_store(st + ((q->seq & 63ull) * 32 + slot), (long long)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); return; }
        if (first) __hip_atomic_store(st + ((q->seq & 63ull) * 32 + slot), (long long)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (last) __hip_atomic_store(st + ((q->seq & 63ull) * 32 + 8 + slot), (long long)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
#else
__device__ inline void arm_stamp(KargBytes, int) {}
#endif

// Phase 2 geometry: lanes per keypoint and keypoints per phase-2 wave for a workgroup of kpb keypoints.  One lane per
// keypoint from 48 keypoints on.  (Half-filled phase-2 waves -- 32 keypoints each, twice as many waves so that two dependent
// FP64 chains interleave per SIMD -- were measured in round 3: no gain)
#ifndef SRL_P2_POLICY
#define SRL_P2_POLICY 1
#endif
#if SRL_P2_POLICY == 0
__host__ __device__ constexpr int p2_lanes_per_keypoint(int kpb) { return kpb >= 48 ? 1 : (kpb >= 32 ? 2 : 4); }     // rounds 2-3
#else
// Round 4: small workgroups spread a keypoint over four lanes (the neighbour loops are a 20-deep FP64 dependency chain on a wave that
// has its SIMD to itself; with four lanes it is 5 deep + two quad-permute steps, and the phase occupies 4-6 of the 16 waves instead
// of 1-2).  Workgroups of >= 192 keypoints keep one lane per keypoint: there the phase is bound by instruction issue, not by the chain.
__host__ __device__ constexpr int p2_lanes_per_keypoint(int kpb) { return kpb >= 192 ? 1 : (kpb >= 128 ? 2 : 4); }
#endif
__host__ __device__ constexpr int p2_keypoints_per_wave(int kpb) { return 64 / p2_lanes_per_keypoint(kpb); }

// ARMED = 1: the pass as an armed launch (assoc_body's prologue): the pose sits in LDS, not in the kernarg segment.
// DBG = 1: the instantiation the profiling tools run (srl_debug_set_ablate: parts of the kernel switched off at run time, workgroup
// time stamps).  The production instantiations carry none of those tests: read in the pair loop they cost ~10 lane reads of
// spilled flags per keypoint pair (headline launch 49.4 -> 48.4 us together with the probe reordering below).
template <int NB, int FAST, int KPW, int WPB, int DBG = 0, int ARMED = 0>
__device__ __forceinline__ bool assoc_tile(KargBytes karg, const SrlAssocArgs &A, const int tile) {
    constexpr int KPB = WPB * KPW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int Kn = A.K;
    const int abl = DBG ? A.ablate : 0;
    constexpr bool POSE_LDS = ARMED != 0;                                 // the pose of this pass sits in LDS, not in the kernarg
    const LdsLayout L = lds_layout(Kn, NB, KPW, WPB, ARMED);
    const int NB_ROW = L.nb_row;
    const int tid = threadIdx.x;
    const int lane = lane_id();
    const int wave = tid >> 6;
    // workgroup-level keypoint arrays: any wave may search any keypoint pair of the workgroup (phase 1 hands pairs out
    // dynamically), phase 2 then takes the KPW keypoints of its own quarter
    float *s_nb = reinterpret_cast<float *>(smem + L.off_nb);
    const int nb_plane = Kn * NB_ROW;
    double *s_pw = reinterpret_cast<double *>(smem + L.off_pw);
    double *s_pimu = reinterpret_cast<double *>(smem + L.off_pimu);
    float *s_qf = reinterpret_cast<float *>(smem + L.off_qf);
    int *s_kv = reinterpret_cast<int *>(smem + L.off_kv);
    int *s_nfound = reinterpret_cast<int *>(smem + L.off_nfound);
    int *s_ncand = reinterpret_cast<int *>(smem + L.off_ncand);
    int *s_next = reinterpret_cast<int *>(smem + L.off_next);
    unsigned short *s_defer = reinterpret_cast<unsigned short *>(smem + L.off_defer);
    unsigned char *wbase = smem + L.off_wave + wave * L.wave_bytes;
    VoxEnt *vox = reinterpret_cast<VoxEnt *>(wbase + L.off_vox);
    Surv *surv = reinterpret_cast<Surv *>(wbase + L.off_scratch);
    double *s_wpart = reinterpret_cast<double *>(smem + L.off_wpart);     // [4][32]
    int *s_winfo = reinterpret_cast<int *>(smem + L.off_winfo);           // [WPB][8]: accepted, sum_pk, 1 + first NaN keypoint, fallback, planes

    const long long dbg_t0 = (DBG && (abl & 128)) ? (long long)wall_clock64() : 0;   // debug: workgroup start (100 MHz clock)
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const __attribute__((address_space(4))) SrlAssocArgs *KernargPtr;
#endif
    double *s_pose = reinterpret_cast<double *>(smem + L.off_pose);       // armed launch: Rn[9] | R[9] | t[3] of this pass
    // keypoints of this pass: an armed launch learns the count with its pose (it may have been fired for ANOTHER sweep than the one it was
    // armed on -- srl_sweep_swap --, the grid stays the one it was launched with: workgroups behind the sweep's end find empty tiles).
    // Re-read where it is needed: nothing stays live across phase 1.
    auto n_pass = [&]() -> int {
        if constexpr (ARMED) return __builtin_amdgcn_readfirstlane(reinterpret_cast<const int *>(s_pose + SRL_POSE_DOUBLES)[1]);
        else return A.n;
    };
    int n_fallback = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    auto tile_stamp = [&](int slot) {
        if constexpr (ARMED) {
            arm_stamp(karg, slot - 8);                                   // 10..13 -> slots 2..5: phase 0 / 1 / 2 done
        }
    };
#else
    auto tile_stamp = [&](int) {};
#endif
    tile_stamp(10);
    // bounds of the previous pass may be used when this pass runs on the sweep they were written for: not by an armed launch fired for
    // the context's other sweep buffer or for another keypoint count (srl_sweep_swap) -- the host vouches for everything else
    bool use_bounds = A.bound_in != nullptr && A.bound_use > 0;
    if constexpr (ARMED) use_bounds = use_bounds && reinterpret_cast<const int *>(s_pose + SRL_POSE_DOUBLES)[2] == 0 && n_pass() == A.n;
    const int bbase_kp = tile * KPB;                                      // first keypoint of this tile
    const int wbase_kp = bbase_kp + wave * KPW;                           // first keypoint of this wave's quarter (phases 0 and 2)

    // ---------------- phase 0: transformKeypoints (optimize.cpp:30-40), location (optimize.cpp:83), voxel key
    if (tid < 4) s_next[tid] = 0;
    if (lane < KPW) {
        const int kq = wave * KPW + lane;                                  // index inside the workgroup
        const int g = wbase_kp + lane;
        D3 p_imu = d3(0, 0, 0), p_w = d3(0, 0, 0);
        if constexpr (ARMED) {
            // the body-frame point does not depend on the pose: this lane computed it while the launch was waiting (assoc_body's prologue)
            if (g < n_pass()) {
                p_imu = d3(s_pimu[kq * 3 + 0], s_pimu[kq * 3 + 1], s_pimu[kq * 3 + 2]);
                p_w = add(matvec(s_pose, p_imu), d3(s_pose[18], s_pose[19], s_pose[20]));
            }
        } else {
            if (g < A.n) {
                D3 raw;
                if (A.aos != nullptr) {
                    // first pass over a prefetched sweep: the point still lies AoS in the staging buffer -- read it there and file the SoA planes
                    const double *p = A.aos + 3 * (size_t)g;
                    raw = d3(p[0], p[1], p[2]);
                    const_cast<double *>(A.raw_x)[g] = raw.x; const_cast<double *>(A.raw_y)[g] = raw.y; const_cast<double *>(A.raw_z)[g] = raw.z;
                } else {
                    raw = d3(A.raw_x[g], A.raw_y[g], A.raw_z[g]);
                }
                p_imu = add(matvec(A.R_il, raw), d3(A.t_il[0], A.t_il[1], A.t_il[2]));
                p_w = add(matvec(A.Rn, p_imu), d3(A.t[0], A.t[1], A.t[2]));
            }
            s_pimu[kq * 3 + 0] = p_imu.x; s_pimu[kq * 3 + 1] = p_imu.y; s_pimu[kq * 3 + 2] = p_imu.z;
        }
        s_pw[kq * 3 + 0] = p_w.x; s_pw[kq * 3 + 1] = p_w.y; s_pw[kq * 3 + 2] = p_w.z;
        {
            // FP32 prefilter constants of this keypoint (used once per keypoint by every lane of the wave later):
            // error model of select_topk_f32_r, m(T) = 4 a (T + 1)/2 + 8 u T + 4 a^2 at T = 2 tau + 1e-6, thr = (tau + 2 m)(1 + 1e-6)
            // = c0 + c1 tau
            const float qxf = (float)p_w.x, qyf = (float)p_w.y, qzf = (float)p_w.z;
            const float u = 5.9604645e-8f;
            const float am = fmaxf(fmaxf(fabsf(qxf), fabsf(qyf)), fabsf(qzf)) * u + 1e-30f;
            const float h = 4.0f * am * 0.5001f;                                  // 4 a / 2, rounded up
            const float m0 = h * (1.0f + 1e-6f) + 8.0f * u * 1e-6f + 4.0f * am * am;   // m at tau = 0
            const float m1 = 2.0f * h + 16.0f * u;                                  // d m / d tau
            float *qf = s_qf + kq * 8;
            qf[0] = qxf; qf[1] = qyf; qf[2] = qzf;
            qf[3] = (2.0f * m0) * 1.00001f;                                         // c0
            qf[4] = (1.0f + 2.0f * m1) * 1.00001f;                                  // c1
            // squared cull radius from the previous pass's bound (SrlAssocArgs::bound_in): r = sqrt(tau) + |p_w - p_w_prev| + slack for
            // everything that is rounded on the way (the stored FP32 positions -- a point may sit one ulp across its voxel's face --, the
            // FP32 query and box faces of probe_finish, tau's and the distance's own rounding); +inf = visit every voxel
            float r2 = __builtin_huge_valf();
            if (use_bounds && g < A.bound_use) {
                const float4 bp = reinterpret_cast<const float4 *>(A.bound_in)[g];
                const float ex = qxf - bp.x, ey = qyf - bp.y, ez = qzf - bp.z;
                const float mag = fabsf(qxf) + fabsf(qyf) + fabsf(qzf);
                const float r = sqrtf(bp.w) * 1.000001f + sqrtf(ex * ex + ey * ey + ez * ez) * 1.000001f + (1e-3f + 1e-6f * mag);
                r2 = (r * r) * 1.00001f;                                            // (tau = +inf: no K-th neighbour was known -- stays +inf)
            }
            qf[5] = r2;
            *reinterpret_cast<double *>(qf + 6) = __builtin_huge_val();             // this pass's tau: set by whoever finishes the keypoint on the fast path
        }
        // static_cast<short>(point / size_voxel_map): truncation toward zero (optimize.cpp:372-374)
        // (x / 1.0 == x exactly: the shipped size_voxel_map = 1.0 skips three FP64 divisions)
        const bool unit = A.size_voxel == 1.0;
        s_kv[kq * 4 + 0] = (int)(short)(int)(unit ? p_w.x : p_w.x / A.size_voxel);
        s_kv[kq * 4 + 1] = (int)(short)(int)(unit ? p_w.y : p_w.y / A.size_voxel);
        s_kv[kq * 4 + 2] = (int)(short)(int)(unit ? p_w.z : p_w.z / A.size_voxel);
        s_nfound[kq] = 0;
        s_ncand[kq] = 0;
    } else if (wave == WPB - 1 && lane < KPW + 2) {
        const int kq = KPB + (lane - KPW);                                 // the two zero entries behind the last pair
        s_kv[kq * 4 + 0] = 0; s_kv[kq * 4 + 1] = 0; s_kv[kq * 4 + 2] = 0;
    }
    __syncthreads();
    tile_stamp(11);

    // ---------------- phase 1: searchNeighbors, the whole wave on one keypoint at a time
    {
        const int left = n_pass() - bbase_kp;
        const int n_here = __builtin_amdgcn_readfirstlane(left < KPB ? left : KPB);   // keypoints of this workgroup that exist (<= 0: an empty tile)
        auto make_sink = [&](int kl) {
            LdsSink sink;
            sink.col = s_nb + kl;
            sink.row = NB_ROW;
            sink.plane = nb_plane;
            sink.tap_ids = A.tap_ids ? (A.tap_ids + (size_t)(bbase_kp + kl) * Kn) : nullptr;
            return sink;
        };
        // general path for one keypoint (own hash probes; tie = replay the reference's heap directly)
        auto keypoint_general = [&](int kl, bool tie) {
            const double qx = s_pw[kl * 3 + 0], qy = s_pw[kl * 3 + 1], qz = s_pw[kl * 3 + 2];
            LdsSink sink = make_sink(kl);
            int total = 0, fb = 0;
            const int nv = probe_voxels<NB>(qx, qy, qz, A.size_voxel, A.thr_cap, A.table, A.table_mask, vox, lane);
            select_topk(qx, qy, qz, nv, vox, A.slabs, Kn, tie ? 5 : A.select_mode, surv, lane, sink, total, fb);
            n_fallback += (NB == 1) ? 1 : fb;      // r = 1: anything off the fast path counts as a fallback
            if (lane == 0) {
                s_nfound[kl] = total < Kn ? total : Kn;
                s_ncand[kl] = total;
            }
        };
        if constexpr (NB == 1 && FAST != 0) {
            // The waves of the workgroup take keypoint PAIRS from a shared counter: a wave whose keypoints were cheap
            // simply takes more of them (per-keypoint cost varies ~3x with the number of occupied voxels; with static
            // shares the slowest wave sets the pace).  Hash lookups run
            // for the pair (one keypoint per half-wave) and one pair ahead, so their L2 round trip overlaps the selection
            // of the current pair.  Results do not depend on who searched what: phase 2 is static.
            // Keypoints the fast path cannot finish (> 64 survivors, or a tie among the K+1 smallest distances) are only
            // NOTED here (workgroup list) and handled after the loop: the hot loop carries no general-path code.
            const LaneRole role = lane_role(lane);
            const int npairs = n_here > 0 ? (n_here + 1) >> 1 : 0;            // (an empty tile: a workgroup behind the end of a shorter sweep)
            auto take = [&]() { int p = 0; if (lane == 0) p = atomicAdd(s_next, 1); return __builtin_amdgcn_readfirstlane(p); };
            int cur = take();
            ProbeReq preq;
            if (!(abl & 32)) preq = probe_issue(s_kv, 2 * (cur < npairs ? cur : npairs), role, A.table, A.table_mask, lane);
            while (cur < npairs) {
                const int nxt = take();
                // this pair's probes are consumed BEFORE the next pair's are issued: one probe state live at a time (no copy
                // of the 11-register request per pair); the next pair's table loads still have the whole selection to land
                const int nv_pair = __builtin_amdgcn_readfirstlane((abl & 8) ? 0 : probe_finish(preq, A.thr_cap, A.table, A.table_mask, vox, lane, s_ncand + 2 * cur, use_bounds, s_qf + 16 * cur,
                                                                                                               s_kv + 8 * cur, &role, (float)A.size_voxel));
                if (!(abl & 32)) preq = probe_issue(s_kv, 2 * (nxt < npairs ? nxt : npairs), role, A.table, A.table_mask, lane);
                auto file = [&](int kl, int done, int total) {        // lane 0: result of one keypoint
                    (void)total;
                    if (done == SEL_DONE) {
                        // (the candidate total P_k was filed by probe_finish -- every found voxel, also those the bounds let this pass skip;
                        //  the loads of cand_issue count only what was visited)
                    } else {
                        s_defer[atomicAdd(s_next + 1, 1)] = (unsigned short)(kl | (done == SEL_TIE ? 0x8000 : 0));
                    }
                };
                // ceil(nv / 3) for nv <= 27 as a multiply-shift (the signed / 3 went through s_mul_hi)
                const int r_a = (int)((((unsigned)nv_pair & 0xFFu) + 2u) * 0x5556u >> 16), r_b = (int)(((((unsigned)nv_pair >> 8) & 0xFFu) + 2u) * 0x5556u >> 16);
                const int r_max = r_a > r_b ? r_a : r_b;
                if (2 * cur + 1 < n_here && r_max <= SRL_PAIR_MAX_ROUNDS && !(abl & (4 | 256))) {
                    // both keypoints exist and their candidate rounds fit in registers together: B's loads fly while A is selected
                    const int kl = 2 * cur;
                    LdsSink sink_a = make_sink(kl), sink_b = make_sink(kl + 1);
                    int total_a = 0, total_b = 0, done;
                    // (two rounds: what a pass that starts from the previous pass's bounds typically has left -- ~6 of ~12 occupied voxels)
                    if (r_max <= 2) done = select_pair_f32_r<2>(s_pw + kl * 3, s_pw + kl * 3 + 3, s_qf + kl * 8, s_qf + kl * 8 + 8, vox, A.slabs, A.inf_off, Kn, surv, lane, role, sink_a, sink_b, total_a, total_b, abl);
                    else if (r_max <= 3) done = select_pair_f32_r<3>(s_pw + kl * 3, s_pw + kl * 3 + 3, s_qf + kl * 8, s_qf + kl * 8 + 8, vox, A.slabs, A.inf_off, Kn, surv, lane, role, sink_a, sink_b, total_a, total_b, abl);
                    else done = select_pair_f32_r<4>(s_pw + kl * 3, s_pw + kl * 3 + 3, s_qf + kl * 8, s_qf + kl * 8 + 8, vox, A.slabs, A.inf_off, Kn, surv, lane, role, sink_a, sink_b, total_a, total_b, abl);
                    // (candidate totals: accumulated by probe_finish; neighbour counts: derived from them in phase 2 -- a pair that
                    // finished on the fast path files nothing)
                    (void)total_a; (void)total_b;
                    if (done != (SEL_DONE | (SEL_DONE << 8))) {
                        if (lane == 0) {
                            if ((done & 0xFF) != SEL_DONE) s_defer[atomicAdd(s_next + 1, 1)] = (unsigned short)(kl | ((done & 0xFF) == SEL_TIE ? 0x8000 : 0));
                            if ((done >> 8) != SEL_DONE) s_defer[atomicAdd(s_next + 1, 1)] = (unsigned short)((kl + 1) | ((done >> 8) == SEL_TIE ? 0x8000 : 0));
                        }
                    }
                } else {
#pragma nounroll
                    for (int h = 0; h < 2; ++h) {
                        const int kl = 2 * cur + h;
                        if (kl >= n_here) break;
                        const int nv_fast = h ? (nv_pair >> 8) : (nv_pair & 0xFF);
                        const double qx = s_pw[kl * 3 + 0], qy = s_pw[kl * 3 + 1], qz = s_pw[kl * 3 + 2];
                        LdsSink sink = make_sink(kl);
                        int total = nv_fast;
                        int done = SEL_DONE;
                        if (!(abl & 4)) done = select_topk_f32(qx, qy, qz, s_qf + kl * 8, nv_fast, vox + 32 * h, A.slabs, A.inf_off, Kn, surv, lane, role, sink, total, abl);
                        if (lane == 0) file(kl, done, total);
                    }
                }
                cur = nxt;
            }
        } else if constexpr (NB == 2 && FAST != 0) {
            // init mode (r = 2): single keypoints from the shared counter, own 125 probes, looped fast path
            const LaneRole role = lane_role(lane);
            auto take = [&]() { int p = 0; if (lane == 0) p = atomicAdd(s_next, 1); return __builtin_amdgcn_readfirstlane(p); };
            for (int kl = take(); kl < n_here; kl = take()) {
                const double qx = s_pw[kl * 3 + 0], qy = s_pw[kl * 3 + 1], qz = s_pw[kl * 3 + 2];
                int total_all = 0;
                const int nv = probe_voxels<NB>(qx, qy, qz, A.size_voxel, A.thr_cap, A.table, A.table_mask, vox, lane, use_bounds ? s_qf + kl * 8 : nullptr, &total_all);
                if (lane < 3) { VoxEnt z; z.slab = 0u; z.count = 0u; vox[nv + lane] = z; }      // branch-free reads up to 3 * rounds
                __builtin_amdgcn_wave_barrier();
                LdsSink sink = make_sink(kl);
                int total = 0;
                const int done = select_topk_f32_loop(qx, qy, qz, s_qf + kl * 8, nv, vox, A.slabs, A.inf_off, Kn, surv, lane, role, sink, total);
                total = total_all;           // P_k: every found voxel counts, also those the bounds let this pass skip
                if (lane == 0) {
                    if (done == SEL_DONE) {
                        s_nfound[kl] = total < Kn ? total : Kn;
                        s_ncand[kl] = total;
                    } else {
                        s_defer[atomicAdd(s_next + 1, 1)] = (unsigned short)(kl | (done == SEL_TIE ? 0x8000 : 0));
                    }
                }
            }
        } else {
            // general path only (forced modes): static quarters
            for (int kl = wave * KPW; kl < (wave + 1) * KPW && kl < n_here; ++kl) keypoint_general(kl, false);
        }
        if constexpr (FAST != 0) {
            __syncthreads();
            // deferred keypoints (rare; none on tie-free sweeps): handed out one by one to whichever wave comes first
            const int n_defer = __builtin_amdgcn_readfirstlane(s_next[1]);
            if (n_defer > 0) {
                auto take2 = [&]() { int p = 0; if (lane == 0) p = atomicAdd(s_next + 2, 1); return __builtin_amdgcn_readfirstlane(p); };
                for (int i = take2(); i < n_defer; i = take2()) {
                    const int e = __builtin_amdgcn_readfirstlane((int)s_defer[i]);
                    keypoint_general(e & 0x7FFF, (e & 0x8000) != 0);
                }
            }
        }
    }
    __syncthreads();
    tile_stamp(12);

    if (abl & 64) return true;                              // debug: phase 0 + loop skeleton only
    // Phase 2 re-reads its parameters from the kernarg segment through a laundered pointer: kept live across phase 1
    // they cost ~70 SGPRs and pushed the selection loop into SGPR spills (v_readlane / v_writelane).
#if defined(__HIP_DEVICE_COMPILE__)
    KernargPtr bp = (KernargPtr)karg;     // the struct is the kernel's first argument
    asm volatile("" : "+s"(bp));
    const __attribute__((address_space(4))) SrlAssocArgs &b = *bp;
#else
    const SrlAssocArgs &b = A;
#endif
    // ---------------- phase 2: plane fit + residual + Jacobian.  LPK lanes per keypoint: ONE when the workgroup has at least
    // 48 keypoints (its KPB keypoints then fill ceil(KPB / 64) waves and the other waves have nothing to do here), two / four
    // for workgroups of 32 / 16 keypoints (one wave either way; the lanes of a keypoint split its neighbours and butterfly the
    // sums, so the serial chain stays short where there is nothing to amortise).  Until round 2 every keypoint had four
    // lanes that ran the scalar part -- eigen-decomposition, weights, gate, Jacobian: most of this phase -- redundantly, so 16
    // keypoints cost a wave ~800 instructions: 16 waves x 800 per 256 keypoints against 4 x ~700 now (headline: -3 us).
    // Which waves work rotates with the workgroup index, so that the phase-2 waves of workgroups sharing a CU sit on
    // different SIMDs.  With one lane per keypoint the neighbour-plane reads are conflict-free and the barycentre / scatter
    // sums run sequentially over the neighbours, the reference's own order (optimize.cpp:320-337).
    constexpr int LPK = p2_lanes_per_keypoint(KPB);                                  // lanes per keypoint
    constexpr int KP2 = p2_keypoints_per_wave(KPB);                                  // keypoints per phase-2 wave
    constexpr int P2W = (KPB + KP2 - 1) / KP2;
    const int w2 = (wave + WPB - (int)(blockIdx.x % WPB)) % WPB;                    // phase-2 slot of this wave
    const bool p2_wave = w2 < P2W;
    const int klw = lane / LPK, sl = lane % LPK;                                    // keypoint inside this wave, sub-lane
    const int kl = (p2_wave ? w2 : 0) * KP2 + klw;                                  // keypoint inside the workgroup
    const bool owner_lane = p2_wave && klw < KP2 && kl < KPB;
    const int n2 = ARMED ? n_pass() : b.n;                                          // keypoints of this pass (phase 2's copy)
    const int g = owner_lane ? bbase_kp + kl : n2;
    // sum over the LPK lanes of a keypoint (butterfly: all of them end with the same bits)
    // (quad permutes on the 32-bit halves: two v_mov_dpp per step -- __shfl_xor goes through the LDS crossbar, ds_bpermute x 2)
    auto quad_xor = [](double v, auto ctrl) {
#if defined(__HIP_DEVICE_COMPILE__)
        const long long b = __double_as_longlong(v);
        const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)b, decltype(ctrl)::value, 0xF, 0xF, true);
        const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)((unsigned long long)b >> 32), decltype(ctrl)::value, 0xF, 0xF, true);
        return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
#else
        return v;
#endif
    };
    auto lpk_sum = [&](double v) {
        if (LPK >= 2) v += quad_xor(v, std::integral_constant<int, 0xB1>{});      // quad_perm [1, 0, 3, 2]: lane ^ 1
        if (LPK >= 4) v += quad_xor(v, std::integral_constant<int, 0x4E>{});      // quad_perm [2, 3, 0, 1]: lane ^ 2
        return v;
    };
    int status = 3;
    bool nan_bad = false;
    double J[6] = {0, 0, 0, 0, 0, 0};
    double dist = 0.0, weight = 0.0;
    // neighbours found = min(candidates visited, K): every path leaves the candidate count in s_ncand
    const int nc2 = owner_lane ? s_ncand[kl] : 0;
    const int nf = nc2 < Kn ? nc2 : Kn;
    if (g < n2) status = 0;
    const bool fit = (g < n2) && (nf >= b.min_nb) && !(DBG && (b.ablate & 1));
    if (fit) {
#pragma clang fp contract(fast)      // plane fit / weights / Jacobian are tolerance-bound (1e-9 vs the oracle): products may fuse
        const D3 p_imu = d3(s_pimu[kl * 3 + 0], s_pimu[kl * 3 + 1], s_pimu[kl * 3 + 2]);
        const D3 p_w = d3(s_pw[kl * 3 + 0], s_pw[kl * 3 + 1], s_pw[kl * 3 + 2]);
        // barycenter (optimize.cpp:320-325)
        D3 bc = d3(0, 0, 0);
        for (int i = sl; i < nf; i += LPK) {
            const float *p = s_nb + i * NB_ROW + kl;
            bc = add(bc, d3((double)p[0], (double)p[nb_plane], (double)p[2 * nb_plane]));
        }
        bc = d3(lpk_sum(bc.x), lpk_sum(bc.y), lpk_sum(bc.z));
        const double icnt = rcp_nr((double)nf);
        bc = d3(bc.x * icnt, bc.y * icnt, bc.z * icnt);
        // scatter matrix, upper triangle (optimize.cpp:328-337)
        double c00 = 0, c01 = 0, c02 = 0, c11 = 0, c12 = 0, c22 = 0;
        for (int i = sl; i < nf; i += LPK) {
            const float *p = s_nb + i * NB_ROW + kl;
            const double ex = (double)p[0] - bc.x, ey = (double)p[nb_plane] - bc.y, ez = (double)p[2 * nb_plane] - bc.z;
            c00 += ex * ex; c01 += ex * ey; c02 += ex * ez;
            c11 += ey * ey; c12 += ey * ez;
            c22 += ez * ez;
        }
        double C[3][3];
        C[0][0] = lpk_sum(c00); C[0][1] = lpk_sum(c01); C[0][2] = lpk_sum(c02);
        C[1][1] = lpk_sum(c11); C[1][2] = lpk_sum(c12); C[2][2] = lpk_sum(c22);
        C[1][0] = C[0][1]; C[2][0] = C[0][2]; C[2][1] = C[1][2];
        if constexpr (ARMED) arm_stamp(karg, 21);
        double ev[3];
        D3 nrm;
        // Closed form first (5x fewer FP64 instructions, no data-dependent loop).  It resolves the two SMALLER eigenvalues only down to
        // ~sqrt(eps) of the largest (a near-double root of the characteristic cubic: error ~ eps p^2 / gap): on a line-like neighbourhood
        // -- a cable, a pole: e0 ~ e1 << e2 -- the eigenvector of e0 inside that near-null plane comes out arbitrary, while the
        // reference's iterative solver (optimize.cpp:339) still resolves it, and such a keypoint is NOT weightless (optimize.cpp:87-88:
        // lambda_neighborhood * exp(...) does not vanish with a2D).  There the Jacobi sweeps run -- high relative accuracy on the small
        // eigenvalues; never on a planar patch (e1 ~ e2), so the benchmark scenes do not pay for it.
        bool jacobi = b.select_mode == 4;
        if (!jacobi) {
            eig3_closed(C, ev, nrm);                               // already unit length (re-normalised once more at :93 below)
            jacobi = (ev[1] - ev[0]) < 1e-3 * ev[2];
        }
        if (jacobi) { eig3_jacobi(C, ev, nrm); nrm = normalized3(nrm); }   // .col(0).normalized() (optimize.cpp:340)
        if constexpr (ARMED) arm_stamp(karg, 22);
        const double sigma_1 = sqrt(fabs(ev[2]));
        const double sigma_2 = sqrt(fabs(ev[1]));
        const double sigma_3 = sqrt(fabs(ev[0]));
        const double a2D = (sigma_2 - sigma_3) * rcp_nr(sigma_1);  // optimize.cpp:343-346 (0 / 0 still yields NaN)
        if (a2D != a2D) nan_bad = true;                           // optimize.cpp:348-350 throws here: no residual from this keypoint
        const double w_plan = (b.power_planarity == 2.0) ? a2D * a2D : pow(a2D, b.power_planarity);
        // normal flip: world-frame last translation minus body-frame location (optimize.cpp:49-51)
        const D3 tl = POSE_LDS ? d3(s_pose[21], s_pose[22], s_pose[23]) : d3(b.t_last[0], b.t_last[1], b.t_last[2]);   // (armed: t_last came with the pose)
        if (dot3(nrm, sub(tl, p_imu)) < 0.0) nrm = d3(-1.0 * nrm.x, -1.0 * nrm.y, -1.0 * nrm.z);
        const D3 nn0 = d3((double)s_nb[kl], (double)s_nb[nb_plane + kl], (double)s_nb[2 * nb_plane + kl]);
        const D3 dq = sub(nn0, p_w);
        weight = b.lambda_w * w_plan + b.lambda_n * exp(-sqrt(dot3(dq, dq)) * rcp_nr(b.nbr_scale));   // optimize.cpp:87-88
        const D3 nv = normalized3_fast(nrm);                       // optimize.cpp:93
        const double off = -dot3(nv, nn0);                         // optimize.cpp:94
        // the residual uses the un-normalised rotation (optimize.cpp:95,101); armed launch: the pose block in LDS
        double Rm[9], tv[3];
#pragma unroll
        for (int i = 0; i < 9; ++i) Rm[i] = POSE_LDS ? s_pose[9 + i] : b.R[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) tv[i] = POSE_LDS ? s_pose[18 + i] : b.t[i];
        const D3 pe = add(matvec(Rm, p_imu), d3(tv[0], tv[1], tv[2]));
        dist = dot3(nv, pe) + off;                                 // optimize.cpp:95
        status = 1;
        if (b.tap_normal && sl == 0) {
            b.tap_normal[(size_t)g * 3 + 0] = nv.x; b.tap_normal[(size_t)g * 3 + 1] = nv.y; b.tap_normal[(size_t)g * 3 + 2] = nv.z;
            b.tap_a2d[g] = a2D;
            b.tap_offset[g] = off;
        }
        if (dist < b.max_dist && !nan_bad) {                       // signed gate (optimize.cpp:98)
            status = 2;
            J[0] = nv.x * weight; J[1] = nv.y * weight; J[2] = nv.z * weight;
            // - n^T * R * skew(p_imu) * weight, left to right (optimize.cpp:101)
            const double m0 = -nv.x, m1 = -nv.y, m2 = -nv.z;
            const double r0 = (m0 * Rm[0] + m1 * Rm[3]) + m2 * Rm[6];
            const double r1 = (m0 * Rm[1] + m1 * Rm[4]) + m2 * Rm[7];
            const double r2 = (m0 * Rm[2] + m1 * Rm[5]) + m2 * Rm[8];
            // skew(p) = [[0,-pz,py],[pz,0,-px],[-py,px,0]]
            const double s0 = (r0 * 0.0 + r1 * p_imu.z) + r2 * (-p_imu.y);
            const double s1 = (r0 * (-p_imu.z) + r1 * 0.0) + r2 * p_imu.x;
            const double s2 = (r0 * p_imu.y + r1 * (-p_imu.x)) + r2 * 0.0;
            J[3] = s0 * weight; J[4] = s1 * weight; J[5] = s2 * weight;
        }
    }
    if constexpr (ARMED) arm_stamp(karg, 23);
    if (g < n2 && sl == 0 && b.bound_out != nullptr) {
        // what the next pass over this sweep may start from (SrlAssocArgs::bound_in): this pass's world position and the exact squared
        // distance of the K-th nearest neighbour, rounded UP to FP32 (+inf where the fast path did not finish the keypoint)
        const float *qf = reinterpret_cast<const float *>(smem + L.off_qf) + kl * 8;
        const double tau = *reinterpret_cast<const double *>(qf + 6);
        float4 bo;
        bo.x = qf[0]; bo.y = qf[1]; bo.z = qf[2];
        bo.w = (float)tau * 1.0000002f;
        reinterpret_cast<float4 *>(b.bound_out)[g] = bo;
    }
    if (g < n2 && b.write_rec && sl == 0) {
        // per-keypoint record {J[6], distance, weight} (ordered cut-off path + taps; never on the throughput path)
        double2 *r = reinterpret_cast<double2 *>(b.rec + (size_t)g * 8);
        double2 v;
        v.x = J[0]; v.y = J[1]; r[0] = v;
        v.x = J[2]; v.y = J[3]; r[1] = v;
        v.x = J[4]; v.y = J[5]; r[2] = v;
        v.x = dist; v.y = weight; r[3] = v;
        b.status[g] = (unsigned char)status;
        if (b.tap_ncand) b.tap_ncand[g] = s_ncand[kl];
    }

    if (b.rec_granules != nullptr && owner_lane && sl == 0) {
        // fused ordered cut (4.2): the record {J[6], distance, weight} of every keypoint of this workgroup travels as 16 tagged
        // granules (double d = granules 2d, 2d + 1); zeros unless accepted -- also for the keypoints behind the end of the sweep
        typedef __attribute__((address_space(1))) unsigned long long gu64r;
        const bool accd = status == 2;
        const double rec8[8] = {J[0], J[1], J[2], J[3], J[4], J[5], dist, weight};
        const unsigned long long tag = (unsigned long long)(unsigned)b.seq << 32;   // the epoch of this pass
        gu64r *dst = (gu64r *)(b.rec_granules + ((size_t)blockIdx.x * KPB + kl) * 16);
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            const unsigned long long bits = accd ? (unsigned long long)__double_as_longlong(rec8[d]) : 0ull;
            __hip_atomic_store(dst + 2 * d, tag | (bits & 0xFFFFFFFFull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(dst + 2 * d + 1, tag | (bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }

    // ---- wave partial of H^T H (21 upper-tri), H^T h (6), loss (1).  Every keypoint of the wave leaves the row
    // {J[0..5], h, distance} in LDS (zeros unless accepted); lane c < 28 then owns component c and walks the KP2 rows in
    // keypoint order -- two LDS reads, one multiply, one add each, no cross-lane traffic, fixed summation order.
    if (p2_wave) {
        // KP2 rows x 8 doubles per phase-2 wave in the per-wave regions of phase 1 (free behind the barrier above)
        static_assert(P2W * KP2 * 64 <= WPB * (64 * 8 + SRL_WAVE_SCRATCH), "phase-2 rows must fit in the phase-1 per-wave regions");
        double *s_row = reinterpret_cast<double *>(smem + L.off_wave) + w2 * (KP2 * 8);
        const bool accd = status == 2;
        if (sl == 0 && klw < KP2) {
            const double h = dist * weight;                      // optimize.cpp:169
            double4 v;
            v.x = accd ? J[0] : 0.0; v.y = accd ? J[1] : 0.0; v.z = accd ? J[2] : 0.0; v.w = accd ? J[3] : 0.0;
            *reinterpret_cast<double4 *>(s_row + klw * 8) = v;
            v.x = accd ? J[4] : 0.0; v.y = accd ? J[5] : 0.0; v.z = accd ? h : 0.0; v.w = accd ? dist : 0.0;
            *reinterpret_cast<double4 *>(s_row + klw * 8 + 4) = v;
        }
        __builtin_amdgcn_wave_barrier();
        if (lane < 28) {
            // (row, column) of component c: upper triangle of H^T H, then J_i h, then distance^2
            int ia, ib;
            if (lane < 21) {
                // triangular index -> (ia, ib), ia <= ib < 6
                int c = lane, r = 0;
                while (c >= 6 - r) { c -= 6 - r; ++r; }
                ia = r; ib = r + c;
            } else if (lane < 27) { ia = lane - 21; ib = 6; }
            else { ia = 7; ib = 7; }
            double acc = 0.0;
#pragma unroll 16
            for (int k = 0; k < KP2; ++k) acc += s_row[k * 8 + ia] * s_row[k * 8 + ib];
            s_wpart[w2 * 32 + lane] = acc;
        }
    }
    if constexpr (ARMED) arm_stamp(karg, 24);
    {
        // one bit per keypoint: the ballot has the keypoint's bit at lane klw * LPK
        auto per_keypoint = [](unsigned long long m) {
            if (LPK == 2) {
                m &= 0x5555555555555555ull;
                m = (m | (m >> 1)) & 0x3333333333333333ull;
                m = (m | (m >> 2)) & 0x0F0F0F0F0F0F0F0Full;
                m = (m | (m >> 4)) & 0x00FF00FF00FF00FFull;
                m = (m | (m >> 8)) & 0x0000FFFF0000FFFFull;
                m = (m | (m >> 16)) & 0x00000000FFFFFFFFull;
            } else if (LPK == 4) {
                m &= 0x1111111111111111ull;
                m = (m | (m >> 3)) & 0x0303030303030303ull;
                m = (m | (m >> 6)) & 0x000F000F000F000Full;
                m = (m | (m >> 12)) & 0x000000FF000000FFull;
                m = (m | (m >> 24)) & 0xFFFFull;
            }
            return m;
        };
        const unsigned long long acc_mask = __ballot(status == 2 && sl == 0);
        const unsigned long long nan_mask = __ballot(nan_bad);
        const unsigned long long pln_mask = __ballot((status == 1 || status == 2) && sl == 0);
        int pk = (g < n2 && sl == 0) ? s_ncand[kl] : 0;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) pk += __shfl_xor(pk, off);
        if (lane == 0) {
            s_winfo[wave * 8 + 3] = n_fallback;                 // phase-1 work of THIS wave
            if (p2_wave) {                                       // phase-2 results, filed under the phase-2 slot
                s_winfo[w2 * 8 + 0] = __popcll(acc_mask);
                s_winfo[w2 * 8 + 1] = pk;
                // first NaN keypoint of this wave as 1 + index inside the workgroup
                s_winfo[w2 * 8 + 2] = nan_mask ? 1 + w2 * KP2 + (int)__builtin_ctzll(nan_mask) / LPK : 0;
                s_winfo[w2 * 8 + 4] = __popcll(pln_mask);
                // accepted keypoints of this wave (bit i = keypoint w2 * KP2 + i), as two 32-bit words
                const unsigned long long km = per_keypoint(acc_mask);
                s_winfo[w2 * 8 + 5] = (int)(unsigned)km;
                s_winfo[w2 * 8 + 6] = (int)(unsigned)(km >> 32);
            }
        }
    }
    __syncthreads();
    tile_stamp(13);

    if (DBG && (b.ablate & 128) && (tid == 28 || tid == 29 || tid == 30))        // debug: start / end stamps of this workgroup in the spare slots
        b.partials[(size_t)blockIdx.x * SRL_PART_STRIDE + tid] =
            (tid == 28) ? (double)dbg_t0 : ((tid == 29) ? (double)(long long)wall_clock64() : (double)__builtin_amdgcn_s_getreg(6164) /* XCC_ID */);
    // ---- block partial = wave partials added in wave order (deterministic)
    if (tid < 28 && b.granules == nullptr) {
        double v = s_wpart[tid];
#pragma unroll
        for (int w = 1; w < P2W; ++w) v += s_wpart[w * 32 + tid];
        b.partials[(size_t)blockIdx.x * SRL_PART_STRIDE + tid] = v;
    }
    if (tid == 0) {
        SrlBlockInfo bi;
        bi.accepted = 0; bi.sum_pk = 0; bi.nan_first = 0; bi.num_fallback = 0; bi.planes = 0;
        bi.pad[0] = bi.pad[1] = bi.pad[2] = 0;
        for (int w = 0; w < WPB; ++w) bi.num_fallback += s_winfo[w * 8 + 3];
        for (int w = 0; w < P2W; ++w) {
            bi.accepted += s_winfo[w * 8 + 0]; bi.sum_pk += (unsigned)s_winfo[w * 8 + 1];
            if (bi.nan_first == 0) bi.nan_first = s_winfo[w * 8 + 2];       // slots in keypoint order: the first one wins
            bi.planes += s_winfo[w * 8 + 4];
        }
        b.binfo[blockIdx.x] = bi;
    }
    return b.granules == nullptr;       // true: not fused -- the reduce kernel takes it from here
}

// ---------------------------------------------------------------------------------------------------------------------
// Direct peer exchange of one row (SrlPeerTable, srl_device.h), run by ONE wave: lane t < nwords stores the 64-bit word
// `mine` of its rank's row, as two tagged granules, into the inbox of every rank (its own included), then collects word t
// of every rank's row from its own inbox -- rank 0 first -- and hands it to take(rank, word).  System-scope stores and
// loads on fine-grained memory: a peer's store over xGMI is visible to the polling load without any cache maintenance.
// Returns false (wave-uniform) when some row did not arrive within the bounded spin (a peer died or never launched).
// ---------------------------------------------------------------------------------------------------------------------
template <class Take>
__device__ __forceinline__ bool peer_exchange(const SrlPeerTable *pt, unsigned epoch, int slot, int lane, int nwords,
                                              unsigned long long mine, Take take) {
    typedef __attribute__((address_space(1))) unsigned long long gu64;
    const int G = pt->nranks, me = pt->rank;
    const bool act = lane < nwords;
    const unsigned long long tag = (unsigned long long)epoch << 32;
    const size_t slot_off = (size_t)slot * SRL_MAX_PEERS * (2 * SRL_PEER_ROW);
    if (act) {
        const unsigned long long lo = tag | (unsigned)mine, hi = tag | (unsigned)(mine >> 32);
        for (int r = 0; r < G; ++r) {
            unsigned long long *dst = pt->inbox[r] + slot_off + (size_t)me * (2 * SRL_PEER_ROW);
            __hip_atomic_store((gu64 *)(dst + lane), lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store((gu64 *)(dst + SRL_PEER_ROW + lane), hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    bool ok = true;
    const unsigned long long *own = pt->inbox[me] + slot_off;
    for (int r = 0; r < G; ++r) {
        if (act) {
            const unsigned long long *src = own + (size_t)r * (2 * SRL_PEER_ROW);
            unsigned long long x0, x1;
            unsigned spins = 0;
            for (;;) {
                x0 = __hip_atomic_load((gu64 *)(src + lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                x1 = __hip_atomic_load((gu64 *)(src + SRL_PEER_ROW + lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if ((unsigned)(x0 >> 32) == epoch && (unsigned)(x1 >> 32) == epoch) break;
                if (++spins > (1u << 20)) { ok = false; x0 = x1 = 0ull; break; }     // ~1 s
                __builtin_amdgcn_s_sleep(8);
            }
            take(r, (x1 << 32) | (unsigned long long)(unsigned)x0);
        }
    }
    return __ballot(!ok) == 0ull;
}

// ---------------------------------------------------------------------------------------------------------------------
// The finishing workgroup: sums the published rows (with the ordered cut of optimize.cpp:107 when max_num_residuals can
// bind) and sends the normal equations to the host mailbox.
// ---------------------------------------------------------------------------------------------------------------------
template <int KPW, int WPB>
__device__ __forceinline__ void finish_rows(KargBytes karg, const unsigned epoch, const int n_total) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int KPB = WPB * KPW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef const __attribute__((address_space(4))) SrlAssocArgs *KernargPtr;
    typedef __attribute__((address_space(1))) unsigned long long gu64;
    KernargPtr bq = (KernargPtr)karg;
    asm volatile("" : "+s"(bq));
    const __attribute__((address_space(4))) SrlAssocArgs &b = *bq;
    const int tid = threadIdx.x;
    const int lane = lane_id();
    const int wave = tid >> 6;
    (void)lane; (void)wave;
    // The normal equations are assembled in LDS (every writer sits in wave 0) and leave in ONE step: mail_out() below.
    auto put_f = [](double *p, double x) { *p = x; };
    auto put_i = [](long long *p, long long x) { *p = x; };
    // mail_out: wave 0 sends the 52 words of the staged SrlDevOut.  Tagged form (the host mailbox of a single-context pass): word w
    // travels as two granules {sequence number, 32-bit half} -- the host checks the tags, so no store has to wait for another (the
    // plain form pays a drain + a second PCIe write for the sequence word: ~1.2 us).  Plain form: a device-side mailbox the RCCL
    // all-reduce works on, read as doubles.
    auto mail_out = [&](const SrlDevOut *staged) {
        constexpr int NW = (int)(sizeof(SrlDevOut) / 8);
        if (tid < 64) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const unsigned long long w = tid < NW ? reinterpret_cast<const unsigned long long *>(staged)[tid] : 0ull;
            if (b.mail_tagged) {
                if (tid < NW) store_granule_pair(&b.mailbox->g[2 * tid], (unsigned)b.seq, w);
            } else {
                if (tid < NW) __hip_atomic_store(reinterpret_cast<unsigned long long *>(&b.mailbox->out) + tid, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (tid == 0) __hip_atomic_store(&b.mailbox->seq, b.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    };
    if (b.cut_max > 0) {
        // ---- finisher WITH the ordered cut (optimize.cpp:107: the sequential loop stops at the max-th accepted residual).
        // (a) every thread t < #workgroups reads the counters of row t; a prefix over the accepted counts finds the workgroup c that
        // holds the max-th accepted residual and how many of its residuals still count; (b) rows r < c are summed in the same
        // fixed order as without a cut (counters over all rows); (c) workgroup c's acceptance mask gives the stop keypoint and
        // its records -- re-read as tagged granules, one per thread -- are re-accumulated in keypoint order up to it.
        // Same results as srl_reduce_kernel on the same launch shape (tests/test_gpu_parity.py runs both).
        if constexpr (KPB <= SRL_FUSED_CUT_MAX_KPB && WPB == 16) {
            constexpr int NT = 64 * WPB, NPART = NT / 32, INF = 8;
            __syncthreads();
            double *s_part = reinterpret_cast<double *>(smem);                           // [NPART][32]
            double *s_recd = reinterpret_cast<double *>(smem + NPART * 32 * 8);          // [KPB][8]
            long long *s_wacc = reinterpret_cast<long long *>(smem + NPART * 32 * 8 + SRL_FUSED_CUT_MAX_KPB * 64);   // [WPB]
            int *s_i = reinterpret_cast<int *>(smem + NPART * 32 * 8 + SRL_FUSED_CUT_MAX_KPB * 64 + WPB * 8);      // bad, c, allowed, pos, nan_min, -, -, -, mask[8]
            const int nbk = (int)gridDim.x;
            const long long cut_max = b.cut_max;
            if (tid == 0) { s_i[0] = 0; s_i[1] = nbk; s_i[2] = 0; s_i[3] = -1; s_i[4] = 0x7fffffff; }
            __syncthreads();
            bool timed_out = false;
            auto ld = [](const unsigned long long *p) { return __hip_atomic_load((gu64 *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
            auto fresh = [epoch](unsigned long long x) { return (unsigned)(x >> 32) == epoch; };
            auto as_double = [](unsigned lo, unsigned hi) { return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo)); };
            // (a) four granules per thread, requested together (every poll round is one memory round trip, not four)
            const __amdgpu_buffer_rsrc_t rs = rows_rsrc(b.granules);
            const double *s_own = reinterpret_cast<const double *>(smem + SRL_OWN_ROW_OFFSET);      // this workgroup's own row (assoc_body)
            long long my_acc = 0;
            if (tid < nbk) {
                double d_acc, d_nan;
                if (tid == nbk - 1) { d_acc = s_own[28]; d_nan = s_own[30]; }
                else {
                    v4u32 x0, x1;
                    unsigned spins = 0;
                    for (;;) {
                        x0 = row_load(rs, tid, 28); x1 = row_load(rs, tid, 30);
                        if (x0.y == epoch && x0.w == epoch && x1.y == epoch && x1.w == epoch) break;
                        if (++spins > (1u << 18)) { timed_out = true; x0 = x1 = v4u32{0u, 0u, 0u, 0u}; break; }
                        __builtin_amdgcn_s_sleep(4);
                    }
                    d_acc = as_double(x0.x, x0.z); d_nan = as_double(x1.x, x1.z);
                }
                my_acc = (long long)d_acc;
                const int nanf = (int)d_nan;
                if (nanf > 0) atomicMin(&s_i[4], tid * KPB + nanf - 1);
            }
            long long incl = my_acc;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const long long o = __shfl_up(incl, off); if (lane >= off) incl += o; }
            if (lane == 63) s_wacc[wave] = incl;
            __syncthreads();
            {
                long long before = incl - my_acc, total = 0;
                for (int w = 0; w < WPB; ++w) { const long long t = s_wacc[w]; total += t; if (w < wave) before += t; }
                if (total >= cut_max && tid < nbk && before < cut_max && before + my_acc >= cut_max) {
                    s_i[1] = tid;
                    s_i[2] = (int)(cut_max - before);
                }
            }
            __syncthreads();
            const int c = s_i[1];
            const bool cut = c < nbk;
            // (c) workgroup c's acceptance mask and records: requested here, looked at after the row sums below (same round trip)
            const unsigned long long *p_rec = b.rec_granules + (size_t)(cut ? c : 0) * KPB * 16 + (tid < KPB * 16 ? tid : 0);
            const unsigned long long *p_msk = b.granules + (size_t)(cut ? c : 0) * SRL_ROW_GRANULES + 64 + (tid & 7);
            unsigned long long x_rec = 0ull, x_msk = 0ull;
            if (cut) { x_rec = ld(p_rec); x_msk = ld(p_msk); }
            // (b)
            const int comp = tid & 31, part = tid >> 5;
            double s0 = 0.0;
            for (int r0 = part; r0 < nbk; r0 += NPART * INF) {
                unsigned lo[INF], hi[INF];
                unsigned spins = 0;
                for (;;) {
                    bool ok = true;
#pragma unroll
                    for (int k = 0; k < INF; ++k) {
                        const int r = r0 + NPART * k;
                        if (r < nbk - 1 && (comp >= 28 || r < c)) {
                            const v4u32 x = row_load(rs, r, comp);
                            ok = ok && x.y == epoch && x.w == epoch;
                            lo[k] = x.x; hi[k] = x.z;
                        } else if (r == nbk - 1 && (comp >= 28 || r < c)) {        // the own row: from LDS
                            const unsigned long long ob = (unsigned long long)__double_as_longlong(s_own[comp]);
                            lo[k] = (unsigned)ob; hi[k] = (unsigned)(ob >> 32);
                        } else { lo[k] = 0u; hi[k] = 0u; }
                    }
                    if (ok) break;
                    if (++spins > (1u << 18)) { timed_out = true; break; }
                    __builtin_amdgcn_s_sleep(8);
                }
#pragma unroll
                for (int k = 0; k < INF; ++k) s0 += as_double(lo[k], hi[k]);
            }
            if (cut) {
                unsigned spins = 0;
                while (!(fresh(x_rec) && fresh(x_msk))) {
                    if (++spins > (1u << 18)) { timed_out = true; x_rec = 0ull; x_msk = 0ull; break; }
                    __builtin_amdgcn_s_sleep(8);
                    x_rec = ld(p_rec); x_msk = ld(p_msk);
                }
                if (tid < 8) s_i[8 + tid] = (int)(unsigned)x_msk;
                const unsigned hi = __shfl_down((unsigned)x_rec, 1);            // granule 2d + half of keypoint tid >> 4
                if (tid < KPB * 16 && (tid & 1) == 0) s_recd[tid >> 1] = as_double((unsigned)x_rec, hi);
            }
            if (timed_out) atomicOr(&s_i[0], 1);
            s_part[part * 32 + comp] = s0;
            __syncthreads();
            if (tid < 32) {
                double sum = s_part[tid];
                for (int p = 1; p < NPART; ++p) sum += s_part[p * 32 + tid];
                if (tid < 28 && cut) {
                    // the stop keypoint: the `allowed`-th accepted one of workgroup c
                    int need = s_i[2], pos = -1;
                    for (int w = 0; w < 8 && pos < 0; ++w) {
                        unsigned m = (unsigned)s_i[8 + w];
                        const int pc = __popc(m);
                        if (need > pc) { need -= pc; continue; }
                        for (int k = 1; k < need; ++k) m &= m - 1u;
                        pos = w * 32 + (int)__builtin_ctz(m);
                    }
                    if (tid == 0) s_i[3] = pos;
                    int ia = 0, ib = 0;
                    if (tid < 21) { int cc = tid; int rowlen = 6; while (cc >= rowlen) { cc -= rowlen; ia++; rowlen--; } ib = ia + cc; }
                    else if (tid < 27) ia = tid - 21;
                    // keypoints that were not accepted published all-zero records: they add +0.0, i.e. nothing, so the walk needs no
                    // mask test and its LDS reads do not depend on anything (the same sum as srl_reduce_kernel's, which skips them)
                    double accd = 0.0;
                    if (tid < 21) {
#pragma unroll 8
                        for (int k = 0; k <= pos; ++k) accd += s_recd[k * 8 + ia] * s_recd[k * 8 + ib];
                    } else if (tid < 27) {
#pragma unroll 8
                        for (int k = 0; k <= pos; ++k) accd += s_recd[k * 8 + ia] * (s_recd[k * 8 + 6] * s_recd[k * 8 + 7]);
                    } else {
#pragma unroll 8
                        for (int k = 0; k <= pos; ++k) accd += s_recd[k * 8 + 6] * s_recd[k * 8 + 6];
                    }
                    sum += accd;
                }
                s_part[tid] = sum;
            }
            __syncthreads();
            SrlDevOut *out = reinterpret_cast<SrlDevOut *>(smem + NPART * 32 * 8 + SRL_FUSED_CUT_MAX_KPB * 64 + WPB * 8 + 64);
            if (tid < 21) {
                int ia = 0, cc = tid, rowlen = 6;
                while (cc >= rowlen) { cc -= rowlen; ia++; rowlen--; }
                const int ib = ia + cc;
                put_f(&out->HtH[ia * 6 + ib], s_part[tid]);
                if (ib != ia) put_f(&out->HtH[ib * 6 + ia], s_part[tid]);
            } else if (tid < 27) {
                put_f(&out->Hth[tid - 21], s_part[tid]);
            } else if (tid == 27) {
                put_f(&out->loss, s_part[27]);
            } else if (tid == 32) {
                const long long last_visited = cut ? (long long)c * KPB + s_i[3] : (long long)n_total - 1;
                put_f(&out->d_num_res, cut ? (double)cut_max : s_part[28]);
                put_f(&out->d_total_accepted, s_part[28]);
                put_f(&out->d_sum_pk, s_part[29]);
                put_f(&out->d_nan, (long long)s_i[4] <= last_visited ? 1.0 : 0.0);   // NaN planarity only counts for visited keypoints
                put_f(&out->d_fallback, s_part[31]);
                put_f(&out->d_visited, (double)(last_visited + 1));
                put_f(&out->d_timeout, s_i[0] ? 1.0 : 0.0);
                put_i(&out->last_visited, last_visited);
                put_i(&out->pad, s_i[0] ? 0x7117ll : 0ll);      // time-out marker
            }
            mail_out(out);
        }
        return;
    } else {
        // deterministic: part p sums rows p, p + NPART, ... ascending; parts are then added in order
        constexpr int NT = 64 * WPB, NPART = NT / 32, INF = 8;
        __syncthreads();                                                   // phase-2 LDS reads are done: smem is free
        double *s_part = reinterpret_cast<double *>(smem);                 // [NPART][32]
        int *s_bad = reinterpret_cast<int *>(smem + NPART * 32 * 8);
        if (tid == 0) *s_bad = 0;
        const int comp = tid & 31, part = tid >> 5;
        // Grids of more than two groups of SRL_FUSED_GROUP workgroups (sweeps beyond 128k keypoints) reduce in two levels: one compute unit
        // pulls tagged rows at ~65 GB/s -- the 1 024 rows of a 256k-keypoint pass alone cost the single finisher 8 us (measured: no gain
        // over the separate reduce kernel).  The last workgroup of every group sums the group's rows -- while the later rounds are still
        // computing -- and publishes a "super row"; the grid's last workgroup adds its own group's rows and the super rows before it.
        const bool two_level = (int)gridDim.x > 2 * SRL_FUSED_GROUP;
        const int my_group = two_level ? (int)blockIdx.x / SRL_FUSED_GROUP : 0;
        const int row_lo = my_group * SRL_FUSED_GROUP;
        const int nbk = (int)blockIdx.x + 1;                               // rows row_lo .. blockIdx.x (the own row last: from LDS)
        const bool is_last = blockIdx.x == gridDim.x - 1;
        const __amdgpu_buffer_rsrc_t rs = rows_rsrc(b.granules);
        const double *s_own = reinterpret_cast<const double *>(smem + SRL_OWN_ROW_OFFSET);          // this workgroup's own row (assoc_body)
        double s0 = 0.0;
        bool timed_out = false;
        for (int r0 = row_lo + part; r0 < nbk; r0 += NPART * INF) {
            unsigned lo[INF], hi[INF];
            unsigned spins = 0;
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int k = 0; k < INF; ++k) {
                    const int r = r0 + NPART * k;
                    if (r < nbk - 1) {
                        const v4u32 x = row_load(rs, r, comp);
                        ok = ok && x.y == epoch && x.w == epoch;
                        lo[k] = x.x; hi[k] = x.z;
                    } else if (r == nbk - 1) {                                 // the own row: from LDS
                        const unsigned long long ob = (unsigned long long)__double_as_longlong(s_own[comp]);
                        lo[k] = (unsigned)ob; hi[k] = (unsigned)(ob >> 32);
                    } else { lo[k] = 0u; hi[k] = 0u; }
                }
                if (ok) break;
                if (++spins > (1u << 18)) { timed_out = true; break; }     // ~0.3 s: something died; do not hang the GPU
                __builtin_amdgcn_s_sleep(4);
            }
#pragma unroll
            for (int k = 0; k < INF; ++k) s0 += __longlong_as_double((long long)(((unsigned long long)hi[k] << 32) | lo[k]));
        }
        if (timed_out) atomicOr(s_bad, 1);
        arm_stamp(karg, 16);
        s_part[part * 32 + comp] = s0;
        __syncthreads();
        arm_stamp(karg, 17);
        if (tid < 32) {
            double sum = s_part[tid];
            for (int p = 1; p < NPART; ++p) sum += s_part[p * 32 + tid];
            if (two_level && !is_last) {
                // a group's finisher: the group's sum leaves as a super row (time-out: the row stays stale, the grid's finisher times out on it)
                if (!*s_bad) row_store(rs, SRL_FUSED_MAX_BLOCKS + my_group, tid, epoch, sum);
            } else if (two_level) {
                // the grid's finisher: the super rows of the groups before its own, in group order
                for (int g = 0; g < my_group; ++g) {
                    unsigned spins = 0;
                    v4u32 x;
                    for (;;) {
                        x = row_load(rs, SRL_FUSED_MAX_BLOCKS + g, tid);
                        if (x.y == epoch && x.w == epoch) break;
                        if (++spins > (1u << 18)) { atomicOr(s_bad, 1); x = v4u32{0u, 0u, 0u, 0u}; break; }
                        __builtin_amdgcn_s_sleep(4);
                    }
                    sum += __longlong_as_double((long long)(((unsigned long long)x.z << 32) | x.x));
                }
            }
            s_part[tid] = sum;                                             // row 0 = the totals
        }
        __syncthreads();
        if (two_level && !is_last) return;
        arm_stamp(karg, 18);
        SrlDevOut *out = reinterpret_cast<SrlDevOut *>(smem + NPART * 32 * 8 + 64);
        bool peer_done = false;
        {
            if (b.peer) {
                // sharded sweep with direct peer exchange: wave 0 lays this rank's totals out as SrlDevOut's leading doubles,
                // exchanges them with the other ranks' finishing workgroups and publishes the sum -- still one kernel per pass
                peer_done = true;
                if (tid < 64) {
                    double v = 0.0;
                    if (tid < 36) {
                        const int i0 = tid / 6, i1 = tid % 6;
                        const int ia = i0 < i1 ? i0 : i1, ib = i0 < i1 ? i1 : i0;
                        v = s_part[ia * 6 - (ia * (ia - 1)) / 2 + (ib - ia)];
                    } else if (tid < 42) v = s_part[21 + (tid - 36)];
                    else if (tid == 42) v = s_part[27];
                    else if (tid == 43 || tid == 44) v = s_part[28];
                    else if (tid == 45) v = s_part[29];
                    else if (tid == 46) v = s_part[30] > 0.0 ? 1.0 : 0.0;
                    else if (tid == 47) v = s_part[31];
                    else if (tid == 48) v = (double)n_total;
                    else if (tid == 49) v = *s_bad ? 1.0 : 0.0;
                    double sum = 0.0;
                    const bool ok = peer_exchange(b.peer, b.peer_epoch, b.peer_slot, lane, SRL_REDUCED_DOUBLES, (unsigned long long)__double_as_longlong(v),
                                                  [&](int, unsigned long long w) { sum += __longlong_as_double((long long)w); });
                    double *od = reinterpret_cast<double *>(out);
                    if (tid < SRL_REDUCED_DOUBLES) put_f(od + tid, sum);
                    else if (tid == SRL_REDUCED_DOUBLES) put_i(&out->last_visited, (long long)n_total - 1);
                    else if (tid == SRL_REDUCED_DOUBLES + 1) put_i(&out->pad, ok ? 0ll : SRL_PEER_TIMEOUT_MARK);
                }
            }
        }
        if (peer_done) {
        } else if (tid < 21) {
            int ia = 0, c = tid, rowlen = 6;
            while (c >= rowlen) { c -= rowlen; ia++; rowlen--; }
            const int ib = ia + c;
            put_f(&out->HtH[ia * 6 + ib], s_part[tid]);
            if (ib != ia) put_f(&out->HtH[ib * 6 + ia], s_part[tid]);
        } else if (tid < 27) {
            put_f(&out->Hth[tid - 21], s_part[tid]);
        } else if (tid == 27) {
            put_f(&out->loss, s_part[27]);
        } else if (tid == 32) {
            put_f(&out->d_num_res, s_part[28]);                            // no ordered cut possible here: every accepted residual counts
            put_f(&out->d_total_accepted, s_part[28]);
            put_f(&out->d_sum_pk, s_part[29]);
            put_f(&out->d_nan, s_part[30] > 0.0 ? 1.0 : 0.0);              // every keypoint is visited
            put_f(&out->d_fallback, s_part[31]);
            put_f(&out->d_visited, (double)n_total);
            put_f(&out->d_timeout, *s_bad ? 1.0 : 0.0);
            put_i(&out->last_visited, (long long)n_total - 1);
            put_i(&out->pad, *s_bad ? 0x7117ll : 0ll);          // time-out marker
        }
        arm_stamp(karg, 19);
        mail_out(out);                                                     // every writer of the staged record sits in wave 0
        arm_stamp(karg, 20);
    }
#endif
}

template <int NB, int FAST, int KPW, int WPB, int DBG = 0, int ARMED = 0>
__device__ __forceinline__ void assoc_body(const SrlAssocArgs &a) {
    constexpr int KPB = WPB * KPW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    if (a.ablate & 16) return;                                            // debug: launch/drain floor
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const __attribute__((address_space(4))) SrlAssocArgs *KernargPtr;
#endif
    // the LDS carve is re-derived from max_number_neighbors wherever it is needed: kept in registers its dozen offsets come out of
    // the selection loop's budget (the kernel runs at 121 of 128 VGPRs and at the SGPR limit)
    auto carve = [&]() -> LdsLayout { return lds_layout(a.K, NB, KPW, WPB, ARMED); };
    if constexpr (ARMED) {
#if defined(__HIP_DEVICE_COMPILE__)
        // ---- ARMED launch (srl_capi.cpp: arm_next): this kernel was enqueued while the pass before it was still running, before
        // its pose existed; its workgroups are resident and waiting when the host has finished the 17-dim update, so the launch call,
        // the dispatch and the ramp of this pass are off the per-iteration critical path.  The pose (Rn[9] R[9] t[3], the only
        // arguments that differ from the pass before) arrives through the POSE BOX: 8-byte granules {epoch, 32-bit half} the host
        // writes ("the data is the flag": no ordering between the stores is needed) + one control granule {epoch, GO | CANCEL}.
        // Wave 0 of every workgroup polls: the box itself, or -- when a relay is given (box in host memory: 256 pollers across PCIe
        // would be 256 round trips per poll) -- workgroup 0 polls the box and republishes into device memory for the others.
        // A tag NEWER than this launch's epoch means the host has moved on: cancelled.  The wait is bounded (arm_linger_ticks of
        // the 100 MHz clock; the host never fires a launch that old, so the bound is a safety net, not a protocol step).
        typedef __attribute__((address_space(1))) unsigned long long gu64a;
        const LdsLayout L = carve();
        double *s_pose = reinterpret_cast<double *>(smem + L.off_pose);
        int *s_ctrl = reinterpret_cast<int *>(s_pose + SRL_POSE_DOUBLES);
        arm_stamp((KargBytes)__builtin_amdgcn_kernarg_segment_ptr(), 0);
        {
            // what phase 0 can do without the pose is done now: raw point -> body frame (optimize.cpp:83), every lane for the keypoint
            // it owns in phase 0 (it reads its own LDS words back there: no barrier in between)
            const int lane0 = tid & 63, wave0 = tid >> 6;
            if (lane0 < KPW) {
                const int kq = wave0 * KPW + lane0;
                const int g = (int)blockIdx.x * KPB + kq;
                double *s_pimu = reinterpret_cast<double *>(smem + L.off_pimu);
                D3 p_imu = d3(0, 0, 0);
                if (g < a.n) p_imu = add(matvec(a.R_il, d3(a.raw_x[g], a.raw_y[g], a.raw_z[g])), d3(a.t_il[0], a.t_il[1], a.t_il[2]));
                s_pimu[kq * 3 + 0] = p_imu.x; s_pimu[kq * 3 + 1] = p_imu.y; s_pimu[kq * 3 + 2] = p_imu.z;
            }
        }
        if (tid < 64) {
            const int lane = tid;
            const unsigned epoch = a.pose_epoch;
            const bool relayed = a.pose_relayed != 0;
            const bool leader = !relayed || blockIdx.x == 0;
            const bool act = lane < SRL_POSE_BOX_USED;
            const unsigned long long *src = (leader ? a.pose_box : (const unsigned long long *)a.pose_relay) + (act ? lane : 0);
            const long long t0 = (long long)wall_clock64();
            const long long limit = leader ? (long long)a.arm_linger_ticks : 2 * (long long)a.arm_linger_ticks + 100000;
            unsigned long long x = 0ull;
            unsigned code = 0u;
            // a grid of more workgroups than the chip holds runs in rounds: a workgroup of a later round starts when the launch has long been
            // fired (the box holds its pose: the first poll succeeds) -- or when an earlier round gave up waiting: its note in the relay ends
            // this one at once (the bound of a waiting launch holds for the launch, not per round)
            const unsigned long long note = relayed ? 0ull : __hip_atomic_load((gu64a *)(a.pose_relay + SRL_POSE_BOX_CTRL), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((unsigned)(note >> 32) == epoch && (unsigned)note == SRL_ARM_EXPIRED) code = SRL_ARM_EXPIRED;
            else for (;;) {
                x = leader ? __hip_atomic_load((gu64a *)src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
                           : __hip_atomic_load((gu64a *)src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned tag = (unsigned)(x >> 32);
                if (__ballot(act && tag != epoch) == 0ull) { code = (unsigned)__shfl((unsigned)x, SRL_POSE_BOX_CTRL); break; }
                if (__ballot(act && (int)(tag - epoch) > 0) != 0ull) { code = SRL_ARM_CANCEL; break; }
                if ((long long)wall_clock64() - t0 > limit) { code = SRL_ARM_EXPIRED; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            if (!relayed && code == SRL_ARM_EXPIRED && lane == 0)
                __hip_atomic_store((gu64a *)(a.pose_relay + SRL_POSE_BOX_CTRL), ((unsigned long long)epoch << 32) | SRL_ARM_EXPIRED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (relayed && leader && act) {
                // the verdict of the poll for the other workgroups: the pose as it arrived, or a control granule that ends them
                unsigned long long y = x;
                // (a launch fired for the swapped-in sweep carries GO | ALT: the flag travels on with the pose)
                if ((code & SRL_ARM_CODE_MASK) != SRL_ARM_GO) y = ((unsigned long long)epoch << 32) | (lane == SRL_POSE_BOX_CTRL ? code : 0u);
                __hip_atomic_store((gu64a *)(a.pose_relay + lane), y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            const unsigned hi = (unsigned)__shfl_down((unsigned)x, 1);
            // granules 2d, 2d + 1 -> pose double d (d < 21); granules 44..49 -> t_last = doubles 21..23
            if (!(lane & 1) && (lane < SRL_POSE_BOX_CTRL || (lane >= SRL_POSE_BOX_TLAST && lane < SRL_POSE_BOX_USED)))
                s_pose[lane < SRL_POSE_BOX_CTRL ? (lane >> 1) : 21 + ((lane - SRL_POSE_BOX_TLAST) >> 1)] = __longlong_as_double((long long)(((unsigned long long)hi << 32) | (unsigned)x));
            if (lane == 0) { s_ctrl[0] = (int)(code & SRL_ARM_CODE_MASK); s_ctrl[2] = (code & SRL_ARM_ALT) ? 1 : 0; }
            if (lane == SRL_POSE_BOX_N) s_ctrl[1] = (code & SRL_ARM_CODE_MASK) == SRL_ARM_GO ? (int)(unsigned)x : a.n;     // keypoints of this pass
        }
        __syncthreads();
        arm_stamp((KargBytes)__builtin_amdgcn_kernarg_segment_ptr(), 1);
        {
            const int code = *s_ctrl;
            if (code == (int)SRL_ARM_GO) {
                // Fired for the context's OTHER sweep buffer (srl_sweep_swap kept this launch: it is the first pass of the NEXT sweep) or with
                // another keypoint count: the body-frame points precomputed above belong to the sweep this launch was armed on -- redo them.
                // (Same lane, same LDS words as above and as phase 0 reads back: no barrier.)
                const int n_new = s_ctrl[1];
                const bool alt = s_ctrl[2] != 0;
                if (alt || n_new != a.n) {
                    const int lane0 = tid & 63, wave0 = tid >> 6;
                    if (lane0 < KPW) {
                        const int kq = wave0 * KPW + lane0;
                        const int g = (int)blockIdx.x * KPB + kq;
                        double *s_pimu = reinterpret_cast<double *>(smem + L.off_pimu);
                        D3 p_imu = d3(0, 0, 0);
                        if (g < n_new) {
                            D3 raw;
                            if (alt) {
                                // the other buffer's points arrived AoS by DMA (srl_sweep_prefetch): this pass files its SoA planes (SrlAssocArgs::aos)
                                // (system-scope loads: this kernel was resident BEFORE the DMA that wrote them had finished -- no kernel-boundary
                                //  acquire lies between the copy engine's write and these reads, so they must not be served by a cache line an
                                //  earlier reader of the same staging buffer left behind)
                                typedef const __attribute__((address_space(1))) double gf64c;
                                gf64c *p = (gf64c *)(a.alt_aos + 3 * (size_t)g);
                                raw = d3(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM),
                                         __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM),
                                         __hip_atomic_load(p + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
                                const_cast<double *>(a.alt_x)[g] = raw.x; const_cast<double *>(a.alt_y)[g] = raw.y; const_cast<double *>(a.alt_z)[g] = raw.z;
                            } else {
                                raw = d3(a.raw_x[g], a.raw_y[g], a.raw_z[g]);
                            }
                            p_imu = add(matvec(a.R_il, raw), d3(a.t_il[0], a.t_il[1], a.t_il[2]));
                        }
                        s_pimu[kq * 3 + 0] = p_imu.x; s_pimu[kq * 3 + 1] = p_imu.y; s_pimu[kq * 3 + 2] = p_imu.z;
                    }
                }
            }
            if (code != (int)SRL_ARM_GO) {
                if (code == (int)SRL_ARM_EXPIRED && blockIdx.x == gridDim.x - 1 && tid == 0) {
                    // nobody is listening any more: a host that fires this launch after all learns it from the mailbox's `expired` word and
                    // relaunches.  (NOT through the result record: this launch leaves on its own schedule, possibly while the host -- descheduled
                    // for longer than the bound -- has not yet read the result of the pass before it; found by the 10^6-launch soak.)
                    __hip_atomic_store(&a.mailbox->expired, a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                if (code == (int)SRL_ARM_EXPIRED && blockIdx.x == gridDim.x - 1 && tid < 64 && !a.mail_tagged && a.peer == nullptr) {
                    // the RCCL form (plain record in DEVICE memory, an all-reduce behind this kernel on the stream): the host cannot see the
                    // word above, and the collective runs whatever this launch did -- so the launch contributes an empty record whose
                    // time-out flag is part of the reduced range: EVERY rank sees it in the sum and repeats the pass, together
                    constexpr int NW = (int)(sizeof(SrlDevOut) / 8);
                    unsigned long long w = 0ull;
                    if (tid == (int)(offsetof(SrlDevOut, d_timeout) / 8)) w = (unsigned long long)__double_as_longlong(1.0);
                    if (tid == (int)(offsetof(SrlDevOut, pad) / 8)) w = 0x7117ull;
                    if (tid == (int)(offsetof(SrlDevOut, last_visited) / 8)) w = ~0ull;
                    // (system-scope stores like mail_out's: plain stores of this early-exit path were observed NOT to reach the all-reduce's read --
                    //  the record word written through, d_mail->expired above, arrived; the plain ones did not)
                    if (tid < NW) __hip_atomic_store(reinterpret_cast<unsigned long long *>(&a.mailbox->out) + tid, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                return;
            }
        }
#endif
    }
    if (assoc_tile<NB, FAST, KPW, WPB, DBG, ARMED>((KargBytes)__builtin_amdgcn_kernarg_segment_ptr(), a, (int)blockIdx.x)) return;
    double row_v = 0.0;                                                   // tid < 32: component tid of this workgroup's row
    constexpr int P2W_T = (KPB + p2_keypoints_per_wave(KPB) - 1) / p2_keypoints_per_wave(KPB);
    // ---- the workgroup's row: 28 partial sums + {accepted, candidates visited, NaN flag,
    // off-fast-path keypoints} carried as doubles (threads 0..31)
    if (tid < 32) {
        const LdsLayout L = carve();
        const double *s_wpart = reinterpret_cast<const double *>(smem + L.off_wpart);     // [P2W][32]
        const int *s_winfo = reinterpret_cast<const int *>(smem + L.off_winfo);           // [WPB][8]: accepted, sum_pk, 1 + first NaN keypoint, fallback, planes
        double v = 0.0;
        if (tid < 28) {
            v = s_wpart[tid];
#pragma unroll
            for (int w = 1; w < P2W_T; ++w) v += s_wpart[w * 32 + tid];
        } else {
            int acc = 0, pk = 0, nanf = 0, fb = 0;
            for (int w = 0; w < WPB; ++w) fb += s_winfo[w * 8 + 3];
            // nanf: 1 + index inside the tile of its first NaN-planarity keypoint (slots are in keypoint order), 0 = none
            for (int w = 0; w < P2W_T; ++w) { acc += s_winfo[w * 8 + 0]; pk += s_winfo[w * 8 + 1]; if (nanf == 0) nanf = s_winfo[w * 8 + 2]; }
            v = tid == 28 ? (double)acc : (tid == 29 ? (double)(unsigned)pk : (tid == 30 ? (double)nanf : (double)fb));
        }
        row_v = v;
    }
#if defined(__HIP_DEVICE_COMPILE__)
    KernargPtr bq = (KernargPtr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(bq));
    const __attribute__((address_space(4))) SrlAssocArgs &b = *bq;
#else
    const SrlAssocArgs &b = a;
#endif
    const LdsLayout L = lds_layout(b.K, NB, KPW, WPB, ARMED);
    const int *s_winfo = reinterpret_cast<const int *>(smem + L.off_winfo);

    // ---------------- fused final reduction: every workgroup publishes its row, the LAST workgroup of the grid finishes.
    // Row = 28 partial sums + {accepted, candidates visited, NaN flag, off-fast-path keypoints} carried as doubles.
    // Hand-off across the 8 XCDs (private L2s) in the "data is the flag" form of guide G16 (R2): every double travels as
    // two 8-byte granules {epoch, 32-bit half}, stored write-through at agent scope -- ONE store instruction per workgroup,
    // no drain, no counter, no fence; the finisher re-reads the granules it needs (agent-scope loads) until every tag
    // carries this launch's epoch, then sums the rows in a fixed order.  (Round-2 first version: drained row + two-level
    // arrival counters = three dependent memory round trips behind the last workgroup, +5.6 us on the 64k launch.)
    // A stale granule has an older epoch (the epoch is the context's launch sequence number), so nothing is reset
    // between launches.  The finisher only waits for results every other workgroup produces without it: no co-residency
    // assumption; its spin is bounded (time-out marker in the mailbox, the host turns it into an error).
    typedef __attribute__((address_space(1))) unsigned long long gu64;
    const unsigned epoch = (unsigned)b.seq;
    constexpr int KP2 = p2_keypoints_per_wave(KPB);                       // keypoints per phase-2 wave (as in phase 2)
    constexpr int P2W = (KPB + KP2 - 1) / KP2;
    if (tid < 32) {
        // lane l publishes component l of the row (row_store: one 16-byte write-through store); the finishing workgroup keeps its own
        // row in LDS -- read back through memory it was the row the finisher's first poll always missed
#if defined(__HIP_DEVICE_COMPILE__)
        const bool finishes = blockIdx.x == gridDim.x - 1 ||
                              (b.cut_max == 0 && (int)gridDim.x > 2 * SRL_FUSED_GROUP && ((int)blockIdx.x % SRL_FUSED_GROUP) == SRL_FUSED_GROUP - 1);
        if (finishes) reinterpret_cast<double *>(smem + SRL_OWN_ROW_OFFSET)[tid] = row_v;
        else row_store(rows_rsrc(b.granules), (int)blockIdx.x, tid, epoch, row_v);
#endif
    } else if (tid >= 64 && tid < 72 && b.cut_max > 0) {
        // granules 64..71: which keypoints of this workgroup were accepted (bit i = keypoint i), 32 per granule
        const int j = tid - 64;
        unsigned word = 0u;
        if (KP2 >= 32) {
            const int w = (32 * j) / KP2;
            if (w < P2W) word = (unsigned)s_winfo[w * 8 + 5 + (((32 * j) % KP2) >> 5)];
        } else {
#pragma unroll
            for (int t = 0; t < 32 / KP2; ++t) {
                const int w = (32 * j) / KP2 + t;
                if (w < P2W) word |= (unsigned)s_winfo[w * 8 + 5] << (t * KP2);
            }
        }
        __hip_atomic_store((gu64 *)(b.granules + (size_t)blockIdx.x * SRL_ROW_GRANULES + tid), ((unsigned long long)epoch << 32) | word,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if constexpr (ARMED) arm_stamp((KargBytes)__builtin_amdgcn_kernarg_segment_ptr(), 6);
    // the grid's last workgroup finishes; in a grid of more than two groups also the last workgroup of every group (two-level reduction)
    const bool group_finisher = b.cut_max == 0 && (int)gridDim.x > 2 * SRL_FUSED_GROUP && ((int)blockIdx.x % SRL_FUSED_GROUP) == SRL_FUSED_GROUP - 1;
    if (blockIdx.x != gridDim.x - 1 && !group_finisher) return;
    int n_total = b.n;
    if constexpr (ARMED) n_total = __builtin_amdgcn_readfirstlane(reinterpret_cast<const int *>(smem + L.off_pose + SRL_POSE_DOUBLES * 8)[1]);   // the count that came with the pose
    finish_rows<KPW, WPB>((KargBytes)__builtin_amdgcn_kernarg_segment_ptr(), epoch, n_total);
    if constexpr (ARMED) arm_stamp((KargBytes)__builtin_amdgcn_kernarg_segment_ptr(), 7);
}

template <int NB, int FAST, int KPW, int WPB, int DBG = 0>
__global__ void __launch_bounds__(64 * WPB, WPB == 16 ? 1 : SRL_ASSOC_WAVES_PER_SIMD) srl_assoc_kernel(const SrlAssocArgs a) {
    assoc_body<NB, FAST, KPW, WPB, DBG>(a);
}
// the same pass as an ARMED launch (16-wave workgroups, fast paths): enqueued before its pose exists, the pose arrives through
// the pose box (assoc_body's prologue); everything behind the prologue is the one-shot kernel
template <int NB, int KPW>
__global__ void __launch_bounds__(1024, 1) srl_assoc_armed_kernel(const SrlAssocArgs a) {
    assoc_body<NB, 1, KPW, 16, 0, 1>(a);
}
// ---------------------------------------------------------------------------------------------
// ordered cut-off + final reduction (single workgroup)
// mode: 0 = budget max_res >= 1; 1 = max_num_residuals <= 0: the loop stops at the first keypoint that reaches the
// break test (optimize.cpp:107 sits behind the `continue` of :78-79, so keypoints with too few neighbours are passed
// over); 2 = visit nothing (budget spent, or the stop keypoint found, in earlier shards).
// With a.gather the kernel derives mode and budget itself from the per-rank counts gathered on the stream.
// NaN planarity (optimize.cpp:348-350) is an error only for keypoints the sequential loop reaches (<= last_visited).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) srl_reduce_kernel(const SrlReduceArgs a, int mode_in) {
    __shared__ long long s_wtot[16];        // accepted residuals per wave (inclusive prefix input)
    __shared__ double s_rec[256 * 8];       // records of the re-accumulated keypoints (<= one workgroup of the association pass)
    __shared__ unsigned char s_stat[256];
    __shared__ int s_wcnt[4];
    __shared__ int s_first;                 // mode 1: first keypoint with a plane
    __shared__ double s_part[32][SRL_PART_STRIDE];
    __shared__ long long s_tot[4];          // total accepted, sum_pk, (unused), fallback
    __shared__ int s_cut[4];                // cut block, allowed in cut block, last visited local idx, num_res
    __shared__ int s_nan_min;               // smallest keypoint index with NaN planarity
    const int tid = threadIdx.x;
    const int nb = a.nblocks;

    int mode = mode_in;
    long long max_res = a.max_res;
    if (a.gather) {
        long long prior = 0;
        for (int r = 0; r < a.rank; ++r) prior += a.gather[r];
        if (a.max_num_residuals > 0) {
            max_res = (long long)a.max_num_residuals - prior;      // what the shards before this one left of the budget
            mode = max_res <= 0 ? 2 : 0;
        } else {
            mode = prior > 0 ? 2 : 1;                              // gathered: keypoints with a plane per rank
        }
    }

    // integer totals
    long long acc = 0, pk = 0, fb = 0;
    int nan_min = 0x7fffffff;
    const int per = (nb + 1023) / 1024;
    const int b0 = tid * per;
    const int b1 = (b0 + per < nb) ? b0 + per : nb;
    for (int b = b0; b < b1; ++b) {
        const SrlBlockInfo bi = a.binfo[b];
        acc += bi.accepted; pk += bi.sum_pk; fb += bi.num_fallback;
        if (bi.nan_first > 0) { const int g = b * a.kpb + bi.nan_first - 1; nan_min = g < nan_min ? g : nan_min; }
    }
    if (tid < 4) s_tot[tid] = 0;
    if (tid == 0) { s_nan_min = 0x7fffffff; s_first = 0x7fffffff; }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    // inclusive prefix of the per-thread accepted counts over the wave (thread t owns blocks [t * per, (t + 1) * per))
    long long incl = acc;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const long long o = __shfl_up(incl, off); if (lane >= off) incl += o; }
    if (lane == 63) s_wtot[wave] = incl;
    {   // wave-level sums first (integers: order irrelevant), then one LDS atomic per wave and counter
        long long w1 = pk, w3 = fb;
        int w2 = nan_min;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            w1 += __shfl_xor(w1, off); w3 += __shfl_xor(w3, off);
            const int o2 = __shfl_xor(w2, off); w2 = o2 < w2 ? o2 : w2;
        }
        if (lane == 63) {
            if (incl) atomicAdd((unsigned long long *)&s_tot[0], (unsigned long long)incl);
            if (w1) atomicAdd((unsigned long long *)&s_tot[1], (unsigned long long)w1);
            if (w2 != 0x7fffffff) atomicMin(&s_nan_min, w2);
            if (w3) atomicAdd((unsigned long long *)&s_tot[3], (unsigned long long)w3);
        }
    }
    __syncthreads();

    // Where does the sequential loop stop (optimize.cpp:107)?  All 1 024 threads take part: no serial chain of dependent
    // global loads (round-2 first version: one thread walked chunk sums, block counts and the cut block's status bytes one
    // load after the other -- 12.8 us for the 70-workgroup prefix pass of the shipped max_num_residuals = 600).
    const long long total = s_tot[0];
    if (tid == 0) {
        s_cut[0] = nb; s_cut[1] = 0; s_cut[2] = a.n - 1; s_cut[3] = (int)total;          // no cut: everything is visited
        if (mode == 2) { s_cut[0] = 0; s_cut[2] = -1; s_cut[3] = 0; }
    }
    __syncthreads();
    if (mode == 0 && total >= max_res) {
        // the thread whose blocks hold the max_res-th accepted residual finds the block (exactly one: the prefix is monotone)
        long long before = incl - acc;
        for (int w = 0; w < wave; ++w) before += s_wtot[w];
        if (before < max_res && before + acc >= max_res) {
            int b = b0;
            while (b < b1 && before + a.binfo[b].accepted < max_res) { before += a.binfo[b].accepted; ++b; }
            s_cut[0] = b;
            s_cut[1] = (int)(max_res - before);
            s_cut[3] = (int)max_res;
        }
    } else if (mode == 1) {
        // first keypoint with a plane (status 1 or 2): it is the last one visited, and the only possible residual
        for (int base = 0; base < a.n; base += 1024) {
            const int k = base + tid;
            const bool hit = k < a.n && (a.status[k] == 1 || a.status[k] == 2);
            if (hit) atomicMin(&s_first, k);
            if (__syncthreads_or(hit ? 1 : 0)) break;
        }
        if (tid == 0) {
            const int k = s_first;
            s_cut[0] = 0; s_cut[1] = 0;
            s_cut[2] = (k < a.n) ? k : a.n - 1;
            s_cut[3] = (k < a.n && a.status[k] == 2) ? 1 : 0;
        }
    }
    __syncthreads();
    const int cut_block = s_cut[0];
    if (mode == 0 && cut_block < nb) {
        // the keypoint of the cut block that holds its `allowed`-th accepted residual: one status byte per thread, ballot prefix
        const int k0 = cut_block * a.kpb;
        const int kend = (k0 + a.kpb < a.n) ? k0 + a.kpb : a.n;
        const int allowed = s_cut[1];
        const int k = k0 + tid;
        const bool is2 = tid < a.kpb && k < kend && a.status[k] == 2;        // kpb <= 256: waves 0..3
        const unsigned long long m = __ballot(is2);
        const int upto = __popcll(m & ((2ull << lane) - 1ull));              // accepted among lanes 0..lane of this wave
        if (lane == 0 && wave < 4) s_wcnt[wave] = __popcll(m);
        if (tid == 0) s_cut[2] = kend;                                       // (not reached: the block holds >= allowed accepted keypoints)
        __syncthreads();
        if (wave < 4 && is2) {
            int pre = 0;
            for (int w = 0; w < wave; ++w) pre += s_wcnt[w];
            if (pre + upto == allowed) s_cut[2] = k;
        }
        __syncthreads();
    }
    const int last_visited = s_cut[2];
    // records of the keypoints that are re-accumulated one by one (the cut block up to the stop keypoint; mode 1: the stop
    // keypoint, nothing before it has a plane): staged in LDS by all threads
    int nrows = 0;
    if (mode != 2 && cut_block < nb) {
        const int row0 = mode == 1 ? last_visited : cut_block * a.kpb;
        nrows = last_visited - row0 + 1;
        nrows = nrows < 0 ? 0 : (nrows > 256 ? 256 : nrows);
        for (int i = tid; i < nrows * 8; i += 1024) s_rec[i] = a.rec[(size_t)row0 * 8 + i];
        if (tid < nrows) s_stat[tid] = a.status[row0 + tid];
    }

    // deterministic sum of the partials of all blocks strictly before the cut block:
    // part p sums blocks b == p (mod 32) ascending; parts are then added in order 0..31.
    // 32 independent coalesced loads are in flight per thread before the first add (one memory round trip).
    {
        const int comp = tid & 31, part = tid >> 5;
        double s0 = 0.0;
        for (int b0 = part; b0 < cut_block; b0 += 32 * 32) {
            double v[32];
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                const int b = b0 + 32 * k;
                v[k] = (b < cut_block) ? a.partials[(size_t)b * SRL_PART_STRIDE + comp] : 0.0;
            }
#pragma unroll
            for (int k = 0; k < 32; ++k) s0 += v[k];
        }
        s_part[part][comp] = s0;
    }
    __syncthreads();
    // results: to the host-mapped mailbox with system-scope (write-through) stores when given, else to device memory
    SrlDevOut *out = a.mailbox ? &a.mailbox->out : a.out;
    const bool to_host = a.mailbox != nullptr;
    auto put_f = [to_host](double *p, double x) {
        if (to_host) __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); else *p = x;
    };
    auto put_i = [to_host](long long *p, long long x) {
        if (to_host) __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); else *p = x;
    };
    if (tid < 28) {
        double s = s_part[0][tid];
        for (int p = 1; p < 32; ++p) s += s_part[p][tid];
        // cut block (or the keypoints up to the stop keypoint in mode 1): re-accumulate from the records, in order
        if (mode != 2 && cut_block < nb) {
            int ia = 0, ib = 0;
            if (tid < 21) { int c = tid; int rowlen = 6; while (c >= rowlen) { c -= rowlen; ia++; rowlen--; } ib = ia + c; }
            else if (tid < 27) ia = tid - 21;
            double accd = 0.0;
            for (int k = 0; k < nrows; ++k) {
                if (s_stat[k] != 2) continue;
                const double *r = s_rec + k * 8;
                if (tid < 21) accd += r[ia] * r[ib];
                else if (tid < 27) accd += r[ia] * (r[6] * r[7]);
                else accd += r[6] * r[6];
            }
            s += accd;
        }
        if (tid < 21) {
            int ia = 0, c = tid, rowlen = 6;
            while (c >= rowlen) { c -= rowlen; ia++; rowlen--; }
            const int ib = ia + c;
            put_f(&out->HtH[ia * 6 + ib], s);
            if (ib != ia) put_f(&out->HtH[ib * 6 + ia], s);
        } else if (tid < 27) {
            put_f(&out->Hth[tid - 21], s);
        } else {
            put_f(&out->loss, s);
        }
    }
    if (tid == 32) {
        put_f(&out->d_num_res, (double)s_cut[3]);
        put_f(&out->d_total_accepted, (double)s_tot[0]);
        put_f(&out->d_sum_pk, (double)s_tot[1]);
        put_f(&out->d_nan, (s_nan_min <= last_visited) ? 1.0 : 0.0);
        put_f(&out->d_fallback, (double)s_tot[3]);
        put_f(&out->d_visited, (double)(last_visited + 1));
        put_f(&out->d_timeout, 0.0);
        put_i(&out->last_visited, (long long)last_visited);
        put_i(&out->pad, 0);
    }
    if (to_host && tid < 64) {
        // every writer sits in wave 0: drain the write-through stores, then publish the sequence word
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tid == 0) __hip_atomic_store(&a.mailbox->seq, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// per-rank count for the ordered cut across shards: accepted residuals, or (max_num_residuals <= 0) keypoints with a plane
__global__ void __launch_bounds__(256) srl_count_kernel(const SrlBlockInfo *binfo, int nblocks, int count_planes, long long *out_total) {
    __shared__ long long s[256];
    long long acc = 0;
    for (int b = threadIdx.x; b < nblocks; b += 256) acc += count_planes ? binfo[b].planes : binfo[b].accepted;
    s[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x == 0) { long long t = 0; for (int i = 0; i < 256; ++i) t += s[i]; *out_total = t; }
}

// ---------------------------------------------------------------------------------------------
// searchNeighbors for a batch of world points (one wave per query)
// ---------------------------------------------------------------------------------------------
struct GlobalSink {
    int *ids;
    float *xyz;
    __device__ __forceinline__ void put(int rank, float x, float y, float z, unsigned id) {
        ids[rank] = (int)id;
        if (xyz) { xyz[rank * 3 + 0] = x; xyz[rank * 3 + 1] = y; xyz[rank * 3 + 2] = z; }
    }
};

template <int NB>
__global__ void __launch_bounds__(SRL_BLOCK) srl_search_kernel(const SrlSearchArgs a) {
    __shared__ __attribute__((aligned(16))) VoxEnt s_vox[4][128];
    __shared__ __attribute__((aligned(16))) unsigned char s_scr[4][SRL_WAVE_SCRATCH];     // survivors [64] + sorted [K + 1]
    const int lane = lane_id();
    const int wave = threadIdx.x >> 6;
    const int nwaves = gridDim.x * 4;
    for (int q = blockIdx.x * 4 + wave; q < a.n; q += nwaves) {
        const double qx = a.q[(size_t)q * 3 + 0], qy = a.q[(size_t)q * 3 + 1], qz = a.q[(size_t)q * 3 + 2];
        const int nv = probe_voxels<NB>(qx, qy, qz, a.size_voxel, a.thr_cap, a.table, a.table_mask, s_vox[wave], lane);
        GlobalSink sink;
        sink.ids = a.ids + (size_t)q * a.K;
        sink.xyz = a.nb_xyz ? a.nb_xyz + (size_t)q * a.K * 3 : nullptr;
        int total = 0, fb = 0;
        select_topk(qx, qy, qz, nv, s_vox[wave], a.slabs, a.K, a.select_mode, reinterpret_cast<Surv *>(s_scr[wave]), lane, sink, total, fb);
        if (lane == 0) a.num_found[q] = total < a.K ? total : a.K;
    }
}

// transformPoint (utility.cpp:314-318) for the post-solve loop (optimize.cpp:441-445)
__global__ void srl_transform_kernel(const double *raw, int n, const SrlXform X, double *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const D3 r = d3(raw[(size_t)i * 3], raw[(size_t)i * 3 + 1], raw[(size_t)i * 3 + 2]);
    const D3 pi = add(matvec(X.R_il, r), d3(X.t_il[0], X.t_il[1], X.t_il[2]));
    const D3 pw = add(matvec(X.R, pi), d3(X.t[0], X.t[1], X.t[2]));
    out[(size_t)i * 3] = pw.x; out[(size_t)i * 3 + 1] = pw.y; out[(size_t)i * 3 + 2] = pw.z;
}

__global__ void srl_aos_to_soa_kernel(const double *aos, int n, double *x, double *y, double *z) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    x[i] = aos[(size_t)i * 3];
    y[i] = aos[(size_t)i * 3 + 1];
    z[i] = aos[(size_t)i * 3 + 2];
}

// debug: the device's sqrt(double) (the heap replay relies on it being correctly rounded; tests compare with the host's)
__global__ void srl_sqrt_kernel(double *io, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) io[i] = sqrt(io[i]);
}

}  // namespace

hipError_t srl_launch_sqrt(double *io, int n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(srl_sqrt_kernel, dim3((n + 255) / 256), dim3(256), 0, s, io, n);
    return hipGetLastError();
}

template <int KPW, int WPB>
static hipError_t launch_assoc_cfg(const SrlAssocArgs &a, int nb_voxels, hipStream_t s) {
    const int nblocks = (a.n + WPB * KPW - 1) / (WPB * KPW);
    const bool armed = a.pose_box != nullptr;
    const LdsLayout L = lds_layout(a.K, nb_voxels, KPW, WPB, armed ? 1 : 0);
    const dim3 blk(64 * WPB);
    auto launch = [&](auto kern) {
        if (L.total > 64 * 1024) {      // more dynamic LDS than the default limit: opt in (once per kernel, device and size)
            static thread_local const void *done_fn = nullptr;
            static thread_local int done_bytes = 0, done_dev = -1;
            int dev = -1;
            hipGetDevice(&dev);
            const void *fn = reinterpret_cast<const void *>(kern);
            if (fn != done_fn || L.total > done_bytes || dev != done_dev) {
                hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, SRL_LDS_LIMIT);
                if (e != hipSuccess) return e;
                done_fn = fn; done_bytes = SRL_LDS_LIMIT; done_dev = dev;
            }
        }
        hipLaunchKernelGGL(kern, dim3(nblocks), blk, L.total, s, a);
        return hipGetLastError();
    };
    // select_mode 0 / 4: fast paths; 1, 2, 5: general path only
    const bool fast = a.select_mode == 0 || a.select_mode == 4;
    if (armed) {
        if constexpr (WPB == 16) { if (fast && a.ablate == 0) return nb_voxels == 1 ? launch(srl_assoc_armed_kernel<1, KPW>) : launch(srl_assoc_armed_kernel<2, KPW>); }
        return hipErrorInvalidConfiguration;      // the host arms only what has an armed instantiation
    }
    // the debug switches exist in ONE instantiation (r = 1 fast path, 16 x 16 keypoints per workgroup: the host forces that shape
    // while srl_debug_set_ablate is non-zero); everywhere else a non-zero a.ablate is ignored
    if constexpr (KPW == 16 && WPB == 16) { if (a.ablate != 0 && nb_voxels == 1 && fast) return launch(srl_assoc_kernel<1, 1, 16, 16, 1>); }
    if (nb_voxels == 1) return fast ? launch(srl_assoc_kernel<1, 1, KPW, WPB>) : launch(srl_assoc_kernel<1, 0, KPW, WPB>);
    return fast ? launch(srl_assoc_kernel<2, 1, KPW, WPB>) : launch(srl_assoc_kernel<2, 0, KPW, WPB>);
}
// kpw = keypoints per wave (16-wave workgroups: 2 / 3 / 4 / 6 / 8 / 12 / 16; 4-wave: 4 / 8 / 16), wpb = waves per workgroup
hipError_t srl_launch_assoc(const SrlAssocArgs &a, int nb_voxels, int kpw, int wpb, hipStream_t s) {
    if (a.n <= 0) return hipSuccess;
    if (wpb == 16) {
        if (kpw == 2) return launch_assoc_cfg<2, 16>(a, nb_voxels, s);
        if (kpw == 3) return launch_assoc_cfg<3, 16>(a, nb_voxels, s);
        if (kpw == 4) return launch_assoc_cfg<4, 16>(a, nb_voxels, s);
        if (kpw == 6) return launch_assoc_cfg<6, 16>(a, nb_voxels, s);
        if (kpw == 8) return launch_assoc_cfg<8, 16>(a, nb_voxels, s);
        if (kpw == 12) return launch_assoc_cfg<12, 16>(a, nb_voxels, s);
        return launch_assoc_cfg<16, 16>(a, nb_voxels, s);
    }
    if (kpw != 4 && kpw != 8 && kpw != 16) return hipErrorInvalidConfiguration;   // 4-wave workgroups: phase 2 needs whole waves of 16 keypoints
    if (kpw == 4) return launch_assoc_cfg<4, 4>(a, nb_voxels, s);
    if (kpw == 8) return launch_assoc_cfg<8, 4>(a, nb_voxels, s);
    return launch_assoc_cfg<16, 4>(a, nb_voxels, s);
}
// LDS bytes of a configuration (host: does the 16-wave workgroup fit?)
int srl_assoc_lds_bytes(int K, int nb_voxels, int kpw, int wpb) { return lds_layout(K, nb_voxels, kpw, wpb).total; }

hipError_t srl_launch_reduce(const SrlReduceArgs &a, int mode, hipStream_t s) {
    hipLaunchKernelGGL(srl_reduce_kernel, dim3(1), dim3(1024), 0, s, a, mode);
    return hipGetLastError();
}

// After the all-reduce: forward the reduced result (device memory) to the host-mapped mailbox, same protocol as the
// single-rank reduce kernel -- the host spins on the sequence word instead of a D2H copy + stream synchronisation.
__global__ void __launch_bounds__(64) srl_publish_kernel(const SrlDevOut *src, SrlMailbox *mb, unsigned long long seq) {
    const int tid = threadIdx.x;
    constexpr int NW = (int)(sizeof(SrlDevOut) / 8);
    const unsigned long long *s = reinterpret_cast<const unsigned long long *>(src);
    unsigned long long *d = reinterpret_cast<unsigned long long *>(&mb->out);
    if (tid < NW) __hip_atomic_store(d + tid, s[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0) __hip_atomic_store(&mb->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
hipError_t srl_launch_publish(const SrlDevOut *src, SrlMailbox *mb, unsigned long long seq, hipStream_t s) {
    static_assert(sizeof(SrlDevOut) / 8 <= 64, "one wave publishes the result");
    hipLaunchKernelGGL(srl_publish_kernel, dim3(1), dim3(64), 0, s, src, mb, seq);
    return hipGetLastError();
}

// Direct peer exchange behind an un-fused pass: the reduce kernel left this rank's result in device memory; one wave
// exchanges its leading doubles with the peers and publishes the sum (rank order) into the host mailbox.
__global__ void __launch_bounds__(64) srl_peer_rows_kernel(const SrlPeerTable *pt, unsigned epoch, int slot, const SrlDevOut *src, SrlMailbox *mb,
                                                            unsigned long long seq, const long long *gather_check) {
    const int tid = threadIdx.x;
    const unsigned long long *sw = reinterpret_cast<const unsigned long long *>(src);
    double sum = 0.0;
    bool ok = peer_exchange(pt, epoch, slot, tid, SRL_REDUCED_DOUBLES, tid < SRL_REDUCED_DOUBLES ? sw[tid] : 0ull,
                            [&](int, unsigned long long w) { sum += __longlong_as_double((long long)w); });
    if (gather_check)                      // the counts of the ordered cut came through a peer exchange too: a missing one poisons the pass
        for (int r = 0; r < pt->nranks; ++r) ok = ok && gather_check[r] != (long long)0x8000000000000000ull;
    unsigned long long *d = reinterpret_cast<unsigned long long *>(&mb->out);
    if (tid < SRL_REDUCED_DOUBLES) __hip_atomic_store(d + tid, (unsigned long long)__double_as_longlong(sum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else if (tid == SRL_REDUCED_DOUBLES) __hip_atomic_store(&mb->out.last_visited, src->last_visited, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else if (tid == SRL_REDUCED_DOUBLES + 1) __hip_atomic_store(&mb->out.pad, ok ? 0ll : SRL_PEER_TIMEOUT_MARK, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0) __hip_atomic_store(&mb->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// ... and of the per-rank counts the ordered cut across shards starts from (one word per rank; a missing one = INT64_MIN)
__global__ void __launch_bounds__(64) srl_peer_counts_kernel(const SrlPeerTable *pt, unsigned epoch, int slot, const long long *count, long long *gather_out) {
    const int tid = threadIdx.x;
    if (tid == 0) for (int r = 0; r < pt->nranks; ++r) gather_out[r] = (long long)0x8000000000000000ull;
    __builtin_amdgcn_wave_barrier();
    peer_exchange(pt, epoch, slot, tid, 1, tid == 0 ? (unsigned long long)*count : 0ull,
                  [&](int r, unsigned long long w) { gather_out[r] = (long long)w; });
}
hipError_t srl_launch_peer_rows(const SrlPeerTable *peer, unsigned epoch, int slot, const SrlDevOut *src, SrlMailbox *mb, unsigned long long seq,
                                const long long *gather_check, hipStream_t s) {
    static_assert(SRL_REDUCED_DOUBLES + 2 <= 64 && SRL_REDUCED_DOUBLES <= SRL_PEER_ROW, "one wave exchanges the row");
    static_assert(offsetof(SrlDevOut, last_visited) == SRL_REDUCED_DOUBLES * 8, "the reduced range ends where last_visited starts");
    hipLaunchKernelGGL(srl_peer_rows_kernel, dim3(1), dim3(64), 0, s, peer, epoch, slot, src, mb, seq, gather_check);
    return hipGetLastError();
}
hipError_t srl_launch_peer_counts(const SrlPeerTable *peer, unsigned epoch, int slot, const long long *count, long long *gather_out, hipStream_t s) {
    hipLaunchKernelGGL(srl_peer_counts_kernel, dim3(1), dim3(64), 0, s, peer, epoch, slot, count, gather_out);
    return hipGetLastError();
}

hipError_t srl_launch_count(const SrlBlockInfo *binfo, int nblocks, int count_planes, long long *out_total, hipStream_t s) {
    hipLaunchKernelGGL(srl_count_kernel, dim3(1), dim3(256), 0, s, binfo, nblocks, count_planes, out_total);
    return hipGetLastError();
}

hipError_t srl_launch_search(const SrlSearchArgs &a, int nb_voxels, hipStream_t s) {
    if (a.n <= 0) return hipSuccess;
    int nblocks = (a.n + 3) / 4;
    if (nblocks > 4096) nblocks = 4096;
    if (nb_voxels == 1) hipLaunchKernelGGL(srl_search_kernel<1>, dim3(nblocks), dim3(SRL_BLOCK), 0, s, a);
    else hipLaunchKernelGGL(srl_search_kernel<2>, dim3(nblocks), dim3(SRL_BLOCK), 0, s, a);
    return hipGetLastError();
}

hipError_t srl_launch_transform(const double *raw_aos, int n, const SrlXform &X, double *out_aos, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(srl_transform_kernel, dim3((n + 255) / 256), dim3(256), 0, s, raw_aos, n, X, out_aos);
    return hipGetLastError();
}

hipError_t srl_launch_aos_to_soa(const double *aos, int n, double *x, double *y, double *z, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(srl_aos_to_soa_kernel, dim3((n + 255) / 256), dim3(256), 0, s, aos, n, x, y, z);
    return hipGetLastError();
}
