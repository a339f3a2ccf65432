/*
 * srlivo_hip_debug.h -- debug, parity and tuning hooks of libsrlivo_hip.so (srl_debug_*).
 *
 * Kept apart from the product ABI (srlivo_hip.h): nothing on the product path -- the host mirror, integration/optimize_hip.cpp, a node
 * that links the library -- calls any of these.  They exist for tests/, tools/ and bench.py's A/B legs: switching parts of the kernel
 * off, forcing selection paths and launch shapes, reading time lines.  Same conventions as srlivo_hip.h.
 */
#ifndef SRLIVO_HIP_DEBUG_H
#define SRLIVO_HIP_DEBUG_H

#include "srlivo_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Selection path of srl_build_residuals on this context (until round 4 a field of srl_icp_opts; that struct now holds the fields of the
 * reference's icpOptions only): 0 = automatic (the default); 1 streaming extraction, 2 general two-pass, 3 FP64-retained fast path,
 * 4 Jacobi eigen-solver in phase 2, 5 heap replay (the reference's literal priority_queue sequence) for every keypoint. */
int srl_debug_set_select_mode(srl_ctx *ctx, int select_mode);

/* Where srl_frame_select_keypoints orders the keypoints (the iteration order of gridSampling's std::tr1::unordered_map,
 * utility.cpp:167-201): 0 = on the device where it applies (the default: frames whose bucket table fits one scan launch), 1 = always the
 * host replay of csrc/host/tr1_order.h (what rounds 3-4 shipped; still the path of larger frames and of a bucket with more than 16 voxels).
 * srl_debug_frame_order_used: what the last selection did -- 1 device order, 2 host replay, 3 device order found an overfull bucket and
 * the host replay ran behind it.  Both orders are the same permutation (tests/test_gpu_frame_order.py). */
/* The frame path's own stable sort of (key, position) pairs over the low `bits` (1..18) key bits, n <= 131072 (two one-launch radix
 * passes, csrc/srl_frame_scratch.h; what srl_frame_commit's addPointsToMap groups a frame's points with) on caller data: keys_sorted and
 * positions_sorted (the position 0..n-1 each key came from) as a stable sort by (key & ((1 << bits) - 1)) would leave them. */
int srl_debug_radix_sort_pairs(srl_ctx *ctx, const uint32_t *keys, int n, int bits, uint32_t *keys_sorted, uint32_t *positions_sorted);

int srl_debug_set_frame_order_mode(srl_ctx *ctx, int mode);
int srl_debug_frame_order_used(srl_ctx *ctx, int *used);

/* test hook for the ON-DEVICE budget derivation of the sharded ordered cut (optimize.cpp:107 across ordered shards): the
 * context acts as rank `rank` of `nranks` whose on-stream all-gather of per-rank counts (accepted residuals; keypoints with a
 * plane when max_num_residuals <= 0) has already delivered `counts`; the all-reduce is the identity, so srl_build_residuals
 * returns THIS rank's contribution.  Lets one GPU exercise the reduce kernel's rank > 0 branches, which a 1-rank communicator
 * cannot reach; the same result must come out of the host-side srl_shard_budget path (srl_comm_set_host_callbacks).
 * counts = NULL switches the hook off.  Upload the sweep after switching (the shard range changes). */
int srl_debug_set_gather_counts(srl_ctx *ctx, int nranks, int rank, const int64_t *counts);

/* srl_debug_set_ablate: bit mask that switches parts of the association kernel off (profiling tools only; results are
 *   wrong while it is set).  Replaces the SRL_ABLATE environment variable of round 1 -- the library reads no environment
 *   variable on the per-iteration path.  The switches live in ONE extra instantiation of the kernel (r = 1 fast path, 16 x 16
 *   keypoints per workgroup); while the mask is non-zero every pass is launched in that shape, whatever the sweep size.  The
 *   production instantiations carry no trace of them.
 * srl_debug_set_search_select_mode: selection path used by srl_search_neighbors (0 default, 1 extraction, 5 heap replay).
 * srl_debug_heap_topk: the device kernels' restatement of libstdc++'s push_heap / pop_heap (csrc/srl_heap.h, the
 *   std::priority_queue of optimize.cpp:355-363,394-404,411-422) run on the host: offers distances[0..n) in order to a
 *   bounded max-heap of K, writes the read-out order (candidate indices, ascending distance) and returns its size.  No GPU.
 * srl_debug_device_sqrt: out[i] = the device's sqrt(in[i]) (the tie replay relies on it being correctly rounded). */
int srl_debug_set_ablate(srl_ctx *ctx, int bits);
/* 0: always run the separate ordered-cut / reduce kernel.  Default 1: on an unsharded context without taps the last workgroup
 * of the association kernel sums the published rows and writes the normal equations itself (one kernel per ESIKF iteration)
 * -- also WITH the ordered cut of a finite max_num_residuals when the pass runs in workgroups of <= 64 keypoints (the
 * prefix pass of the shipped 600).  A finisher that gives up waiting for a row (bounded spin) makes srl_build_residuals repeat
 * the pass once with the separate reduce kernel instead of failing. */
int srl_debug_set_fused_reduce(srl_ctx *ctx, int enable);
/* Neighbourhood bounds (default on): every pass leaves, per keypoint, its world position and the exact squared distance of its K-th
 * nearest neighbour; the next pass over the same sweep and map does not visit voxels that lie further from the keypoint than
 * sqrt(tau) + |movement| -- the same neighbours (searchNeighbors, optimize.cpp:365-426, keeps the K nearest whatever else it looked at),
 * fewer candidates evaluated.  0 switches the culling off (A/B in the tests: every bit must agree). */
int srl_debug_set_bound_culling(srl_ctx *ctx, int enable);
/* Where the pose box of the armed launches lives.  kind 1: fine-grained DEVICE memory the host writes through the PCIe BAR (every
 * workgroup polls it locally; SRL_ERR_UNSUPPORTED when device memory is not CPU-visible on this system).  kind 0: page-locked host
 * memory -- workgroup 0 of the waiting kernel polls it across PCIe and republishes the pose into device memory for the other
 * workgroups (~0.7 us per pass slower).  kind -1 (the default): 1 where possible, else 0.
 * srl_debug_set_arm_linger: the age (us) beyond which a call cancels an armed launch instead of firing it (default 150), and the
 * kernel-side bound (us, default 300) after which a waiting launch leaves by itself -- tests drive both paths with it. */
int srl_debug_set_pose_box(srl_ctx *ctx, int kind);
int srl_debug_set_arm_linger(srl_ctx *ctx, double host_linger_us, double kernel_linger_us);
/* Time line of the armed passes (tools/arm_timeline.py): enable allocates a host-mapped stamp buffer the armed kernels file into
 * (kernels built with -DSRL_ARM_STAMPS only: the product build has no stamp sites and leaves the buffer at zero);
 * gpu_out[64 * 32] (optional): per pass (row = sequence number & 63) the 100 MHz device clock at {entry, pose received, tile start,
 * phase 0 / 1 / 2 done, row published, finisher done} of workgroup 0 (slots 0..7) and of the finishing workgroup (8..15), slots
 * 16..24 inside the finisher and phase 2 (-DSRL_STAMP_DETAIL);
 * host_out[64 * 4] (optional): steady-clock ns at {call entry, pose written or launch returned, result seen} and a fired flag. */
int srl_debug_pass_stamps(srl_ctx *ctx, int enable, long long *gpu_out, long long *host_out);
/* Stage times of the frame pipeline (bench.py's `pipeline` leg, tools/pipeline_probe.py).  While enabled the stream is synchronised at
 * every stage boundary and the wall time of each stage is ACCUMULATED in us; out16 (optional) receives the sums since the last call,
 * which also clears them: [0] srl_frame_upload; srl_frame_select_keypoints: [1] grouping kernels (transform, voxel key, first point per
 * voxel), [2] download of the voxel list, [3] host: std::tr1::unordered_map iteration order, [4] gather into the resident sweep;
 * srl_frame_commit: [5] re-transform, [6] download of point3D::point; the map insertion behind it (also srl_map_insert): [7] keys +
 * (key, index) sort, [8] segments, [9] lookup + creation of new voxels, [10] per-voxel replay + counters. */
int srl_debug_frame_timing(srl_ctx *ctx, int enable, double out16[16]);
/* The scratch hash tables of the frame pipeline are never cleared per frame: entries carry a 16-bit epoch (csrc/srl_frame_scratch.h) and
 * the tables are cleared when it wraps, every 65 535 frames (1.8 h of a 10 Hz sensor).  Test hook: set the epoch counters of both tables so
 * that the wrap happens `frames_to_wrap` frames from now. */
int srl_debug_set_frame_epoch(srl_ctx *ctx, int frames_to_wrap);
/* tuning experiments: force the association kernel's launch shape -- keypoints per wave (16-wave workgroups: 2 / 3 / 4 / 6 / 8 /
 * 12 / 16; 4-wave workgroups: 4 / 8 / 16) and waves per workgroup (4 / 16); 0, 0 = automatic (by sweep size).  Results do not
 * depend on the shape beyond FP64 summation order. */
int srl_debug_set_launch_shape(srl_ctx *ctx, int keypoints_per_wave, int waves_per_workgroup);
int srl_debug_set_search_select_mode(srl_ctx *ctx, int select_mode);
int srl_debug_heap_topk(const double *distances, int n, int K, int32_t *out_index);
int srl_debug_device_sqrt(srl_ctx *ctx, const double *in, int n, double *out);
/* debug (srl_debug_set_ablate(ctx, 128)): {start, end, xcc id} stamps of every workgroup of the last association launch, in ticks of
 * the 100 MHz wall clock; out = max_blocks x 3 doubles.  Used by tools/block_times.py. */
int srl_debug_block_times(srl_ctx *ctx, double *out, int max_blocks, int *nblocks);

#ifdef __cplusplus
}
#endif
#endif
