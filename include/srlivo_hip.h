/*
 * srlivo_hip.h -- C-ABI of libsrlivo_hip.so: the MI355X (gfx950) implementation of SR-LIVO's LIO
 * scan-matching hot path (reference: ZikangYuan/sr_livo, src/optimize.cpp).
 *
 * The reference has no FFI/plugin API for this path: it sits behind member functions of
 * `class lioOptimization` (include/lioOptimization.h:334-357).  Each entry point below names the
 * reference interface it replaces (paths relative to the reference root).  The C++ host mirror that
 * keeps the reference's class surfaces (lioOptimization / eskfEstimator / cloudMap types) lives in
 * sr_livo_amd/csrc/host/ and forwards to this ABI; INTEGRATION.md shows the binding a maintainer adds.
 *
 * Conventions: plain pointers and sizes, caller-owned HOST buffers unless a parameter says
 * "device"; all floating point is FP64 except stored map positions (FP32, cloudMap.h:54) and voxel
 * keys (int16, cloudMap.h:124-145); matrices ROW-MAJOR; quaternions (w,x,y,z).  Every function
 * returns SRL_OK (0) or a negative srl_status; no exceptions cross the ABI.  There is NO CPU
 * fallback: without a HIP device srl_ctx_create fails with SRL_ERR_NO_DEVICE.
 */
#ifndef SRLIVO_HIP_H
#define SRLIVO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SRL_VOXEL_CAP 20        /* max_num_points_in_voxel of the LIO map (config/r3live.yaml:50, ntu.yaml:49) */
#define SRL_MAX_NEIGHBORS 32    /* upper bound supported for icpOptions::max_number_neighbors (shipped: 20) */

typedef enum srl_status {
    SRL_OK = 0,
    SRL_ERR_NO_DEVICE = -1,     /* no HIP device / HIP runtime failure at create */
    SRL_ERR_HIP = -2,           /* a HIP call failed (see srl_last_error) */
    SRL_ERR_BAD_ARG = -3,
    SRL_ERR_UNSUPPORTED = -4,   /* option outside the supported envelope (cap != 20, K > 32, nb_voxels > 2) */
    SRL_ERR_NO_MAP = -5,
    SRL_ERR_NO_SWEEP = -6,
    SRL_ERR_COMM = -7,          /* RCCL failure */
    SRL_ERR_NAN_PLANARITY = -8, /* optimize.cpp:348-350: a2D is NaN -> the reference throws std::runtime_error("error") */
    SRL_ERR_NOT_ENOUGH_RESIDUALS = -9  /* optimize.cpp:110-123: summary.success = false */
} srl_status;

typedef struct srl_ctx srl_ctx;

/* the fields of icpOptions the path reads (include/parameters.h:8-56) -- those and nothing else */
typedef struct srl_icp_opts {
    int32_t threshold_voxel_occupancy;  /* parameters.h:13 */
    int32_t init_num_frames;            /* parameters.h:15 */
    double  size_voxel_map;             /* parameters.h:17 */
    int32_t num_iters_icp;              /* parameters.h:19 */
    int32_t min_number_neighbors;       /* parameters.h:21 */
    int32_t voxel_neighborhood;         /* parameters.h:23 */
    double  power_planarity;            /* parameters.h:25 */
    int32_t max_number_neighbors;       /* parameters.h:30 */
    double  max_dist_to_plane_icp;      /* parameters.h:32 */
    double  threshold_orientation_norm; /* parameters.h:34 */
    double  threshold_translation_norm; /* parameters.h:36 */
    int32_t max_num_residuals;          /* parameters.h:40 */
    double  weight_alpha;               /* parameters.h:46 */
    double  weight_neighborhood;        /* parameters.h:48 */
} srl_icp_opts;

/* per-iteration pose + frame constants read by buildPlaneResiduals (optimize.cpp:21-28,83) */
typedef struct srl_frame {
    double  q[4];        /* p_frame->p_state->rotation (w,x,y,z), NOT normalised by the caller */
    double  t[3];        /* p_frame->p_state->translation */
    double  t_last[3];   /* all_cloud_frame[id-1]->p_state->translation (optimize.cpp:25) */
    double  R_il[9];     /* R_imu_lidar (lioOptimization.h:227) */
    double  t_il[3];     /* t_imu_lidar (lioOptimization.h:228) */
    int32_t frame_id;    /* p_frame->frame_id (selects init mode, optimize.cpp:21-23) */
} srl_frame;

/* what one buildPlaneResiduals pass hands to updateIEKF (optimize.cpp:160-170,235,239) */
typedef struct srl_normal_eq {
    double  HtH[36];         /* H_x^T H_x, 6x6 row-major */
    double  Hth[6];          /* H_x^T h */
    double  loss_sum;        /* sum of distance^2 over accepted residuals (optimize.cpp:104) */
    int32_t num_residuals;   /* optimizeSummary.num_residuals_used */
    int32_t success;         /* optimizeSummary.success (optimize.cpp:110) */
    int64_t sum_candidates;  /* sum over keypoints of resident points visited (P_k), whole sweep */
    int64_t last_visited;    /* global index of the last keypoint the sequential loop would visit (cut-off) */
    int32_t nan_error;       /* 1 if a NaN planarity was met among visited keypoints */
    int32_t num_fallback;    /* keypoints that took the streaming-extraction selection path */
} srl_normal_eq;

/* ------------------------------------------------------------------ context */
int         srl_device_count(int *count);
int         srl_ctx_create(int device, srl_ctx **out);
int         srl_ctx_destroy(srl_ctx *ctx);
const char *srl_last_error(const srl_ctx *ctx);      /* text of the last failure on this context */
const char *srl_status_str(int status);
void        srl_icp_opts_default(srl_icp_opts *o);   /* effective values of config/r3live.yaml:57-69 */

/* ------------------------------------------------------------------ voxel map
 * replaces: voxelHashMap voxel_map (lioOptimization.h:274; cloudMap.h:171 tsl::robin_map<voxel, voxelBlock>).
 * srl_map_upload installs a map built elsewhere (voxels in creation order; point id = voxel*cap + slot). */
int srl_map_upload(srl_ctx *ctx, const int16_t *keys_xyz /* V x 3 */, const int32_t *counts /* V */,
                   const float *xyz /* V x cap x 3, AoS */, int num_voxels, int cap);
/* replaces lioOptimization::addPointsToMap / addPointToMap (lioOptimization.cpp:520-554, 400-446):
 * order-dependent insert of world points (AoS n x 3, FP64) into the device-resident map. */
int srl_map_insert(srl_ctx *ctx, const double *world_xyz, int n, double voxel_size, int cap,
                   double min_distance_points, int min_num_points, int *num_added);
/* replaces lioOptimization::mapSize (lioOptimization.cpp:574-581) */
int srl_map_size(srl_ctx *ctx, int64_t *num_points, int32_t *num_voxels);
/* copies the device map back in creation order (same layout as srl_map_upload) */
int srl_map_download(srl_ctx *ctx, int16_t *keys_xyz, int32_t *counts, float *xyz, int max_voxels);
/* A cheap fingerprint of the part of the map a set of points falls into -- what a caller that keeps a map of its OWN (the node's
 * tsl::robin_map, lioOptimization.h:274) compares frame by frame instead of walking both maps: for every point the voxel it belongs to
 * (key = short(float(p) / voxel_size), lioOptimization.cpp:403-405) contributes srl_probe_mix(key, points in the voxel, position of the
 * voxel's LAST stored point); a point without a voxel contributes 0; the checksum is the sum modulo 2^64 (points of one voxel count
 * once each: no deduplication on either side).  world_xyz = host points (n x 3), or NULL: the world points the last srl_frame_commit
 * left in HBM (n is then ignored; the empty sum when a newer frame has been uploaded since).  stride >= 1: only the points 0, stride,
 * 2 stride, ... take part (a sample keeps the caller's side of the comparison cheap).  Waits for a deferred insertion to finish. */
#if defined(__HIPCC__)
__host__ __device__
#endif
static inline uint64_t srl_probe_mix(int16_t kx, int16_t ky, int16_t kz, int32_t count, float lx, float ly, float lz) {
    union { float f; uint32_t u; } a, b, c;
    a.f = lx; b.f = ly; c.f = lz;
    uint64_t h = (uint64_t)(uint16_t)kx | ((uint64_t)(uint16_t)ky << 16) | ((uint64_t)(uint16_t)kz << 32) | ((uint64_t)(uint32_t)count << 48);
    h = (h ^ (h >> 31)) * 0x9E3779B97F4A7C15ull;
    h ^= ((uint64_t)a.u << 32) | b.u;
    h = (h ^ (h >> 29)) * 0xBF58476D1CE4E5B9ull;
    h ^= c.u;
    h = (h ^ (h >> 32)) * 0x94D049BB133111EBull;
    return h ^ (h >> 30);
}
int srl_map_probe_checksum(srl_ctx *ctx, const double *world_xyz, int n, int stride, double voxel_size, uint64_t *checksum);

/* ------------------------------------------------------------------ sweep
 * replaces: the `keypoints` vector handed to updateIEKF (optimize.cpp:133; point3D::raw_point,
 * cloudMap.h:40).  AoS n x 3 FP64 in the lidar frame, in keypoint order.  Uploaded once per sweep.
 * With a communicator attached this rank keeps the contiguous range [rank*n/R, (rank+1)*n/R). */
int srl_sweep_upload(srl_ctx *ctx, const double *raw_xyz, int n);
/* Page-locked host memory for the sweep: srl_sweep_upload DMAs straight out of a buffer obtained here (or registered with
 * srl_host_register) -- one asynchronous copy on the context's stream, no staging, no synchronisation; such a buffer must stay
 * untouched until the next call that returns results.  From pageable memory the upload goes through a pinned ring inside
 * the context (CPU copy of a chunk overlapped with the DMA of the previous one) and the buffer is free on return. */
int srl_pinned_alloc(size_t bytes, void **out);
/* Blocks until the DMA of the last srl_sweep_upload / srl_sweep_prefetch that read a PAGE-LOCKED caller buffer has finished:
 * after it the buffer may be refilled.  (A result returned by srl_build_residuals implies the same for the
 * sweep it was computed on; a caller that refills its page-locked buffer earlier than that calls this first.  Uploads from
 * pageable memory never need it: the buffer is consumed on return.)  No-op when nothing is pending. */
int srl_sweep_wait(srl_ctx *ctx);
int srl_pinned_free(void *p);
int srl_host_register(void *p, size_t bytes);
int srl_host_unregister(void *p);
/* Thread placement helper for the node's estimation thread (the reference runs it as the ROS node's main thread,
 * src/lioOptimization.cpp:1587-1611): restricts the CALLING thread to the CPUs of the NUMA node the context's GPU hangs off
 * (/sys/bus/pci/devices/<gpu>/local_cpulist).  The solve is a latency-bound host loop -- one mailbox read and one doorbell
 * write per ESIKF iteration --, and from the other socket of a two-socket host every iteration costs ~3 us more.  Optional;
 * returns SRL_ERR_UNSUPPORTED when the topology cannot be read (nothing is changed then).  *numa_node (may be NULL) receives
 * the node. */
int srl_thread_pin_to_gpu_numa(srl_ctx *ctx, int *numa_node);
/* The NEXT sweep while the current one is being solved (a node receives sweep k + 1 during the solve of sweep k):
 * srl_sweep_prefetch uploads it on the context's copy stream into a second sweep buffer and returns at once; the current
 * sweep stays valid.  srl_sweep_swap makes the prefetched sweep current -- the compute stream waits for the upload's event,
 * the host does not; when the upload has already landed, a launch armed behind the previous sweep's last pass is kept and serves
 * as the new sweep's first pass (see ARMED LAUNCHES).  Same buffer rules as srl_sweep_upload (a page-locked source must stay untouched until the first
 * result computed on the swapped-in sweep has been returned). */
int srl_sweep_prefetch(srl_ctx *ctx, const double *raw_xyz, int n);
int srl_sweep_swap(srl_ctx *ctx);
int srl_sweep_shard(srl_ctx *ctx, int *begin, int *count, int *total);

/* ------------------------------------------------------------------ frame-resident pipeline (optional)
 * Keeps the whole reconstructed sweep in HBM from keypoint selection to map insertion.
 * srl_frame_upload            replaces handing p_frame->point_frame (raw points) to optimize() (optimize.cpp:428)
 * srl_frame_select_keypoints  replaces gridSampling / subSampleFrame (utility.cpp:167-201) applied to
 *                             point = R(q) (R_il raw + t_il) + t (utility.cpp:314-318): same keypoints in the same
 *                             (std::tr1::unordered_map iteration) order; they become the resident sweep.  The order is
 *                             computed on the device (csrc/host/tr1_relation.h; frames beyond 131 072 points or buckets, or
 *                             with more than 16 voxels in one bucket of the container, through the host replay of
 *                             csrc/host/tr1_order.h).  keypoint_index == NULL: only the count comes back (the call returns
 *                             while the last kernels of the selection still run; the passes queue behind them) -- the
 *                             index list costs a copy and a stream synchronisation, ask for it only if the caller needs it.
 * srl_frame_commit            replaces the re-transform loop (optimize.cpp:441-445) + addPointsToMap
 *                             (lioOptimization.cpp:520-554) with the final pose, without leaving the device.
 * A page-locked raw_xyz (srl_pinned_alloc) is copied by the DMA engine on the context's copy stream, beside whatever the compute stream
 * still does for the previous frame; the buffer may be refilled once a later call on the context has returned results (or after
 * srl_sweep_wait).  A pageable raw_xyz is consumed before the call returns. */
int srl_frame_upload(srl_ctx *ctx, const double *raw_xyz, int n);

/* Sweep reconstruction: the per-point stages of buildFrame (lioOptimization.cpp:833-850).
 * srl_imu_state mirrors imuState (cloudMap.h:110-122), quat as w x y z.
 * srl_frame_undistort replaces distortFrameByConstant / distortFrameByImu (utility.cpp:203-306) followed by
 *   transformAllImuPoint (utility.cpp:320-332) over ALL n points of the cut sweep: relative_time_ms is
 *   point3D::relative_time (makePointTimestamp, lioOptimization.cpp:786-819), motion_compensation the enum of
 *   include/utility.h:82-86 (SRL_MC_NONE = neither branch of lioOptimization.cpp:833-836 runs).  imu_point_in (or NULL
 *   = zeros) is what imu_point holds before: distortFrameByImu leaves points its interval walk does not reach untouched.
 *   Outputs (optional): point3D::imu_point and the corrected point3D::raw_point, n x 3 each.  The corrected sweep
 *   stays in HBM.
 * srl_frame_take makes the points index[0..m) of the corrected sweep the resident frame (what srl_frame_upload
 *   would upload) -- the order buildFrame's shuffle / subSampleFrame / shuffle leaves (a host decision). */
typedef struct srl_imu_state {
    double timestamp;
    double un_acc[3], un_gyr[3], trans[3];
    double quat[4];
    double vel[3];
} srl_imu_state;
enum { SRL_MC_IMU = 0, SRL_MC_CONSTANT_VELOCITY = 1, SRL_MC_NONE = 2 };
int srl_frame_undistort(srl_ctx *ctx, const double *raw_xyz, const double *relative_time_ms, const double *imu_point_in,
                        int n, const srl_imu_state *imu_states, int n_states, double time_frame_begin,
                        int motion_compensation, const double R_il[9], const double t_il[3], double *imu_point_out,
                        double *raw_out);
int srl_frame_take(srl_ctx *ctx, const int32_t *index, int m);
int srl_frame_size(srl_ctx *ctx, int *n);      /* points of the resident frame (capacity needed for keypoint_index) */
int srl_frame_select_keypoints(srl_ctx *ctx, const double q[4], const double t[3], const double R_il[9],
                               const double t_il[3], double sample_voxel_size,
                               int32_t *keypoint_index /* capacity n, or NULL */, int *num_keypoints);
int srl_frame_commit(srl_ctx *ctx, const double q[4], const double t[3], const double R_il[9], const double t_il[3],
                     double voxel_size, int cap, double min_distance_points, int min_num_points,
                     double *world_out /* n x 3 or NULL */, int *num_added /* or NULL */);
/* num_added == NULL (addPointsToMap returns nothing either): the insertion is enqueued behind the re-transform and the call returns as
 * soon as world_out (if asked for) has arrived; the passes of the next sweep are ordered behind the insertion on the context's stream,
 * and the map's totals are brought up to date by the next call that reads them (srl_map_size, srl_map_download, the next insertion). */

/* ------------------------------------------------------------------ hot path
 * replaces: lioOptimization::buildPlaneResiduals (optimize.cpp:18-131) incl. searchNeighbors
 * (:365-426), computeNeighborhoodDistribution (:316-353), and the H_x^T H_x / H_x^T h contraction
 * of updateIEKF (:235,:239).  One call per ESIKF iteration.  With a communicator the result is
 * all-reduced (RCCL) and identical on every rank.  Returns SRL_OK even when out->success == 0. */
int srl_build_residuals(srl_ctx *ctx, const srl_frame *frame, const srl_icp_opts *opts, srl_normal_eq *out);
/* The same call with a host callback that runs ONCE while the kernels are in flight (after the launches are enqueued,
 * before the result is awaited): updateIEKF's part of the 17-dim algebra that does not depend on H_x -- the prior error
 * state, the covariance projection and the first of the two 17x17 inverses, src/optimize.cpp:172-234 -- fits there, so
 * only the second inverse and the gain (src/optimize.cpp:235-244) remain behind the kernel.  fn may be NULL. */
typedef void (*srl_overlap_fn)(void *user);
int srl_build_residuals_overlap(srl_ctx *ctx, const srl_frame *frame, const srl_icp_opts *opts, srl_normal_eq *out,
                                srl_overlap_fn fn, void *user);
/* ARMED LAUNCHES.  The loop of updateIEKF (src/optimize.cpp:147-312) alternates kernel and host: buildPlaneResiduals' normal
 * equations (:153,:235,:239) -> 17-dim update (:172-261) -> next pose -> buildPlaneResiduals.  With armed launches on (the
 * default) an srl_build_residuals call on an unsharded context, besides running its own pass, enqueues the kernel of the NEXT
 * pass while the current one is in flight -- same map and options; the pose, which does not exist yet, arrives later through a
 * small host-written "pose box" the waiting workgroups poll, together with the keypoint count and which of the context's two
 * sweep buffers (srl_sweep_prefetch / srl_sweep_swap) the pass runs on.  The next call, if its arguments are those of the armed
 * launch (pose, sweep buffer and count may differ: that is the point), only writes the box: the launch call, the dispatch and the
 * ramp of the kernel are off the per-iteration critical path -- also across srl_sweep_swap, where the launch armed behind the last
 * pass of sweep k becomes the first pass of sweep k + 1 (src/lioOptimization.cpp:1003-1027: one optimize() per sweep).  Any other
 * call on the context, or a pass with other arguments, cancels the armed launch first (one 384-byte write; the waiting kernel
 * exits), so nothing observable changes: same kernels, same arithmetic, same results.
 * PROCESS-WIDE EFFECT: a waiting launch keeps one workgroup per compute unit resident.  Other work for the same GPU -- another
 * context or process, a foreign hipDeviceSynchronize / hipFree -- waits until the launch is fired, cancelled or leaves by itself
 * (300 us after it started waiting; a call arriving more than 150 us after arming cancels instead of firing).  Therefore a launch
 * is only armed where it is likely to fire: not behind the pass expected to be the last of a solve (the pass count of the
 * previous solve, told by srl_solve_end) unless a prefetched sweep is waiting, and not while a second context of this process
 * lives on the device.
 *   srl_set_armed_launch(ctx, 0 | 1 | 2)  off / on with the policy above (default) / armed behind every eligible pass
 *   srl_solve_end(ctx)                 the caller's ESIKF loop on the current sweep has ended (converged, iteration cap, failure):
 *                                      remembers how many passes it took and cancels an armed launch unless a prefetched sweep
 *                                      is waiting.  Optional (without it every eligible pass arms, as with mode 2).
 *   srl_disarm(ctx)                    cancel an armed launch now (optional: e.g. before the thread goes idle)
 *   srl_get_arm_stats                  counters {armed, fired, cancelled, expired} since context creation */
int srl_set_armed_launch(srl_ctx *ctx, int mode);
int srl_solve_end(srl_ctx *ctx);
int srl_disarm(srl_ctx *ctx);
int srl_get_arm_stats(srl_ctx *ctx, uint64_t out[4]);

/* enable/disable the per-keypoint parity taps written by srl_build_residuals (off by default) */
int srl_set_taps(srl_ctx *ctx, int enable);

/* parity taps for the LAST srl_build_residuals on this rank's shard (any pointer may be NULL).
 * status: 0 = < min_number_neighbors, 1 = plane built but distance gate rejected, 2 = accepted,
 * 3 = not visited by the sequential loop (after the max_num_residuals cut-off). */
int srl_fetch_neighbors(srl_ctx *ctx, int32_t *ids /* n x K, -1 padded */, uint8_t *status /* n */,
                        int32_t *num_candidates /* n */);
int srl_fetch_residuals(srl_ctx *ctx, double *normal /* n x 3 */, double *a2D, double *weight,
                        double *norm_offset, double *distance, double *jacobian /* n x 6 */);

/* replaces: lioOptimization::searchNeighbors for a batch of WORLD points (optimize.cpp:365-426) */
int srl_search_neighbors(srl_ctx *ctx, const double *world_xyz, int n, int nb_voxels_visited,
                         double size_voxel_map, int max_num_neighbors, int threshold_voxel_capacity,
                         int32_t *ids /* n x K */, float *nb_xyz /* n x K x 3, may be NULL */,
                         int32_t *num_found /* n */);

/* replaces: the re-transform loop at the end of optimize() (optimize.cpp:441-445 -> utility.cpp:314-318):
 * out = R(q) * (R_il * raw + t_il) + t for n points (AoS, host buffers). */
int srl_transform_points(srl_ctx *ctx, const double *raw_xyz, int n, const double q[4], const double t[3],
                         const double R_il[9], const double t_il[3], double *out_xyz);

/* ------------------------------------------------------------------ multi-GPU (RCCL over xGMI)
 * one context per process/GPU; the only exchange step is the all-reduce of the normal equations. */
#define SRL_COMM_ID_BYTES 128
int srl_comm_unique_id(void *id /* SRL_COMM_ID_BYTES */);
/* debug / test hook: take the nccl* entry points of this process from the given shared object instead of the process's RCCL.  Only
 * before the first communicator call (SRL_ERR_BAD_ARG afterwards: one instance per process).  tests/fake_rccl/libfake_rccl.so -- ranks
 * as processes meeting in shared memory, collectives stream-ordered through host callbacks -- lets the N > 1 sequencing of
 * srl_build_residuals (count all-gather -> reduce kernel -> all-reduce of 50 doubles -> publish, src/optimize.cpp:107,235,239 across
 * shards) run on a one-GPU box, where RCCL itself refuses two ranks per device. */
int srl_comm_set_library(const char *path);
int srl_comm_init_rank(srl_ctx *ctx, int nranks, int rank, const void *id);
int srl_comm_destroy(srl_ctx *ctx);
/* Which RCCL the communicator calls run on.  The library does not link librccl: it uses the RCCL instance the process has
 * already loaded (a PyTorch process: torch/lib/librccl.so) or, when there is none, dlopens librccl.so.1 -- one instance per
 * process, never two.  origin = path of that shared object, version = ncclGetVersion, preloaded = 1 when it was found in the
 * process.  SRL_ERR_COMM (origin = reason) when no RCCL is available; single-GPU use never needs one. */
int srl_comm_backend_info(char *origin, int origin_len, int *version, int *preloaded);
/* What the sharded path of this context runs on, for a run that has to explain itself (bench.py prints it for every --gpus N line):
 * transport 0 none (unsharded), 1 RCCL all-reduce, 2 direct peer exchange, 3 host callbacks; nranks / rank as attached; ranks_seen = the
 * communicator's own count (ncclCommCount) or the number of peer inboxes mapped -- equal to nranks when every rank really joined;
 * passes_armed = armed launches fired on this context so far (0 on a path that never arms).  Any pointer may be NULL. */
int srl_comm_info(srl_ctx *ctx, int *transport, int *nranks, int *rank, int *ranks_seen, int64_t *passes_armed);
/* suspend != 0: run unsharded (whole sweep, no collective) while keeping the communicator; 0: back to sharded mode.
 * Re-upload the sweep after switching. */
int srl_comm_suspend(srl_ctx *ctx, int suspend);
/* test hook: a host all-reduce callback (sum over ranks of `count` doubles, in place) used INSTEAD of
 * RCCL when set -- lets the sharded logic run over gloo/MPI or G logical shards on one device. */
typedef int (*srl_allreduce_fn)(double *buf, int count, void *user);
typedef int (*srl_allgather_i64_fn)(const int64_t *mine, int64_t *all /* nranks */, void *user);
int srl_comm_set_host_callbacks(srl_ctx *ctx, int nranks, int rank, srl_allreduce_fn ar,
                                srl_allgather_i64_fn ag, void *user);

/* Direct peer exchange: the sum over the point-range shards (optimize.cpp:235,239 see the sum of all residuals) WITHOUT an
 * RCCL call on the data path -- the alternative of SURVEY.md 5 for one node.  Every rank owns an inbox in fine-grained device
 * memory; per pass each rank's finishing workgroup stores its 50-double row, as tagged 8-byte granules, into the inbox of every
 * rank over xGMI and adds the rows it received in rank order (the same bits on every rank).  A sharded pass then is still ONE
 * kernel; with the ordered cut (max_num_residuals can bind) the per-rank counts travel the same way, followed by the reduce
 * kernel and a one-wave exchange kernel.
 *   srl_peer_export : creates the inbox on first use and RESETS it (a new session: call it before every attachment, on every
 *                     rank, before the handles are exchanged -- never while peers are attached); ipc_handle
 *                     (SRL_PEER_HANDLE_BYTES, may be NULL) receives its HIP IPC handle for peers in OTHER processes,
 *                     *local_ptr (may be NULL) the device pointer for peers in the SAME process.
 *   srl_peer_attach : ipc_handles = nranks x SRL_PEER_HANDLE_BYTES gathered from all ranks (how they travel is the caller's
 *                     business: torch.distributed.all_gather_object, MPI, a file) and / or local_ptrs[nranks] for same-process
 *                     peers (NULL entries fall back to the handle).  Sets the shard layout like srl_comm_init_rank: upload the
 *                     sweep afterwards.  nranks <= 8.  Mutually exclusive with srl_comm_init_rank / host callbacks.
 *   srl_peer_detach : back to an unsharded context (unmaps the peers' inboxes).
 * All ranks must call srl_build_residuals the same number of times with the same options (as with any collective).
 *
 * A late rank is not a failure.  The kernel that waits for the rows gives up after a bounded spin (0.3-1 s: it must not hold the GPU for
 * ever), but what follows is decided on the HOST: srl_build_residuals repeats the pass with the same exchange tags -- this rank's rows are
 * in every inbox already, the repeat only polls again; nobody can run ahead, the next exchange needs a row of this rank -- until the
 * missing row is there (every rank then returns SRL_OK, the late one included) or until the wall-clock deadline of
 *   srl_peer_set_deadline_ms (default 10 000 ms, measured from the start of the pass; 0 = give up at the first time-out)
 * has passed.  Then the SESSION is given up for every rank: this rank sets the poison word of every inbox and returns SRL_ERR_COMM; a rank
 * whose own pass times out looks at its word first and returns SRL_ERR_COMM at once instead of waiting for its own deadline (a rank that
 * had already completed the exchange -- the rows were there -- fails at its next one).  After that every srl_build_residuals on the session
 * returns SRL_ERR_COMM until srl_peer_detach + srl_peer_export + srl_peer_attach on every rank start a new one.
 *   srl_peer_stats : passes repeated because a row had not arrived within one kernel's spin (since the attach), whether the session failed. */
#define SRL_PEER_HANDLE_BYTES 64
int srl_peer_export(srl_ctx *ctx, void *ipc_handle, void **local_ptr);
int srl_peer_attach(srl_ctx *ctx, int nranks, int rank, const void *ipc_handles, void *const *local_ptrs);
int srl_peer_set_deadline_ms(srl_ctx *ctx, int deadline_ms);
int srl_peer_stats(srl_ctx *ctx, int64_t *passes_repeated, int *session_failed);
int srl_peer_detach(srl_ctx *ctx);

/* pure helpers of the sharded path (also used internally): the contiguous point range of a rank
 * (SURVEY.md 8(e)), and the residual budget a rank may still spend given the accepted counts of all
 * shards (reproduces the sequential early exit, optimize.cpp:107, across ordered shards).
 * mode: 0 = spend up to *budget, 1 = visit only the first keypoint, 2 = visit nothing. */
void srl_shard_range(int n, int nranks, int rank, int *begin, int *count);
void srl_shard_budget(int max_num_residuals, const int64_t *accepted_per_rank, int nranks, int rank,
                      int64_t *budget, int *mode);

/* ------------------------------------------------------------------ measurement
 * HIP-event timings (ms) of the last srl_build_residuals on the context's own stream. */
typedef struct srl_timing {
    float   assoc_ms;          /* last call: association + plane fit + residual kernel */
    float   reduce_ms;         /* last call: ordered cut-off + final reduction kernel(s) */
    float   total_ms;          /* last call: first launch -> results on host */
    int32_t calls;             /* srl_build_residuals calls since profiling was switched on */
    int64_t algorithmic_bytes; /* last call: sum_k (24 + 12*(2r+1)^3 + 12*P_k) for this rank's shard */
    double  sum_assoc_ms;      /* accumulated over `calls` */
    double  sum_reduce_ms;
    double  sum_total_ms;
    int64_t sum_algorithmic_bytes;
    int64_t sum_keypoints;
    double  sum_host_launch_us; /* host wall: call entry -> both kernels enqueued */
    double  sum_host_wait_us;   /* host wall: enqueue done -> results on the host (copy + stream sync [+ all-reduce]) */
    double  sum_host_total_us;  /* host wall: whole srl_build_residuals call */
    int64_t sum_passes;         /* buildPlaneResiduals passes the timed launches ran (1 per launch) */
} srl_timing;
int srl_get_timing(srl_ctx *ctx, srl_timing *t);
/* (debug / parity / tuning hooks -- srl_debug_* -- are declared in srlivo_hip_debug.h: no product code path calls them) */
int srl_set_profiling(srl_ctx *ctx, int mode);     /* 0 off (default); 1 full: four events + a sync per call (kernel, reduce,
                                                      * device total, host splits); 2 light: one event pair around the association
                                                      * kernel, read back lazily (calls / sum_assoc_ms / sum_algorithmic_bytes only);
                                                      * 3 host stamps only, no events: sum_host_launch_us = the launch call,
                                                      * sum_host_wait_us = enqueued -> result (incl. the overlap callback),
                                                      * sum_host_total_us = the whole call, sum_assoc_ms / sum_reduce_ms = argument
                                                      * preparation / launch returned -> everything enqueued (tools/host_hop_probe.py).
                                                      * Switching on resets the sums. */
/* Mode 2 looks at every association launch by default (one event record behind each: ~2.5 us of the loop per launch).  With a period
 * P > 1 only every P-th launch is timed (the launch before it leaves its end event as the start): 2 records per P launches.  calls,
 * sum_assoc_ms, sum_algorithmic_bytes, sum_passes then cover the timed launches and their passes only.  1 <= P <= 64; pick an odd P where
 * the launches of a solve alternate (first / second iteration), so that both kinds are sampled. */
int srl_set_profiling_period(srl_ctx *ctx, int period);
/* "The sums of srl_get_timing start HERE" (mode 2) without a read-back: no event is waited for, an armed launch stays armed.  Launches
 * enqueued before the mark -- the one armed behind the last pass included -- are left out of calls / sum_assoc_ms when their events are
 * read later, and their passes out of the byte sums.  With a period P > 1 the launches timed behind the mark are the 4th, the (4 + P)-th,
 * ... (the first step behind a barrier -- an un-armed launch into an idle GPU -- is no steady-state sample).  (srl_get_timing itself waits for every outstanding event and cancels an armed
 * launch: called between warm-up and a timed region it idles the GPU for a few hundred microseconds.) */
int srl_timing_mark(srl_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif
