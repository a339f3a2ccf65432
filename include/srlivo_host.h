/*
 * srlivo_host.h -- C handles onto the C++ host mirror (sr_livo_amd/csrc/host/: lioOptimization,
 * eskfEstimator, cloudMap types with the reference's member names).  A C++ consumer (the ROS node)
 * includes the mirror headers directly; these handles exist so that non-C++ harnesses (the Python
 * parity tests and bench.py, via ctypes) drive exactly the same C++ code.  Exported by
 * libsrlivo_hip.so next to the kernel-level ABI of srlivo_hip.h.
 */
#ifndef SRLIVO_HOST_H
#define SRLIVO_HOST_H

#include "srlivo_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct srl_lio srl_lio;

/* device >= 0: lioOptimization on the HIP backend (fails with SRL_ERR_NO_DEVICE without a GPU).
 * device <  0: host-only object; only srl_lio_update_iekf_provided() may be used on it. */
int      srl_lio_create(int device, srl_lio **out);
int      srl_lio_destroy(srl_lio *lio);
srl_ctx *srl_lio_ctx(srl_lio *lio);                         /* the context behind voxel_map (NULL if host-only) */
const char *srl_lio_last_error(srl_lio *lio);

/* members of class lioOptimization the path reads (lioOptimization.h:221,227-228) */
int srl_lio_set_extrinsics(srl_lio *lio, const double R_il[9], const double t_il[3]);
int srl_lio_set_laser_point_cov(srl_lio *lio, double cov);
/* srl_lio_last_solve_launches: kernel launches the last solve enqueued for its passes (one per ESIKF iteration; with armed launches
 * all but the first are enqueued ahead of time, include/srlivo_hip.h). */
int srl_lio_last_solve_launches(srl_lio *lio, int *launches);
/* eskfEstimator::observe() calls of the last solve (src/optimize.cpp:253).  0 = every pass ran into the step guard (:248-251) or
 * failed before it: the reference then never wrote p_frame->p_state, G, G_norm (:255-261) -- a binding must not either. */
int srl_lio_last_solve_observed(srl_lio *lio, int *observed);

/* eskfEstimator accessors (eskfEstimator.h:74-108).  state = p(3) q(wxyz,4) v(3) ba(3) bg(3) g(3) */
int srl_lio_eskf_get_state(srl_lio *lio, double s[19]);
int srl_lio_eskf_set_state(srl_lio *lio, const double s[19]);
int srl_lio_eskf_get_cov(srl_lio *lio, double P[289]);
int srl_lio_eskf_set_cov(srl_lio *lio, const double P[289]);
int srl_lio_eskf_set_noise(srl_lio *lio, double acc_cov, double gyr_cov, double b_acc_cov, double b_gyr_cov);
int srl_lio_eskf_init_imu(srl_lio *lio, const double acc0[3], const double gyr0[3]);
int srl_lio_eskf_scale_init_cov(srl_lio *lio);
int srl_lio_eskf_predict(srl_lio *lio, double dt, const double acc1[3], const double gyr1[3]);
int srl_lio_eskf_observe(srl_lio *lio, const double dx[17]);

/* lioOptimization::addPointsToMap / mapSize (lioOptimization.h:347-357) on the device map */
int srl_lio_add_points_to_map(srl_lio *lio, const double *world_xyz, int n, double voxel_size,
                              int max_num_points_in_voxel, double min_distance_points, int min_num_points);
int srl_lio_map_size(srl_lio *lio, int64_t *num_points);
/* srl_map_probe_checksum over the world points the last srl_lio_commit_frame left in HBM (the frame the node inserts next) */
int srl_lio_probe_checksum_of_committed_frame(srl_lio *lio, int stride, double voxel_size, uint64_t *checksum, int32_t *num_voxels);

/* keep a sweep resident in HBM for the next srl_lio_update_iekf (pass raw_xyz = NULL there) */
int srl_lio_resident_sweep(srl_lio *lio, const double *raw_xyz, int n);
/* srl_sweep_prefetch / srl_sweep_swap behind the handle: upload the next sweep while srl_lio_update_iekf (raw_xyz = NULL)
 * solves the current one; the swapped-in sweep becomes the resident one. */
int srl_lio_prefetch_sweep(srl_lio *lio, const double *raw_xyz, int n);
int srl_lio_swap_sweep(srl_lio *lio);
/* The same prefetch, issued BY the next srl_lio_update_iekf while the kernel of its first pass is in flight (a node receives sweep k + 1
 * during the solve of sweep k, src/lioOptimization.cpp:1003-1027): the host time of the upload call (a memcpy enqueue and two event
 * records, ~5 us) then lies beside the association kernel instead of between two solves.  raw_xyz must stay valid and untouched until that
 * solve has returned (page-locked memory: until srl_sweep_wait / the first result on the swapped-in sweep, as for srl_sweep_prefetch). */
int srl_lio_prefetch_sweep_during_solve(srl_lio *lio, const double *raw_xyz, int n);

/* lioOptimization::updateIEKF (optimize.cpp:133-314).
 * state_io: p_frame->p_state = q(wxyz) t v ba bg (16 doubles) in/out; t_last = previous frame's
 * translation; log (optional): per iteration HtH(36) Hth(6) d_x(17) num_residuals loss = 61 doubles.
 * Returns SRL_OK with *iters >= 1, or SRL_ERR_NOT_ENOUGH_RESIDUALS / SRL_ERR_NAN_PLANARITY / ... */
int srl_lio_update_iekf(srl_lio *lio, const srl_icp_opts *opts, const double *raw_xyz, int n,
                        double state_io[16], const double t_last[3], int frame_id, double *log,
                        int max_log_iters, int *iters, int *num_residuals_used);

/* same update, the per-iteration normal equations coming from `provider` (multi-process CPU tests of
 * the sharded host logic; never used by the product path). */
typedef int (*srl_normal_eq_provider)(const srl_frame *frame, const srl_icp_opts *opts, srl_normal_eq *out, void *user);
/* The body of a node's loop over a stream of sweeps (src/lioOptimization.cpp:1003-1027: one optimize() per sweep) as ONE call, for replay
 * and benchmark loops whose own language would otherwise sit between the calls: reset the filter to the given prior (eskf_state[19],
 * eskf_cov[289]; NULL: keep what the filter holds), register the upload of the NEXT sweep with the solve (next_raw_xyz / next_n; NULL: none),
 * solve the resident sweep of n keypoints (srl_lio_update_iekf with raw_xyz = NULL), and make the next sweep the resident one
 * (srl_lio_swap_sweep; only when one was given).  Same statuses as the calls it stands for; the first failure is returned. */
int srl_lio_stream_step(srl_lio *lio, const srl_icp_opts *opts, const double eskf_state[19], const double eskf_cov[289], int n, double state_io[16],
                        const double t_last[3], int frame_id, const double *next_raw_xyz, int next_n, int *iters, int *num_residuals_used);

int srl_lio_update_iekf_provided(srl_lio *lio, const srl_icp_opts *opts, srl_normal_eq_provider provider,
                                 void *user, int n, double state_io[16], const double t_last[3], int frame_id,
                                 double *log, int max_log_iters, int *iters, int *num_residuals_used);

/* lioOptimization::optimize (optimize.cpp:428-448): gridSampling -> updateIEKF -> re-transform.
 * frame_raw / frame_world: point3D::raw_point / ::point of p_frame->point_frame (n x 3 each);
 * frame_world is overwritten with the re-transformed points on success; keypoint_index (optional,
 * capacity n) receives the indices gridSampling selected, in keypoint order. */
int srl_lio_optimize(srl_lio *lio, const srl_icp_opts *opts, double sample_voxel_size, const double *frame_raw,
                     double *frame_world, int n, double state_io[16], const double t_last[3], int frame_id,
                     int32_t *keypoint_index, int *num_keypoints, int *iters, int *num_residuals_used);

/* eskfEstimator::tryInit (eskfEstimator.cpp:43-118): imu_meas as t (n), gyr (n x 3), acc (n x 3).
 * *initialized: 1 initialised by this call (initial_flag set), 0 wait for more, -1 / -2 gyro / accel variance too large.
 * get_init_stats: mean_gyr, mean_acc, gyr_cov, acc_cov, num_init_meas, initial_flag (14 doubles).
 * initial_flag is process-global, as in the reference (utility.cpp:11). */
int srl_lio_eskf_try_init(srl_lio *lio, const double *t, const double *gyr, const double *acc, int n, int *initialized);
int srl_lio_eskf_get_init_stats(srl_lio *lio, double out[14]);
int srl_lio_set_initial_flag(srl_lio *lio, int flag);
/* lioOptimization::stateInitialization (lioOptimization.cpp:895-990): pose prior (q wxyz, t) of frame index_frame from
 * prev2 / prev1 = poses of all_cloud_frame[size-2] / [size-1]; initialization: 0 INIT_IMU, 1 INIT_CONSTANT_VELOCITY. */
int srl_lio_state_initialization(srl_lio *lio, int index_frame, int initialization, const double prev2[7],
                                 const double prev1[7], double out[7]);

/* ---- ROS-free replay driver: the LIO part of lioOptimization::run()'s loop body (lioOptimization.cpp:1427-1584):
 * tryInit while the filter is uninitialised; afterwards IMU propagation (predict per sample, imu_states), then
 * process() = stateInitialization -> buildFrame (device undistortion) -> stateEstimation (device keypoints, ESIKF,
 * device map insert) and the sliding window.  One call = one `Measurements` element: time_frame = time_image,
 * IMU samples (time, linear_acceleration, angular_velocity), the cut sweep (sensor-frame points + absolute point
 * timestamps) and time_sweep = (begin, offset).  srl_odometry_opts carries the odometryOptions fields the path
 * reads (parameters.h:58-88) plus the IMU noise parameters (setAccCov ...) and isPointTimeEnable(). */
typedef struct srl_odometry_opts {
    double init_voxel_size, init_sample_voxel_size;
    int init_num_frames, num_for_initialization;
    double voxel_size, sample_voxel_size;
    int max_num_points_in_voxel;
    double min_distance_points;
    int motion_compensation;          /* include/utility.h:82-86: 0 IMU, 1 CONSTANT_VELOCITY */
    int initialization;               /* include/utility.h:88-92: 0 INIT_IMU, 1 INIT_CONSTANT_VELOCITY */
    int point_time_enable;
    double acc_cov, gyr_cov, b_acc_cov, b_gyr_cov;
    srl_icp_opts icp;
} srl_odometry_opts;
typedef struct srl_replay_result {
    int processed;                    /* 0: the call only fed tryInit */
    int initialized;                  /* initial_flag after the call */
    int index_frame;                  /* after the call */
    int success, num_residuals_used, iterations;
    int frame_points, keypoints, points_added;
    double state[16];                 /* p_state of the frame: q wxyz, t, v, ba, bg */
} srl_replay_result;
int srl_lio_set_odometry_options(srl_lio *lio, const srl_odometry_opts *opts);
int srl_lio_run_measurement(srl_lio *lio, double time_frame, const double *imu_t, const double *imu_acc,
                            const double *imu_gyr, int n_imu, const double *pts_raw, const double *pts_timestamp,
                            int n_pts, double time_sweep_begin, double time_sweep_offset, srl_replay_result *out);
/* point3D::raw_point / ::point / ::imu_point of the newest frame in all_cloud_frame (any pointer may be NULL) */
int srl_lio_last_frame(srl_lio *lio, int capacity, double *raw_point, double *point, double *imu_point, int *n);

/* Frame-resident form of optimize() + the map update that follows it in stateEstimation
 * (lioOptimization.cpp:1027,1051): the raw frame (n x 3) is uploaded once; keypoints are selected on the device
 * from point = R(q)(R_il raw + t_il) + t with the PRIOR pose in state_io (what point3D::point holds when
 * optimize() is entered, lioOptimization.cpp:981-1001 -> utility.cpp:314-318) -- same keypoints, same order as
 * gridSampling -- and the ESIKF runs on them in place.  srl_lio_commit_frame then re-transforms the frame with
 * `state` (optimize.cpp:441-445) and inserts it (addPointsToMap) without the points leaving HBM;
 * world_out (n x 3, optional) receives point3D::point. */
int srl_lio_optimize_resident(srl_lio *lio, const srl_icp_opts *opts, double sample_voxel_size, const double *frame_raw,
                              int n, double state_io[16], const double t_last[3], int frame_id, int32_t *keypoint_index,
                              int *num_keypoints, int *iters, int *num_residuals_used);
int srl_lio_commit_frame(srl_lio *lio, const double state[16], double voxel_size, int max_num_points_in_voxel,
                         double min_distance_points, int min_num_points, double *world_out, int *num_added /* or NULL: see srl_frame_commit */);

/* lioOptimization::searchNeighbors / computeNeighborhoodDistribution single-call forms */
int srl_lio_search_neighbors(srl_lio *lio, const double point[3], int nb_voxels_visited, double size_voxel_map,
                             int max_num_neighbors, int threshold_voxel_capacity, double *out_xyz /* K x 3 */,
                             int16_t *out_voxels /* K x 3 or NULL */, int *num_found);
int srl_lio_neighborhood(srl_lio *lio, const double *pts, int n, double center[3], double normal[3],
                         double cov[9], double *a2D);

/* lioOptimization::buildPlaneResiduals, signature-compatible form (materialises the residual list):
 * outputs for the accepted residuals in order: raw_point(3) norm_vector(3) jacobians(6) norm_offset
 * distance weight = 15 doubles each, capacity max_out rows. */
int srl_lio_build_plane_residuals(srl_lio *lio, const srl_icp_opts *opts, const double *raw_xyz, int n,
                                  const double state[16], const double t_last[3], int frame_id,
                                  double *out_rows, int max_out, int *num_out, double *loss_sum,
                                  int *success, double *keypoint_world /* n x 3 or NULL */);

/* gridSampling (utility.cpp:188-201) alone: indices of the selected points, in output order */
int srl_grid_sampling(const double *world_xyz, int n, double size_voxel, int32_t *index_out, int *num_out);
/* debug / parity hook: iteration order of the std::tr1::unordered_map<voxel, ...> of subSampleFrame (utility.cpp:169-185) after
 * inserting n DISTINCT voxel keys in the given order, by the flat replay of the container's bucket moves that
 * srl_frame_select_keypoints uses (csrc/host/tr1_order.h; no container is built).  order_out[r] = index of the r-th element. */
int srl_debug_tr1_order(const int16_t *keys_xyz, int n, int32_t *order_out);
/* ... and the same order computed the way the DEVICE does it for frames that fit (csrc/host/tr1_relation.h: the container's iteration
 * order as a relation between two elements -- no replay; the functions the kernels call, compiled for the host). */
int srl_debug_tr1_order_by_relation(const int16_t *keys_xyz, int n, int32_t *order_out);

#ifdef __cplusplus
}
#endif
#endif
