/*
 * srl_oracle.h -- C ABI of the CPU ORACLE for the SR-LIVO LIO scan-matching hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and there
 * only as the checker / the timed CPU baseline -- never as the thing shipped.
 *
 * PINNED AGAINST THE REFERENCE'S OWN SOURCE: the reference (ZikangYuan/sr_livo) ships no tests, golden vectors or
 * fixtures for this path and its build needs Eigen/ROS/PCL/OpenCV/Ceres, which are absent
 * here -- but src/{optimize,lioOptimization,eskfEstimator,utility,state,cloudMap,parameters}.cpp compile where they lie
 * against stand-in third-party headers (oracle/ref_shim/, oracle/ref_harness.cpp -> oracle/_ref/libref_path.so), and
 * tests/test_reference_tu.py requires this restatement to equal that library BITWISE (residual fields, neighbour lists on
 * tied distances, solved state and covariance, the whole node's run() over 40 sweeps, the final map).  What is restated
 * rather than compiled is the third-party arithmetic only (Eigen 3.3.7's SelfAdjointEigenSolver and PartialPivLU inverse:
 * orc_eigen337.h; quaternion helpers; product / reduction order).  The oracle restates
 *   src/optimize.cpp:18-448, include/cloudMap.h:37-184, src/cloudMap.cpp:5-29,
 *   src/lioOptimization.cpp:400-446,520-554,574-581,786-990, src/eskfEstimator.cpp:3-21,43-230,
 *   include/utility.h:191-331, src/utility.cpp:146-332
 * and is additionally checked by (1) an independent NumPy implementation in tests/, (2) analytic cases, (3) the
 * known-answer values of SURVEY.md Appendix D, and (4) a build against the real vendored tsl::robin_map
 * (oracle/_ref/liboracle_tsl.so, see oracle/Makefile).
 *
 * All matrices are ROW-MAJOR in this ABI.  Quaternions are (w, x, y, z).
 */
#ifndef SRL_ORACLE_H
#define SRL_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* mirrors icpOptions (include/parameters.h:8-56); defaults = effective r3live.yaml values */
typedef struct orc_icp_opts {
    int    threshold_voxel_occupancy;   /* 1   */
    int    init_num_frames;             /* 20  */
    double size_voxel_map;              /* 1.0 */
    int    num_iters_icp;               /* 5   */
    int    min_number_neighbors;        /* 20  */
    int    voxel_neighborhood;          /* 1   */
    double power_planarity;             /* 2.0 */
    int    estimate_normal_from_neighborhood; /* 1 (always true in the reference) */
    int    max_number_neighbors;        /* 20  */
    double max_dist_to_plane_icp;       /* 0.3 */
    double threshold_orientation_norm;  /* 0.1 deg */
    double threshold_translation_norm;  /* 0.01 m  */
    int    max_num_residuals;           /* 600 (shipped) */
    double weight_alpha;                /* 0.9 */
    double weight_neighborhood;         /* 0.1 */
} orc_icp_opts;

void orc_icp_opts_default(orc_icp_opts *o);

/* ---- voxel map (cloudMap.h:124-184 + lioOptimization.cpp:400-446,520-554) ---- */
typedef struct orc_map orc_map;
orc_map *orc_map_create(void);
void     orc_map_destroy(orc_map *m);
/* addPointsToMap: xyz = AoS world points (n x 3 f64), inserted in order.  Returns #points added. */
int      orc_map_add_points(orc_map *m, const double *xyz, int n, double voxel_size,
                            int max_num_points_in_voxel, double min_distance_points, int min_num_points);
size_t   orc_map_size(const orc_map *m);          /* mapSize(): total points */
int      orc_map_num_voxels(const orc_map *m);
/* export in voxel CREATION order: keys[V*3], counts[V], xyz[V*cap*3] (AoS f32, unused slots 0) */
void     orc_map_export(const orc_map *m, int cap, int16_t *keys, int32_t *counts, float *xyz);
/* tests only: rebuild the map from exported arrays without the insertion rules (maps addPointToMap cannot produce) */
int      orc_map_import(orc_map *m, const int16_t *keys, const int32_t *counts, const float *xyz, int V, int cap);
/* std::hash<voxel> (cloudMap.h:173-184) */
uint64_t orc_voxel_hash(int16_t x, int16_t y, int16_t z);
/* static_cast<short>(v / size) (optimize.cpp:372) */
int16_t  orc_voxel_coord(double v, double size);
const char *orc_map_backend(void);                /* "tsl::robin_map" or "std::unordered_map" */

/* ---- searchNeighbors (optimize.cpp:365-426) for one query point.
 * out_xyz: up to K x 3 f64 ascending by distance, out_ids: voxel_index*cap + slot, out_dist: distances.
 * *tie_flag = 1 when the K+1 smallest candidate distances contain an exact tie.
 * *num_candidates = number of resident points visited (P_k).  Returns number of neighbours. */
int orc_search_neighbors(orc_map *m, const double p[3], int nb_voxels_visited, double size_voxel_map,
                         int max_num_neighbors, int threshold_voxel_capacity, int cap,
                         double *out_xyz, int32_t *out_ids, double *out_dist, int *tie_flag,
                         int *num_candidates);

/* ---- computeNeighborhoodDistribution (optimize.cpp:316-353).  Returns 0, or -1 on NaN a2D. */
int orc_neighborhood(const double *pts, int n, double center[3], double normal[3], double cov[9],
                     double *a2D, double eigenvalues[3]);

/* ---- buildPlaneResiduals (optimize.cpp:18-131) ----
 * status per keypoint: 0 = fewer than min_number_neighbors, 1 = plane found but distance gate
 * rejected, 2 = accepted residual, 3 = not visited (loop left at max_num_residuals).
 * Any per-keypoint output pointer may be NULL. */
typedef struct orc_residual_out {
    uint8_t *status;      /* N      */
    int32_t *ids;         /* N x K  (-1 padded) */
    uint8_t *tie;         /* N      */
    double  *point_world; /* N x 3  keypoint.point (optimize.cpp:38) */
    double  *normal;      /* N x 3  unit normal after flip */
    double  *a2D;         /* N      */
    double  *weight;      /* N      */
    double  *norm_offset; /* N      */
    double  *distance;    /* N      */
    double  *jacobian;    /* N x 6  */
} orc_residual_out;

typedef struct orc_normal_eq {
    double  HtH[36];      /* H_x^T H_x   (optimize.cpp:235) */
    double  Hth[6];       /* H_x^T h     (optimize.cpp:239) */
    double  loss_sum;     /* sum d^2     (optimize.cpp:104) */
    int32_t num_residuals;
    int32_t success;      /* optimizeSummary.success (optimize.cpp:110-130) */
    int64_t sum_candidates;   /* sum of P_k over visited keypoints */
    int32_t num_visited;
    int32_t num_ties;
    int32_t nan_error;    /* NaN a2D encountered (optimize.cpp:348-350 would throw) */
} orc_normal_eq;

int orc_build_plane_residuals(orc_map *m, const orc_icp_opts *o, const double *raw_xyz, int n,
                              const double q_wxyz[4], const double t[3], const double t_last[3],
                              const double R_il[9], const double t_il[3], int frame_id, int cap,
                              orc_residual_out *out, orc_normal_eq *neq);

/* ---- eskfEstimator (eskfEstimator.cpp) ---- */
typedef struct orc_eskf orc_eskf;
orc_eskf *orc_eskf_create(void);                   /* ctor state, eskfEstimator.cpp:3-21 */
void      orc_eskf_destroy(orc_eskf *e);
/* state vector layout: p(3) q(wxyz,4) v(3) ba(3) bg(3) g(3) = 19 doubles */
void orc_eskf_get_state(const orc_eskf *e, double s[19]);
void orc_eskf_set_state(orc_eskf *e, const double s[19]);
void orc_eskf_get_cov(const orc_eskf *e, double P[289]);
void orc_eskf_set_cov(orc_eskf *e, const double P[289]);
void orc_eskf_set_noise(orc_eskf *e, double acc_cov, double gyr_cov, double b_acc_cov, double b_gyr_cov);
void orc_eskf_init_imu(orc_eskf *e, const double acc0[3], const double gyr0[3]);
void orc_eskf_scale_init_cov(orc_eskf *e);         /* tryInit covariance scaling, eskfEstimator.cpp:74-76 */
/* tryInit (eskfEstimator.cpp:43-118): t (n), gyr / acc (n x 3) = imu_meas {time, {gyr, acc}}.  Returns 1 when the
 * filter became initialised in this call (initial_flag = true), 0 "wait more", -1 / -2 gyro / accelerometer
 * variance too large.  acc_cov / gyr_cov start at zero here (uninitialised Eigen members upstream; the first
 * sample multiplies them by 0).  get_init_stats: mean_gyr, mean_acc, gyr_cov, acc_cov, num_init_meas, initial_flag. */
int  orc_eskf_try_init(orc_eskf *e, const double *t, const double *gyr, const double *acc, int n);
void orc_eskf_get_init_stats(const orc_eskf *e, double out[14]);
void orc_eskf_set_g_norm(orc_eskf *e, double g_norm);
/* stateInitialization (lioOptimization.cpp:895-990): prev2 / prev1 = (q wxyz, t) of all_cloud_frame[size-2] / [size-1];
 * initialization: 0 INIT_IMU, 1 INIT_CONSTANT_VELOCITY (utility.h:88-92), other = copy the last pose. */
void orc_state_initialization(int index_frame, int initialization, int initial_flag, const double prev2[7],
                              const double prev1[7], const double eskf_q[4], const double eskf_t[3], double out[7]);
void orc_eskf_predict(orc_eskf *e, double dt, const double acc1[3], const double gyr1[3]);
void orc_eskf_observe(orc_eskf *e, const double dx[17]);

/* ---- updateIEKF (optimize.cpp:133-314) ----
 * state_io: the frame's p_state (q wxyz, t, v, ba, bg) = 16 doubles, in/out.
 * log (optional): per iteration HtH(36) Hth(6) dx(17) num_residuals loss = 61 doubles, max_log_iters rows.
 * Returns number of iterations executed (>=1), negative on failure:
 *   -1 not enough residuals (summary.success=false), -2 NaN planarity. */
int orc_update_iekf(orc_map *m, orc_eskf *e, const orc_icp_opts *o, const double *raw_xyz, int n,
                    double state_io[16], const double t_last[3], const double R_il[9],
                    const double t_il[3], int frame_id, int cap, double laser_point_cov,
                    double *log, int max_log_iters, int *num_residuals_used);

/* ---- frame side of optimize() (rows f1/f2) ----
 * orc_transform_points: transformPoint (utility.cpp:314-318) over n raw points:
 *     point = q.toRotationMatrix() * (R_il * raw + t_il) + t   (q used as is, not normalised)
 * orc_grid_sampling: gridSampling -> subSampleFrame (utility.cpp:167-201): key = short(point / size), first point
 *     of every voxel, emitted in std::tr1::unordered_map iteration order.  idx_out (capacity n) receives the
 *     frame indices of the keypoints, in keypoint order; returns their number. */
void orc_transform_points(const double *raw_xyz, int n, const double q_wxyz[4], const double t[3], const double R_il[9],
                          const double t_il[3], double *world_xyz);
int  orc_grid_sampling(const double *world_xyz, int n, double size_voxel, int32_t *idx_out);

/* ---- sweep reconstruction (row f4): buildFrame's per-point stages (lioOptimization.cpp:821-850) ----
 * imu_states: n_states x 17 doubles = timestamp, un_acc(3), un_gyr(3), trans(3), quat wxyz(4), vel(3) (cloudMap.h:110-122).
 * relative_time in milliseconds (makePointTimestamp, lioOptimization.cpp:786-819).
 * orc_distort_frame_by_constant / _by_imu (utility.cpp:203-306): write imu_point (n x 3); _by_imu leaves the points
 *   the sequential interval walk never reaches untouched (imu_point is in/out) and returns how many it wrote.
 * orc_transform_all_imu_point (utility.cpp:320-332): raw_point <- lidar frame at the sweep end.
 * orc_build_frame_order (lioOptimization.cpp:840-848): the index order buildFrame leaves the frame in:
 *   std::shuffle with a default-seeded 64-bit Mersenne twister (boost::mt19937_64 == std::mt19937_64, seed 5489),
 *   subSampleFrame on point3D::point (= the sensor-frame point, cloudProcessing.cpp:143), the same engine shuffles again.
 *   Returns the number of kept points; order_out has capacity n.
 * orc_make_point_timestamp (lioOptimization.cpp:786-819): point_time_enable != 0 keeps every point (alpha clamp
 *   1 - 1e-5), == 0 erases points outside [time_begin, time_end]; keep_out (n) flags the survivors. */
void orc_distort_frame_by_constant(const double *raw_xyz, const double *relative_time, int n, const double *imu_states,
                                   int n_states, double time_frame_begin, const double R_il[9], const double t_il[3],
                                   double *imu_point);
int  orc_distort_frame_by_imu(const double *raw_xyz, const double *relative_time, int n, const double *imu_states,
                              int n_states, double time_frame_begin, const double R_il[9], const double t_il[3],
                              double *imu_point);
void orc_transform_all_imu_point(const double *imu_point, int n, const double *imu_states, int n_states,
                                 const double R_il[9], const double t_il[3], double *raw_xyz);
int  orc_build_frame_order(const double *point_xyz, int n, double sample_size, int do_subsample, int32_t *order_out);
int  orc_make_point_timestamp(const double *timestamp, int n, double time_begin, double time_end, int point_time_enable,
                              double *relative_time, double *alpha_time, uint8_t *keep_out);
uint64_t orc_mt19937_64_nth(int nth);            /* KAT: the 10000th output of a default-seeded engine */

/* small numeric helpers exported for unit tests (utility.h numType, utility.cpp:146-153) */
void   orc_quat_to_rot(const double q_wxyz[4], double R[9]);
void   orc_rot_to_quat(const double R[9], double q_wxyz[4]);
void   orc_so3_to_rot(const double w[3], double R[9]);
void   orc_so3_to_quat(const double w[3], double q_wxyz[4]);
void   orc_rot_to_so3(const double R[9], double w[3]);
double orc_angular_distance_so3(const double w[3]);
void   orc_derivative_s2(const double g[3], double B[6]);
int    orc_inverse17(const double A[289], double Ainv[289]);
void   orc_eig3(const double A[9], double evals[3], double evecs[9]); /* ascending; evecs columns, row-major; current solver */
/* SelfAdjointEigenSolver<Matrix3d> (optimize.cpp:339).  solver 0 = Eigen 3.3.7's own algorithm restated (3x3
 * tridiagonalisation + implicit symmetric QR with Wilkinson shift; the default everywhere in the oracle), 1 = FP64 cyclic
 * Jacobi (an independent second solver used to bound the Eigen-boundary uncertainty).  orc_eig3_solver returns -1 when
 * the QR iteration did not converge (Eigen: info() == NoConvergence). */
int    orc_eig3_solver(int solver, const double A[9], double evals[3], double evecs[9]);
void   orc_set_eig_solver(int solver);     /* process-wide: solver used by computeNeighborhoodDistribution */
int    orc_get_eig_solver(void);
/* ORACLE ADDITION: threads of the keypoint loop of buildPlaneResiduals.  1 (default) = the reference's own single-threaded
 * loop.  > 1 = the all-cores CPU baseline: blocks of keypoints are visited in parallel (OpenMP) and committed in keypoint
 * order, so the ordered cut-off at max_num_residuals and every sum are bit-identical to the single-threaded run. */
void   orc_set_threads(int threads);
int    orc_get_threads(void);
/* the literal std::priority_queue sequence of searchNeighbors (optimize.cpp:394-404, 411-422) on a list of distances
 * offered in index order; out_index (capacity max_num_neighbors) = surviving indices in read-out order */
int    orc_heap_topk(const double *distances, int n, int max_num_neighbors, int32_t *out_index);

#ifdef __cplusplus
}
#endif
#endif
