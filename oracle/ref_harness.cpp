// oracle/ref_harness.cpp -- C ABI around the REFERENCE'S OWN translation units (TEST INFRASTRUCTURE ONLY).
//
// oracle/_ref/libref_path.so = this file + /root/reference/src/{optimize,lioOptimization,eskfEstimator,utility,state,
// cloudMap,parameters}.cpp compiled WHERE THEY LIE (never copied) against the stand-in third-party headers of
// oracle/ref_shim/ and the real vendored tsl::robin_map (recipe: oracle/Makefile, target `refpath`).  What runs behind every
// ref_* entry point is therefore the reference's source: lioOptimization::buildPlaneResiduals / updateIEKF / searchNeighbors
// / computeNeighborhoodDistribution / optimize (src/optimize.cpp); the node itself -- constructor, readParameters,
// imuHandler, getMeasurements, run, process, stateInitialization, makePointTimestamp, buildFrame, stateEstimation,
// addPointToMap / addPointsToMap (src/lioOptimization.cpp); eskfEstimator (src/eskfEstimator.cpp); gridSampling /
// subSampleFrame / transformPoint / distortFrameBy* / transformAllImuPoint / AngularDistance (src/utility.cpp); numType
// (include/utility.h); rgbPoint (src/cloudMap.cpp).  Ours are only (a) the third-party arithmetic (oracle/ref_shim/Eigen/
// Core: Eigen is absent from this image), (b) inert ROS / PCL / OpenCV stand-ins (publishers drop their messages, images
// are empty, parameters come from a stand-in parameter server the tests fill), and (c) no-op definitions of the sensor
// decoder (src/cloudProcessing.cpp) and of the vision stage's entry points, both out of scope: the harness
// puts decoded points straight into the node's point_buffer, exactly where cloudProcessing would.
// The ABI mirrors oracle/srl_oracle.h (row-major matrices, quaternions w,x,y,z) so tests/test_reference_tu.py can run the
// restatement and the reference side by side.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <iostream>
#include <map>
#include <memory>
#include <queue>
#include <sstream>
#include <stdexcept>
#include <string>
#include <tuple>
#include <unordered_map>
#include <vector>
#include <tr1/unordered_map>

#include "srl_oracle.h"      // orc_icp_opts (plain C struct shared with the restatement's ABI)

// the harness plays the role of the node: it has to set the members the node sets (all_cloud_frame, eskf_pro, voxel_map,
// extrinsics, laser_point_cov are private).  Access specifiers do not change layout or mangling.
#define private public
#define protected public
#include "lioOptimization.h"
#undef private
#undef protected

// ---- out-of-scope collaborators of the node, defined as no-ops so that src/lioOptimization.cpp links ----
// sensor decoder (src/cloudProcessing.cpp needs the ROS / Livox message layouts): the tests hand over decoded points
cloudProcessing::cloudProcessing() {
    lidar_type = LIVOX; N_SCANS = 6; SCAN_RATE = 10; time_unit = US; time_interval_sweep = 0.1; time_unit_scale = 1.e-3; blind = 0.01;
    given_offset_time = true; point_filter_num = 1; sweep_id = 0; delta_cut_time = 0.0; last_end_time = 0.0;
    R_imu_lidar = Eigen::Matrix3d::Identity(); t_imu_lidar = Eigen::Vector3d::Zero();
}
void cloudProcessing::setLidarType(int para) { lidar_type = para; }
void cloudProcessing::setNumScans(int para) { N_SCANS = para; }
void cloudProcessing::setScanRate(int para) { SCAN_RATE = para; time_interval_sweep = 1 / double(SCAN_RATE); }   // src/cloudProcessing.cpp:38-42
void cloudProcessing::setTimeUnit(int para) { time_unit = para; }
void cloudProcessing::setBlind(double para) { blind = para; }
void cloudProcessing::setExtrinR(Eigen::Matrix3d &R) { R_imu_lidar = R; }
void cloudProcessing::setExtrinT(Eigen::Vector3d &t) { t_imu_lidar = t; }
void cloudProcessing::setUseFeature(bool) {}
void cloudProcessing::setPointFilterNum(int para) { point_filter_num = para; }
void cloudProcessing::process(const sensor_msgs::PointCloud2::ConstPtr &, std::queue<point3D> &) {}
void cloudProcessing::livoxHandler(const livox_ros_driver::CustomMsg::ConstPtr &, std::queue<point3D> &) {}
// vision stage (oracle/ref_shim/local/imageProcessing.h)
imageProcessing::imageProcessing() { map_tracker = new rgbMapTracker(); }
void imageProcessing::setImageWidth(int &) {}
void imageProcessing::setImageHeight(int &) {}
void imageProcessing::setCameraIntrinsic(std::vector<double> &) {}
void imageProcessing::setCameraDistCoeffs(std::vector<double> &) {}
void imageProcessing::setExtrinR(Eigen::Matrix3d &) {}
void imageProcessing::setExtrinT(Eigen::Vector3d &) {}
Eigen::Matrix3d imageProcessing::getCameraIntrinsic() { return Eigen::Matrix3d::Identity(); }
void imageProcessing::process(voxelHashMap &, cloudFrame *) {}
void imageProcessing::printParameter() {}

// what a launch file would provide and readParameters() cannot default (vec3FromArray / mat33FromArray index their arrays)
static void ensure_default_params() {
    auto &n = ros::standin::num_params();
    auto &s = ros::standin::str_params();
    auto put = [&](const char *k, std::vector<double> v) { if (!n.count(k)) n[k] = v; };
    put("common/gravity_acc", {0.0, 0.0, 9.81});
    put("extrinsic_parameter/extrinsic_t_imu_lidar", {0, 0, 0});
    put("extrinsic_parameter/extrinsic_R_imu_lidar", {1, 0, 0, 0, 1, 0, 0, 0, 1});
    put("extrinsic_parameter/extrinsic_t_imu_camera", {0, 0, 0});
    put("extrinsic_parameter/extrinsic_R_imu_camera", {1, 0, 0, 0, 1, 0, 0, 0, 1});
    put("camera_parameter/camera_intrinsic", {1, 0, 0, 0, 1, 0, 0, 0, 1});
    put("camera_parameter/camera_dist_coeffs", {0, 0, 0, 0, 0});
    if (!s.count("output_path")) s["output_path"] = "/nonexistent_srl_ref_output";     // recordSinglePose appends to files there: must not exist
}

namespace {

inline Eigen::Vector3d v3(const double *p) { return Eigen::Vector3d(p[0], p[1], p[2]); }
inline Eigen::Matrix3d m3(const double *p) {
    Eigen::Matrix3d m;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) m(i, j) = p[3 * i + j];
    return m;
}
inline icpOptions to_opts(const orc_icp_opts *o) {
    icpOptions r;
    r.threshold_voxel_occupancy = o->threshold_voxel_occupancy;
    r.init_num_frames = o->init_num_frames;
    r.size_voxel_map = o->size_voxel_map;
    r.num_iters_icp = o->num_iters_icp;
    r.min_number_neighbors = o->min_number_neighbors;
    r.voxel_neighborhood = o->voxel_neighborhood;
    r.power_planarity = o->power_planarity;
    r.estimate_normal_from_neighborhood = o->estimate_normal_from_neighborhood != 0;
    r.max_number_neighbors = o->max_number_neighbors;
    r.max_dist_to_plane_icp = o->max_dist_to_plane_icp;
    r.threshold_orientation_norm = o->threshold_orientation_norm;
    r.threshold_translation_norm = o->threshold_translation_norm;
    r.max_num_residuals = o->max_num_residuals;
    r.weight_alpha = o->weight_alpha;
    r.weight_neighborhood = o->weight_neighborhood;
    r.debug_print = false;
    return r;
}

// the node object plus the two frames the path dereferences (all_cloud_frame[p_frame->id - 1] and p_frame)
struct ParamGuard { ParamGuard() { ensure_default_params(); } };
struct Node {
    ParamGuard guard;                     // before the node's constructor reads its parameters
    lioOptimization lio;
    state last_state, cur_state;
    std::vector<point3D> no_points;
    cloudFrame last_frame, cur_frame;
    Node() : last_frame(no_points, &last_state), cur_frame(no_points, &cur_state) {
        last_frame.id = 0; last_frame.sub_id = 0; last_frame.frame_id = 0;
        cur_frame.id = 1; cur_frame.sub_id = 0; cur_frame.frame_id = 1;
        lio.all_cloud_frame.push_back(&last_frame);
        lio.all_cloud_frame.push_back(&cur_frame);
    }
};

inline std::vector<point3D> make_keypoints(const double *raw_xyz, int n) {
    std::vector<point3D> kp((size_t)n);
    for (int i = 0; i < n; i++) {
        kp[i].raw_point = v3(raw_xyz + 3 * (size_t)i);
        kp[i].point = kp[i].raw_point;
        kp[i].imu_point = Eigen::Vector3d::Zero();
    }
    return kp;
}

}  // namespace

struct ref_map { voxelHashMap map; };
struct ref_eskf { eskfEstimator e; };

extern "C" {

const char *ref_describe(void) {
    return "reference translation units src/{optimize,eskfEstimator,utility,state,cloudMap}.cpp compiled in place; "
           "third-party: stand-in Eigen (oracle/ref_shim), vendored tsl::robin_map";
}

// ---- voxel map: the reference's own container and point type (include/cloudMap.h:147-184, src/cloudMap.cpp:5-29) ----
ref_map *ref_map_create(void) { return new ref_map(); }
void ref_map_destroy(ref_map *m) { delete m; }
// rebuilds a map from the arrays orc_map_export writes (voxel creation order; cap slots per voxel, AoS f32):
// voxelBlock::AddPoint(rgbPoint(position)) per resident point, in slot order
int ref_map_import(ref_map *m, const int16_t *keys, const int32_t *counts, const float *xyz, int V, int cap) {
    m->map.clear();
    for (int v = 0; v < V; v++) {
        voxel key(keys[3 * v], keys[3 * v + 1], keys[3 * v + 2]);
        if (m->map.find(key) != m->map.end()) return -1;
        voxelBlock block(cap);
        for (int s = 0; s < counts[v]; s++) {
            const float *p = xyz + ((size_t)v * cap + s) * 3;
            rgbPoint pt(Eigen::Vector3d((double)p[0], (double)p[1], (double)p[2]));
            block.AddPoint(pt);
        }
        m->map[key] = std::move(block);
    }
    return 0;
}
int ref_map_num_voxels(const ref_map *m) { return (int)m->map.size(); }
uint64_t ref_voxel_hash(int16_t x, int16_t y, int16_t z) { return (uint64_t)std::hash<voxel>()(voxel(x, y, z)); }

// ---- lioOptimization::searchNeighbors (src/optimize.cpp:365-426) for one query point ----
// out_xyz: up to K x 3 (ascending by distance, as returned), out_voxel: K x 3 shorts (the `voxels` out-parameter)
int ref_search_neighbors(ref_map *m, const double p[3], int nb_voxels_visited, double size_voxel_map, int max_num_neighbors,
                         int threshold_voxel_capacity, double *out_xyz, int16_t *out_voxel) {
    static Node node;                         // searchNeighbors reads no member
    std::vector<voxel> voxels;
    auto nb = node.lio.searchNeighbors(m->map, v3(p), nb_voxels_visited, size_voxel_map, max_num_neighbors, threshold_voxel_capacity,
                                       out_voxel ? &voxels : nullptr);
    for (size_t i = 0; i < nb.size(); i++) {
        out_xyz[3 * i] = nb[i][0]; out_xyz[3 * i + 1] = nb[i][1]; out_xyz[3 * i + 2] = nb[i][2];
        if (out_voxel) { out_voxel[3 * i] = voxels[i].x; out_voxel[3 * i + 1] = voxels[i].y; out_voxel[3 * i + 2] = voxels[i].z; }
    }
    return (int)nb.size();
}

// ---- lioOptimization::computeNeighborhoodDistribution (src/optimize.cpp:316-353).  Returns 0, -1 when it throws ----
int ref_neighborhood(const double *pts, int n, double center[3], double normal[3], double cov[9], double *a2D) {
    static Node node;                         // computeNeighborhoodDistribution reads no member
    std::vector<Eigen::Vector3d, Eigen::aligned_allocator<Eigen::Vector3d>> points;
    for (int i = 0; i < n; i++) points.push_back(v3(pts + 3 * (size_t)i));
    try {
        Neighborhood nb = node.lio.computeNeighborhoodDistribution(points);
        for (int i = 0; i < 3; i++) { center[i] = nb.center[i]; normal[i] = nb.normal[i]; for (int j = 0; j < 3; j++) cov[3 * i + j] = nb.covariance(i, j); }
        *a2D = nb.a2D;
        return 0;
    } catch (const std::runtime_error &) {
        return -1;
    }
}

// ---- lioOptimization::buildPlaneResiduals (src/optimize.cpp:18-131) ----
// point_world (n x 3): keypoint.point after the call.  The accepted residuals come back in the order the reference
// pushes them: res (capacity n x 15) = raw_point(3) norm_vector(3) jacobians(6) norm_offset distance weight.
// Returns the number of residuals pushed (plane_residuals.size()), -2 when computeNeighborhoodDistribution threw.
int ref_build_plane_residuals(ref_map *m, const orc_icp_opts *o, const double *raw_xyz, int n, const double q_wxyz[4],
                              const double t[3], const double t_last[3], const double R_il[9], const double t_il[3],
                              int frame_id, double *point_world, double *res, int *success, int *num_residuals_used,
                              double *loss_sum) {
    Node node;
    node.lio.R_imu_lidar = m3(R_il);
    node.lio.t_imu_lidar = v3(t_il);
    node.cur_frame.frame_id = frame_id;
    node.cur_state.rotation = Eigen::Quaterniond(q_wxyz[0], q_wxyz[1], q_wxyz[2], q_wxyz[3]);
    node.cur_state.translation = v3(t);
    node.last_state.translation = v3(t_last);
    const icpOptions opts = to_opts(o);
    std::vector<point3D> keypoints = make_keypoints(raw_xyz, n);
    std::vector<planeParam> plane_residuals;
    double loss = 0.0;
    optimizeSummary summary;
    try {
        summary = node.lio.buildPlaneResiduals(opts, m->map, keypoints, plane_residuals, &node.cur_frame, loss);
    } catch (const std::runtime_error &) {
        return -2;
    }
    if (point_world) for (int i = 0; i < n; i++) for (int c = 0; c < 3; c++) point_world[3 * (size_t)i + c] = keypoints[i].point[c];
    for (size_t i = 0; i < plane_residuals.size(); i++) {
        const planeParam &pp = plane_residuals[i];
        double *r = res + 15 * i;
        for (int c = 0; c < 3; c++) { r[c] = pp.raw_point[c]; r[3 + c] = pp.norm_vector[c]; }
        for (int c = 0; c < 6; c++) r[6 + c] = pp.jacobians(0, c);
        r[12] = pp.norm_offset; r[13] = pp.distance; r[14] = pp.weight;
    }
    *success = summary.success ? 1 : 0;
    *num_residuals_used = summary.num_residuals_used;
    *loss_sum = loss;
    return (int)plane_residuals.size();
}

// ---- eskfEstimator (src/eskfEstimator.cpp) ----
ref_eskf *ref_eskf_create(void) {
    ref_eskf *r = new ref_eskf();
    // acc_cov / gyr_cov are uninitialised Eigen members upstream (include/eskfEstimator.h:25); start them at zero like
    // the restatement does (the first sample multiplies them by (1 - 1.0))
    r->e.acc_cov = Eigen::Vector3d::Zero(); r->e.gyr_cov = Eigen::Vector3d::Zero();
    r->e.acc_cov_scale = Eigen::Vector3d::Zero(); r->e.gyr_cov_scale = Eigen::Vector3d::Zero();
    r->e.b_acc_cov = Eigen::Vector3d::Zero(); r->e.b_gyr_cov = Eigen::Vector3d::Zero();
    r->e.acc_0 = Eigen::Vector3d::Zero(); r->e.gyr_0 = Eigen::Vector3d::Zero();
    r->e.acc_1 = Eigen::Vector3d::Zero(); r->e.gyr_1 = Eigen::Vector3d::Zero();
    r->e.lxly = Eigen::Matrix<double, 3, 2>::Zero();
    r->e.time_first_imu = 0.0;
    r->e.dt = 0.0;
    return r;
}
void ref_eskf_destroy(ref_eskf *e) { delete e; }
// state vector layout: p(3) q(wxyz,4) v(3) ba(3) bg(3) g(3) = 19 doubles
void ref_eskf_get_state(ref_eskf *e, double s[19]) {
    const Eigen::Vector3d p = e->e.getTranslation(), v = e->e.getVelocity(), ba = e->e.getBa(), bg = e->e.getBg(), g = e->e.getGravity();
    const Eigen::Quaterniond q = e->e.getRotation();
    for (int i = 0; i < 3; i++) { s[i] = p[i]; s[7 + i] = v[i]; s[10 + i] = ba[i]; s[13 + i] = bg[i]; s[16 + i] = g[i]; }
    s[3] = q.w(); s[4] = q.x(); s[5] = q.y(); s[6] = q.z();
}
void ref_eskf_set_state(ref_eskf *e, const double s[19]) {
    e->e.setTranslation(v3(s));
    e->e.setRotation(Eigen::Quaterniond(s[3], s[4], s[5], s[6]));
    e->e.setVelocity(v3(s + 7)); e->e.setBa(v3(s + 10)); e->e.setBg(v3(s + 13)); e->e.setGravity(v3(s + 16));
}
void ref_eskf_get_cov(ref_eskf *e, double P[289]) {
    const Eigen::Matrix<double, 17, 17> c = e->e.getCovariance();
    for (int i = 0; i < 17; i++) for (int j = 0; j < 17; j++) P[17 * i + j] = c(i, j);
}
void ref_eskf_set_cov(ref_eskf *e, const double P[289]) {
    Eigen::Matrix<double, 17, 17> c;
    for (int i = 0; i < 17; i++) for (int j = 0; j < 17; j++) c(i, j) = P[17 * i + j];
    e->e.setCovariance(c);
}
// the node's sequence (src/lioOptimization.cpp: setAccCov ... ; tryInit copies the scales into acc_cov / gyr_cov and
// calls initializeNoise): here the noise matrix is filled directly from the four values, like orc_eskf_set_noise
void ref_eskf_set_noise(ref_eskf *e, double acc_cov, double gyr_cov, double b_acc_cov, double b_gyr_cov) {
    e->e.setAccCov(acc_cov); e->e.setGyrCov(gyr_cov); e->e.setBiasAccCov(b_acc_cov); e->e.setBiasGyrCov(b_gyr_cov);
    e->e.acc_cov = e->e.acc_cov_scale; e->e.gyr_cov = e->e.gyr_cov_scale;
    e->e.initializeNoise();
}
void ref_eskf_set_cov_scales(ref_eskf *e, double acc_cov, double gyr_cov, double b_acc_cov, double b_gyr_cov) {
    e->e.setAccCov(acc_cov); e->e.setGyrCov(gyr_cov); e->e.setBiasAccCov(b_acc_cov); e->e.setBiasGyrCov(b_gyr_cov);
}
void ref_eskf_init_imu(ref_eskf *e, const double acc0[3], const double gyr0[3]) { e->e.initializeImuData(v3(acc0), v3(gyr0)); }
void ref_eskf_predict(ref_eskf *e, double dt, const double acc1[3], const double gyr1[3]) { e->e.predict(dt, v3(acc1), v3(gyr1)); }
void ref_eskf_observe(ref_eskf *e, const double dx[17]) {
    Eigen::Matrix<double, 17, 1> d;
    for (int i = 0; i < 17; i++) d(i) = dx[i];
    e->e.observe(d);
}
// tryInit (src/eskfEstimator.cpp:43-91).  G_norm / initial_flag are globals of src/utility.cpp.
// Returns 1 when initial_flag became true in this call, else 0.  stats: mean_gyr, mean_acc, gyr_cov, acc_cov, num_init_meas, initial_flag
int ref_eskf_try_init(ref_eskf *e, const double *t, const double *gyr, const double *acc, int n, double g_norm, double stats[14]) {
    std::vector<std::pair<double, std::pair<Eigen::Vector3d, Eigen::Vector3d>>> meas;
    for (int i = 0; i < n; i++) meas.push_back({t[i], {v3(gyr + 3 * (size_t)i), v3(acc + 3 * (size_t)i)}});
    G_norm = g_norm;
    const bool before = initial_flag;
    std::streambuf *old = std::cout.rdbuf(nullptr);       // tryInit prints the initial gravity / bias
    std::streambuf *olde = std::cerr.rdbuf(nullptr);
    e->e.tryInit(meas);
    std::cout.rdbuf(old);
    std::cerr.rdbuf(olde);
    const int became = (!before && initial_flag) ? 1 : 0;
    for (int i = 0; i < 3; i++) { stats[i] = e->e.mean_gyr[i]; stats[3 + i] = e->e.mean_acc[i]; stats[6 + i] = e->e.gyr_cov[i]; stats[9 + i] = e->e.acc_cov[i]; }
    stats[12] = (double)e->e.num_init_meas;
    stats[13] = initial_flag ? 1.0 : 0.0;
    return became;
}
void ref_reset_globals(void) { initial_flag = false; G = Eigen::Vector3d::Zero(); G_norm = 0.0; }

// ---- lioOptimization::updateIEKF (src/optimize.cpp:133-314) ----
// state_io: the frame's p_state (q wxyz, t, v, ba, bg) = 16 doubles, in/out.  Returns 1 on success, -1 when
// summary.success is false, -2 when computeNeighborhoodDistribution threw.  keypoint world points come back in point_world.
int ref_update_iekf(ref_map *m, ref_eskf *e, const orc_icp_opts *o, const double *raw_xyz, int n, double state_io[16],
                    const double t_last[3], const double R_il[9], const double t_il[3], int frame_id, double laser_point_cov,
                    int *num_residuals_used, double *point_world) {
    Node node;
    node.lio.R_imu_lidar = m3(R_il);
    node.lio.t_imu_lidar = v3(t_il);
    node.lio.laser_point_cov = laser_point_cov;
    delete node.lio.eskf_pro;
    node.lio.eskf_pro = &e->e;
    node.cur_frame.frame_id = frame_id;
    node.cur_state.rotation = Eigen::Quaterniond(state_io[0], state_io[1], state_io[2], state_io[3]);
    node.cur_state.translation = v3(state_io + 4);
    node.cur_state.velocity = v3(state_io + 7);
    node.cur_state.ba = v3(state_io + 10);
    node.cur_state.bg = v3(state_io + 13);
    node.last_state.translation = v3(t_last);
    const icpOptions opts = to_opts(o);
    std::vector<point3D> keypoints = make_keypoints(raw_xyz, n);
    optimizeSummary summary;
    try {
        summary = node.lio.updateIEKF(opts, m->map, keypoints, &node.cur_frame);
    } catch (const std::runtime_error &) {
        return -2;
    }
    const state &s = node.cur_state;
    state_io[0] = s.rotation.w(); state_io[1] = s.rotation.x(); state_io[2] = s.rotation.y(); state_io[3] = s.rotation.z();
    for (int i = 0; i < 3; i++) { state_io[4 + i] = s.translation[i]; state_io[7 + i] = s.velocity[i]; state_io[10 + i] = s.ba[i]; state_io[13 + i] = s.bg[i]; }
    if (num_residuals_used) *num_residuals_used = summary.num_residuals_used;
    if (point_world) for (int i = 0; i < n; i++) for (int c = 0; c < 3; c++) point_world[3 * (size_t)i + c] = keypoints[i].point[c];
    return summary.success ? 1 : -1;
}

// ---- lioOptimization::optimize (src/optimize.cpp:428-447): gridSampling + updateIEKF + the re-transform of the frame ----
// frame_raw / frame_point (n x 3): point_frame's raw_point / point on entry; frame_point_out: point after the call.
// keypoint_index_out (capacity n): which frame points gridSampling picked, in keypoint order (matched by raw_point).
int ref_optimize(ref_map *m, ref_eskf *e, const orc_icp_opts *o, const double *frame_raw, const double *frame_point, int n,
                 double sample_voxel_size, double state_io[16], const double t_last[3], const double R_il[9], const double t_il[3],
                 int frame_id, double laser_point_cov, int *num_residuals_used, double *frame_point_out) {
    Node node;
    node.lio.R_imu_lidar = m3(R_il);
    node.lio.t_imu_lidar = v3(t_il);
    node.lio.laser_point_cov = laser_point_cov;
    delete node.lio.eskf_pro;
    node.lio.eskf_pro = &e->e;
    node.lio.voxel_map = m->map;
    node.cur_frame.frame_id = frame_id;
    node.cur_state.rotation = Eigen::Quaterniond(state_io[0], state_io[1], state_io[2], state_io[3]);
    node.cur_state.translation = v3(state_io + 4);
    node.cur_state.velocity = v3(state_io + 7);
    node.cur_state.ba = v3(state_io + 10);
    node.cur_state.bg = v3(state_io + 13);
    node.last_state.translation = v3(t_last);
    node.cur_frame.point_frame.resize((size_t)n);
    for (int i = 0; i < n; i++) {
        node.cur_frame.point_frame[i].raw_point = v3(frame_raw + 3 * (size_t)i);
        node.cur_frame.point_frame[i].point = v3(frame_point + 3 * (size_t)i);
        node.cur_frame.point_frame[i].imu_point = Eigen::Vector3d::Zero();
    }
    const icpOptions opts = to_opts(o);
    optimizeSummary summary;
    try {
        summary = node.lio.optimize(&node.cur_frame, opts, sample_voxel_size);
    } catch (const std::runtime_error &) {
        return -2;
    }
    const state &s = node.cur_state;
    state_io[0] = s.rotation.w(); state_io[1] = s.rotation.x(); state_io[2] = s.rotation.y(); state_io[3] = s.rotation.z();
    for (int i = 0; i < 3; i++) { state_io[4 + i] = s.translation[i]; state_io[7 + i] = s.velocity[i]; state_io[10 + i] = s.ba[i]; state_io[13 + i] = s.bg[i]; }
    if (num_residuals_used) *num_residuals_used = summary.num_residuals_used;
    for (int i = 0; i < n; i++) for (int c = 0; c < 3; c++) frame_point_out[3 * (size_t)i + c] = node.cur_frame.point_frame[i].point[c];
    return summary.success ? 1 : -1;
}

// ---- src/utility.cpp ----
// transformPoint (utility.cpp:314-318) over n raw points
void ref_transform_points(const double *raw_xyz, int n, const double q_wxyz[4], const double t[3], const double R_il[9],
                          const double t_il[3], double *world_xyz) {
    Eigen::Quaterniond q(q_wxyz[0], q_wxyz[1], q_wxyz[2], q_wxyz[3]);
    Eigen::Vector3d tt = v3(t), til = v3(t_il);
    Eigen::Matrix3d Ril = m3(R_il);
    for (int i = 0; i < n; i++) {
        point3D p;
        p.raw_point = v3(raw_xyz + 3 * (size_t)i);
        transformPoint(p, q, tt, Ril, til);
        for (int c = 0; c < 3; c++) world_xyz[3 * (size_t)i + c] = p.point[c];
    }
}
// gridSampling (utility.cpp:186-201): the frame index travels in point3D::index_frame
int ref_grid_sampling(const double *world_xyz, int n, double size_voxel, int32_t *idx_out) {
    std::vector<point3D> frame((size_t)n), keypoints;
    for (int i = 0; i < n; i++) { frame[i].point = v3(world_xyz + 3 * (size_t)i); frame[i].raw_point = frame[i].point; frame[i].imu_point = frame[i].point; frame[i].index_frame = i; }
    gridSampling(frame, keypoints, size_voxel);
    for (size_t i = 0; i < keypoints.size(); i++) idx_out[i] = keypoints[i].index_frame;
    return (int)keypoints.size();
}
namespace {
std::vector<imuState> make_imu_states(const double *s, int n_states) {
    std::vector<imuState> v((size_t)n_states);
    for (int i = 0; i < n_states; i++) {
        const double *p = s + 17 * (size_t)i;
        v[i].timestamp = p[0];
        v[i].un_acc = v3(p + 1); v[i].un_gyr = v3(p + 4); v[i].trans = v3(p + 7);
        v[i].quat = Eigen::Quaterniond(p[10], p[11], p[12], p[13]);
        v[i].vel = v3(p + 14);
    }
    return v;
}
}  // namespace
void ref_distort_frame_by_constant(const double *raw_xyz, const double *relative_time, int n, const double *imu_states, int n_states,
                                   double time_frame_begin, const double R_il[9], const double t_il[3], double *imu_point) {
    std::vector<point3D> pts((size_t)n);
    for (int i = 0; i < n; i++) { pts[i].raw_point = v3(raw_xyz + 3 * (size_t)i); pts[i].relative_time = relative_time[i]; pts[i].imu_point = v3(imu_point + 3 * (size_t)i); }
    std::vector<imuState> st = make_imu_states(imu_states, n_states);
    Eigen::Matrix3d Ril = m3(R_il); Eigen::Vector3d til = v3(t_il);
    distortFrameByConstant(pts, st, time_frame_begin, Ril, til);
    for (int i = 0; i < n; i++) for (int c = 0; c < 3; c++) imu_point[3 * (size_t)i + c] = pts[i].imu_point[c];
}
void ref_distort_frame_by_imu(const double *raw_xyz, const double *relative_time, int n, const double *imu_states, int n_states,
                              double time_frame_begin, const double R_il[9], const double t_il[3], double *imu_point) {
    std::vector<point3D> pts((size_t)n);
    for (int i = 0; i < n; i++) { pts[i].raw_point = v3(raw_xyz + 3 * (size_t)i); pts[i].relative_time = relative_time[i]; pts[i].imu_point = v3(imu_point + 3 * (size_t)i); }
    std::vector<imuState> st = make_imu_states(imu_states, n_states);
    Eigen::Matrix3d Ril = m3(R_il); Eigen::Vector3d til = v3(t_il);
    distortFrameByImu(pts, st, time_frame_begin, Ril, til);
    for (int i = 0; i < n; i++) for (int c = 0; c < 3; c++) imu_point[3 * (size_t)i + c] = pts[i].imu_point[c];
}
void ref_transform_all_imu_point(const double *imu_point, int n, const double *imu_states, int n_states, const double R_il[9],
                                 const double t_il[3], double *raw_xyz) {
    std::vector<point3D> pts((size_t)n);
    for (int i = 0; i < n; i++) { pts[i].imu_point = v3(imu_point + 3 * (size_t)i); pts[i].raw_point = Eigen::Vector3d::Zero(); }
    std::vector<imuState> st = make_imu_states(imu_states, n_states);
    Eigen::Matrix3d Ril = m3(R_il); Eigen::Vector3d til = v3(t_il);
    transformAllImuPoint(pts, st, Ril, til);
    for (int i = 0; i < n; i++) for (int c = 0; c < 3; c++) raw_xyz[3 * (size_t)i + c] = pts[i].raw_point[c];
}
double ref_angular_distance_so3(const double w[3]) { return AngularDistance(v3(w)); }

// ---- numType (include/utility.h:191-418), the reference's own templates ----
void ref_so3_to_rot(const double w[3], double R[9]) { Eigen::Matrix3d m = numType::so3ToRotation(v3(w)); for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R[3 * i + j] = m(i, j); }
void ref_so3_to_quat(const double w[3], double q[4]) { Eigen::Quaterniond r = numType::so3ToQuat(v3(w)); q[0] = r.w(); q[1] = r.x(); q[2] = r.y(); q[3] = r.z(); }
void ref_rot_to_so3(const double R[9], double w[3]) { Eigen::Vector3d r = numType::rotationToSo3(m3(R)); w[0] = r[0]; w[1] = r[1]; w[2] = r[2]; }
void ref_derivative_s2(const double g[3], double B[6]) { Eigen::Matrix<double, 3, 2> b = numType::derivativeS2(v3(g)); for (int i = 0; i < 3; i++) for (int j = 0; j < 2; j++) B[2 * i + j] = b(i, j); }

// =====================================================================================================================
// The node itself (src/lioOptimization.cpp): constructed by its own constructor from the stand-in parameter server, fed
// through its own imuHandler and buffers, advanced by its own run().
// =====================================================================================================================
void ref_param_clear(void) { ros::standin::num_params().clear(); ros::standin::str_params().clear(); }
void ref_param_set_num(const char *name, const double *v, int n) { ros::standin::num_params()[name] = std::vector<double>(v, v + n); }
void ref_param_set_str(const char *name, const char *v) { ros::standin::str_params()[name] = v; }

struct ref_node { ParamGuard guard; lioOptimization lio; };

// ---- wall time of every lioOptimization::optimize call the node makes (tests/test_gpu_integration.py compares the all-CPU node with
// the node that carries integration/optimize_hip.cpp).  Both node libraries are linked with
//   -Wl,--wrap=_ZN15lioOptimization8optimizeEP10cloudFrameRK10icpOptionsd
// so that stateEstimation's call (src/lioOptimization.cpp:1009, in another object file) lands in __wrap_..., which stamps the clock
// around __real_... = whichever optimize() the library was built with.  A non-static member function is called like a free function
// with `this` first (Itanium C++ ABI), including the hidden result pointer of the class it returns.
#include <chrono>
static std::vector<double> g_optimize_us;
extern "C" optimizeSummary __real__ZN15lioOptimization8optimizeEP10cloudFrameRK10icpOptionsd(lioOptimization *, cloudFrame *, const icpOptions &, double);
extern "C" optimizeSummary __wrap__ZN15lioOptimization8optimizeEP10cloudFrameRK10icpOptionsd(lioOptimization *self, cloudFrame *f, const icpOptions &o, double s) {
    const auto t0 = std::chrono::steady_clock::now();
    optimizeSummary r = __real__ZN15lioOptimization8optimizeEP10cloudFrameRK10icpOptionsd(self, f, o, s);
    g_optimize_us.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    return r;
}
int ref_optimize_times(double *out, int cap) {
    const int n = (int)g_optimize_us.size();
    for (int i = 0; i < n && i < cap; i++) out[i] = g_optimize_us[i];
    return n;
}
void ref_optimize_times_reset(void) { g_optimize_us.clear(); }

// point_time_enable: what cloudProcessing derives from the message fields (given_offset_time, src/cloudProcessing.cpp:228-233)
ref_node *ref_node_create(int point_time_enable) {
    std::streambuf *old = std::cout.rdbuf(nullptr);
    ref_node *n = new ref_node();
    std::cout.rdbuf(old);
    n->lio.cloud_pro->given_offset_time = point_time_enable != 0;
    // uninitialised Eigen members of eskfEstimator upstream (include/eskfEstimator.h:21-28): zero, like the restatement
    eskfEstimator &e = *n->lio.eskf_pro;
    e.acc_cov = Eigen::Vector3d::Zero(); e.gyr_cov = Eigen::Vector3d::Zero();
    e.acc_0 = Eigen::Vector3d::Zero(); e.gyr_0 = Eigen::Vector3d::Zero(); e.acc_1 = Eigen::Vector3d::Zero(); e.gyr_1 = Eigen::Vector3d::Zero();
    e.lxly = Eigen::Matrix<double, 3, 2>::Zero(); e.time_first_imu = 0.0; e.dt = 0.0;
    return n;
}
void ref_node_destroy(ref_node *n) { delete n; }

// sensor_msgs::Imu through the node's own imuHandler (src/lioOptimization.cpp:606-626)
void ref_node_push_imu(ref_node *n, double t, const double acc[3], const double gyr[3]) {
    sensor_msgs::Imu::Ptr msg(new sensor_msgs::Imu());
    msg->header.stamp = ros::Time().fromSec(t);
    msg->linear_acceleration.x = acc[0]; msg->linear_acceleration.y = acc[1]; msg->linear_acceleration.z = acc[2];
    msg->angular_velocity.x = gyr[0]; msg->angular_velocity.y = gyr[1]; msg->angular_velocity.z = gyr[2];
    n->lio.imuHandler(msg);
}
// what imageHandler leaves behind (src/lioOptimization.cpp:628-641): an (empty) image and its time stamp
void ref_node_push_image_time(ref_node *n, double t) {
    n->lio.img_buffer.push(cv::Mat());
    n->lio.time_img_buffer.push(t);
    n->lio.last_time_img = t;
}
// decoded LiDAR points as cloudProcessing::livoxHandler queues them (src/cloudProcessing.cpp:139-146):
// raw_point, point = raw_point, timestamp, alpha_time = 0
void ref_node_push_points(ref_node *n, const double *raw_xyz, const double *timestamp, int count) {
    for (int i = 0; i < count; i++) {
        point3D p;
        p.raw_point = v3(raw_xyz + 3 * (size_t)i);
        p.point = p.raw_point;
        p.imu_point = Eigen::Vector3d::Zero();
        p.timestamp = timestamp[i];
        p.relative_time = 0.0;
        p.alpha_time = 0.0;
        n->lio.point_buffer.push(p);
    }
}
// one call of lioOptimization::run() (src/lioOptimization.cpp:1427-1584).  info: index_frame, initial_flag, #frames in the
// window, frames ever built (all_cloud_frame.back()->frame_id), points left in point_buffer, map points, map voxels.
// Returns 0, or -2 when computeNeighborhoodDistribution threw.
int ref_node_run(ref_node *n, double info[8]) {
    std::streambuf *old = std::cout.rdbuf(nullptr);
    std::streambuf *olde = std::cerr.rdbuf(nullptr);
    int rc = 0;
    try { n->lio.run(); } catch (const std::runtime_error &) { rc = -2; }
    std::cout.rdbuf(old);
    std::cerr.rdbuf(olde);
    lioOptimization &l = n->lio;
    info[0] = l.index_frame; info[1] = initial_flag ? 1 : 0; info[2] = (double)l.all_cloud_frame.size();
    info[3] = l.all_cloud_frame.empty() ? 0 : l.all_cloud_frame.back()->frame_id;
    info[4] = (double)l.point_buffer.size(); info[5] = (double)l.mapSize(l.voxel_map); info[6] = (double)l.voxel_map.size();
    info[7] = l.current_time;
    return rc;
}
// the newest frame of the window: state16 (q wxyz, t, v, ba, bg), times (sweep begin, sweep end), frame_id, #points
int ref_node_last_frame_info(ref_node *n, double state16[16], double times[2], int *frame_id) {
    if (n->lio.all_cloud_frame.empty()) return -1;
    cloudFrame *f = n->lio.all_cloud_frame.back();
    const state &s = *f->p_state;
    state16[0] = s.rotation.w(); state16[1] = s.rotation.x(); state16[2] = s.rotation.y(); state16[3] = s.rotation.z();
    for (int i = 0; i < 3; i++) { state16[4 + i] = s.translation[i]; state16[7 + i] = s.velocity[i]; state16[10 + i] = s.ba[i]; state16[13 + i] = s.bg[i]; }
    times[0] = f->time_sweep_begin; times[1] = f->time_sweep_end;
    *frame_id = f->frame_id;
    return (int)f->point_frame.size();
}
// its points: raw_point, point, imu_point (n x 3 each), alpha_time, relative_time, timestamp (n each)
void ref_node_last_frame_points(ref_node *n, double *raw, double *point, double *imu_point, double *alpha, double *rel, double *ts) {
    cloudFrame *f = n->lio.all_cloud_frame.back();
    for (size_t i = 0; i < f->point_frame.size(); i++) {
        const point3D &p = f->point_frame[i];
        for (int c = 0; c < 3; c++) { raw[3 * i + c] = p.raw_point[c]; point[3 * i + c] = p.point[c]; imu_point[3 * i + c] = p.imu_point[c]; }
        alpha[i] = p.alpha_time; rel[i] = p.relative_time; ts[i] = p.timestamp;
    }
}
void ref_node_eskf_get(ref_node *n, double s[19], double P[289]) {
    eskfEstimator &e = *n->lio.eskf_pro;
    const Eigen::Vector3d p = e.getTranslation(), v = e.getVelocity(), ba = e.getBa(), bg = e.getBg(), g = e.getGravity();
    const Eigen::Quaterniond q = e.getRotation();
    for (int i = 0; i < 3; i++) { s[i] = p[i]; s[7 + i] = v[i]; s[10 + i] = ba[i]; s[13 + i] = bg[i]; s[16 + i] = g[i]; }
    s[3] = q.w(); s[4] = q.x(); s[5] = q.y(); s[6] = q.z();
    const Eigen::Matrix<double, 17, 17> c = e.getCovariance();
    for (int i = 0; i < 17; i++) for (int j = 0; j < 17; j++) P[17 * i + j] = c(i, j);
}
// the node object itself (the drop-in test asks integration/optimize_hip.cpp for the context bound to it)
const void *ref_node_lio_ptr(ref_node *n) { return &n->lio; }
// the LiDAR voxel map, in the container's iteration order: keys (V x 3), counts (V), xyz (V x cap x 3 f32); returns V
int ref_node_map_num_voxels(ref_node *n) { return (int)n->lio.voxel_map.size(); }
int ref_node_map_export(ref_node *n, int cap, int16_t *keys, int32_t *counts, float *xyz) {
    int v = 0;
    for (auto it = n->lio.voxel_map.begin(); it != n->lio.voxel_map.end(); ++it, ++v) {
        keys[3 * v] = it->first.x; keys[3 * v + 1] = it->first.y; keys[3 * v + 2] = it->first.z;
        auto &block = it.value();
        counts[v] = block.NumPoints();
        for (int s = 0; s < block.NumPoints() && s < cap; s++) {
            const Eigen::Vector3d p = block.points[s].getPosition();
            float *o = xyz + ((size_t)v * cap + s) * 3;
            o[0] = (float)p[0]; o[1] = (float)p[1]; o[2] = (float)p[2];
        }
    }
    return v;
}
// lioOptimization::addPointsToMap (src/lioOptimization.cpp:520-554) -> addPointToMap (:399-446) on world points, in order
int ref_node_add_points_to_map(ref_node *n, const double *world_xyz, int count, double voxel_size, int max_num_points_in_voxel,
                               double min_distance_points, int min_num_points) {
    lioOptimization &l = n->lio;
    state st;
    std::vector<point3D> pts((size_t)count);
    for (int i = 0; i < count; i++) { pts[i].point = v3(world_xyz + 3 * (size_t)i); pts[i].raw_point = pts[i].point; pts[i].imu_point = pts[i].point; }
    cloudFrame frame(pts, &st);
    frame.time_sweep_end = 1.0 + (double)l.voxel_map.size();     // only read by the colour map's bookkeeping
    const size_t before = l.mapSize(l.voxel_map);
    l.addPointsToMap(l.voxel_map, &frame, voxel_size, max_num_points_in_voxel, min_distance_points, min_num_points, false);
    return (int)(l.mapSize(l.voxel_map) - before);
}
// lioOptimization::stateInitialization (src/lioOptimization.cpp:895-990).  prev2 / prev1 = (q wxyz, t) of
// all_cloud_frame[size-2] / [size-1]; the node's `initialization` comes from the parameter server
void ref_node_state_initialization(ref_node *n, int index_frame, int initial_flag_, const double prev2[7], const double prev1[7],
                                   const double eskf_q[4], const double eskf_t[3], double out[7]) {
    lioOptimization &l = n->lio;
    state s2, s1, cur;
    s2.rotation = Eigen::Quaterniond(prev2[0], prev2[1], prev2[2], prev2[3]); s2.translation = v3(prev2 + 4);
    s1.rotation = Eigen::Quaterniond(prev1[0], prev1[1], prev1[2], prev1[3]); s1.translation = v3(prev1 + 4);
    std::vector<point3D> none;
    cloudFrame f2(none, &s2), f1(none, &s1);
    std::vector<cloudFrame *> saved = l.all_cloud_frame;
    l.all_cloud_frame.clear(); l.all_cloud_frame.push_back(&f2); l.all_cloud_frame.push_back(&f1);
    const int saved_index = l.index_frame; const bool saved_flag = initial_flag;
    l.index_frame = index_frame; initial_flag = initial_flag_ != 0;
    l.eskf_pro->setRotation(Eigen::Quaterniond(eskf_q[0], eskf_q[1], eskf_q[2], eskf_q[3])); l.eskf_pro->setTranslation(v3(eskf_t));
    l.stateInitialization(&cur);
    out[0] = cur.rotation.w(); out[1] = cur.rotation.x(); out[2] = cur.rotation.y(); out[3] = cur.rotation.z();
    for (int i = 0; i < 3; i++) out[4 + i] = cur.translation[i];
    l.all_cloud_frame = saved; l.index_frame = saved_index; initial_flag = saved_flag;
}
// lioOptimization::makePointTimestamp (src/lioOptimization.cpp:786-819): returns the number of points kept; keep_index (n)
int ref_node_make_point_timestamp(ref_node *n, const double *timestamp, int count, double time_begin, double time_end,
                                  double *relative_time, double *alpha_time, int32_t *keep_index) {
    std::vector<point3D> sweep((size_t)count);
    for (int i = 0; i < count; i++) { sweep[i].timestamp = timestamp[i]; sweep[i].index_frame = i; sweep[i].raw_point = Eigen::Vector3d::Zero(); sweep[i].point = sweep[i].raw_point; sweep[i].imu_point = sweep[i].raw_point; }
    n->lio.makePointTimestamp(sweep, time_begin, time_end);
    for (size_t i = 0; i < sweep.size(); i++) { relative_time[i] = sweep[i].relative_time; alpha_time[i] = sweep[i].alpha_time; keep_index[i] = sweep[i].index_frame; }
    return (int)sweep.size();
}

}  // extern "C"
