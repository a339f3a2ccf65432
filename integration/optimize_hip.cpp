// integration/optimize_hip.cpp -- the MI355X binding of SR-LIVO's LIO scan-matching path, as a REPLACEMENT FOR src/optimize.cpp
// of ZikangYuan/sr_livo.  Build the node with this file instead of src/optimize.cpp, add include/ of this repository to the
// include path and link libsrlivo_hip.so: nothing else changes -- no header of the reference is edited, lioOptimization.cpp
// (constructor, run, process, buildFrame, stateEstimation, addPointsToMap ...) is compiled as it stands.
//
// It defines the five member functions src/optimize.cpp defines (declared at include/lioOptimization.h:334-343), with the
// reference's signatures and Eigen types:
//     optimize                          optimize.cpp:428-448   gridSampling -> updateIEKF -> re-transform of the frame
//     updateIEKF                        optimize.cpp:133-314   the ESIKF loop: association passes on the GPU, 17-dim update
//     buildPlaneResiduals               optimize.cpp:18-131    one pass with the residual list materialised
//     computeNeighborhoodDistribution   optimize.cpp:316-353
//     searchNeighbors                   optimize.cpp:365-426
// Each forwards to the C handles of include/srlivo_host.h (one srl_lio = one GPU context + the host mirror of the update).
// The filter stays the node's own eskfEstimator: its state and covariance are handed to the call and read back
// (eskfEstimator.h:86-108), p_frame->p_state and the globals G / G_norm are written as optimize.cpp:255-261 writes them.
//
// The voxel map.  The node keeps inserting sweeps into its tsl::robin_map (addPointsToMap, lioOptimization.cpp:1027 -- still
// needed by the colour map / rendering side); the path reads the DEVICE map.  Both are fed the same points in the same order:
// before a solve, every frame of the sliding window that the node has inserted since the last solve -- the frames before
// frame_id 2, which stateEstimation inserts without calling optimize, and every frame whose optimize() succeeded
// (lioOptimization.cpp:1003-1027) -- is inserted on the device with the parameters stateEstimation uses (srl_map_insert
// reproduces addPointToMap's order-dependent semantics bit for bit).  mapSize() of the two maps is compared after every
// sync: a difference is a hard error, not a silent drift.
//
// tests/test_gpu_integration.py compiles this file together with the reference's own translation units
// (oracle/Makefile, target refnode_hip) and drives the reference's run() over the 40-sweep replay stream.
#include <cmath>
#include <cstdint>
#include <map>
#include <mutex>
#include <sstream>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include <Eigen/Core>

#include "cloudMap.h"
#include "lioOptimization.h"
#include "parameters.h"
#include "utility.h"

#include "srlivo_host.h"

namespace {

struct HipBinding {
    srl_lio *lio = nullptr;
    int last_frame_handled = -1;              // frame_id up to which the window's frames have been looked at for the map sync
    std::map<int, bool> solved;               // frame_id -> optimize() succeeded (the node inserts exactly those, lioOptimization.cpp:1011-1027)
};

std::mutex g_mutex;
std::unordered_map<const lioOptimization *, HipBinding> g_bindings;

[[noreturn]] void fail(srl_lio *lio, int rc, const char *what) {
    std::string msg = std::string(what) + ": " + srl_status_str(rc);
    if (lio) { msg += " ("; msg += srl_lio_last_error(lio); msg += ")"; }
    throw std::runtime_error(msg);
}

HipBinding &binding_of(const lioOptimization *self) {
    std::lock_guard<std::mutex> lk(g_mutex);
    HipBinding &b = g_bindings[self];
    if (!b.lio) {
        const int rc = srl_lio_create(0, &b.lio);          // no CPU fallback: without a GPU the node cannot run this path
        if (rc != SRL_OK) fail(nullptr, rc, "srl_lio_create");
    }
    return b;
}

srl_icp_opts to_abi(const icpOptions &o) {
    srl_icp_opts s;
    srl_icp_opts_default(&s);
    s.threshold_voxel_occupancy = o.threshold_voxel_occupancy;
    s.init_num_frames = o.init_num_frames;
    s.size_voxel_map = o.size_voxel_map;
    s.num_iters_icp = o.num_iters_icp;
    s.min_number_neighbors = o.min_number_neighbors;
    s.voxel_neighborhood = o.voxel_neighborhood;
    s.power_planarity = o.power_planarity;
    s.max_number_neighbors = o.max_number_neighbors;
    s.max_dist_to_plane_icp = o.max_dist_to_plane_icp;
    s.threshold_orientation_norm = o.threshold_orientation_norm;
    s.threshold_translation_norm = o.threshold_translation_norm;
    s.max_num_residuals = o.max_num_residuals;
    s.weight_alpha = o.weight_alpha;
    s.weight_neighborhood = o.weight_neighborhood;
    s.select_mode = 0;
    return s;
}

void pack_state16(const state *s, double st[16]) {        // q (w x y z), t, v, ba, bg
    st[0] = s->rotation.w(); st[1] = s->rotation.x(); st[2] = s->rotation.y(); st[3] = s->rotation.z();
    for (int a = 0; a < 3; a++) { st[4 + a] = s->translation[a]; st[7 + a] = s->velocity[a]; st[10 + a] = s->ba[a]; st[13 + a] = s->bg[a]; }
}
void unpack_state16(const double st[16], state *s) {
    s->rotation = Eigen::Quaterniond(st[0], st[1], st[2], st[3]);
    s->translation = Eigen::Vector3d(st[4], st[5], st[6]);
    s->velocity = Eigen::Vector3d(st[7], st[8], st[9]);
    s->ba = Eigen::Vector3d(st[10], st[11], st[12]);
    s->bg = Eigen::Vector3d(st[13], st[14], st[15]);
}

std::vector<double> raw_points_of(const std::vector<point3D> &pts) {
    std::vector<double> raw(3 * pts.size());
    for (size_t k = 0; k < pts.size(); ++k) for (int d = 0; d < 3; d++) raw[3 * k + d] = pts[k].raw_point[d];
    return raw;
}

}  // namespace

// test / diagnostics hook: the context behind a node's binding (NULL before the first call into the path)
extern "C" srl_ctx *srl_integration_ctx(const void *node) {
    std::lock_guard<std::mutex> lk(g_mutex);
    auto it = g_bindings.find(static_cast<const lioOptimization *>(node));
    return it == g_bindings.end() || !it->second.lio ? nullptr : srl_lio_ctx(it->second.lio);
}
extern "C" void srl_integration_release(const void *node) {
    std::lock_guard<std::mutex> lk(g_mutex);
    auto it = g_bindings.find(static_cast<const lioOptimization *>(node));
    if (it == g_bindings.end()) return;
    if (it->second.lio) srl_lio_destroy(it->second.lio);
    g_bindings.erase(it);
}

// ---- the device map follows the node's map (see the header of this file) ----
static void sync_device_map(lioOptimization *self, HipBinding &b, const std::vector<cloudFrame *> &window, const cloudFrame *current,
                            const odometryOptions &oo, voxelHashMap &host_map) {
    for (const cloudFrame *f : window) {
        if (f == current || f->frame_id <= b.last_frame_handled) continue;
        const bool inserted_by_node = f->frame_id <= 1 || (b.solved.count(f->frame_id) && b.solved[f->frame_id]);
        if (inserted_by_node && !f->point_frame.empty()) {
            std::vector<double> xyz(3 * f->point_frame.size());
            for (size_t k = 0; k < f->point_frame.size(); ++k) for (int d = 0; d < 3; d++) xyz[3 * k + d] = f->point_frame[k].point[d];
            const int rc = srl_lio_add_points_to_map(b.lio, xyz.data(), (int)f->point_frame.size(), oo.optimize_options.size_voxel_map,
                                                     oo.max_num_points_in_voxel, oo.min_distance_points, 0);       // lioOptimization.cpp:996-998,1027
            if (rc != SRL_OK) fail(b.lio, rc, "srl_lio_add_points_to_map");
        }
        b.last_frame_handled = f->frame_id;
    }
    int64_t on_device = 0;
    const int rc = srl_lio_map_size(b.lio, &on_device);
    if (rc != SRL_OK) fail(b.lio, rc, "srl_lio_map_size");
    if ((size_t)on_device != self->mapSize(host_map)) {
        std::stringstream ss;
        ss << "device map (" << on_device << " points) and voxel_map (" << self->mapSize(host_map) << " points) differ";
        throw std::runtime_error(ss.str());
    }
}

// ---------------------------------------------------------------------------------------------------- optimize.cpp:428-448
optimizeSummary lioOptimization::optimize(cloudFrame *p_frame, const icpOptions &cur_icp_options, double sample_voxel_size)
{
    HipBinding &b = binding_of(this);
    sync_device_map(this, b, all_cloud_frame, p_frame, odometry_options, voxel_map);

    std::vector<point3D> keypoints;
    gridSampling(p_frame->point_frame, keypoints, sample_voxel_size);                 // utility.cpp:188-201 (the node's own)

    optimizeSummary optimize_summary = updateIEKF(cur_icp_options, voxel_map, keypoints, p_frame);
    b.solved[p_frame->frame_id] = optimize_summary.success;
    if (b.solved.size() > 64) b.solved.erase(b.solved.begin());
    if (!optimize_summary.success) return optimize_summary;

    // transformPoint over the whole frame with the final pose (optimize.cpp:441-445 -> utility.cpp:314-318), on the device
    const int n = (int)p_frame->point_frame.size();
    if (n > 0) {
        const std::vector<double> raw = raw_points_of(p_frame->point_frame);
        std::vector<double> world(3 * (size_t)n);
        const Eigen::Quaterniond &q_end = p_frame->p_state->rotation;
        const double qv[4] = {q_end.w(), q_end.x(), q_end.y(), q_end.z()};
        double t[3], R_il[9], t_il[3];
        for (int i = 0; i < 3; i++) { t[i] = p_frame->p_state->translation[i]; t_il[i] = t_imu_lidar[i]; for (int j = 0; j < 3; j++) R_il[3 * i + j] = R_imu_lidar(i, j); }
        const int rc = srl_transform_points(srl_lio_ctx(b.lio), raw.data(), n, qv, t, R_il, t_il, world.data());
        if (rc != SRL_OK) fail(b.lio, rc, "srl_transform_points");
        for (int k = 0; k < n; k++) p_frame->point_frame[k].point = Eigen::Vector3d(world[3 * (size_t)k], world[3 * (size_t)k + 1], world[3 * (size_t)k + 2]);
    }
    return optimize_summary;
}

// ---------------------------------------------------------------------------------------------------- optimize.cpp:133-314
optimizeSummary lioOptimization::updateIEKF(const icpOptions &cur_icp_options, voxelHashMap &voxel_map_temp, std::vector<point3D> &keypoints, cloudFrame *p_frame)
{
    (void)voxel_map_temp;                                    // the path reads the device map (kept equal to voxel_map, see sync_device_map)
    HipBinding &b = binding_of(this);
    srl_lio *lio = b.lio;

    // members of the node the path reads (lioOptimization.h:216-228)
    double R_il[9], t_il[3];
    for (int i = 0; i < 3; i++) { t_il[i] = t_imu_lidar[i]; for (int j = 0; j < 3; j++) R_il[3 * i + j] = R_imu_lidar(i, j); }
    srl_lio_set_extrinsics(lio, R_il, t_il);
    srl_lio_set_laser_point_cov(lio, laser_point_cov);

    // the filter: eskf_pro's state and covariance in, the updated ones out
    double es[19], P[289];
    {
        const Eigen::Vector3d p = eskf_pro->getTranslation(), v = eskf_pro->getVelocity(), ba = eskf_pro->getBa(), bg = eskf_pro->getBg(), g = eskf_pro->getGravity();
        const Eigen::Quaterniond q = eskf_pro->getRotation();
        for (int a = 0; a < 3; a++) { es[a] = p[a]; es[7 + a] = v[a]; es[10 + a] = ba[a]; es[13 + a] = bg[a]; es[16 + a] = g[a]; }
        es[3] = q.w(); es[4] = q.x(); es[5] = q.y(); es[6] = q.z();
        const Eigen::Matrix<double, 17, 17> cov = eskf_pro->getCovariance();
        for (int i = 0; i < 17; i++) for (int j = 0; j < 17; j++) P[17 * i + j] = cov(i, j);
    }
    srl_lio_eskf_set_state(lio, es);
    srl_lio_eskf_set_cov(lio, P);

    const srl_icp_opts abi = to_abi(cur_icp_options);
    const std::vector<double> raw = raw_points_of(keypoints);
    double st[16], t_last[3];
    pack_state16(p_frame->p_state, st);
    const state *last_state = all_cloud_frame[p_frame->id - 1]->p_state;             // optimize.cpp:25
    for (int a = 0; a < 3; a++) t_last[a] = last_state->translation[a];
    int iters = 0, num_residuals = 0;
    const int rc = srl_lio_update_iekf(lio, &abi, raw.data(), (int)keypoints.size(), st, t_last, p_frame->frame_id, nullptr, 0, &iters, &num_residuals);
    if (rc == SRL_ERR_NAN_PLANARITY) throw std::runtime_error("error");              // optimize.cpp:348-350
    if (rc != SRL_OK && rc != SRL_ERR_NOT_ENOUGH_RESIDUALS) fail(lio, rc, "srl_lio_update_iekf");

    // what the loop leaves behind: the filter (observe(), optimize.cpp:253; setCovariance, :307), the frame's state (:255-259),
    // the gravity globals (:260-261)
    srl_lio_eskf_get_state(lio, es);
    srl_lio_eskf_get_cov(lio, P);
    eskf_pro->setTranslation(Eigen::Vector3d(es[0], es[1], es[2]));
    eskf_pro->setRotation(Eigen::Quaterniond(es[3], es[4], es[5], es[6]));
    eskf_pro->setVelocity(Eigen::Vector3d(es[7], es[8], es[9]));
    eskf_pro->setBa(Eigen::Vector3d(es[10], es[11], es[12]));
    eskf_pro->setBg(Eigen::Vector3d(es[13], es[14], es[15]));
    eskf_pro->setGravity(Eigen::Vector3d(es[16], es[17], es[18]));
    {
        Eigen::Matrix<double, 17, 17> cov;
        for (int i = 0; i < 17; i++) for (int j = 0; j < 17; j++) cov(i, j) = P[17 * i + j];
        eskf_pro->setCovariance(cov);
    }
    unpack_state16(st, p_frame->p_state);
    if (iters > 0) {
        G = eskf_pro->getGravity();
        G_norm = G.norm();
    }

    optimizeSummary summary;
    summary.num_residuals_used = num_residuals;
    if (rc == SRL_ERR_NOT_ENOUGH_RESIDUALS) {                                        // optimize.cpp:110-123
        std::stringstream ss_out;
        ss_out << "[Optimization] Error : not enough keypoints selected in ct-icp !" << std::endl;
        ss_out << "[Optimization] number_of_residuals : " << num_residuals << std::endl;
        summary.success = false;
        summary.error_log = ss_out.str();
        return summary;
    }
    summary.success = true;
    return summary;
}

// ---------------------------------------------------------------------------------------------------- optimize.cpp:18-131
optimizeSummary lioOptimization::buildPlaneResiduals(const icpOptions &cur_icp_options, voxelHashMap &voxel_map_temp, std::vector<point3D> &keypoints,
    std::vector<planeParam> &plane_residuals, cloudFrame *p_frame, double &loss_sum)
{
    (void)voxel_map_temp;
    HipBinding &b = binding_of(this);
    srl_lio *lio = b.lio;
    double R_il[9], t_il[3];
    for (int i = 0; i < 3; i++) { t_il[i] = t_imu_lidar[i]; for (int j = 0; j < 3; j++) R_il[3 * i + j] = R_imu_lidar(i, j); }
    srl_lio_set_extrinsics(lio, R_il, t_il);

    const srl_icp_opts abi = to_abi(cur_icp_options);
    const int n = (int)keypoints.size();
    const std::vector<double> raw = raw_points_of(keypoints);
    double st[16], t_last[3];
    pack_state16(p_frame->p_state, st);
    const state *last_state = all_cloud_frame[p_frame->id - 1]->p_state;
    for (int a = 0; a < 3; a++) t_last[a] = last_state->translation[a];
    std::vector<double> rows((size_t)n * 15), world((size_t)n * 3);
    int num_out = 0, success = 0;
    const int rc = srl_lio_build_plane_residuals(lio, &abi, raw.data(), n, st, t_last, p_frame->frame_id, rows.data(), n, &num_out, &loss_sum, &success,
                                                 world.data());
    if (rc == SRL_ERR_NAN_PLANARITY) throw std::runtime_error("error");
    if (rc != SRL_OK) fail(lio, rc, "srl_lio_build_plane_residuals");
    for (int k = 0; k < n; k++) keypoints[k].point = Eigen::Vector3d(world[3 * (size_t)k], world[3 * (size_t)k + 1], world[3 * (size_t)k + 2]);   // transformKeypoints (optimize.cpp:30-40)
    for (int r = 0; r < num_out; r++) {
        const double *row = rows.data() + (size_t)r * 15;
        planeParam pl;
        pl.raw_point = Eigen::Vector3d(row[0], row[1], row[2]);
        pl.norm_vector = Eigen::Vector3d(row[3], row[4], row[5]);
        for (int c = 0; c < 6; c++) pl.jacobians(0, c) = row[6 + c];
        pl.norm_offset = row[12];
        pl.distance = row[13];
        pl.weight = row[14];
        plane_residuals.push_back(pl);
    }
    optimizeSummary summary;
    summary.num_residuals_used = num_out;
    summary.success = success != 0;
    if (!summary.success) {
        std::stringstream ss_out;
        ss_out << "[Optimization] Error : not enough keypoints selected in ct-icp !" << std::endl;
        ss_out << "[Optimization] number_of_residuals : " << num_out << std::endl;
        summary.error_log = ss_out.str();
    }
    return summary;
}

// ---------------------------------------------------------------------------------------------------- optimize.cpp:316-353
Neighborhood lioOptimization::computeNeighborhoodDistribution(const std::vector<Eigen::Vector3d, Eigen::aligned_allocator<Eigen::Vector3d>> &points)
{
    HipBinding &b = binding_of(this);
    std::vector<double> pts(3 * points.size());
    for (size_t k = 0; k < points.size(); ++k) for (int d = 0; d < 3; d++) pts[3 * k + d] = points[k][d];
    double center[3], normal[3], cov[9], a2D = 0.0;
    const int rc = srl_lio_neighborhood(b.lio, pts.data(), (int)points.size(), center, normal, cov, &a2D);
    if (rc == SRL_ERR_NAN_PLANARITY) throw std::runtime_error("error");              // optimize.cpp:348-350
    if (rc != SRL_OK) fail(b.lio, rc, "srl_lio_neighborhood");
    Neighborhood neighborhood;
    neighborhood.center = Eigen::Vector3d(center[0], center[1], center[2]);
    neighborhood.normal = Eigen::Vector3d(normal[0], normal[1], normal[2]);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) neighborhood.covariance(i, j) = cov[3 * i + j];
    neighborhood.a2D = a2D;
    return neighborhood;
}

// ---------------------------------------------------------------------------------------------------- optimize.cpp:365-426
std::vector<Eigen::Vector3d, Eigen::aligned_allocator<Eigen::Vector3d>> lioOptimization::searchNeighbors(voxelHashMap &map, const Eigen::Vector3d &point,
        int nb_voxels_visited, double size_voxel_map, int max_num_neighbors, int threshold_voxel_capacity, std::vector<voxel> *voxels)
{
    (void)map;
    HipBinding &b = binding_of(this);
    const double p[3] = {point[0], point[1], point[2]};
    std::vector<double> xyz(3 * (size_t)max_num_neighbors);
    std::vector<int16_t> vox(3 * (size_t)max_num_neighbors);
    int found = 0;
    const int rc = srl_lio_search_neighbors(b.lio, p, nb_voxels_visited, size_voxel_map, max_num_neighbors, threshold_voxel_capacity, xyz.data(),
                                            voxels ? vox.data() : nullptr, &found);
    if (rc != SRL_OK) fail(b.lio, rc, "srl_lio_search_neighbors");
    std::vector<Eigen::Vector3d, Eigen::aligned_allocator<Eigen::Vector3d>> closest_neighbors((size_t)found);
    if (voxels) voxels->resize((size_t)found);
    for (int i = 0; i < found; i++) {
        closest_neighbors[i] = Eigen::Vector3d(xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2]);
        if (voxels) (*voxels)[i] = voxel(vox[3 * (size_t)i], vox[3 * (size_t)i + 1], vox[3 * (size_t)i + 2]);
    }
    return closest_neighbors;
}
