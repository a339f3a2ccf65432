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
// The frame stays on the device.  optimize() uploads the frame's raw points ONCE, from a page-locked buffer the binding keeps across
// calls; keypoints are selected on the device (same keypoints, same order as gridSampling, utility.cpp:167-201), the ESIKF runs on
// them in place, and the frame is re-transformed with the solved pose (optimize.cpp:441-445) and inserted into the DEVICE map without
// leaving HBM (srl_lio_optimize_resident / srl_lio_commit_frame); one download returns point3D::point for the node.
//
// The voxel map.  The node keeps inserting sweeps into its own tsl::robin_map (addPointsToMap, lioOptimization.cpp:1027 -- the colour
// map / rendering side needs it); the path reads the DEVICE map.  Both receive the same points in the same order: a frame whose
// optimize() succeeded is committed on the device inside optimize(), from the very world points the node inserts right after it
// (lioOptimization.cpp:1003-1027); the frames before frame_id 2, which stateEstimation inserts without calling optimize, are
// uploaded and inserted before the next solve (srl_map_insert reproduces addPointToMap's order-dependent semantics bit for bit).
// EVERY call compares the two maps where the frame inserted last touched them (a sample of its points: voxel, point count, last stored
// point -- srl_map_probe_checksum -- and the voxel totals); mapSize() of both whole maps is compared during the first calls and every
// 32nd afterwards (the node's mapSize walks every voxel).  On a difference the device copy is rebuilt from voxel_map (the node's map is
// the truth), with one line on stderr -- never a silent drift, and never one that outlives the next frame.
//
// tests/test_gpu_integration.py compiles this file together with the reference's own translation units
// (oracle/Makefile, target refnode_hip) and drives the reference's run() over the 40-sweep replay stream.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
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
    std::map<int, bool> committed;            // frame_id -> already inserted into the device map by optimize() itself
    double *pinned_raw = nullptr;             // page-locked upload buffer for the frame's raw points, kept across calls (srl_pinned_alloc)
    size_t pinned_cap = 0;                    // points
    double *pinned_world = nullptr;           // point3D::point of the committed frame (download target), page-locked, kept across calls
    long calls = 0;
    long resyncs = 0;                         // times the device map had to be rebuilt from voxel_map (sync_device_map)
    int last_commit_frame_id = -1;            // the frame optimize() committed last on the device (the node inserts it right after the call) ...
    int last_commit_n = 0;                    // ... and its number of points
    // (freed by srl_integration_release, not by a destructor: a binding still alive at process exit would call into a HIP runtime that is
    // already shutting down)
    void free_buffers() { if (pinned_raw) srl_pinned_free(pinned_raw); if (pinned_world) srl_pinned_free(pinned_world); pinned_raw = pinned_world = nullptr; pinned_cap = 0; }
};

std::mutex g_mutex;
std::unordered_map<const lioOptimization *, HipBinding> g_bindings;
// the node calls from ONE thread (the ROS main thread, lioOptimization.cpp:1596-1604): the binding of the node seen last is remembered,
// so that a call costs neither the lock nor the map lookup
// -- valid only while no binding has been released since it was remembered: srl_integration_release (any thread) advances g_generation,
// and a thread whose cached generation is stale looks the binding up under the lock again (a released node's address may be reused)
std::atomic<unsigned long> g_generation{1};
thread_local const lioOptimization *t_self = nullptr;
thread_local HipBinding *t_binding = nullptr;
thread_local unsigned long t_generation = 0;

[[noreturn]] void fail(srl_lio *lio, int rc, const char *what) {
    std::string msg = std::string(what) + ": " + srl_status_str(rc);
    if (lio) { msg += " ("; msg += srl_lio_last_error(lio); msg += ")"; }
    throw std::runtime_error(msg);
}

HipBinding &binding_of(const lioOptimization *self) {
    if (t_self == self && t_binding && t_generation == g_generation.load(std::memory_order_acquire)) return *t_binding;
    std::lock_guard<std::mutex> lk(g_mutex);
    HipBinding &b = g_bindings[self];                      // (references into an unordered_map stay valid across insertions)
    if (!b.lio) {
        const int rc = srl_lio_create(0, &b.lio);          // no CPU fallback: without a GPU the node cannot run this path
        if (rc != SRL_OK) fail(nullptr, rc, "srl_lio_create");
    }
    t_self = self; t_binding = &b; t_generation = g_generation.load(std::memory_order_acquire);
    return b;
}

// the node's members the path reads (lioOptimization.h:216-228) and its filter, handed to the mirror before a solve ...
void push_node_state(lioOptimization *self, srl_lio *lio, const Eigen::Matrix3d &R_imu_lidar, const Eigen::Vector3d &t_imu_lidar, double laser_point_cov,
                     eskfEstimator *eskf_pro) {
    (void)self;
    double R_il[9], t_il[3];
    for (int i = 0; i < 3; i++) { t_il[i] = t_imu_lidar[i]; for (int j = 0; j < 3; j++) R_il[3 * i + j] = R_imu_lidar(i, j); }
    srl_lio_set_extrinsics(lio, R_il, t_il);
    srl_lio_set_laser_point_cov(lio, laser_point_cov);
    double es[19], P[289];
    const Eigen::Vector3d p = eskf_pro->getTranslation(), v = eskf_pro->getVelocity(), ba = eskf_pro->getBa(), bg = eskf_pro->getBg(), g = eskf_pro->getGravity();
    const Eigen::Quaterniond q = eskf_pro->getRotation();
    for (int a = 0; a < 3; a++) { es[a] = p[a]; es[7 + a] = v[a]; es[10 + a] = ba[a]; es[13 + a] = bg[a]; es[16 + a] = g[a]; }
    es[3] = q.w(); es[4] = q.x(); es[5] = q.y(); es[6] = q.z();
    const Eigen::Matrix<double, 17, 17> cov = eskf_pro->getCovariance();
    for (int i = 0; i < 17; i++) for (int j = 0; j < 17; j++) P[17 * i + j] = cov(i, j);
    srl_lio_eskf_set_state(lio, es);
    srl_lio_eskf_set_cov(lio, P);
}
// ... and what the loop leaves behind read back: the filter (observe(), optimize.cpp:253; setCovariance, :307) -- ALWAYS, also when
// the solve ended in the NaN throw of optimize.cpp:348-350 (the reference keeps the observe() calls of the passes before it)
void pull_filter(srl_lio *lio, eskfEstimator *eskf_pro) {
    double es[19], P[289];
    srl_lio_eskf_get_state(lio, es);
    srl_lio_eskf_get_cov(lio, P);
    eskf_pro->setTranslation(Eigen::Vector3d(es[0], es[1], es[2]));
    eskf_pro->setRotation(Eigen::Quaterniond(es[3], es[4], es[5], es[6]));
    eskf_pro->setVelocity(Eigen::Vector3d(es[7], es[8], es[9]));
    eskf_pro->setBa(Eigen::Vector3d(es[10], es[11], es[12]));
    eskf_pro->setBg(Eigen::Vector3d(es[13], es[14], es[15]));
    eskf_pro->setGravity(Eigen::Vector3d(es[16], es[17], es[18]));
    Eigen::Matrix<double, 17, 17> cov;
    for (int i = 0; i < 17; i++) for (int j = 0; j < 17; j++) cov(i, j) = P[17 * i + j];
    eskf_pro->setCovariance(cov);
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
extern "C" long srl_integration_resyncs(const void *node) {
    std::lock_guard<std::mutex> lk(g_mutex);
    auto it = g_bindings.find(static_cast<const lioOptimization *>(node));
    return it == g_bindings.end() ? -1 : it->second.resyncs;
}
extern "C" void srl_integration_release(const void *node) {
    std::lock_guard<std::mutex> lk(g_mutex);
    auto it = g_bindings.find(static_cast<const lioOptimization *>(node));
    if (it == g_bindings.end()) return;
    it->second.free_buffers();
    if (it->second.lio) srl_lio_destroy(it->second.lio);
    if (t_binding == &it->second) { t_self = nullptr; t_binding = nullptr; }
    g_generation.fetch_add(1, std::memory_order_acq_rel);          // every thread's remembered binding is void from here on
    g_bindings.erase(it);
}

// ---- the device map follows the node's map (see the header of this file) ----
static void sync_device_map(lioOptimization *self, HipBinding &b, const std::vector<cloudFrame *> &window, const cloudFrame *current,
                            const odometryOptions &oo, voxelHashMap &host_map) {
    for (const cloudFrame *f : window) {
        if (f == current || f->frame_id <= b.last_frame_handled) continue;
        const bool inserted_by_node = f->frame_id <= 1 || (b.solved.count(f->frame_id) && b.solved[f->frame_id]);
        const bool already_on_device = b.committed.count(f->frame_id) != 0;       // optimize() committed it itself
        if (inserted_by_node && !already_on_device && !f->point_frame.empty()) {
            std::vector<double> xyz(3 * f->point_frame.size());
            for (size_t k = 0; k < f->point_frame.size(); ++k) for (int d = 0; d < 3; d++) xyz[3 * k + d] = f->point_frame[k].point[d];
            const int rc = srl_lio_add_points_to_map(b.lio, xyz.data(), (int)f->point_frame.size(), oo.optimize_options.size_voxel_map,
                                                     oo.max_num_points_in_voxel, oo.min_distance_points, 0);       // lioOptimization.cpp:996-998,1027
            if (rc != SRL_OK) fail(b.lio, rc, "srl_lio_add_points_to_map");
        }
        b.last_frame_handled = f->frame_id;
    }
    // EVERY call: the frame committed by the previous optimize() has since been inserted by the node too (lioOptimization.cpp:1027).  Both
    // maps are probed with a sample of that frame's points -- every 16th: the voxel it falls into, how many points the voxel holds and
    // where its last stored point lies (srl_map_probe_checksum; the device side is one small kernel over the world points still in HBM,
    // the node's side ~n/16 finds) -- and the voxel counts are compared (tsl::robin_map::size() is O(1)).  A divergence is caught on the
    // frame after it happened; the walk over both whole maps (mapSize) stays as the slow, exhaustive check every 32nd call.
    bool differ = false;
    if (b.last_commit_frame_id >= 0) {
        const cloudFrame *prev = nullptr;
        for (const cloudFrame *f : window) if (f != current && f->frame_id == b.last_commit_frame_id) prev = f;
        if (prev && (int)prev->point_frame.size() == b.last_commit_n) {
            constexpr int STRIDE = 16;
            uint64_t on_dev = 0, on_host = 0;
            int32_t dev_voxels = 0;
            const int rcp = srl_lio_probe_checksum_of_committed_frame(b.lio, STRIDE, oo.optimize_options.size_voxel_map, &on_dev, &dev_voxels);
            if (rcp != SRL_OK) fail(b.lio, rcp, "srl_lio_probe_checksum_of_committed_frame");
            const double vs = oo.optimize_options.size_voxel_map;
            for (size_t k = 0; k < prev->point_frame.size(); k += STRIDE) {
                const float fx = (float)prev->point_frame[k].point[0], fy = (float)prev->point_frame[k].point[1], fz = (float)prev->point_frame[k].point[2];
                const short kx = static_cast<short>((double)fx / vs), ky = static_cast<short>((double)fy / vs), kz = static_cast<short>((double)fz / vs);
                auto it = host_map.find(voxel(kx, ky, kz));
                if (it == host_map.end()) continue;
                voxelBlock &blk = it.value();                 // (getPosition() is not const-qualified in the reference's rgbPoint)
                const int c = blk.NumPoints();
                if (c <= 0) continue;
                const Eigen::Vector3d last = blk.points[c - 1].getPosition();
                on_host += srl_probe_mix(kx, ky, kz, c, (float)last[0], (float)last[1], (float)last[2]);
            }
            differ = on_dev != on_host || (size_t)dev_voxels != host_map.size();
        }
        b.last_commit_frame_id = -1;
    }
    int64_t on_device = 0;
    if (!differ) {
        if (!(b.calls < 16 || b.calls % 32 == 0)) return;       // the node's mapSize() walks every voxel of its map
        const int rc = srl_lio_map_size(b.lio, &on_device);
        if (rc != SRL_OK) fail(b.lio, rc, "srl_lio_map_size");
        differ = (size_t)on_device != self->mapSize(host_map);
    } else {
        (void)srl_lio_map_size(b.lio, &on_device);
    }
    if (differ) {
        // The device copy has fallen out of step with voxel_map (something other than addPointsToMap touched one of them): the node's map is
        // the truth -- rebuild the device copy from it, voxel by voxel in the stored point order, and say so once.
        static bool said = false;
        if (!said) {
            std::fprintf(stderr, "[srlivo_hip] device map (%lld points) and voxel_map (%zu points) differ: device map rebuilt from voxel_map\n",
                         (long long)on_device, (size_t)self->mapSize(host_map));
            said = true;
        }
        const int cap = oo.max_num_points_in_voxel;
        const size_t V = host_map.size();
        std::vector<int16_t> keys(3 * V);
        std::vector<int32_t> counts(V);
        std::vector<float> xyz((size_t)3 * cap * V, 0.0f);
        size_t v = 0;
        for (auto it = host_map.begin(); it != host_map.end(); ++it, ++v) {
            keys[3 * v] = it.key().x; keys[3 * v + 1] = it.key().y; keys[3 * v + 2] = it.key().z;
            voxelBlock &blk = it.value();
            const int c = std::min(blk.NumPoints(), cap);
            counts[v] = c;
            for (int i = 0; i < c; i++) {
                const Eigen::Vector3d pnt = blk.points[i].getPosition();
                for (int d = 0; d < 3; d++) xyz[((size_t)v * cap + i) * 3 + d] = (float)pnt[d];      // stored as float (cloudMap.cpp:7,28): exact
            }
        }
        const int rcu = srl_map_upload(srl_lio_ctx(b.lio), keys.data(), counts.data(), xyz.data(), (int)V, cap);
        if (rcu != SRL_OK) fail(b.lio, rcu, "srl_map_upload (resynchronisation)");
        b.resyncs++;
    }
}

// ---------------------------------------------------------------------------------------------------- optimize.cpp:428-448
optimizeSummary lioOptimization::optimize(cloudFrame *p_frame, const icpOptions &cur_icp_options, double sample_voxel_size)
{
    HipBinding &b = binding_of(this);
    srl_lio *lio = b.lio;
    sync_device_map(this, b, all_cloud_frame, p_frame, odometry_options, voxel_map);
    b.calls++;

    // the frame's raw points into the page-locked buffer (the upload is one DMA out of it)
    const int n = (int)p_frame->point_frame.size();
    if ((size_t)n > b.pinned_cap) {
        if (b.pinned_raw) srl_pinned_free(b.pinned_raw);
        if (b.pinned_world) srl_pinned_free(b.pinned_world);
        b.pinned_raw = b.pinned_world = nullptr;
        b.pinned_cap = (size_t)n + (size_t)n / 2 + 1024;
        void *p = nullptr, *w = nullptr;
        const int rca = srl_pinned_alloc(b.pinned_cap * 3 * sizeof(double), &p);
        const int rcw = rca == SRL_OK ? srl_pinned_alloc(b.pinned_cap * 3 * sizeof(double), &w) : rca;
        if (rca != SRL_OK || rcw != SRL_OK) { if (p) srl_pinned_free(p); b.pinned_cap = 0; fail(lio, rca != SRL_OK ? rca : rcw, "srl_pinned_alloc"); }
        b.pinned_raw = static_cast<double *>(p);
        b.pinned_world = static_cast<double *>(w);
    }
    for (int k = 0; k < n; k++) for (int d = 0; d < 3; d++) b.pinned_raw[3 * (size_t)k + d] = p_frame->point_frame[k].raw_point[d];

    push_node_state(this, lio, R_imu_lidar, t_imu_lidar, laser_point_cov, eskf_pro);
    const srl_icp_opts abi = to_abi(cur_icp_options);
    double st[16], t_last[3];
    pack_state16(p_frame->p_state, st);
    const state *last_state = all_cloud_frame[p_frame->id - 1]->p_state;             // optimize.cpp:25
    for (int a = 0; a < 3; a++) t_last[a] = last_state->translation[a];

    // gridSampling (utility.cpp:188-201: same keypoints, same order, selected on the device from point = R(q)(R_il raw + t_il) + t at the
    // prior pose -- what point3D::point holds here) + updateIEKF on them in place
    int iters = 0, num_residuals = 0, observed = 0;
    const int rc = srl_lio_optimize_resident(lio, &abi, sample_voxel_size, b.pinned_raw, n, st, t_last, p_frame->frame_id, nullptr, nullptr, &iters,
                                             &num_residuals);
    pull_filter(lio, eskf_pro);
    srl_lio_last_solve_observed(lio, &observed);
    if (observed > 0) {                                                              // optimize.cpp:255-261 ran at least once
        unpack_state16(st, p_frame->p_state);
        G = eskf_pro->getGravity();
        G_norm = G.norm();
    }
    if (rc == SRL_ERR_NAN_PLANARITY) throw std::runtime_error("error");              // optimize.cpp:348-350
    if (rc != SRL_OK && rc != SRL_ERR_NOT_ENOUGH_RESIDUALS) fail(lio, rc, "srl_lio_optimize_resident");

    optimizeSummary optimize_summary;
    optimize_summary.num_residuals_used = num_residuals;
    optimize_summary.success = rc == SRL_OK;
    b.solved[p_frame->frame_id] = optimize_summary.success;
    if (b.solved.size() > 64) b.solved.erase(b.solved.begin());
    if (!optimize_summary.success) {                                                 // optimize.cpp:110-123
        std::stringstream ss_out;
        ss_out << "[Optimization] Error : not enough keypoints selected in ct-icp !" << std::endl;
        ss_out << "[Optimization] number_of_residuals : " << num_residuals << std::endl;
        optimize_summary.error_log = ss_out.str();
        return optimize_summary;
    }

    // transformPoint over the whole frame with the final pose (optimize.cpp:441-445 -> utility.cpp:314-318) and the insertion the node
    // performs right after this call (addPointsToMap, lioOptimization.cpp:1027), both on the frame resident in HBM; the world points
    // come back once, for point3D::point
    // (num_added = NULL: addPointsToMap returns nothing; the insertion runs on behind this call, the next sweep's passes are ordered behind it)
    const int rcc = srl_lio_commit_frame(lio, st, odometry_options.optimize_options.size_voxel_map, odometry_options.max_num_points_in_voxel,
                                         odometry_options.min_distance_points, 0, n > 0 ? b.pinned_world : nullptr, nullptr);
    if (rcc != SRL_OK) fail(lio, rcc, "srl_lio_commit_frame");
    b.committed[p_frame->frame_id] = true;
    b.last_commit_frame_id = p_frame->frame_id;
    b.last_commit_n = n;
    if (b.committed.size() > 64) b.committed.erase(b.committed.begin());
    for (int k = 0; k < n; k++) p_frame->point_frame[k].point = Eigen::Vector3d(b.pinned_world[3 * (size_t)k], b.pinned_world[3 * (size_t)k + 1], b.pinned_world[3 * (size_t)k + 2]);
    return optimize_summary;
}

// ---------------------------------------------------------------------------------------------------- optimize.cpp:133-314
optimizeSummary lioOptimization::updateIEKF(const icpOptions &cur_icp_options, voxelHashMap &voxel_map_temp, std::vector<point3D> &keypoints, cloudFrame *p_frame)
{
    (void)voxel_map_temp;                                    // the path reads the device map (kept equal to voxel_map, see sync_device_map)
    HipBinding &b = binding_of(this);
    srl_lio *lio = b.lio;

    push_node_state(this, lio, R_imu_lidar, t_imu_lidar, laser_point_cov, eskf_pro);

    const srl_icp_opts abi = to_abi(cur_icp_options);
    const std::vector<double> raw = raw_points_of(keypoints);
    double st[16], t_last[3];
    pack_state16(p_frame->p_state, st);
    const state *last_state = all_cloud_frame[p_frame->id - 1]->p_state;             // optimize.cpp:25
    for (int a = 0; a < 3; a++) t_last[a] = last_state->translation[a];
    int iters = 0, num_residuals = 0;
    const int rc = srl_lio_update_iekf(lio, &abi, raw.data(), (int)keypoints.size(), st, t_last, p_frame->frame_id, nullptr, 0, &iters, &num_residuals);
    // what the loop leaves behind: the filter -- also behind the NaN throw --, the frame's state (:255-259) and the gravity globals
    // (:260-261), the last two only if observe() ran at all (a solve whose every step hit the guard of :248-251 writes neither)
    pull_filter(lio, eskf_pro);
    int observed = 0;
    srl_lio_last_solve_observed(lio, &observed);
    if (observed > 0) {
        unpack_state16(st, p_frame->p_state);
        G = eskf_pro->getGravity();
        G_norm = G.norm();
    }
    if (rc == SRL_ERR_NAN_PLANARITY) throw std::runtime_error("error");              // optimize.cpp:348-350
    if (rc != SRL_OK && rc != SRL_ERR_NOT_ENOUGH_RESIDUALS) fail(lio, rc, "srl_lio_update_iekf");

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
