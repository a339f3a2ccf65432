// srl_host_capi.cpp -- C handles (include/srlivo_host.h) onto the C++ host mirror, so that ctypes
// harnesses drive the same lioOptimization / eskfEstimator code a C++ consumer links.
#include "../../../include/srlivo_host.h"
#include "lioOptimization.h"
#include "tr1_order.h"
#include "tr1_relation.h"

#include <cstring>
#include <new>
#include <stdexcept>
#include <string>

using namespace srlivo;

struct srl_lio {
    lioOptimization *lio = nullptr;
    std::string err;
    state prev_state, cur_state;
};

namespace {

int status_from_exception(srl_lio *h, const std::exception &e) {
    h->err = e.what();
    if (h->err == "error") return SRL_ERR_NAN_PLANARITY;   // optimize.cpp:348-350
    return SRL_ERR_HIP;
}

void state_from(const double s[16], state &st) {
    st.rotation = srl::Quat(s[0], s[1], s[2], s[3]);
    st.translation = srl::vec3(s[4], s[5], s[6]);
    st.velocity = srl::vec3(s[7], s[8], s[9]);
    st.ba = srl::vec3(s[10], s[11], s[12]);
    st.bg = srl::vec3(s[13], s[14], s[15]);
}
void state_to(const state &st, double s[16]) {
    s[0] = st.rotation.w; s[1] = st.rotation.x; s[2] = st.rotation.y; s[3] = st.rotation.z;
    for (int i = 0; i < 3; i++) { s[4 + i] = st.translation[i]; s[7 + i] = st.velocity[i]; s[10 + i] = st.ba[i]; s[13 + i] = st.bg[i]; }
}

// builds the two-frame window updateIEKF needs: all_cloud_frame[id-1] (previous) and the current frame
struct FrameWindow {
    std::vector<point3D> no_points;
    cloudFrame prev, cur;
    srl_lio *owner;
    std::vector<cloudFrame *> saved;          // the replay driver's own window, put back on exit
    ~FrameWindow() { owner->lio->all_cloud_frame.swap(saved); }
    FrameWindow(srl_lio *h, const double state_io[16], const double t_last[3], int frame_id, std::vector<point3D> &pts)
        : prev(no_points, &h->prev_state), cur(pts, &h->cur_state), owner(h) {
        saved.swap(h->lio->all_cloud_frame);
        h->prev_state = state();
        h->prev_state.translation = srl::vec3(t_last[0], t_last[1], t_last[2]);
        state_from(state_io, h->cur_state);
        prev.id = 0; prev.frame_id = frame_id - 1;
        cur.id = 1; cur.frame_id = frame_id;
        h->lio->all_cloud_frame.clear();
        h->lio->all_cloud_frame.push_back(&prev);
        h->lio->all_cloud_frame.push_back(&cur);
    }
};

void write_log(const lioOptimization &L, double *log, int max_log_iters) {
    if (!log) return;
    const int m = std::min<int>((int)L.iteration_log.size(), max_log_iters);
    for (int i = 0; i < m; i++) {
        double *row = log + (size_t)i * 61;
        std::memcpy(row, L.iteration_log[i].neq.HtH, 36 * sizeof(double));
        std::memcpy(row + 36, L.iteration_log[i].neq.Hth, 6 * sizeof(double));
        for (int a = 0; a < 17; a++) row[42 + a] = L.iteration_log[i].d_x[a];
        row[59] = (double)L.iteration_log[i].neq.num_residuals;
        row[60] = L.iteration_log[i].neq.loss_sum;
    }
}

int run_update(srl_lio *h, const srl_icp_opts *opts, std::vector<point3D> &keypoints, double state_io[16],
               const double t_last[3], int frame_id, double *log, int max_log_iters, int *iters, int *num_res,
               bool resident = false) {
    FrameWindow w(h, state_io, t_last, frame_id, keypoints);
    h->lio->record_iterations = (log != nullptr);
    const icpOptions o = icpOptions::fromAbi(*opts);
    optimizeSummary s;
    try {
        s = resident ? h->lio->solveIEKF(o, &w.cur) : h->lio->updateIEKF(o, h->lio->voxel_map, keypoints, &w.cur);
    } catch (const std::exception &e) {
        // the NaN throw of optimize.cpp:348-350 leaves p_frame->p_state as the passes before it wrote it (:255-259): hand that back too
        h->lio->all_cloud_frame.clear();
        state_to(h->cur_state, state_io);
        if (iters) *iters = h->lio->last_num_iterations;
        return status_from_exception(h, e);
    }
    h->lio->all_cloud_frame.clear();
    state_to(h->cur_state, state_io);
    write_log(*h->lio, log, max_log_iters);
    if (iters) *iters = h->lio->last_num_iterations;
    if (num_res) *num_res = s.num_residuals_used;
    if (!s.success) { h->err = s.error_log; return SRL_ERR_NOT_ENOUGH_RESIDUALS; }
    return SRL_OK;
}

}  // namespace

extern "C" {

int srl_lio_create(int device, srl_lio **out) {
    if (!out) return SRL_ERR_BAD_ARG;
    *out = nullptr;
    srl_lio *h = new (std::nothrow) srl_lio();
    if (!h) return SRL_ERR_HIP;
    try {
        h->lio = new lioOptimization(device);
    } catch (const std::exception &) {
        delete h;
        return SRL_ERR_NO_DEVICE;
    }
    *out = h;
    return SRL_OK;
}

int srl_lio_destroy(srl_lio *h) {
    if (!h) return SRL_OK;
    delete h->lio;
    delete h;
    return SRL_OK;
}

srl_ctx *srl_lio_ctx(srl_lio *h) { return h ? h->lio->context() : nullptr; }
const char *srl_lio_last_error(srl_lio *h) { return h ? h->err.c_str() : "null handle"; }

int srl_lio_set_extrinsics(srl_lio *h, const double R_il[9], const double t_il[3]) {
    if (!h || !R_il || !t_il) return SRL_ERR_BAD_ARG;
    for (int i = 0; i < 9; i++) h->lio->R_imu_lidar.a[i] = R_il[i];
    for (int i = 0; i < 3; i++) h->lio->t_imu_lidar.a[i] = t_il[i];
    return SRL_OK;
}
int srl_lio_set_laser_point_cov(srl_lio *h, double cov) { if (!h) return SRL_ERR_BAD_ARG; h->lio->laser_point_cov = cov; return SRL_OK; }
int srl_lio_last_solve_observed(srl_lio *h, int *observed) {
    if (!h || !observed) return SRL_ERR_BAD_ARG;
    *observed = h->lio->last_num_observed;
    return SRL_OK;
}
int srl_lio_last_solve_launches(srl_lio *h, int *launches) { if (!h || !launches) return SRL_ERR_BAD_ARG; *launches = h->lio->last_solve_launches; return SRL_OK; }

int srl_lio_eskf_get_state(srl_lio *h, double s[19]) {
    if (!h || !s) return SRL_ERR_BAD_ARG;
    eskfEstimator *e = h->lio->eskf_pro;
    const srl::Vec3 p = e->getTranslation(), v = e->getVelocity(), ba = e->getBa(), bg = e->getBg(), g = e->getGravity();
    const srl::Quat q = e->getRotation();
    for (int i = 0; i < 3; i++) { s[i] = p[i]; s[7 + i] = v[i]; s[10 + i] = ba[i]; s[13 + i] = bg[i]; s[16 + i] = g[i]; }
    s[3] = q.w; s[4] = q.x; s[5] = q.y; s[6] = q.z;
    return SRL_OK;
}
int srl_lio_eskf_set_state(srl_lio *h, const double s[19]) {
    if (!h || !s) return SRL_ERR_BAD_ARG;
    eskfEstimator *e = h->lio->eskf_pro;
    e->setTranslation(srl::vec3(s[0], s[1], s[2]));
    e->setRotation(srl::Quat(s[3], s[4], s[5], s[6]));
    e->setVelocity(srl::vec3(s[7], s[8], s[9]));
    e->setBa(srl::vec3(s[10], s[11], s[12]));
    e->setBg(srl::vec3(s[13], s[14], s[15]));
    e->setGravity(srl::vec3(s[16], s[17], s[18]));
    return SRL_OK;
}
int srl_lio_eskf_get_cov(srl_lio *h, double P[289]) {
    if (!h || !P) return SRL_ERR_BAD_ARG;
    const srl::Mat17 c = h->lio->eskf_pro->getCovariance();
    std::memcpy(P, c.a, sizeof c.a);
    return SRL_OK;
}
int srl_lio_eskf_set_cov(srl_lio *h, const double P[289]) {
    if (!h || !P) return SRL_ERR_BAD_ARG;
    srl::Mat17 c;
    std::memcpy(c.a, P, sizeof c.a);
    h->lio->eskf_pro->setCovariance(c);
    return SRL_OK;
}
int srl_lio_eskf_set_noise(srl_lio *h, double acc_cov, double gyr_cov, double b_acc_cov, double b_gyr_cov) {
    if (!h) return SRL_ERR_BAD_ARG;
    eskfEstimator *e = h->lio->eskf_pro;
    e->setAccCov(acc_cov); e->setGyrCov(gyr_cov); e->setBiasAccCov(b_acc_cov); e->setBiasGyrCov(b_gyr_cov);
    e->useScaleCovAsCov();
    e->initializeNoise();
    return SRL_OK;
}
int srl_lio_eskf_init_imu(srl_lio *h, const double acc0[3], const double gyr0[3]) {
    if (!h || !acc0 || !gyr0) return SRL_ERR_BAD_ARG;
    h->lio->eskf_pro->initializeImuData(srl::vec3(acc0[0], acc0[1], acc0[2]), srl::vec3(gyr0[0], gyr0[1], gyr0[2]));
    return SRL_OK;
}
int srl_lio_eskf_scale_init_cov(srl_lio *h) { if (!h) return SRL_ERR_BAD_ARG; h->lio->eskf_pro->scaleInitialCovariance(); return SRL_OK; }
int srl_lio_eskf_predict(srl_lio *h, double dt, const double acc1[3], const double gyr1[3]) {
    if (!h || !acc1 || !gyr1) return SRL_ERR_BAD_ARG;
    h->lio->eskf_pro->predict(dt, srl::vec3(acc1[0], acc1[1], acc1[2]), srl::vec3(gyr1[0], gyr1[1], gyr1[2]));
    return SRL_OK;
}
int srl_lio_eskf_observe(srl_lio *h, const double dx[17]) {
    if (!h || !dx) return SRL_ERR_BAD_ARG;
    srl::Vec17 d;
    std::memcpy(d.a, dx, sizeof d.a);
    h->lio->eskf_pro->observe(d);
    return SRL_OK;
}

int srl_lio_add_points_to_map(srl_lio *h, const double *world_xyz, int n, double voxel_size, int max_num_points_in_voxel,
                              double min_distance_points, int min_num_points) {
    if (!h || n < 0 || (n > 0 && !world_xyz)) return SRL_ERR_BAD_ARG;
    std::vector<point3D> pts((size_t)n);
    for (int k = 0; k < n; k++) pts[k].point = srl::vec3(world_xyz[(size_t)k * 3], world_xyz[(size_t)k * 3 + 1], world_xyz[(size_t)k * 3 + 2]);
    state st;
    cloudFrame frame(pts, &st);
    try {
        h->lio->addPointsToMap(h->lio->voxel_map, &frame, voxel_size, max_num_points_in_voxel, min_distance_points, min_num_points, false);
    } catch (const std::exception &e) { return status_from_exception(h, e); }
    return SRL_OK;
}
int srl_lio_map_size(srl_lio *h, int64_t *num_points) {
    if (!h || !num_points) return SRL_ERR_BAD_ARG;
    try { *num_points = (int64_t)h->lio->mapSize(h->lio->voxel_map); }
    catch (const std::exception &e) { return status_from_exception(h, e); }
    return SRL_OK;
}

int srl_lio_probe_checksum_of_committed_frame(srl_lio *h, int stride, double voxel_size, uint64_t *checksum, int32_t *num_voxels) {
    if (!h || !checksum) return SRL_ERR_BAD_ARG;
    srl_ctx *ctx = h->lio->context();
    if (!ctx) return SRL_ERR_NO_DEVICE;
    int rc = srl_map_probe_checksum(ctx, nullptr, 0, stride, voxel_size, checksum);
    if (rc == SRL_OK && num_voxels) rc = srl_map_size(ctx, nullptr, num_voxels);
    return rc;
}

int srl_lio_resident_sweep(srl_lio *h, const double *raw_xyz, int n) {
    if (!h || n < 0 || (n > 0 && !raw_xyz)) return SRL_ERR_BAD_ARG;
    return h->lio->residentSweep(raw_xyz, n);
}

int srl_lio_prefetch_sweep(srl_lio *h, const double *raw_xyz, int n) {
    if (!h || n < 0 || (n > 0 && !raw_xyz)) return SRL_ERR_BAD_ARG;
    return h->lio->prefetchSweep(raw_xyz, n);
}
int srl_lio_prefetch_sweep_during_solve(srl_lio *h, const double *raw_xyz, int n) {
    if (!h || n < 0 || (n > 0 && !raw_xyz)) return SRL_ERR_BAD_ARG;
    if (!h->lio->context()) return SRL_ERR_NO_DEVICE;
    h->lio->prefetchSweepDuringSolve(raw_xyz, n);
    return SRL_OK;
}
int srl_lio_swap_sweep(srl_lio *h) {
    if (!h) return SRL_ERR_BAD_ARG;
    return h->lio->swapSweep();
}

int srl_lio_update_iekf(srl_lio *h, const srl_icp_opts *opts, const double *raw_xyz, int n, double state_io[16],
                        const double t_last[3], int frame_id, double *log, int max_log_iters, int *iters,
                        int *num_residuals_used) {
    if (!h || !opts || !state_io || !t_last || n < 0) return SRL_ERR_BAD_ARG;
    if (!h->lio->context()) { h->err = "host-only handle: use srl_lio_update_iekf_provided"; return SRL_ERR_NO_DEVICE; }
    h->lio->setNormalEqProvider(nullptr, nullptr);
    std::vector<point3D> keypoints;
    if (raw_xyz) {
        h->lio->releaseSweep();
        keypoints.resize((size_t)n);
        for (int k = 0; k < n; k++) keypoints[k].raw_point = srl::vec3(raw_xyz[(size_t)k * 3], raw_xyz[(size_t)k * 3 + 1], raw_xyz[(size_t)k * 3 + 2]);
        return run_update(h, opts, keypoints, state_io, t_last, frame_id, log, max_log_iters, iters, num_residuals_used);
    }
    if (!h->lio->sweepPinned(n)) { h->err = "no resident sweep of this size"; return SRL_ERR_NO_SWEEP; }
    return run_update(h, opts, keypoints, state_io, t_last, frame_id, log, max_log_iters, iters, num_residuals_used, true);
}

int srl_lio_stream_step(srl_lio *h, const srl_icp_opts *opts, const double eskf_state[19], const double eskf_cov[289], int n, double state_io[16],
                        const double t_last[3], int frame_id, const double *next_raw_xyz, int next_n, int *iters, int *num_residuals_used) {
    if (!h || !opts || !state_io || !t_last || n < 0 || next_n < 0) return SRL_ERR_BAD_ARG;
    int rc = SRL_OK;
    if (eskf_state && (rc = srl_lio_eskf_set_state(h, eskf_state)) != SRL_OK) return rc;
    if (eskf_cov && (rc = srl_lio_eskf_set_cov(h, eskf_cov)) != SRL_OK) return rc;
    if (next_raw_xyz && (rc = srl_lio_prefetch_sweep_during_solve(h, next_raw_xyz, next_n)) != SRL_OK) return rc;
    if ((rc = srl_lio_update_iekf(h, opts, nullptr, n, state_io, t_last, frame_id, nullptr, 0, iters, num_residuals_used)) != SRL_OK) return rc;
    return next_raw_xyz ? srl_lio_swap_sweep(h) : SRL_OK;
}

int srl_lio_update_iekf_provided(srl_lio *h, const srl_icp_opts *opts, srl_normal_eq_provider provider, void *user, int n,
                                 double state_io[16], const double t_last[3], int frame_id, double *log, int max_log_iters,
                                 int *iters, int *num_residuals_used) {
    if (!h || !opts || !provider || !state_io || !t_last || n < 0) return SRL_ERR_BAD_ARG;
    (void)n;
    h->lio->setNormalEqProvider(provider, user);
    std::vector<point3D> keypoints;
    int rc = run_update(h, opts, keypoints, state_io, t_last, frame_id, log, max_log_iters, iters, num_residuals_used, true);
    h->lio->setNormalEqProvider(nullptr, nullptr);
    return rc;
}

int srl_lio_optimize(srl_lio *h, const srl_icp_opts *opts, double sample_voxel_size, const double *frame_raw,
                     double *frame_world, int n, double state_io[16], const double t_last[3], int frame_id,
                     int32_t *keypoint_index, int *num_keypoints, int *iters, int *num_residuals_used) {
    if (!h || !opts || !frame_raw || !frame_world || !state_io || !t_last || n < 0) return SRL_ERR_BAD_ARG;
    if (!h->lio->context()) return SRL_ERR_NO_DEVICE;
    h->lio->setNormalEqProvider(nullptr, nullptr);
    std::vector<point3D> pts((size_t)n);
    for (int k = 0; k < n; k++) {
        pts[k].raw_point = srl::vec3(frame_raw[(size_t)k * 3], frame_raw[(size_t)k * 3 + 1], frame_raw[(size_t)k * 3 + 2]);
        pts[k].point = srl::vec3(frame_world[(size_t)k * 3], frame_world[(size_t)k * 3 + 1], frame_world[(size_t)k * 3 + 2]);
        pts[k].index_frame = k;
    }
    if (keypoint_index || num_keypoints) {
        std::vector<point3D> kp;
        gridSampling(pts, kp, sample_voxel_size);
        if (num_keypoints) *num_keypoints = (int)kp.size();
        if (keypoint_index) for (size_t i = 0; i < kp.size(); i++) keypoint_index[i] = kp[i].index_frame;
    }
    FrameWindow w(h, state_io, t_last, frame_id, pts);
    const icpOptions o = icpOptions::fromAbi(*opts);
    optimizeSummary s;
    try {
        s = h->lio->optimize(&w.cur, o, sample_voxel_size);
    } catch (const std::exception &e) {
        h->lio->all_cloud_frame.clear();
        return status_from_exception(h, e);
    }
    h->lio->all_cloud_frame.clear();
    state_to(h->cur_state, state_io);
    if (iters) *iters = h->lio->last_num_iterations;
    if (num_residuals_used) *num_residuals_used = s.num_residuals_used;
    if (!s.success) { h->err = s.error_log; return SRL_ERR_NOT_ENOUGH_RESIDUALS; }
    for (int k = 0; k < n; k++) for (int d = 0; d < 3; d++) frame_world[(size_t)k * 3 + d] = w.cur.point_frame[k].point[d];
    return SRL_OK;
}

int srl_lio_eskf_try_init(srl_lio *h, const double *t, const double *gyr, const double *acc, int n, int *initialized) {
    if (!h || n < 0 || (n > 0 && (!t || !gyr || !acc))) return SRL_ERR_BAD_ARG;
    std::vector<imuMeas> meas((size_t)n);
    for (int i = 0; i < n; i++) {
        meas[i].first = t[i];
        meas[i].second.first = srl::vec3(gyr[3 * i], gyr[3 * i + 1], gyr[3 * i + 2]);
        meas[i].second.second = srl::vec3(acc[3 * i], acc[3 * i + 1], acc[3 * i + 2]);
    }
    const int r = h->lio->eskf_pro->tryInit(meas);
    if (initialized) *initialized = r;
    return SRL_OK;
}
int srl_lio_eskf_get_init_stats(srl_lio *h, double out[14]) {
    if (!h || !out) return SRL_ERR_BAD_ARG;
    const eskfEstimator *e = h->lio->eskf_pro;
    const srl::Vec3 v[4] = {e->getMeanGyr(), e->getMeanAcc(), e->getGyrCov(), e->getAccCov()};
    for (int k = 0; k < 4; k++) for (int d = 0; d < 3; d++) out[3 * k + d] = v[k][d];
    out[12] = (double)e->getNumInitMeas();
    out[13] = initial_flag ? 1.0 : 0.0;
    return SRL_OK;
}
int srl_lio_set_initial_flag(srl_lio *h, int flag) {
    if (!h) return SRL_ERR_BAD_ARG;
    initial_flag = flag != 0;
    return SRL_OK;
}
int srl_lio_state_initialization(srl_lio *h, int index_frame, int initialization, const double prev2[7], const double prev1[7],
                                 double out[7]) {
    if (!h || !out || (index_frame > 2 && (!prev2 || !prev1))) return SRL_ERR_BAD_ARG;
    state s2, s1, cur;
    if (index_frame > 2) {
        s2.rotation = srl::Quat(prev2[0], prev2[1], prev2[2], prev2[3]); s2.translation = srl::vec3(prev2[4], prev2[5], prev2[6]);
        s1.rotation = srl::Quat(prev1[0], prev1[1], prev1[2], prev1[3]); s1.translation = srl::vec3(prev1[4], prev1[5], prev1[6]);
    }
    std::vector<point3D> none;
    cloudFrame f2(none, &s2), f1(none, &s1);
    std::vector<cloudFrame *> saved;
    saved.swap(h->lio->all_cloud_frame);
    h->lio->all_cloud_frame.push_back(&f2);
    h->lio->all_cloud_frame.push_back(&f1);
    const int saved_index = h->lio->index_frame, saved_init = h->lio->initialization;
    h->lio->index_frame = index_frame;
    h->lio->initialization = initialization;
    h->lio->stateInitialization(&cur);
    h->lio->all_cloud_frame.swap(saved);
    h->lio->index_frame = saved_index;
    h->lio->initialization = saved_init;
    out[0] = cur.rotation.w; out[1] = cur.rotation.x; out[2] = cur.rotation.y; out[3] = cur.rotation.z;
    for (int d = 0; d < 3; d++) out[4 + d] = cur.translation[d];
    return SRL_OK;
}

// ---- ROS-free replay driver
int srl_lio_set_odometry_options(srl_lio *h, const srl_odometry_opts *o) {
    if (!h || !o) return SRL_ERR_BAD_ARG;
    lioOptimization &L = *h->lio;
    L.init_voxel_size = o->init_voxel_size; L.init_sample_voxel_size = o->init_sample_voxel_size;
    L.init_num_frames = o->init_num_frames; L.num_for_initialization = o->num_for_initialization;
    L.voxel_size = o->voxel_size; L.sample_voxel_size = o->sample_voxel_size;
    L.max_num_points_in_voxel = o->max_num_points_in_voxel; L.min_distance_points = o->min_distance_points;
    L.motion_compensation = o->motion_compensation; L.initialization = o->initialization;
    L.point_time_enable = o->point_time_enable != 0;
    L.optimize_options = icpOptions::fromAbi(o->icp);
    L.eskf_pro->setAccCov(o->acc_cov); L.eskf_pro->setGyrCov(o->gyr_cov);
    L.eskf_pro->setBiasAccCov(o->b_acc_cov); L.eskf_pro->setBiasGyrCov(o->b_gyr_cov);
    return SRL_OK;
}

int srl_lio_run_measurement(srl_lio *h, double time_frame, const double *imu_t, const double *imu_acc, const double *imu_gyr,
                            int n_imu, const double *pts_raw, const double *pts_timestamp, int n_pts, double time_sweep_begin,
                            double time_sweep_offset, srl_replay_result *out) {
    if (!h || n_imu < 0 || n_pts < 0 || (n_imu > 0 && (!imu_t || !imu_acc || !imu_gyr)) || (n_pts > 0 && (!pts_raw || !pts_timestamp)))
        return SRL_ERR_BAD_ARG;
    lioOptimization &L = *h->lio;
    lioOptimization::Measurement m;
    m.time_frame = time_frame; m.time_sweep_begin = time_sweep_begin; m.time_sweep_offset = time_sweep_offset;
    m.imu.resize((size_t)n_imu);
    for (int i = 0; i < n_imu; i++) {
        m.imu[i].time = imu_t[i];
        m.imu[i].acc = srl::vec3(imu_acc[3 * i], imu_acc[3 * i + 1], imu_acc[3 * i + 2]);
        m.imu[i].gyr = srl::vec3(imu_gyr[3 * i], imu_gyr[3 * i + 1], imu_gyr[3 * i + 2]);
    }
    m.lidar_points.resize((size_t)n_pts);
    for (int i = 0; i < n_pts; i++) {
        point3D &p = m.lidar_points[i];
        p.raw_point = srl::vec3(pts_raw[3 * (size_t)i], pts_raw[3 * (size_t)i + 1], pts_raw[3 * (size_t)i + 2]);
        p.point = p.raw_point;                                   // cloudProcessing.cpp:143
        p.timestamp = pts_timestamp[i];
    }
    optimizeSummary s;
    bool processed = false;
    if (out) std::memset(out, 0, sizeof *out);
    try {
        if (initial_flag && !L.context()) return SRL_ERR_NO_DEVICE;
        processed = L.runMeasurement(m, &s);
    } catch (const std::exception &e) { return status_from_exception(h, e); }
    if (out) {
        out->processed = processed ? 1 : 0;
        out->initialized = initial_flag ? 1 : 0;
        out->index_frame = L.index_frame;
        if (processed) {
            out->success = s.success ? 1 : 0;
            out->num_residuals_used = s.num_residuals_used;
            out->iterations = L.last_num_iterations;
            out->frame_points = L.last_frame_points;
            out->keypoints = L.last_frame_keypoints;
            out->points_added = L.last_points_added;
            const state *st = L.all_cloud_frame.back()->p_state;
            state_to(*st, out->state);
        }
    }
    if (processed && !s.success) { h->err = s.error_log; return SRL_ERR_NOT_ENOUGH_RESIDUALS; }
    return SRL_OK;
}

int srl_lio_last_frame(srl_lio *h, int capacity, double *raw_point, double *point, double *imu_point, int *n) {
    if (!h || !n) return SRL_ERR_BAD_ARG;
    if (h->lio->all_cloud_frame.empty()) { *n = 0; return SRL_OK; }
    const std::vector<point3D> &f = h->lio->all_cloud_frame.back()->point_frame;
    *n = (int)f.size();
    const int m = std::min(capacity, *n);
    for (int k = 0; k < m; k++)
        for (int d = 0; d < 3; d++) {
            if (raw_point) raw_point[(size_t)k * 3 + d] = f[k].raw_point[d];
            if (point) point[(size_t)k * 3 + d] = f[k].point[d];
            if (imu_point) imu_point[(size_t)k * 3 + d] = f[k].imu_point[d];
        }
    return SRL_OK;
}

int srl_lio_optimize_resident(srl_lio *h, const srl_icp_opts *opts, double sample_voxel_size, const double *frame_raw, int n,
                              double state_io[16], const double t_last[3], int frame_id, int32_t *keypoint_index,
                              int *num_keypoints, int *iters, int *num_residuals_used) {
    if (!h || !opts || (n > 0 && !frame_raw) || !state_io || !t_last || n < 0) return SRL_ERR_BAD_ARG;
    if (!h->lio->context()) return SRL_ERR_NO_DEVICE;
    h->lio->setNormalEqProvider(nullptr, nullptr);
    std::vector<point3D> none;
    FrameWindow w(h, state_io, t_last, frame_id, none);
    const icpOptions o = icpOptions::fromAbi(*opts);
    optimizeSummary s;
    std::vector<int> kidx;
    try {
        s = h->lio->optimizeResident(&w.cur, frame_raw, n, o, sample_voxel_size, keypoint_index || num_keypoints ? &kidx : nullptr);
    } catch (const std::exception &e) {
        h->lio->all_cloud_frame.clear();
        state_to(h->cur_state, state_io);
        if (iters) *iters = h->lio->last_num_iterations;
        return status_from_exception(h, e);
    }
    h->lio->all_cloud_frame.clear();
    if (num_keypoints) *num_keypoints = (int)kidx.size();
    if (keypoint_index) for (size_t i = 0; i < kidx.size(); i++) keypoint_index[i] = kidx[i];
    state_to(h->cur_state, state_io);
    if (iters) *iters = h->lio->last_num_iterations;
    if (num_residuals_used) *num_residuals_used = s.num_residuals_used;
    if (!s.success) { h->err = s.error_log; return SRL_ERR_NOT_ENOUGH_RESIDUALS; }
    return SRL_OK;
}

int srl_lio_commit_frame(srl_lio *h, const double state[16], double voxel_size, int max_num_points_in_voxel,
                         double min_distance_points, int min_num_points, double *world_out, int *num_added) {
    if (!h || !state) return SRL_ERR_BAD_ARG;
    if (!h->lio->context()) return SRL_ERR_NO_DEVICE;
    srlivo::state st;
    state_from(state, st);
    try {
        const int added = h->lio->commitFrame(&st, voxel_size, max_num_points_in_voxel, min_distance_points, min_num_points, world_out, num_added != nullptr);
        if (num_added) *num_added = added;
    } catch (const std::exception &e) { return status_from_exception(h, e); }
    return SRL_OK;
}

int srl_lio_search_neighbors(srl_lio *h, const double point[3], int nb_voxels_visited, double size_voxel_map,
                             int max_num_neighbors, int threshold_voxel_capacity, double *out_xyz, int16_t *out_voxels,
                             int *num_found) {
    if (!h || !point || !out_xyz || !num_found) return SRL_ERR_BAD_ARG;
    try {
        std::vector<voxel> vox;
        const auto nb = h->lio->searchNeighbors(h->lio->voxel_map, srl::vec3(point[0], point[1], point[2]), nb_voxels_visited,
                                                size_voxel_map, max_num_neighbors, threshold_voxel_capacity, out_voxels ? &vox : nullptr);
        *num_found = (int)nb.size();
        for (size_t i = 0; i < nb.size(); i++) {
            for (int d = 0; d < 3; d++) out_xyz[i * 3 + d] = nb[i][d];
            if (out_voxels) { out_voxels[i * 3] = vox[i].x; out_voxels[i * 3 + 1] = vox[i].y; out_voxels[i * 3 + 2] = vox[i].z; }
        }
    } catch (const std::exception &e) { return status_from_exception(h, e); }
    return SRL_OK;
}

int srl_lio_neighborhood(srl_lio *h, const double *pts, int n, double center[3], double normal[3], double cov[9], double *a2D) {
    if (!h || !pts || n <= 0) return SRL_ERR_BAD_ARG;
    std::vector<srl::Vec3> P((size_t)n);
    for (int i = 0; i < n; i++) P[i] = srl::vec3(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
    try {
        const Neighborhood nb = h->lio->computeNeighborhoodDistribution(P);
        for (int d = 0; d < 3; d++) { if (center) center[d] = nb.center[d]; if (normal) normal[d] = nb.normal[d]; }
        if (cov) std::memcpy(cov, nb.covariance.a, 9 * sizeof(double));
        if (a2D) *a2D = nb.a2D;
    } catch (const std::exception &e) { return status_from_exception(h, e); }
    return SRL_OK;
}

int srl_lio_build_plane_residuals(srl_lio *h, const srl_icp_opts *opts, const double *raw_xyz, int n, const double st[16],
                                  const double t_last[3], int frame_id, double *out_rows, int max_out, int *num_out,
                                  double *loss_sum, int *success, double *keypoint_world) {
    if (!h || !opts || !raw_xyz || !st || !t_last || n < 0) return SRL_ERR_BAD_ARG;
    if (!h->lio->context()) return SRL_ERR_NO_DEVICE;
    std::vector<point3D> keypoints((size_t)n);
    for (int k = 0; k < n; k++) keypoints[k].raw_point = srl::vec3(raw_xyz[(size_t)k * 3], raw_xyz[(size_t)k * 3 + 1], raw_xyz[(size_t)k * 3 + 2]);
    double state_io[16];
    std::memcpy(state_io, st, sizeof state_io);
    FrameWindow w(h, state_io, t_last, frame_id, keypoints);
    const icpOptions o = icpOptions::fromAbi(*opts);
    std::vector<planeParam> pr;
    double loss = 0.0;
    optimizeSummary s;
    try {
        s = h->lio->buildPlaneResiduals(o, h->lio->voxel_map, keypoints, pr, &w.cur, loss);
    } catch (const std::exception &e) {
        h->lio->all_cloud_frame.clear();
        return status_from_exception(h, e);
    }
    h->lio->all_cloud_frame.clear();
    if (num_out) *num_out = (int)pr.size();
    if (loss_sum) *loss_sum = loss;
    if (success) *success = s.success ? 1 : 0;
    if (out_rows)
        for (int i = 0; i < (int)pr.size() && i < max_out; i++) {
            double *r = out_rows + (size_t)i * 15;
            for (int d = 0; d < 3; d++) { r[d] = pr[i].raw_point[d]; r[3 + d] = pr[i].norm_vector[d]; }
            for (int c = 0; c < 6; c++) r[6 + c] = pr[i].jacobians(0, c);
            r[12] = pr[i].norm_offset; r[13] = pr[i].distance; r[14] = pr[i].weight;
        }
    if (keypoint_world) for (int k = 0; k < n; k++) for (int d = 0; d < 3; d++) keypoint_world[(size_t)k * 3 + d] = keypoints[k].point[d];
    return SRL_OK;
}

int srl_grid_sampling(const double *world_xyz, int n, double size_voxel, int32_t *index_out, int *num_out) {
    if (n < 0 || (n > 0 && !world_xyz) || !num_out) return SRL_ERR_BAD_ARG;
    std::vector<point3D> pts((size_t)n), kp;
    for (int k = 0; k < n; k++) { pts[k].point = srl::vec3(world_xyz[(size_t)k * 3], world_xyz[(size_t)k * 3 + 1], world_xyz[(size_t)k * 3 + 2]); pts[k].index_frame = k; }
    gridSampling(pts, kp, size_voxel);
    *num_out = (int)kp.size();
    if (index_out) for (size_t i = 0; i < kp.size(); i++) index_out[i] = kp[i].index_frame;
    return SRL_OK;
}

// debug / parity hook: iteration order of std::tr1::unordered_map<voxel, ...> after inserting the given DISTINCT voxel keys
// in order, computed by the flat replay of host/tr1_order.h (what srl_frame_select_keypoints uses); order_out[r] = index
// into keys of the r-th element.  tests compare it with the real container (srl_grid_sampling).
int srl_debug_tr1_order(const int16_t *keys_xyz, int n, int32_t *order_out) {
    if (n < 0 || (n > 0 && (!keys_xyz || !order_out))) return SRL_ERR_BAD_ARG;
    std::vector<std::size_t> h((size_t)n);
    for (int i = 0; i < n; i++) h[(size_t)i] = std::hash<voxel>()(voxel(keys_xyz[3 * i], keys_xyz[3 * i + 1], keys_xyz[3 * i + 2]));
    std::vector<int> out((size_t)n);
    srl::Tr1Order::order(h.data(), n, out.data());
    for (int i = 0; i < n; i++) order_out[i] = out[(size_t)i];
    return SRL_OK;
}

// debug / parity hook: the same order from the RELATION of host/tr1_relation.h, by the steps of the device ordering of
// srl_frame_select_keypoints (bucket of every element at the final level -> exclusive scan over the bucket counts -> rank among the
// elements sharing a bucket), with the very functions the kernels call, compiled for the host.  No bucket-size limit here.
int srl_debug_tr1_order_by_relation(const int16_t *keys_xyz, int n, int32_t *order_out) {
    if (n < 0 || (n > 0 && (!keys_xyz || !order_out))) return SRL_ERR_BAD_ARG;
    if (n == 0) return SRL_OK;
    std::vector<unsigned long long> h((size_t)n);
    for (int i = 0; i < n; i++) h[(size_t)i] = (unsigned long long)std::hash<voxel>()(voxel(keys_xyz[3 * i], keys_xyz[3 * i + 1], keys_xyz[3 * i + 2]));
    SrlTr1Sched S;
    std::memset(&S, 0, sizeof S);
    S.steps = srl::Tr1Order::export_schedule(n, S.first, S.nb, SRL_TR1_MAX_STEPS);
    if (S.steps < 0) return SRL_ERR_UNSUPPORTED;
    const int G = srl_tr1_level(&S, (unsigned)n);
    const unsigned nb = S.nb[G];
    std::vector<std::vector<unsigned>> members(nb);
    for (int e = 0; e < n; e++) members[(size_t)(h[(size_t)e] % nb)].push_back((unsigned)e);
    size_t start = 0;
    for (unsigned b = 0; b < nb; b++) {
        const std::vector<unsigned> &m = members[b];
        for (size_t j = 0; j < m.size(); j++) {
            size_t rank = 0;
            for (size_t i = 0; i < m.size(); i++)
                if (i != j && srl_tr1_before(&S, h.data(), G, m[i], m[j])) ++rank;
            order_out[start + rank] = (int32_t)m[j];
        }
        start += m.size();
    }
    return SRL_OK;
}

}  // extern "C"
