// lioOptimization.cpp (host mirror) -- src/optimize.cpp re-hosted on the HIP backend.
// What stays on the host is what the reference keeps in 17-dim algebra (optimize.cpp:172-310); every
// per-point loop (optimize.cpp:30-40, 68-108, 160-170, 235, 239, 441-445) runs in HIP kernels behind
// the C-ABI.  No CPU fallback: a failing C-ABI call throws.
#include "lioOptimization.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <sstream>
#include <stdexcept>
#include <random>
#include <tr1/unordered_map>

namespace srlivo {

using srl::Mat17;
using srl::Mat2;
using srl::Mat3;
using srl::Mat32;
using srl::Quat;
using srl::Vec17;
using srl::Vec2;
using srl::Vec3;

srl::Vec3 G = srl::vec3(0.0, 0.0, 9.81);
double G_norm = 9.81;

namespace {
void check(srl_ctx *ctx, int rc, const char *what) {
    if (rc == SRL_OK) return;
    std::string msg = std::string(what) + ": " + srl_status_str(rc);
    if (ctx) { msg += " ("; msg += srl_last_error(ctx); msg += ")"; }
    throw std::runtime_error(msg);
}
}  // namespace

// ---------------------------------------------------------------- utility.cpp:167-201
void subSampleFrame(std::vector<point3D> &frame, double size_voxel) {
    std::tr1::unordered_map<voxel, std::vector<point3D>, std::hash<voxel>> grid;
    for (int i = 0; i < (int)frame.size(); i++) {
        auto kx = static_cast<short>(frame[i].point[0] / size_voxel);
        auto ky = static_cast<short>(frame[i].point[1] / size_voxel);
        auto kz = static_cast<short>(frame[i].point[2] / size_voxel);
        grid[voxel(kx, ky, kz)].push_back(frame[i]);
    }
    frame.resize(0);
    for (const auto &n : grid) {
        if (n.second.size() > 0) frame.push_back(n.second[0]);
    }
}

void gridSampling(const std::vector<point3D> &frame, std::vector<point3D> &keypoints, double size_voxel_subsampling) {
    keypoints.resize(0);
    std::vector<point3D> frame_sub(frame.begin(), frame.end());
    subSampleFrame(frame_sub, size_voxel_subsampling);
    keypoints.reserve(frame_sub.size());
    for (int i = 0; i < (int)frame_sub.size(); i++) keypoints.push_back(frame_sub[i]);
}

// ---------------------------------------------------------------- construction
lioOptimization::lioOptimization(int device) {
    eskf_pro = new eskfEstimator();
    if (device >= 0) {
        srl_ctx *ctx = nullptr;
        int rc = srl_ctx_create(device, &ctx);
        if (rc != SRL_OK) {
            delete eskf_pro;
            throw std::runtime_error(std::string("srl_ctx_create: ") + srl_status_str(rc));
        }
        voxel_map.ctx = ctx;
    }
}

lioOptimization::~lioOptimization() {
    releaseFrames();
    if (voxel_map.ctx) srl_ctx_destroy(voxel_map.ctx);
    delete eskf_pro;
}

int lioOptimization::residentSweep(const double *raw_xyz, int n) {
    if (!voxel_map.ctx) return SRL_ERR_NO_DEVICE;
    int rc = srl_sweep_upload(voxel_map.ctx, raw_xyz, n);
    resident_n = (rc == SRL_OK) ? n : -1;
    sweep_pinned = (rc == SRL_OK);
    return rc;
}

int lioOptimization::prefetchSweep(const double *raw_xyz, int n) {
    if (!voxel_map.ctx) return SRL_ERR_NO_DEVICE;
    const int rc = srl_sweep_prefetch(voxel_map.ctx, raw_xyz, n);
    prefetched_n = (rc == SRL_OK) ? n : -1;
    return rc;
}

int lioOptimization::swapSweep() {
    if (!voxel_map.ctx) return SRL_ERR_NO_DEVICE;
    if (pending_prefetch_rc != SRL_OK) {
        // the upload a solve issued on the caller's behalf (prefetchSweepDuringSolve) failed: that status -- and srl_last_error's text of
        // the failed call -- is what the caller gets here, not a "nothing prefetched" from the swap
        const int rc = pending_prefetch_rc;
        pending_prefetch_rc = SRL_OK;
        prefetched_n = -1;
        return rc;
    }
    const int rc = srl_sweep_swap(voxel_map.ctx);
    resident_n = (rc == SRL_OK) ? prefetched_n : -1;
    sweep_pinned = (rc == SRL_OK);
    prefetched_n = -1;
    return rc;
}

void lioOptimization::fillFrame(cloudFrame *p_frame, srl_frame &f) const {
    const state *cur = p_frame->p_state;
    const state *last_state = all_cloud_frame[p_frame->id - 1]->p_state;        // optimize.cpp:25
    f.q[0] = cur->rotation.w; f.q[1] = cur->rotation.x; f.q[2] = cur->rotation.y; f.q[3] = cur->rotation.z;
    for (int i = 0; i < 3; i++) {
        f.t[i] = cur->translation[i];
        f.t_last[i] = last_state->translation[i];
        f.t_il[i] = t_imu_lidar[i];
    }
    for (int i = 0; i < 9; i++) f.R_il[i] = R_imu_lidar.a[i];
    f.frame_id = p_frame->frame_id;
}

int lioOptimization::normalEquations(const icpOptions &o, cloudFrame *p_frame, srl_normal_eq &neq, void (*while_running)(void *), void *user) {
    srl_frame f;
    fillFrame(p_frame, f);
    const srl_icp_opts abi = o.toAbi();
    if (provider) return provider(&f, &abi, &neq, provider_user);
    if (!voxel_map.ctx) return SRL_ERR_NO_DEVICE;
    // while_running: host work that does not need the normal equations, run beside the kernels
    return srl_build_residuals_overlap(voxel_map.ctx, &f, &abi, &neq, while_running, user);
}

// ---------------------------------------------------------------- optimize.cpp:18-131
// Signature-compatible form: the residual list is materialised from the kernel's parity taps.
// updateIEKF below does NOT go through this (it only needs the reduced normal equations).
optimizeSummary lioOptimization::buildPlaneResiduals(const icpOptions &cur_icp_options, voxelHashMap &voxel_map_temp,
                                                     std::vector<point3D> &keypoints, std::vector<planeParam> &plane_residuals,
                                                     cloudFrame *p_frame, double &loss_sum) {
    srl_ctx *ctx = voxel_map_temp.ctx;
    if (!ctx) throw std::runtime_error("buildPlaneResiduals: no HIP context (the product has no CPU path)");
    const int n = (int)keypoints.size();
    std::vector<double> raw((size_t)n * 3);
    for (int k = 0; k < n; k++) for (int d = 0; d < 3; d++) raw[(size_t)k * 3 + d] = keypoints[k].raw_point[d];
    check(ctx, srl_sweep_upload(ctx, raw.data(), n), "srl_sweep_upload");
    releaseSweep();
    srl_frame f;
    fillFrame(p_frame, f);
    const srl_icp_opts abi = cur_icp_options.toAbi();
    srl_normal_eq neq;
    check(ctx, srl_set_taps(ctx, 1), "srl_set_taps");
    int rc = srl_build_residuals(ctx, &f, &abi, &neq);
    if (rc == SRL_ERR_NAN_PLANARITY) { srl_set_taps(ctx, 0); throw std::runtime_error("error"); }   // optimize.cpp:348-350
    check(ctx, rc, "srl_build_residuals");

    // transformKeypoints side effect (optimize.cpp:30-40): keypoint.point = R(q.normalized()) * (...) + t
    {
        const Quat qn = p_frame->p_state->rotation.normalized();
        const double qv[4] = {qn.w, qn.x, qn.y, qn.z};
        std::vector<double> world((size_t)n * 3);
        check(ctx, srl_transform_points(ctx, raw.data(), n, qv, f.t, f.R_il, f.t_il, world.data()), "srl_transform_points");
        for (int k = 0; k < n; k++) keypoints[k].point = srl::vec3(world[(size_t)k * 3], world[(size_t)k * 3 + 1], world[(size_t)k * 3 + 2]);
    }
    std::vector<uint8_t> status(n);
    std::vector<double> normal((size_t)n * 3), weight(n), off(n), dist(n), jac((size_t)n * 6);
    check(ctx, srl_fetch_neighbors(ctx, nullptr, status.data(), nullptr), "srl_fetch_neighbors");
    check(ctx, srl_fetch_residuals(ctx, normal.data(), nullptr, weight.data(), off.data(), dist.data(), jac.data()), "srl_fetch_residuals");
    srl_set_taps(ctx, 0);
    for (int k = 0; k < n; k++) {
        if (status[k] != 2) continue;
        planeParam pl;
        pl.raw_point = R_imu_lidar * keypoints[k].raw_point + t_imu_lidar;     // optimize.cpp:83,91
        pl.norm_vector = srl::vec3(normal[(size_t)k * 3], normal[(size_t)k * 3 + 1], normal[(size_t)k * 3 + 2]);
        for (int c = 0; c < 6; c++) pl.jacobians(0, c) = jac[(size_t)k * 6 + c];
        pl.norm_offset = off[k];
        pl.distance = dist[k];
        pl.weight = weight[k];
        plane_residuals.push_back(pl);
        loss_sum += pl.distance * pl.distance;                                  // optimize.cpp:104
    }
    optimizeSummary summary;
    summary.num_residuals_used = neq.num_residuals;
    if (!neq.success) {                                                         // optimize.cpp:110-123
        std::stringstream ss_out;
        ss_out << "[Optimization] Error : not enough keypoints selected in ct-icp !" << std::endl;
        ss_out << "[Optimization] number_of_residuals : " << neq.num_residuals << std::endl;
        summary.success = false;
        summary.error_log = ss_out.str();
    } else {
        summary.success = true;
    }
    return summary;
}

// ---------------------------------------------------------------- optimize.cpp:133-314
optimizeSummary lioOptimization::updateIEKF(const icpOptions &cur_icp_options, voxelHashMap &voxel_map_temp,
                                            std::vector<point3D> &keypoints, cloudFrame *p_frame) {
    if (!provider) {
        // the keypoints given here are the sweep: always uploaded (a pinned sweep is only ever used by solveIEKF, which
        // takes no keypoint vector -- never guessed from a matching count)
        srl_ctx *ctx = voxel_map_temp.ctx;
        if (!ctx) throw std::runtime_error("updateIEKF: no HIP context (the product has no CPU path)");
        if (ctx != voxel_map.ctx) throw std::runtime_error("updateIEKF: voxel_map_temp must be the node's voxel_map (one device map per lioOptimization)");
        const int n = (int)keypoints.size();
        std::vector<double> raw((size_t)n * 3);
        for (int k = 0; k < n; k++) for (int d = 0; d < 3; d++) raw[(size_t)k * 3 + d] = keypoints[k].raw_point[d];
        check(ctx, srl_sweep_upload(ctx, raw.data(), n), "srl_sweep_upload");
        releaseSweep();
    }
    return solveIEKF(cur_icp_options, p_frame);
}

// the ESIKF loop proper (optimize.cpp:135-313) on whatever sweep is resident in HBM
optimizeSummary lioOptimization::solveIEKF(const icpOptions &cur_icp_options, cloudFrame *p_frame) {
    const int max_num_iter = p_frame->frame_id < cur_icp_options.init_num_frames
                                 ? std::max(15, cur_icp_options.num_iters_icp) : cur_icp_options.num_iters_icp;
    if (!provider && !voxel_map.ctx) throw std::runtime_error("updateIEKF: no HIP context (the product has no CPU path)");

    const Vec3 p_predict = eskf_pro->getTranslation();
    const Quat q_predict = eskf_pro->getRotation();
    const Vec3 v_predict = eskf_pro->getVelocity();
    const Vec3 ba_predict = eskf_pro->getBa();
    const Vec3 bg_predict = eskf_pro->getBg();
    const Vec3 g_predict = eskf_pro->getGravity();

    optimizeSummary summary;
    iteration_log.clear();
    last_num_iterations = 0;
    last_num_observed = 0;
    last_solve_launches = 0;
    // whichever way the loop below is left (converged, iteration cap, failure, exception): tell the context that this solve is over --
    // it remembers the pass count for its arming policy and keeps a launch armed behind the last pass only for a prefetched sweep
    struct SolveEnd { srl_ctx *c; ~SolveEnd() { if (c) srl_solve_end(c); } } solve_end_guard{provider ? nullptr : voxel_map.ctx};
    // a prefetch registered for this solve (prefetchSweepDuringSolve) is issued beside the kernel of the first pass -- or, should no pass
    // get that far, when the solve is left
    auto issue_pending_prefetch = [this]() {
        if (pending_prefetch_n < 0) return;
        const double *raw = pending_prefetch_raw;
        const int n = pending_prefetch_n;
        pending_prefetch_raw = nullptr; pending_prefetch_n = -1;
        pending_prefetch_rc = voxel_map.ctx ? prefetchSweep(raw, n) : SRL_ERR_NO_DEVICE;
    };
    struct PrefetchGuard { decltype(issue_pending_prefetch) &f; ~PrefetchGuard() { f(); } } prefetch_guard{issue_pending_prefetch};

    // covariance projection helpers (optimize.cpp:220-232): rows then columns of the so3 / S2 blocks
    auto left3 = [](Mat17 &dst, const Mat3 &J, const Mat17 &src) {
        for (int j = 0; j < 17; j++) { const Vec3 c = J * src.block<3, 1>(3, j); dst.setBlock<3, 1>(3, j, c); }
    };
    auto left2 = [](Mat17 &dst, const Mat2 &J, const Mat17 &src) {
        for (int j = 0; j < 17; j++) { const Vec2 c = J * src.block<2, 1>(15, j); dst.setBlock<2, 1>(15, j, c); }
    };
    auto right3 = [](Mat17 &dst, const Mat3 &J, const Mat17 &src) {
        const Mat3 Jt = J.transpose();
        for (int j = 0; j < 17; j++) { const srl::Mat<1, 3> r = src.block<1, 3>(j, 3) * Jt; dst.setBlock<1, 3>(j, 3, r); }
    };
    auto right2 = [](Mat17 &dst, const Mat2 &J, const Mat17 &src) {
        const Mat2 Jt = J.transpose();
        for (int j = 0; j < 17; j++) { const srl::Mat<1, 2> r = src.block<1, 2>(j, 15) * Jt; dst.setBlock<1, 2>(j, 15, r); }
    };

    for (int i = -1; i < max_num_iter; i++) {
        // Everything of this iteration that does NOT depend on H_x -- the prior error state (optimize.cpp:172-211), the
        // covariance projection (:220-232) and the first 17x17 inverse (:234) -- runs on the host WHILE the kernels of this
        // iteration are in flight (srl_build_residuals_overlap): the same operations on the same values, only earlier.
        Vec3 d_so3, so3_dg;
        Mat32 B_x_predict;
        Vec17 d_x, d_x_new;
        Mat3 J_k_so3;
        Mat2 J_k_s2;
        Mat17 covariance, temp;
        bool prior_done = false;
        auto prior = [&]() {
        issue_pending_prefetch();            // (first pass only: a no-op afterwards) the next sweep starts crossing PCIe while this pass's kernel runs
        // prior error state (optimize.cpp:172-211)
        const Vec3 d_p = eskf_pro->getTranslation() - p_predict;
        const Quat d_q = q_predict.inverse() * eskf_pro->getRotation();
        d_so3 = numType::quatToSo3(d_q);
        const Vec3 d_v = eskf_pro->getVelocity() - v_predict;
        const Vec3 d_ba = eskf_pro->getBa() - ba_predict;
        const Vec3 d_bg = eskf_pro->getBg() - bg_predict;

        const Vec3 g = eskf_pro->getGravity();
        Vec3 g_predict_normalize = g_predict;
        Vec3 g_normalize = g;
        g_predict_normalize.normalize();
        g_normalize.normalize();

        const Vec3 crs = srl::cross(g_predict_normalize, g_normalize);
        const double dotv = g_predict_normalize.dot(g_normalize);

        Mat3 R_dg;
        if (std::fabs(1.0 - dotv) < 1e-6) R_dg = Mat3::Identity();
        else {
            const Mat3 skew = numType::skewSymmetric(crs);
            R_dg = Mat3::Identity() + skew + ((skew * skew) * (1.0 - dotv)) / (crs[0] * crs[0] + crs[1] * crs[1] + crs[2] * crs[2]);
        }
        so3_dg = numType::rotationToSo3(R_dg);
        B_x_predict = numType::derivativeS2(g_predict);
        const Vec2 d_g = B_x_predict.transpose() * so3_dg;

        for (int a = 0; a < 3; a++) {
            d_x[a] = d_p[a]; d_x[3 + a] = d_so3[a]; d_x[6 + a] = d_v[a]; d_x[9 + a] = d_ba[a]; d_x[12 + a] = d_bg[a];
        }
        d_x[15] = d_g[0];
        d_x[16] = d_g[1];

        J_k_so3 = Mat3::Identity() - 0.5 * numType::skewSymmetric(d_so3);
        J_k_s2 = Mat2::Identity() + ((0.5 * B_x_predict.transpose()) * numType::skewSymmetric(so3_dg)) * B_x_predict;

        d_x_new = d_x;
        {
            const Vec3 t3 = J_k_so3 * d_so3;
            const Vec2 t2 = J_k_s2 * d_g;
            for (int a = 0; a < 3; a++) d_x_new[3 + a] = t3[a];
            d_x_new[15] = t2[0];
            d_x_new[16] = t2[1];
        }

        // covariance projection (optimize.cpp:220-232): rows then columns of the so3 / S2 blocks
        covariance = eskf_pro->getCovariance();
        { Mat17 s = covariance; left3(covariance, J_k_so3, s); }
        { Mat17 s = covariance; left2(covariance, J_k_s2, s); }
        { Mat17 s = covariance; right3(covariance, J_k_so3, s); }
        { Mat17 s = covariance; right2(covariance, J_k_s2, s); }
        // first half of the Kalman gain (optimize.cpp:234)
        srl::inverse<17>(covariance / laser_point_cov, temp);
        prior_done = true;
        };
        // buildPlaneResiduals + H_x^T H_x + H_x^T h (optimize.cpp:153-170,235,239) on the device
        srl_normal_eq neq;
        std::memset(&neq, 0, sizeof neq);
        const int rc = normalEquations(cur_icp_options, p_frame, neq, [](void *u) { (*static_cast<decltype(prior) *>(u))(); }, &prior);
        if (rc == SRL_ERR_NAN_PLANARITY) throw std::runtime_error("error");      // optimize.cpp:348-350
        check(voxel_map.ctx, rc, "srl_build_residuals");
        summary.num_residuals_used = neq.num_residuals;
        if (!neq.success) {                                                      // optimize.cpp:110-123,155-156
            std::stringstream ss_out;
            ss_out << "[Optimization] Error : not enough keypoints selected in ct-icp !" << std::endl;
            ss_out << "[Optimization] number_of_residuals : " << neq.num_residuals << std::endl;
            summary.success = false;
            summary.error_log = ss_out.str();
            return summary;
        }
        summary.success = true;
        summary.error_log.clear();
        last_num_iterations++;
        last_solve_launches++;

        srl::Mat<6, 6> HTH;
        srl::Mat<6, 1> HTh;
        for (int a = 0; a < 36; a++) HTH.a[a] = neq.HtH[a];
        for (int a = 0; a < 6; a++) HTh.a[a] = neq.Hth[a];
        if (!prior_done) prior();      // provider / early paths that did not run it beside the kernels

        // Kalman gain pieces (optimize.cpp:235-244; `temp` = (covariance / laser_point_cov)^-1 comes from prior())
        for (int a = 0; a < 6; a++) for (int b = 0; b < 6; b++) temp(a, b) += HTH(a, b);
        // temp_inv = temp.inverse() is only ever read through temp_inv.block<17, 6>(0, 0) (optimize.cpp:237-242): its first six
        // columns, solved alone, carry the same bits as in the full inverse
        srl::Mat<17, 6> Tl;
        srl::inverse_cols<17, 6>(temp, Tl);
        const Vec17 K_h = Tl * HTh;              // temp_inv.block<17,6>(0,0) * H_x^T * h, with H_x^T h reduced on the device
        Mat17 K_x = Mat17::Zero();
        K_x.setBlock<17, 6>(0, 0, Tl * HTH);

        d_x = -K_h + (K_x - Mat17::Identity()) * d_x_new;

        if (record_iterations) { iterationLog L; L.neq = neq; L.d_x = d_x; iteration_log.push_back(L); }

        const Vec3 g_before = eskf_pro->getGravity();

        const Vec3 dx_p = srl::vec3(d_x[0], d_x[1], d_x[2]);
        const Vec3 dx_r = srl::vec3(d_x[3], d_x[4], d_x[5]);
        if (dx_p.norm() > 100.0 || AngularDistance(dx_r) > 100.0) {             // optimize.cpp:248-251
            continue;
        }

        eskf_pro->observe(d_x);                                                  // optimize.cpp:253
        last_num_observed++;

        p_frame->p_state->translation = eskf_pro->getTranslation();              // optimize.cpp:255-261
        p_frame->p_state->rotation = eskf_pro->getRotation();
        p_frame->p_state->velocity = eskf_pro->getVelocity();
        p_frame->p_state->ba = eskf_pro->getBa();
        p_frame->p_state->bg = eskf_pro->getBg();
        G = eskf_pro->getGravity();
        G_norm = G.norm();

        bool converage = false;
        if (p_frame->frame_id > 1 && dx_p.norm() < cur_icp_options.threshold_translation_norm &&
            AngularDistance(dx_r) < cur_icp_options.threshold_orientation_norm) {
            converage = true;
        }

        if (converage || i == max_num_iter - 1) {                               // optimize.cpp:272-310
            Mat17 covariance_new = covariance;
            const Mat32 B_x_before = numType::derivativeS2(g_before);
            Vec2 dg2;
            dg2[0] = d_x[15];
            dg2[1] = d_x[16];
            J_k_so3 = Mat3::Identity() - 0.5 * numType::skewSymmetric(dx_r);
            J_k_s2 = Mat2::Identity() + ((0.5 * B_x_before.transpose()) * numType::skewSymmetric(B_x_before * dg2)) * B_x_before;

            left3(covariance_new, J_k_so3, covariance);
            left2(covariance_new, J_k_s2, covariance);
            { const Mat17 s = covariance; right3(covariance_new, J_k_so3, s); right3(covariance, J_k_so3, s); }
            { const Mat17 s = covariance; right2(covariance_new, J_k_s2, s); right2(covariance, J_k_s2, s); }

            for (int j = 0; j < 6; j++) { const Vec3 c = J_k_so3 * K_x.block<3, 1>(3, j); K_x.setBlock<3, 1>(3, j, c); }
            for (int j = 0; j < 6; j++) { const Vec2 c = J_k_s2 * K_x.block<2, 1>(15, j); K_x.setBlock<2, 1>(15, j, c); }

            covariance = covariance_new - K_x.block<17, 6>(0, 0) * covariance.block<6, 17>(0, 0);
            eskf_pro->setCovariance(covariance);
            break;
        }
    }
    return summary;
}

// ---------------------------------------------------------------- optimize.cpp:316-353
// Single-neighbourhood utility of the class surface.  The hot path never calls it: the fused kernel
// performs the same computation per keypoint on the device (srl_kernels.hip phase 2).
Neighborhood lioOptimization::computeNeighborhoodDistribution(const std::vector<srl::Vec3> &points) {
    Neighborhood nb;
    Vec3 barycenter = Vec3::Zero();
    for (const auto &p : points) barycenter = barycenter + p;
    barycenter = barycenter / (double)points.size();
    nb.center = barycenter;
    Mat3 cov = Mat3::Zero();
    for (const auto &p : points)
        for (int k = 0; k < 3; ++k)
            for (int l = k; l < 3; ++l) cov(k, l) += (p[k] - barycenter[k]) * (p[l] - barycenter[l]);
    cov(1, 0) = cov(0, 1);
    cov(2, 0) = cov(0, 2);
    cov(2, 1) = cov(1, 2);
    nb.covariance = cov;
    srl::SelfAdjointEigenSolver3 es(cov);                                 // optimize.cpp:339
    nb.normal = es.eigenvector(0).normalized();                          // es.eigenvectors().col(0).normalized()
    const double sigma_1 = std::sqrt(std::abs(es.eigenvalues()[2]));
    const double sigma_2 = std::sqrt(std::abs(es.eigenvalues()[1]));
    const double sigma_3 = std::sqrt(std::abs(es.eigenvalues()[0]));
    nb.a2D = (sigma_2 - sigma_3) / sigma_1;
    if (nb.a2D != nb.a2D) throw std::runtime_error("error");
    return nb;
}

// ---------------------------------------------------------------- optimize.cpp:365-426
std::vector<srl::Vec3> lioOptimization::searchNeighbors(voxelHashMap &map, const srl::Vec3 &point, int nb_voxels_visited,
                                                        double size_voxel_map, int max_num_neighbors,
                                                        int threshold_voxel_capacity, std::vector<voxel> *voxels) {
    if (!map.ctx) throw std::runtime_error("searchNeighbors: no HIP context (the product has no CPU path)");
    const int K = max_num_neighbors;
    std::vector<int32_t> ids(K);
    std::vector<float> xyz((size_t)K * 3);
    int32_t nf = 0;
    check(map.ctx, srl_search_neighbors(map.ctx, point.a, 1, nb_voxels_visited, size_voxel_map, K, threshold_voxel_capacity,
                                        ids.data(), xyz.data(), &nf), "srl_search_neighbors");
    std::vector<srl::Vec3> closest_neighbors(nf);
    if (voxels) voxels->resize(nf);
    for (int i = 0; i < nf; i++) {
        closest_neighbors[i] = srl::vec3((double)xyz[3 * i], (double)xyz[3 * i + 1], (double)xyz[3 * i + 2]);
        if (voxels)   // the voxel a stored point lives in = key of its FP32 position (lioOptimization.cpp:403-405)
            (*voxels)[i] = voxel(static_cast<short>(closest_neighbors[i][0] / size_voxel_map),
                                 static_cast<short>(closest_neighbors[i][1] / size_voxel_map),
                                 static_cast<short>(closest_neighbors[i][2] / size_voxel_map));
    }
    return closest_neighbors;
}

// ---------------------------------------------------------------- lioOptimization.cpp:400-446,520-554,574-581
void lioOptimization::addPointToMap(voxelHashMap &map, const srl::Vec3 &point, double voxel_size, int max_num_points_in_voxel,
                                    double min_distance_points, int min_num_points, cloudFrame *p_frame) {
    (void)p_frame;
    if (!map.ctx) throw std::runtime_error("addPointToMap: no HIP context (the product has no CPU path)");
    check(map.ctx, srl_map_insert(map.ctx, point.a, 1, voxel_size, max_num_points_in_voxel, min_distance_points, min_num_points, nullptr),
          "srl_map_insert");
}

void lioOptimization::addPointsToMap(voxelHashMap &map, cloudFrame *p_frame, double voxel_size, int max_num_points_in_voxel,
                                     double min_distance_points, int min_num_points, bool to_rendering) {
    (void)to_rendering;   // colour map / rendering bookkeeping belongs to the vision stage (out of scope)
    if (!map.ctx) throw std::runtime_error("addPointsToMap: no HIP context (the product has no CPU path)");
    const int n = (int)p_frame->point_frame.size();
    std::vector<double> xyz((size_t)n * 3);
    for (int k = 0; k < n; k++) for (int d = 0; d < 3; d++) xyz[(size_t)k * 3 + d] = p_frame->point_frame[k].point[d];
    check(map.ctx, srl_map_insert(map.ctx, xyz.data(), n, voxel_size, max_num_points_in_voxel, min_distance_points, min_num_points, nullptr),
          "srl_map_insert");
}

size_t lioOptimization::mapSize(const voxelHashMap &map) {
    if (!map.ctx) throw std::runtime_error("mapSize: no HIP context");
    int64_t np = 0;
    check(map.ctx, srl_map_size(map.ctx, &np, nullptr), "srl_map_size");
    return (size_t)np;
}

// ---------------------------------------------------------------- optimize.cpp:428-448
optimizeSummary lioOptimization::optimize(cloudFrame *p_frame, const icpOptions &cur_icp_options, double sample_voxel_size) {
    std::vector<point3D> keypoints;
    gridSampling(p_frame->point_frame, keypoints, sample_voxel_size);

    optimizeSummary optimize_summary = updateIEKF(cur_icp_options, voxel_map, keypoints, p_frame);
    if (!optimize_summary.success) return optimize_summary;

    // transformPoint over the whole frame with the final pose (optimize.cpp:441-445), on the device
    srl_ctx *ctx = voxel_map.ctx;
    const int n = (int)p_frame->point_frame.size();
    if (n > 0) {
        if (!ctx) throw std::runtime_error("optimize: no HIP context (the product has no CPU path)");
        std::vector<double> raw((size_t)n * 3), world((size_t)n * 3);
        for (int k = 0; k < n; k++) for (int d = 0; d < 3; d++) raw[(size_t)k * 3 + d] = p_frame->point_frame[k].raw_point[d];
        const Quat &q_end = p_frame->p_state->rotation;
        const double qv[4] = {q_end.w, q_end.x, q_end.y, q_end.z};
        check(ctx, srl_transform_points(ctx, raw.data(), n, qv, p_frame->p_state->translation.a, R_imu_lidar.a, t_imu_lidar.a, world.data()),
              "srl_transform_points");
        for (int k = 0; k < n; k++) p_frame->point_frame[k].point = srl::vec3(world[(size_t)k * 3], world[(size_t)k * 3 + 1], world[(size_t)k * 3 + 2]);
    }
    return optimize_summary;
}

// ---------------------------------------------------------------- lioOptimization.cpp:895-990
void lioOptimization::stateInitialization(state *cur_state) {
    if (index_frame <= 2) {
        cur_state->rotation = Quat::Identity();
        cur_state->translation = Vec3::Zero();
        return;
    }
    // index_frame == 3 and index_frame > 3 run the same code upstream
    const state *s1 = all_cloud_frame[all_cloud_frame.size() - 1]->p_state;
    const state *s2 = all_cloud_frame[all_cloud_frame.size() - 2]->p_state;
    const bool constant_velocity = initialization == INIT_CONSTANT_VELOCITY || (initialization == INIT_IMU && !initial_flag);
    if (constant_velocity) {
        const Quat d = s1->rotation * s2->rotation.inverse();
        cur_state->rotation = d * s1->rotation;
        cur_state->translation = s1->translation + d * (s1->translation - s2->translation);
    } else if (initialization == INIT_IMU) {
        cur_state->rotation = eskf_pro->getRotation();
        cur_state->translation = eskf_pro->getTranslation();
    } else {
        cur_state->rotation = s1->rotation;
        cur_state->translation = s1->translation;
    }
}

// ---------------------------------------------------------------- lioOptimization.cpp:786-819
void lioOptimization::makePointTimestamp(std::vector<point3D> &sweep, double time_begin, double time_end) {
    const double delta_t = time_end - time_begin;
    if (point_time_enable) {
        for (size_t i = 0; i < sweep.size(); i++) {
            sweep[i].relative_time = sweep[i].timestamp - time_begin;
            sweep[i].alpha_time = sweep[i].relative_time / delta_t;
            sweep[i].relative_time = sweep[i].relative_time * 1000.0;
            if (sweep[i].alpha_time > 1.0) sweep[i].alpha_time = 1.0 - 1e-5;
        }
        return;
    }
    size_t kept = 0;                     // erase-in-loop upstream; one stable compaction here
    for (size_t i = 0; i < sweep.size(); i++) {
        if (sweep[i].timestamp > time_end || sweep[i].timestamp < time_begin) continue;
        point3D p = sweep[i];
        p.relative_time = p.timestamp - time_begin;
        p.alpha_time = p.relative_time / delta_t;
        p.relative_time = p.relative_time * 1000.0;
        sweep[kept++] = p;
    }
    sweep.resize(kept);
}

// ---------------------------------------------------------------- lioOptimization.cpp:821-893
cloudFrame *lioOptimization::buildFrame(std::vector<point3D> &cut_sweep, state *cur_state, double timestamp_begin, double timestamp_offset) {
    srl_ctx *ctx = voxel_map.ctx;
    if (!ctx) throw std::runtime_error("buildFrame: no HIP context (the product has no CPU path)");
    std::vector<point3D> sweep(cut_sweep);

    const double time_sweep_begin = timestamp_begin;
    const double time_frame_begin = timestamp_begin;
    makePointTimestamp(sweep, time_frame_begin, timestamp_begin + timestamp_offset);

    // device: distortFrameBy* + transformAllImuPoint over the whole sweep
    const int n = (int)sweep.size();
    std::vector<double> raw((size_t)n * 3), rel((size_t)n), imu_in((size_t)n * 3), imu_out((size_t)n * 3), raw_out((size_t)n * 3);
    for (int i = 0; i < n; i++) {
        for (int d = 0; d < 3; d++) { raw[(size_t)i * 3 + d] = sweep[i].raw_point[d]; imu_in[(size_t)i * 3 + d] = sweep[i].imu_point[d]; }
        rel[i] = sweep[i].relative_time;
    }
    std::vector<srl_imu_state> st(imu_states.size());
    for (size_t k = 0; k < imu_states.size(); k++) {
        st[k].timestamp = imu_states[k].timestamp;
        for (int d = 0; d < 3; d++) {
            st[k].un_acc[d] = imu_states[k].un_acc[d]; st[k].un_gyr[d] = imu_states[k].un_gyr[d];
            st[k].trans[d] = imu_states[k].trans[d]; st[k].vel[d] = imu_states[k].vel[d];
        }
        st[k].quat[0] = imu_states[k].quat.w; st[k].quat[1] = imu_states[k].quat.x; st[k].quat[2] = imu_states[k].quat.y; st[k].quat[3] = imu_states[k].quat.z;
    }
    const int mode = motion_compensation == CONSTANT_VELOCITY ? SRL_MC_CONSTANT_VELOCITY : motion_compensation == IMU ? SRL_MC_IMU : SRL_MC_NONE;
    check(ctx, srl_frame_undistort(ctx, raw.data(), rel.data(), imu_in.data(), n, st.data(), (int)st.size(), time_frame_begin, mode,
                                   R_imu_lidar.a, t_imu_lidar.a, imu_out.data(), raw_out.data()), "srl_frame_undistort");

    // host: the order (indices only).  subSampleFrame keys on point3D::point, which still holds the sensor-frame
    // point here (cloudProcessing.cpp:143); the engine is shared by both shuffles.
    const double sample_size = index_frame < init_num_frames ? init_voxel_size : voxel_size;
    std::vector<int> order((size_t)n);
    for (int i = 0; i < n; i++) order[i] = i;
    std::mt19937_64 seed;                                                // boost::mt19937_64 seed;
    std::shuffle(order.begin(), order.end(), seed);
    if (voxel_size > 0) {
        std::tr1::unordered_map<voxel, std::vector<int>, std::hash<voxel>> grid;
        for (int i : order) {
            const srl::Vec3 &p = sweep[i].point;
            grid[voxel(static_cast<short>(p[0] / sample_size), static_cast<short>(p[1] / sample_size), static_cast<short>(p[2] / sample_size))].push_back(i);
        }
        order.resize(0);
        for (const auto &kv : grid)
            if (kv.second.size() > 0) order.push_back(kv.second[0]);
        std::shuffle(order.begin(), order.end(), seed);
    }
    std::vector<int32_t> take(order.begin(), order.end());
    check(ctx, srl_frame_take(ctx, take.data(), (int)take.size()), "srl_frame_take");
    releaseSweep();

    std::vector<point3D> frame(order.size());
    for (size_t k = 0; k < order.size(); k++) {
        const int i = order[k];
        frame[k] = sweep[i];
        for (int d = 0; d < 3; d++) { frame[k].imu_point[d] = imu_out[(size_t)i * 3 + d]; frame[k].raw_point[d] = raw_out[(size_t)i * 3 + d]; }
    }

    double dt_offset = 0;
    if (index_frame > 1) dt_offset -= time_frame_begin - all_cloud_frame.back()->time_sweep_end;
    if (index_frame <= 2)
        for (auto &p : frame) p.alpha_time = 1.0;

    // transformPoint (lioOptimization.cpp:864-876): prior pose for index_frame > 2, identity before
    const Quat q_t = index_frame > 2 ? cur_state->rotation : Quat::Identity();
    const Vec3 t_t = index_frame > 2 ? cur_state->translation : Vec3::Zero();
    if (!frame.empty()) {
        std::vector<double> fr(frame.size() * 3), world(frame.size() * 3);
        for (size_t k = 0; k < frame.size(); k++) for (int d = 0; d < 3; d++) fr[k * 3 + d] = frame[k].raw_point[d];
        const double qv[4] = {q_t.w, q_t.x, q_t.y, q_t.z};
        check(ctx, srl_transform_points(ctx, fr.data(), (int)frame.size(), qv, t_t.a, R_imu_lidar.a, t_imu_lidar.a, world.data()), "srl_transform_points");
        for (size_t k = 0; k < frame.size(); k++) frame[k].point = srl::vec3(world[k * 3], world[k * 3 + 1], world[k * 3 + 2]);
    }

    cloudFrame *p_frame = new cloudFrame(frame, cur_state);
    p_frame->time_sweep_begin = time_sweep_begin;
    p_frame->time_sweep_end = timestamp_begin + timestamp_offset;
    p_frame->time_frame_begin = time_frame_begin;
    p_frame->time_frame_end = p_frame->time_sweep_end;
    p_frame->offset_begin = 0;
    p_frame->offset_end = timestamp_offset;
    p_frame->dt_offset = dt_offset;
    p_frame->id = (int)all_cloud_frame.size();
    p_frame->sub_id = 0;
    p_frame->frame_id = index_frame;
    all_cloud_frame.push_back(p_frame);
    return p_frame;
}

optimizeSummary lioOptimization::optimizeBuiltFrame(cloudFrame *p_frame, const icpOptions &cur_icp_options, double sample_voxel_size,
                                                    std::vector<int> *keypoint_index) {
    srl_ctx *ctx = voxel_map.ctx;
    if (!ctx) throw std::runtime_error("optimize: no HIP context (the product has no CPU path)");
    releaseSweep();
    const Quat &q = p_frame->p_state->rotation;
    const double qv[4] = {q.w, q.x, q.y, q.z};
    int frame_n = 0;
    check(ctx, srl_frame_size(ctx, &frame_n), "srl_frame_size");
    // the index list is fetched only for a caller that asks for it: the selection itself stays on the device as the resident sweep
    std::vector<int32_t> idx(keypoint_index ? (size_t)std::max(frame_n, 1) : 0);        // capacity = points of the resident frame
    int m = 0;
    check(ctx, srl_frame_select_keypoints(ctx, qv, p_frame->p_state->translation.a, R_imu_lidar.a, t_imu_lidar.a, sample_voxel_size,
                                          keypoint_index ? idx.data() : nullptr, &m), "srl_frame_select_keypoints");
    if (keypoint_index) keypoint_index->assign(idx.begin(), idx.begin() + m);
    last_frame_keypoints = m;
    resident_n = m;
    sweep_pinned = true;
    optimizeSummary s = solveIEKF(cur_icp_options, p_frame);
    releaseSweep();
    return s;
}

// ---------------------------------------------------------------- lioOptimization.cpp:991-1034 (LIO part)
optimizeSummary lioOptimization::stateEstimation(cloudFrame *p_frame) {
    optimizeSummary optimize_summary;
    const double kSizeVoxelMap = optimize_options.size_voxel_map;
    state commit_pose;                                   // identity: what buildFrame gave point3D::point for index_frame <= 2
    if (p_frame->frame_id > 1) {
        const double svs = p_frame->frame_id < init_num_frames ? init_sample_voxel_size : sample_voxel_size;
        optimize_summary = optimizeBuiltFrame(p_frame, optimize_options, svs, nullptr);      // (sets last_frame_keypoints)
        if (!optimize_summary.success) return optimize_summary;
        commit_pose = *p_frame->p_state;                 // optimize() re-transformed the frame with the final pose
    } else {
        p_frame->p_state->translation = eskf_pro->getTranslation();
        p_frame->p_state->rotation = eskf_pro->getRotation();
        p_frame->p_state->velocity = eskf_pro->getVelocity();
        p_frame->p_state->ba = eskf_pro->getBa();
        p_frame->p_state->bg = eskf_pro->getBg();
        G = eskf_pro->getGravity();
        G_norm = G.norm();
        optimize_summary.success = true;
    }
    // addPointsToMap(voxel_map, p_frame, ...) on the frame resident in HBM
    const int n = (int)p_frame->point_frame.size();
    std::vector<double> world(download_frame_points ? (size_t)n * 3 : 0);
    last_points_added = commitFrame(&commit_pose, kSizeVoxelMap, max_num_points_in_voxel, min_distance_points, 0,
                                    download_frame_points && n > 0 ? world.data() : nullptr);
    if (download_frame_points && p_frame->frame_id > 1)
        for (int k = 0; k < n; k++) p_frame->point_frame[k].point = srl::vec3(world[(size_t)k * 3], world[(size_t)k * 3 + 1], world[(size_t)k * 3 + 2]);
    return optimize_summary;
}

// ---------------------------------------------------------------- lioOptimization.cpp:1036-1133 (LIO part)
void lioOptimization::process(std::vector<point3D> &cut_sweep, double timestamp_begin, double timestamp_offset, optimizeSummary *summary_out) {
    state *cur_state = new state();
    stateInitialization(cur_state);
    std::vector<point3D> const_frame(cut_sweep.begin(), cut_sweep.end());
    cloudFrame *p_frame = buildFrame(const_frame, cur_state, timestamp_begin, timestamp_offset);
    last_frame_points = (int)p_frame->point_frame.size();
    optimizeSummary summary = stateEstimation(p_frame);
    if (summary_out) *summary_out = summary;
    dt_sum = 0;

    int num_remove = 0;
    auto drop_front = [&]() {
        cloudFrame *f = all_cloud_frame[0];
        trajectory.push_back(poseRecord{f->time_sweep_end, f->p_state->translation, f->p_state->rotation});   // recordSinglePose
        delete f->p_state;                                                                              // cloudFrame::release
        delete f;
        all_cloud_frame.erase(all_cloud_frame.begin());
        num_remove++;
    };
    if (initial_flag) {
        if (index_frame > 1)
            while (all_cloud_frame.size() > 2) drop_front();
    } else {
        while ((int)all_cloud_frame.size() > num_for_initialization) drop_front();
    }
    for (size_t i = 0; i < all_cloud_frame.size(); i++) all_cloud_frame[i]->id = all_cloud_frame[i]->id - num_remove;
}

void lioOptimization::releaseFrames() {
    for (cloudFrame *f : all_cloud_frame) { delete f->p_state; delete f; }
    all_cloud_frame.clear();
}

// ---------------------------------------------------------------- lioOptimization.cpp:1427-1584 (loop body of run())
bool lioOptimization::runMeasurement(Measurement &measurement, optimizeSummary *summary) {
    const double time_frame = measurement.time_frame;
    double dx = 0, dy = 0, dz = 0, rx = 0, ry = 0, rz = 0;

    if (!initial_flag) {
        for (const imuSample &imu_msg : measurement.imu) {
            const double time_imu = imu_msg.time;
            if (time_imu <= time_frame) {
                current_time = time_imu;
                dx = imu_msg.acc[0]; dy = imu_msg.acc[1]; dz = imu_msg.acc[2];
                rx = imu_msg.gyr[0]; ry = imu_msg.gyr[1]; rz = imu_msg.gyr[2];
            } else {
                const double dt_1 = time_frame - current_time;
                const double dt_2 = time_imu - time_frame;
                current_time = time_frame;
                const double w1 = dt_2 / (dt_1 + dt_2);
                const double w2 = dt_1 / (dt_1 + dt_2);
                dx = w1 * dx + w2 * imu_msg.acc[0]; dy = w1 * dy + w2 * imu_msg.acc[1]; dz = w1 * dz + w2 * imu_msg.acc[2];
                rx = w1 * rx + w2 * imu_msg.gyr[0]; ry = w1 * ry + w2 * imu_msg.gyr[1]; rz = w1 * rz + w2 * imu_msg.gyr[2];
            }
            imu_meas.emplace_back(current_time, std::make_pair(srl::vec3(rx, ry, rz), srl::vec3(dx, dy, dz)));
        }
        eskf_pro->tryInit(imu_meas);
        imu_meas.clear();
        last_time_frame = time_frame;
        return false;
    }

    auto push_state = [&](const Vec3 &un_acc, const Vec3 &un_gyr, bool before_predict) {
        (void)before_predict;
        imuState s;
        s.timestamp = current_time;
        s.un_acc = un_acc;
        s.un_gyr = un_gyr;
        s.trans = eskf_pro->getTranslation();
        s.quat = eskf_pro->getRotation();
        s.vel = eskf_pro->getVelocity();
        imu_states.push_back(s);
    };
    push_state(eskf_pro->getRotation().toRotationMatrix() * (eskf_pro->getLastAcc() - eskf_pro->getBa()),
               eskf_pro->getLastGyr() - eskf_pro->getBg(), true);

    for (const imuSample &imu_msg : measurement.imu) {
        const double time_imu = imu_msg.time;
        double dt;
        if (time_imu <= time_frame) {
            dt = time_imu - current_time;
            if (dt < -1e-6) continue;
            current_time = time_imu;
            dx = imu_msg.acc[0]; dy = imu_msg.acc[1]; dz = imu_msg.acc[2];
            rx = imu_msg.gyr[0]; ry = imu_msg.gyr[1]; rz = imu_msg.gyr[2];
        } else {
            const double dt_1 = time_frame - current_time;
            const double dt_2 = time_imu - time_frame;
            current_time = time_frame;
            const double w1 = dt_2 / (dt_1 + dt_2);
            const double w2 = dt_1 / (dt_1 + dt_2);
            dx = w1 * dx + w2 * imu_msg.acc[0]; dy = w1 * dy + w2 * imu_msg.acc[1]; dz = w1 * dz + w2 * imu_msg.acc[2];
            rx = w1 * rx + w2 * imu_msg.gyr[0]; ry = w1 * ry + w2 * imu_msg.gyr[1]; rz = w1 * rz + w2 * imu_msg.gyr[2];
            dt = dt_1;
        }
        // un_acc / un_gyr use the filter state BEFORE the predict, trans / quat / vel the state after it
        const Vec3 un_acc = eskf_pro->getRotation().toRotationMatrix() * (0.5 * (eskf_pro->getLastAcc() + srl::vec3(dx, dy, dz)) - eskf_pro->getBa());
        const Vec3 un_gyr = 0.5 * (eskf_pro->getLastGyr() + srl::vec3(rx, ry, rz)) - eskf_pro->getBg();
        dt_sum = dt_sum + dt;
        eskf_pro->predict(dt, srl::vec3(dx, dy, dz), srl::vec3(rx, ry, rz));
        push_state(un_acc, un_gyr, false);
    }

    process(measurement.lidar_points, measurement.time_sweep_begin, measurement.time_sweep_offset, summary);

    imu_states.clear();
    last_time_frame = time_frame;
    index_frame++;
    return true;
}

// ---------------------------------------------------------------- frame-resident optimize() (optimize.cpp:428-448)
optimizeSummary lioOptimization::optimizeResident(cloudFrame *p_frame, const double *frame_raw, int n, const icpOptions &cur_icp_options,
                                                  double sample_voxel_size, std::vector<int> *keypoint_index) {
    srl_ctx *ctx = voxel_map.ctx;
    if (!ctx) throw std::runtime_error("optimize: no HIP context (the product has no CPU path)");
    check(ctx, srl_frame_upload(ctx, frame_raw, n), "srl_frame_upload");
    return optimizeBuiltFrame(p_frame, cur_icp_options, sample_voxel_size, keypoint_index);
}

int lioOptimization::commitFrame(const state *p_state, double voxel_size, int max_num_points_in_voxel, double min_distance_points,
                                 int min_num_points, double *world_out, bool want_added) {
    srl_ctx *ctx = voxel_map.ctx;
    if (!ctx) throw std::runtime_error("addPointsToMap: no HIP context (the product has no CPU path)");
    const double qv[4] = {p_state->rotation.w, p_state->rotation.x, p_state->rotation.y, p_state->rotation.z};
    int added = -1;
    check(ctx, srl_frame_commit(ctx, qv, p_state->translation.a, R_imu_lidar.a, t_imu_lidar.a, voxel_size, max_num_points_in_voxel,
                                min_distance_points, min_num_points, world_out, want_added ? &added : nullptr), "srl_frame_commit");
    return added;
}

}  // namespace srlivo
