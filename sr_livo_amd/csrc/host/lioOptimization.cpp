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

int lioOptimization::normalEquations(const icpOptions &o, cloudFrame *p_frame, srl_normal_eq &neq) {
    srl_frame f;
    fillFrame(p_frame, f);
    const srl_icp_opts abi = o.toAbi();
    if (provider) return provider(&f, &abi, &neq, provider_user);
    if (!voxel_map.ctx) return SRL_ERR_NO_DEVICE;
    return srl_build_residuals(voxel_map.ctx, &f, &abi, &neq);
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
    (void)voxel_map_temp;
    if (!provider) {
        srl_ctx *ctx = voxel_map.ctx;
        if (!ctx) throw std::runtime_error("updateIEKF: no HIP context (the product has no CPU path)");
        if (!sweep_pinned || resident_n != (int)keypoints.size() || keypoints.empty()) {
            const int n = (int)keypoints.size();
            std::vector<double> raw((size_t)n * 3);
            for (int k = 0; k < n; k++) for (int d = 0; d < 3; d++) raw[(size_t)k * 3 + d] = keypoints[k].raw_point[d];
            check(ctx, srl_sweep_upload(ctx, raw.data(), n), "srl_sweep_upload");
        }
        if (!sweep_pinned) resident_n = -1;
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

    for (int i = -1; i < max_num_iter; i++) {
        // buildPlaneResiduals + H_x^T H_x + H_x^T h (optimize.cpp:153-170,235,239) on the device
        srl_normal_eq neq;
        std::memset(&neq, 0, sizeof neq);
        const int rc = normalEquations(cur_icp_options, p_frame, neq);
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

        srl::Mat<6, 6> HTH;
        srl::Mat<6, 1> HTh;
        for (int a = 0; a < 36; a++) HTH.a[a] = neq.HtH[a];
        for (int a = 0; a < 6; a++) HTh.a[a] = neq.Hth[a];

        // prior error state (optimize.cpp:172-211)
        const Vec3 d_p = eskf_pro->getTranslation() - p_predict;
        const Quat d_q = q_predict.inverse() * eskf_pro->getRotation();
        const Vec3 d_so3 = numType::quatToSo3(d_q);
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
        const Vec3 so3_dg = numType::rotationToSo3(R_dg);
        const Mat32 B_x_predict = numType::derivativeS2(g_predict);
        const Vec2 d_g = B_x_predict.transpose() * so3_dg;

        Vec17 d_x;
        for (int a = 0; a < 3; a++) {
            d_x[a] = d_p[a]; d_x[3 + a] = d_so3[a]; d_x[6 + a] = d_v[a]; d_x[9 + a] = d_ba[a]; d_x[12 + a] = d_bg[a];
        }
        d_x[15] = d_g[0];
        d_x[16] = d_g[1];

        Mat3 J_k_so3 = Mat3::Identity() - 0.5 * numType::skewSymmetric(d_so3);
        Mat2 J_k_s2 = Mat2::Identity() + ((0.5 * B_x_predict.transpose()) * numType::skewSymmetric(so3_dg)) * B_x_predict;

        Vec17 d_x_new = d_x;
        {
            const Vec3 t3 = J_k_so3 * d_so3;
            const Vec2 t2 = J_k_s2 * d_g;
            for (int a = 0; a < 3; a++) d_x_new[3 + a] = t3[a];
            d_x_new[15] = t2[0];
            d_x_new[16] = t2[1];
        }

        // covariance projection (optimize.cpp:220-232): rows then columns of the so3 / S2 blocks
        Mat17 covariance = eskf_pro->getCovariance();
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
        { Mat17 s = covariance; left3(covariance, J_k_so3, s); }
        { Mat17 s = covariance; left2(covariance, J_k_s2, s); }
        { Mat17 s = covariance; right3(covariance, J_k_so3, s); }
        { Mat17 s = covariance; right2(covariance, J_k_s2, s); }

        // Kalman gain pieces (optimize.cpp:234-244)
        Mat17 temp, temp_inv;
        srl::inverse<17>(covariance / laser_point_cov, temp);
        for (int a = 0; a < 6; a++) for (int b = 0; b < 6; b++) temp(a, b) += HTH(a, b);
        srl::inverse<17>(temp, temp_inv);

        const srl::Mat<17, 6> Tl = temp_inv.block<17, 6>(0, 0);
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
    // cyclic Jacobi (the restated SelfAdjointEigenSolver<Matrix3d>)
    double a[3][3], V[3][3], scale = 0.0;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) scale = std::max(scale, std::fabs(cov(i, j)));
    if (!(scale > 0.0)) scale = 1.0;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { a[i][j] = cov(i, j) / scale; V[i][j] = i == j ? 1.0 : 0.0; }
    for (int sweep = 0; sweep < 32; sweep++) {
        const double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
        if (off < 1e-40) break;
        for (int p = 0; p < 2; p++)
            for (int q = p + 1; q < 3; q++) {
                const double apq = a[p][q];
                if (apq == 0.0) continue;
                const double theta = (a[q][q] - a[p][p]) / (2.0 * apq);
                double t = 1.0 / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                if (theta < 0.0) t = -t;
                const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                const int r = 3 - p - q;
                const double app = a[p][p], aqq = a[q][q];
                a[p][p] = app - t * apq;
                a[q][q] = aqq + t * apq;
                a[p][q] = a[q][p] = 0.0;
                const double arp = a[r][p], arq = a[r][q];
                a[r][p] = a[p][r] = c * arp - s * arq;
                a[r][q] = a[q][r] = s * arp + c * arq;
                for (int k = 0; k < 3; k++) { const double vkp = V[k][p], vkq = V[k][q]; V[k][p] = c * vkp - s * vkq; V[k][q] = s * vkp + c * vkq; }
            }
    }
    double d[3] = {a[0][0] * scale, a[1][1] * scale, a[2][2] * scale};
    int idx[3] = {0, 1, 2};
    if (d[idx[1]] < d[idx[0]]) std::swap(idx[0], idx[1]);
    if (d[idx[2]] < d[idx[1]]) std::swap(idx[1], idx[2]);
    if (d[idx[1]] < d[idx[0]]) std::swap(idx[0], idx[1]);
    nb.normal = srl::vec3(V[0][idx[0]], V[1][idx[0]], V[2][idx[0]]).normalized();
    const double sigma_1 = std::sqrt(std::abs(d[idx[2]]));
    const double sigma_2 = std::sqrt(std::abs(d[idx[1]]));
    const double sigma_3 = std::sqrt(std::abs(d[idx[0]]));
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

// ---------------------------------------------------------------- frame-resident optimize() (optimize.cpp:428-448)
optimizeSummary lioOptimization::optimizeResident(cloudFrame *p_frame, const double *frame_raw, int n, const icpOptions &cur_icp_options,
                                                  double sample_voxel_size, std::vector<int> *keypoint_index) {
    srl_ctx *ctx = voxel_map.ctx;
    if (!ctx) throw std::runtime_error("optimize: no HIP context (the product has no CPU path)");
    releaseSweep();
    check(ctx, srl_frame_upload(ctx, frame_raw, n), "srl_frame_upload");
    const Quat &q = p_frame->p_state->rotation;
    const double qv[4] = {q.w, q.x, q.y, q.z};
    std::vector<int32_t> idx((size_t)std::max(n, 1));
    int m = 0;
    check(ctx, srl_frame_select_keypoints(ctx, qv, p_frame->p_state->translation.a, R_imu_lidar.a, t_imu_lidar.a, sample_voxel_size,
                                          idx.data(), &m), "srl_frame_select_keypoints");
    if (keypoint_index) keypoint_index->assign(idx.begin(), idx.begin() + m);
    resident_n = m;
    sweep_pinned = true;
    optimizeSummary s = solveIEKF(cur_icp_options, p_frame);
    releaseSweep();
    return s;
}

int lioOptimization::commitFrame(const state *p_state, double voxel_size, int max_num_points_in_voxel, double min_distance_points,
                                 int min_num_points, double *world_out) {
    srl_ctx *ctx = voxel_map.ctx;
    if (!ctx) throw std::runtime_error("addPointsToMap: no HIP context (the product has no CPU path)");
    const double qv[4] = {p_state->rotation.w, p_state->rotation.x, p_state->rotation.y, p_state->rotation.z};
    int added = 0;
    check(ctx, srl_frame_commit(ctx, qv, p_state->translation.a, R_imu_lidar.a, t_imu_lidar.a, voxel_size, max_num_points_in_voxel,
                                min_distance_points, min_num_points, world_out, &added), "srl_frame_commit");
    return added;
}

}  // namespace srlivo
