// eskfEstimator.cpp (host mirror) -- follows src/eskfEstimator.cpp:3-21 (ctor), :23-41 (noise setters),
// :120-126 (initializeNoise), :134-164 (accessors), :166-217 (predict), :219-230 (observe).
#include "eskfEstimator.h"

#include <cmath>

namespace srlivo {

bool initial_flag = false;       // src/utility.cpp:11
extern double G_norm;            // src/utility.cpp:14 (defined in lioOptimization.cpp)

using srl::Mat3;
using srl::Vec3;

eskfEstimator::eskfEstimator() {
    noise = srl::Mat<12, 12>::Zero();
    covariance = srl::Mat17::Identity();
    p = Vec3::Zero();
    q = srl::Quat::Identity();
    v = Vec3::Zero();
    ba = Vec3::Zero();
    bg = Vec3::Zero();
    g = srl::vec3(0.0, 0.0, 9.81);
}

// include/utility.h:28-31
#define MIN_INI_COUNT 10
#define MIN_INI_TIME 3.0
#define MAX_GYR_VAR 0.5
#define MAX_ACC_VAR 0.6

int eskfEstimator::tryInit(const std::vector<imuMeas> &imu_meas) {
    if (imu_meas.empty()) return initial_flag ? 1 : 0;      // front()/back() of an empty vector upstream: undefined
    initialization(imu_meas);

    if (num_init_meas > MIN_INI_COUNT && imu_meas.back().first - time_first_imu > MIN_INI_TIME) {
        acc_cov = acc_cov * std::pow(G_norm / mean_acc.norm(), 2);
        if (gyr_cov.norm() > MAX_GYR_VAR) return -1;
        if (acc_cov.norm() > MAX_ACC_VAR) return -2;

        initial_flag = true;

        gyr_cov = gyr_cov_scale;
        acc_cov = acc_cov_scale;

        const Vec3 init_bg = mean_gyr;
        const Vec3 init_gravity = (mean_acc / mean_acc.norm()) * G_norm;
        setBg(init_bg);
        setGravity(init_gravity);

        scaleInitialCovariance();
        initializeNoise();
        return 1;
    }
    return 0;
}

void eskfEstimator::initialization(const std::vector<imuMeas> &imu_meas) {
    if (is_first_imu_meas) {
        num_init_meas = 1;
        is_first_imu_meas = false;
        time_first_imu = imu_meas.front().first;
        mean_gyr = imu_meas.front().second.first;
        mean_acc = imu_meas.front().second.second;
    }
    for (const auto &imu : imu_meas) {
        const double N = (double)num_init_meas;
        mean_gyr = mean_gyr + (imu.second.first - mean_gyr) / N;
        mean_acc = mean_acc + (imu.second.second - mean_acc) / N;
        const Vec3 dg = imu.second.first - mean_gyr, da = imu.second.second - mean_acc;
        const double NN = (double)(num_init_meas * num_init_meas);
        gyr_cov = (gyr_cov * (N - 1.0)) / N + (srl::vec3(dg[0] * dg[0], dg[1] * dg[1], dg[2] * dg[2]) * (N - 1.0)) / NN;
        acc_cov = (acc_cov * (N - 1.0)) / N + (srl::vec3(da[0] * da[0], da[1] * da[1], da[2] * da[2]) * (N - 1.0)) / NN;
        num_init_meas++;
    }
    gyr_0 = imu_meas.back().second.first;
    acc_0 = imu_meas.back().second.second;
}

void eskfEstimator::setAccCov(double para) { acc_cov_scale = srl::vec3(para, para, para); }
void eskfEstimator::setGyrCov(double para) { gyr_cov_scale = srl::vec3(para, para, para); }
void eskfEstimator::setBiasAccCov(double para) { b_acc_cov = srl::vec3(para, para, para); }
void eskfEstimator::setBiasGyrCov(double para) { b_gyr_cov = srl::vec3(para, para, para); }

void eskfEstimator::useScaleCovAsCov() { gyr_cov = gyr_cov_scale; acc_cov = acc_cov_scale; }

void eskfEstimator::scaleInitialCovariance() {
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            covariance(9 + i, 9 + j) *= 0.001;
            covariance(12 + i, 12 + j) *= 0.0001;
        }
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 2; j++) covariance(15 + i, 15 + j) *= 0.00001;
}

void eskfEstimator::initializeNoise() {
    noise = srl::Mat<12, 12>::Zero();
    for (int i = 0; i < 3; i++) {
        noise(0 + i, 0 + i) = acc_cov[i];
        noise(3 + i, 3 + i) = gyr_cov[i];
        noise(6 + i, 6 + i) = b_acc_cov[i];
        noise(9 + i, 9 + i) = b_gyr_cov[i];
    }
}

void eskfEstimator::initializeImuData(const Vec3 &acc_0_, const Vec3 &gyr_0_) { acc_0 = acc_0_; gyr_0 = gyr_0_; }

void eskfEstimator::setTranslation(const Vec3 &p_) { p = p_; }
void eskfEstimator::setRotation(const srl::Quat &q_) { q = q_; }
void eskfEstimator::setVelocity(const Vec3 &v_) { v = v_; }
void eskfEstimator::setBa(const Vec3 &ba_) { ba = ba_; }
void eskfEstimator::setBg(const Vec3 &bg_) { bg = bg_; }
void eskfEstimator::setGravity(const Vec3 &g_) { g = g_; }
void eskfEstimator::setCovariance(const srl::Mat17 &covariance_) { covariance = covariance_; }

srl::Mat17 eskfEstimator::getCovariance() { return covariance; }
Vec3 eskfEstimator::getTranslation() { return p; }
srl::Quat eskfEstimator::getRotation() { return q; }
Vec3 eskfEstimator::getVelocity() { return v; }
Vec3 eskfEstimator::getBa() { return ba; }
Vec3 eskfEstimator::getBg() { return bg; }
Vec3 eskfEstimator::getGravity() { return g; }
Vec3 eskfEstimator::getLastAcc() { return acc_0; }
Vec3 eskfEstimator::getLastGyr() { return gyr_0; }

void eskfEstimator::predict(double dt_, const Vec3 &acc_1_, const Vec3 &gyr_1_) {
    dt = dt_;
    acc_1 = acc_1_;
    gyr_1 = gyr_1_;

    const srl::Quat q_before = q;

    const Vec3 un_gyr = 0.5 * (gyr_0 + gyr_1) - bg;
    const Vec3 un_acc = 0.5 * (acc_0 + acc_1) - ba;
    q = q * numType::so3ToQuat(un_gyr * dt);
    p = p + v * dt;
    const Mat3 Rb = q_before.toRotationMatrix();
    v = v + (Rb * un_acc) * dt - g * dt;

    const Mat3 R_omega_x = numType::skewSymmetric(un_gyr);
    const Mat3 R_acc_x = numType::skewSymmetric(un_acc);
    const srl::Mat32 B_x = numType::derivativeS2(g);
    const Mat3 I3 = Mat3::Identity();

    srl::Mat17 F_x = srl::Mat17::Zero();
    F_x.setBlock<3, 3>(0, 0, I3);
    F_x.setBlock<3, 3>(0, 6, I3 * dt);
    F_x.setBlock<3, 3>(3, 3, I3 - R_omega_x * dt);
    F_x.setBlock<3, 3>(3, 12, (-I3) * dt);
    F_x.setBlock<3, 3>(6, 3, ((-Rb) * R_acc_x) * dt);
    F_x.setBlock<3, 3>(6, 6, I3);
    F_x.setBlock<3, 3>(6, 9, (-Rb) * dt);
    const Mat3 Sg = numType::skewSymmetric(g);
    F_x.setBlock<3, 2>(6, 15, (Sg * B_x) * dt);
    F_x.setBlock<3, 3>(9, 9, I3);
    F_x.setBlock<3, 3>(12, 12, I3);
    {
        const double gn = g.norm();
        const double f = -1.0 / (gn * gn);
        const srl::Mat<2, 3> Bt = B_x.transpose();
        // - 1.0 / (g.norm() * g.norm()) * B_x^T * skew(g) * skew(g) * B_x, evaluated left to right
        F_x.setBlock<2, 2>(15, 15, (((f * Bt) * Sg) * Sg) * B_x);
    }

    srl::Mat<17, 12> F_w = srl::Mat<17, 12>::Zero();
    F_w.setBlock<3, 3>(6, 0, (-Rb) * dt);
    F_w.setBlock<3, 3>(3, 3, (-I3) * dt);
    F_w.setBlock<3, 3>(9, 6, (-I3) * dt);
    F_w.setBlock<3, 3>(12, 9, (-I3) * dt);

    covariance = (F_x * covariance) * F_x.transpose() + (F_w * noise) * F_w.transpose();

    acc_0 = acc_1;
    gyr_0 = gyr_1;
}

void eskfEstimator::observe(const srl::Vec17 &d_x_) {
    p = p + srl::vec3(d_x_[0], d_x_[1], d_x_[2]);
    q = (q * numType::so3ToQuat(srl::vec3(d_x_[3], d_x_[4], d_x_[5]))).normalized();
    v = v + srl::vec3(d_x_[6], d_x_[7], d_x_[8]);
    ba = ba + srl::vec3(d_x_[9], d_x_[10], d_x_[11]);
    bg = bg + srl::vec3(d_x_[12], d_x_[13], d_x_[14]);

    const srl::Mat32 B_x = numType::derivativeS2(g);
    srl::Vec2 dg;
    dg[0] = d_x_[15];
    dg[1] = d_x_[16];
    const Vec3 so3_dg = B_x * dg;
    g = numType::so3ToRotation(so3_dg) * g;
}

}  // namespace srlivo
