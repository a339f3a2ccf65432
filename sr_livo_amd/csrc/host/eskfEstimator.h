// eskfEstimator.h (host mirror) -- class surface of the reference's 17-dim error-state Kalman filter
// (include/eskfEstimator.h:19-112).  The 17-dim algebra stays on the host exactly as in the reference
// (SURVEY.md 8(a) rows a2/a6); only the per-point sums move to the GPU.  observePose / updateAndReset /
// projectCovariance / calculateLxly have no caller anywhere in the reference (dead code) and are not mirrored.
#pragma once
#include "srl_la.h"
#include "utility.h"

#include <utility>
#include <vector>

namespace srlivo {

// src/utility.cpp:11,14 (globals in the reference)
extern bool initial_flag;

// imu_meas element: {time, {gyr, acc}} (lioOptimization.cpp:1453)
typedef std::pair<double, std::pair<srl::Vec3, srl::Vec3>> imuMeas;

class eskfEstimator {
private:
    double dt = 0.0;
    srl::Vec3 acc_0 = srl::Vec3::Zero(), gyr_0 = srl::Vec3::Zero();
    srl::Vec3 acc_1 = srl::Vec3::Zero(), gyr_1 = srl::Vec3::Zero();
    srl::Vec3 acc_cov = srl::Vec3::Zero(), gyr_cov = srl::Vec3::Zero();
    srl::Vec3 acc_cov_scale = srl::Vec3::Zero(), gyr_cov_scale = srl::Vec3::Zero();
    srl::Vec3 b_acc_cov = srl::Vec3::Zero(), b_gyr_cov = srl::Vec3::Zero();

    srl::Vec3 p, v, ba, bg, g;
    srl::Quat q;

    srl::Mat<12, 12> noise;
    srl::Mat17 covariance;

    srl::Vec3 mean_gyr = srl::vec3(0, 0, 0), mean_acc = srl::vec3(0, 0, 9.81);
    bool is_first_imu_meas = true;
    double time_first_imu = 0.0;
    int num_init_meas = 1;

    void initialization(const std::vector<imuMeas> &imu_meas);       // eskfEstimator.cpp:93-118

public:
    eskfEstimator();

    // eskfEstimator.cpp:43-91.  acc_cov / gyr_cov start at zero (uninitialised members upstream, multiplied by
    // (1 - 1.0) on the first sample).  The outcome is in srlivo::initial_flag, like upstream; the return value
    // adds the reason: 1 initialised now, 0 wait, -1 / -2 gyroscope / accelerometer variance above MAX_*_VAR.
    int tryInit(const std::vector<imuMeas> &imu_meas);
    srl::Vec3 getMeanGyr() const { return mean_gyr; }
    srl::Vec3 getMeanAcc() const { return mean_acc; }
    srl::Vec3 getGyrCov() const { return gyr_cov; }
    srl::Vec3 getAccCov() const { return acc_cov; }
    int getNumInitMeas() const { return num_init_meas; }

    void setAccCov(double para);
    void setGyrCov(double para);
    void setBiasAccCov(double para);
    void setBiasGyrCov(double para);
    void initializeNoise();          // private in the reference (eskfEstimator.cpp:120-126); public here for replay drivers
    void useScaleCovAsCov();         // tryInit: gyr_cov = gyr_cov_scale; acc_cov = acc_cov_scale (eskfEstimator.cpp:65-66)
    void scaleInitialCovariance();   // tryInit covariance scaling (eskfEstimator.cpp:74-76)

    void initializeImuData(const srl::Vec3 &acc_0_, const srl::Vec3 &gyr_0_);

    void setTranslation(const srl::Vec3 &p_);
    void setRotation(const srl::Quat &q_);
    void setVelocity(const srl::Vec3 &v_);
    void setBa(const srl::Vec3 &ba_);
    void setBg(const srl::Vec3 &bg_);
    void setGravity(const srl::Vec3 &g_);
    void setCovariance(const srl::Mat17 &covariance_);

    srl::Mat17 getCovariance();
    srl::Vec3 getTranslation();
    srl::Quat getRotation();
    srl::Vec3 getVelocity();
    srl::Vec3 getBa();
    srl::Vec3 getBg();
    srl::Vec3 getGravity();
    srl::Vec3 getLastAcc();
    srl::Vec3 getLastGyr();

    void predict(double dt_, const srl::Vec3 &acc_1_, const srl::Vec3 &gyr_1_);
    void observe(const srl::Vec17 &d_x_);
};

}  // namespace srlivo
