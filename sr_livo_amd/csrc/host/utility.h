// utility.h (host mirror) -- numType SO(3)/S^2 helpers the ESIKF update needs
// (include/utility.h:191-331) and AngularDistance(Vector3d) (src/utility.cpp:146-153).
#pragma once
#include "srl_la.h"

#include <cmath>

#define THETA_THRESHOLD 0.0001   // include/utility.h:27

namespace srlivo {

class numType {
public:
    SRL_HD static srl::Mat3 normalizeR(const srl::Mat3 &R_in) {                 // utility.h:194-202
        srl::Quat q = srl::Quat::fromRotationMatrix(R_in);
        q.normalize();
        return q.toRotationMatrix();
    }
    SRL_HD static srl::Mat3 skewSymmetric(const srl::Vec3 &m) {                 // utility.h:204-212
        srl::Mat3 s;
        s(0, 0) = 0;     s(0, 1) = -m[2]; s(0, 2) = m[1];
        s(1, 0) = m[2];  s(1, 1) = 0;     s(1, 2) = -m[0];
        s(2, 0) = -m[1]; s(2, 1) = m[0];  s(2, 2) = 0;
        return s;
    }
    SRL_HD static srl::Mat32 derivativeS2(const srl::Vec3 &g_in) {              // utility.h:214-233
        srl::Vec3 g = g_in;
        g.normalize();
        srl::Mat32 B;
        B(0, 0) = 1.0 - g[0] * g[0] / (1.0 + g[2]);
        B(0, 1) = -g[0] * g[1] / (1.0 + g[2]);
        B(1, 0) = B(0, 1);
        B(1, 1) = 1.0 - g[1] * g[1] / (1.0 + g[2]);
        B(2, 0) = -g[0];
        B(2, 1) = -g[1];
        return B;
    }
    SRL_HD static srl::Vec3 rotationToSo3(const srl::Mat3 &R_in) {              // utility.h:266-278
        const srl::Mat3 R = normalizeR(R_in);
        const double theta = std::acos((R(0, 0) + R(1, 1) + R(2, 2) - 1.0) / 2.0);
        const srl::Vec3 a = srl::vec3(R(2, 1) - R(1, 2), R(0, 2) - R(2, 0), R(1, 0) - R(0, 1));
        if (theta < THETA_THRESHOLD) return a / 2.0;
        return (theta * a) / (2.0 * std::sin(theta));
    }
    SRL_HD static srl::Mat3 so3ToRotation(const srl::Vec3 &so3_in) {            // utility.h:280-297
        const double theta = so3_in.norm();
        if (theta < THETA_THRESHOLD) {
            const srl::Mat3 u_x = skewSymmetric(so3_in);
            return srl::Mat3::Identity() + u_x + (0.5 * u_x) * u_x;
        }
        const srl::Mat3 u_x = skewSymmetric(so3_in.normalized());
        return srl::Mat3::Identity() + std::sin(theta) * u_x + ((1.0 - std::cos(theta)) * u_x) * u_x;
    }
    SRL_HD static srl::Quat so3ToQuat(const srl::Vec3 &so3_in) {                // utility.h:299-324
        const double theta = so3_in.norm();
        if (theta < THETA_THRESHOLD) {
            const srl::Vec3 half_so3 = so3_in / 2.0;
            srl::Quat q(1.0, half_so3[0], half_so3[1], half_so3[2]);
            q.normalize();
            return q;
        }
        const srl::Vec3 u = so3_in.normalized();
        const double s = std::sin(0.5 * theta);
        srl::Quat q(std::cos(0.5 * theta), u[0] * s, u[1] * s, u[2] * s);
        q.normalize();
        return q;
    }
    SRL_HD static srl::Vec3 quatToSo3(const srl::Quat &q_in) { return rotationToSo3(q_in.toRotationMatrix()); }   // utility.h:326-330
};

SRL_HD inline double AngularDistance(const srl::Vec3 &d_so3) {                  // src/utility.cpp:146-153 (degrees)
    const srl::Mat3 d_R = numType::so3ToRotation(d_so3);
    double norm = ((d_R(0, 0) + d_R(1, 1) + d_R(2, 2)) - 1) / 2;
    norm = std::acos(norm) * 180 / M_PI;
    return norm;
}

}  // namespace srlivo
