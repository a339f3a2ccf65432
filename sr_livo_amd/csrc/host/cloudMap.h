// cloudMap.h (host mirror) -- the data contract of the hot path, keeping the reference's type names
// (include/cloudMap.h:37-49 point3D, :97-108 planeParam, :124-145 voxel, :171 voxelHashMap) with
// srl:: fixed-size types standing in for Eigen (absent from this image).  The voxel map itself is
// DEVICE-RESIDENT: voxelHashMap is a handle onto the srl_ctx that owns the HBM slabs + hash table
// (replaces tsl::robin_map<voxel, voxelBlock>, cloudMap.h:171).
#pragma once
#include "../../../include/srlivo_hip.h"
#include "srl_la.h"

#include <vector>

namespace srlivo {

struct point3D {                     // cloudMap.h:37-49
    srl::Vec3 raw_point = srl::Vec3::Zero();
    srl::Vec3 point = srl::Vec3::Zero();
    srl::Vec3 imu_point = srl::Vec3::Zero();
    double alpha_time = 0.0;
    double relative_time = 0.0;
    double timestamp = 0.0;
    int index_frame = -1;
};

struct planeParam {                  // cloudMap.h:97-108
    srl::Vec3 raw_point = srl::Vec3::Zero();
    srl::Vec3 norm_vector = srl::Vec3::Zero();
    srl::Mat<1, 6> jacobians = srl::Mat<1, 6>::Zero();
    double norm_offset = 0.0;
    double distance = 0.0;
    double weight = 1.0;
};

struct voxel {                       // cloudMap.h:124-145
    voxel() = default;
    voxel(short x_, short y_, short z_) : x(x_), y(y_), z(z_) {}
    bool operator==(const voxel &vox) const { return x == vox.x && y == vox.y && z == vox.z; }
    inline bool operator<(const voxel &vox) const {
        return x < vox.x || (x == vox.x && y < vox.y) || (x == vox.x && y == vox.y && z < vox.z);
    }
    short x = 0, y = 0, z = 0;
};

struct imuState {                    // cloudMap.h:110-122
    double timestamp = 0.0;
    srl::Vec3 un_acc = srl::Vec3::Zero();
    srl::Vec3 un_gyr = srl::Vec3::Zero();
    srl::Vec3 trans = srl::Vec3::Zero();
    srl::Quat quat;
    srl::Vec3 vel = srl::Vec3::Zero();
};

// handle onto the device-resident map (slabs + open-addressing table in HBM)
struct voxelHashMap {
    srl_ctx *ctx = nullptr;
};

}  // namespace srlivo

namespace std {
template <> struct hash<srlivo::voxel> {      // cloudMap.h:173-184 (used by gridSampling's host grid only)
    std::size_t operator()(const srlivo::voxel &vox) const {
        const size_t kP1 = 73856093;
        const size_t kP2 = 19349669;
        const size_t kP3 = 83492791;
        return vox.x * kP1 + vox.y * kP2 + vox.z * kP3;
    }
};
}  // namespace std
