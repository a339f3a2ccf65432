// NOT PCL: plain-struct stand-ins for the point types and containers the reference's headers, src/utility.cpp,
// src/cloudMap.cpp and src/lioOptimization.cpp name (none is used on the scan-matching path).  Test infrastructure only.
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>
#define PCL_ADD_POINT4D float x, y, z, data_pad_
#define POINT_CLOUD_REGISTER_POINT_STRUCT(name, fields)
namespace pcl {
struct PointXYZ { float x = 0, y = 0, z = 0; };
struct PointXYZI { float x = 0, y = 0, z = 0, intensity = 0; };
struct PointXYZRGB { float x = 0, y = 0, z = 0; std::uint8_t r = 0, g = 0, b = 0, a = 0; };
struct PointXYZINormal { float x = 0, y = 0, z = 0, intensity = 0, normal_x = 0, normal_y = 0, normal_z = 0, curvature = 0; };
template <class T> struct PointCloud {
    typedef std::shared_ptr<PointCloud<T>> Ptr;
    typedef std::shared_ptr<const PointCloud<T>> ConstPtr;
    std::vector<T> points;
    void clear() { points.clear(); }
    void push_back(const T &p) { points.push_back(p); }
    void resize(std::size_t n) { points.resize(n); }
    std::size_t size() const { return points.size(); }
};
struct PCDWriter { template <class C> int writeBinary(const std::string &, const C &) { return 0; } };
template <class C, class M> void toROSMsg(const C &, M &) {}
namespace io { template <class C> int savePCDFileBinary(const std::string &, const C &) { return 0; } }
}  // namespace pcl
