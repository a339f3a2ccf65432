// NOT PCL: plain-struct stand-ins for the point types the reference's headers and src/utility.cpp / src/cloudMap.cpp
// name (none is used on the scan-matching path).  Test infrastructure only.
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
struct PointXYZRGB { float x = 0, y = 0, z = 0; std::uint8_t r = 0, g = 0, b = 0; };
struct PointXYZINormal { float x = 0, y = 0, z = 0, intensity = 0, normal_x = 0, normal_y = 0, normal_z = 0, curvature = 0; };
template <class T> struct PointCloud {
    typedef std::shared_ptr<PointCloud<T>> Ptr;
    typedef std::shared_ptr<const PointCloud<T>> ConstPtr;
    std::vector<T> points;
};
struct PCDWriter { template <class C> int writeBinary(const std::string &, const C &) { return 0; } };
}  // namespace pcl
