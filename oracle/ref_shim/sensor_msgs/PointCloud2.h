// NOT ROS: declaration-only stand-ins (see ros/ros.h in this directory tree).
#pragma once
#include <memory>
namespace sensor_msgs {
struct PointCloud2 { typedef std::shared_ptr<const PointCloud2> ConstPtr; typedef std::shared_ptr<PointCloud2> Ptr; };
typedef PointCloud2::ConstPtr PointCloud2ConstPtr;
struct Image { typedef std::shared_ptr<const Image> ConstPtr; };
typedef Image::ConstPtr ImageConstPtr;
struct CompressedImage { typedef std::shared_ptr<const CompressedImage> ConstPtr; };
typedef CompressedImage::ConstPtr CompressedImageConstPtr;
}  // namespace sensor_msgs
