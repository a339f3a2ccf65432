// NOT ROS: inert stand-ins (see ros/ros.h in this directory tree).
#pragma once
#include <ros/ros.h>
namespace sensor_msgs {
struct PointCloud2 { typedef std::shared_ptr<const PointCloud2> ConstPtr; typedef std::shared_ptr<PointCloud2> Ptr; std_msgs::Header header; };
typedef PointCloud2::ConstPtr PointCloud2ConstPtr;
struct Image { typedef std::shared_ptr<const Image> ConstPtr; std_msgs::Header header; };
typedef Image::ConstPtr ImageConstPtr;
struct CompressedImage { typedef std::shared_ptr<const CompressedImage> ConstPtr; std_msgs::Header header; std::string format; };
typedef CompressedImage::ConstPtr CompressedImageConstPtr;
namespace image_encodings { const std::string BGR8 = "bgr8"; }
}  // namespace sensor_msgs
