// NOT ROS: inert stand-ins (see ros/ros.h).
#pragma once
#include <opencv2/opencv.hpp>
#include <sensor_msgs/PointCloud2.h>
#include <stdexcept>
namespace cv_bridge {
struct CvImage { cv::Mat image; };
typedef std::shared_ptr<CvImage> CvImagePtr;
struct Exception : std::runtime_error { Exception() : std::runtime_error("cv_bridge") {} };
template <class M> CvImagePtr toCvCopy(const M &, const std::string &) { return std::make_shared<CvImage>(); }
}  // namespace cv_bridge
