// NOT ROS: inert stand-ins (see ros/ros.h in this directory tree).
#pragma once
#include <sensor_msgs/Imu.h>
namespace geometry_msgs {
struct Quaternion { double x = 0, y = 0, z = 0, w = 1; };
struct Point { double x = 0, y = 0, z = 0; };
struct Pose { Point position; Quaternion orientation; };
struct PoseStamped { std_msgs::Header header; Pose pose; };
struct PoseWithCovariance { Pose pose; };
}  // namespace geometry_msgs
