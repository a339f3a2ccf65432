// NOT ROS: declaration-only stand-ins (see ros/ros.h in this directory tree).
#pragma once
#include <memory>
namespace sensor_msgs {
struct Imu { typedef std::shared_ptr<const Imu> ConstPtr; typedef std::shared_ptr<Imu> Ptr; };
typedef Imu::ConstPtr ImuConstPtr;
}  // namespace sensor_msgs
