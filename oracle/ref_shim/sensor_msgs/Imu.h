// NOT ROS: inert stand-ins (see ros/ros.h in this directory tree).
#pragma once
#include <ros/ros.h>
namespace geometry_msgs { struct Vector3 { double x = 0, y = 0, z = 0; }; }
namespace sensor_msgs {
struct Imu {
    typedef std::shared_ptr<const Imu> ConstPtr; typedef std::shared_ptr<Imu> Ptr;
    std_msgs::Header header;
    geometry_msgs::Vector3 linear_acceleration, angular_velocity;
};
typedef Imu::ConstPtr ImuConstPtr;
}  // namespace sensor_msgs
