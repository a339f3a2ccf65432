// NOT the Livox driver: inert stand-in (see ros/ros.h).
#pragma once
#include <ros/ros.h>
namespace livox_ros_driver { struct CustomMsg { typedef std::shared_ptr<const CustomMsg> ConstPtr; std_msgs::Header header; }; }
