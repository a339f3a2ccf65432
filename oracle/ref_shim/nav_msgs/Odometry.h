// NOT ROS: inert stand-ins (see ros/ros.h in this directory tree).
#pragma once
#include <geometry_msgs/Vector3.h>
namespace nav_msgs { struct Odometry { std_msgs::Header header; std::string child_frame_id; geometry_msgs::PoseWithCovariance pose; }; }
