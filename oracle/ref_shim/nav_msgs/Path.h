// NOT ROS: inert stand-ins (see ros/ros.h in this directory tree).
#pragma once
#include <geometry_msgs/Vector3.h>
namespace nav_msgs { struct Path { std_msgs::Header header; std::vector<geometry_msgs::PoseStamped> poses; }; }
