// NOT ROS: declaration-only stand-ins (see ros/ros.h).
#pragma once
namespace geometry_msgs { struct Vector3 {}; struct Quaternion {}; struct PoseStamped {}; }
