// NOT ROS: declaration-only stand-ins (see ros/ros.h).
#pragma once
namespace nav_msgs { struct Path {}; }
