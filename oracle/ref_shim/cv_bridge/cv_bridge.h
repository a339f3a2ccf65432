// NOT ROS: declaration-only stand-ins (see ros/ros.h).
#pragma once
#include <sensor_msgs/PointCloud2.h>
