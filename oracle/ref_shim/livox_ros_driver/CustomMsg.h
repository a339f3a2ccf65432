// NOT the Livox driver: declaration-only stand-in (see ros/ros.h).
#pragma once
#include <memory>
namespace livox_ros_driver { struct CustomMsg { typedef std::shared_ptr<const CustomMsg> ConstPtr; }; }
