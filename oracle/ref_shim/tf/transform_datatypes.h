// NOT ROS: declaration-only stand-ins (see ros/ros.h).
#pragma once
namespace tf { class StampedTransform {}; }
