// NOT ROS: declaration-only stand-ins (see ros/ros.h).
#pragma once
#include <tf/transform_datatypes.h>
namespace tf { class TransformBroadcaster {}; }
