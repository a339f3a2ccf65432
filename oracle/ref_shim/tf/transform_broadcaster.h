// NOT ROS: inert stand-ins (see ros/ros.h in this directory tree).
#pragma once
#include <tf/transform_datatypes.h>
namespace tf { class TransformBroadcaster { public: void sendTransform(const StampedTransform &) {} }; }
