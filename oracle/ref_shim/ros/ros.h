// oracle/ref_shim/ros/ros.h -- NOT ROS.  Empty stand-ins for the ROS types the reference's headers name in member
// declarations (include/lioOptimization.h:177-330, include/cloudProcessing.h); nothing on the scan-matching path calls
// into ROS.  Test infrastructure: lets src/optimize.cpp & co. be compiled where they lie (oracle/Makefile: refpath).
#pragma once
#include <memory>
#include <string>
#include <cstdint>
// what a real ROS / PCL install brings in transitively and the reference relies on without including it itself
#include <algorithm>
#include <map>
#include <mutex>
#include <queue>
#include <sstream>
#include <unordered_map>
#include <vector>
namespace ros {
struct Time { double t = 0.0; double toSec() const { return t; } static Time now() { return Time(); } };
class NodeHandle {};
class Publisher {};
class Subscriber {};
}  // namespace ros
#define ROS_INFO(...) ((void)0)
#define ROS_WARN(...) ((void)0)
#define ROS_ERROR(...) ((void)0)
#define ROS_INFO_STREAM(x) ((void)0)
#define ROS_WARN_STREAM(x) ((void)0)
#define ROS_ERROR_STREAM(x) ((void)0)
