// NOT ROS: inert stand-ins (see ros/ros.h in this directory tree).
#pragma once
#include <geometry_msgs/Vector3.h>
namespace tf {
struct Quaternion { Quaternion(double, double, double, double) {} };
struct Vector3 { Vector3(double, double, double) {} };
class StampedTransform {
public:
    std::string frame_id_, child_frame_id_;
    ros::Time stamp_;
    void setRotation(const Quaternion &) {}
    void setOrigin(const Vector3 &) {}
};
inline geometry_msgs::Quaternion createQuaternionMsgFromRollPitchYaw(double, double, double) { return geometry_msgs::Quaternion(); }
}  // namespace tf
