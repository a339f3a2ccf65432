// oracle/ref_shim/ros/ros.h -- NOT ROS.  Inert stand-ins for the ROS names the reference's node file and headers use
// (include/lioOptimization.h:177-330, include/cloudProcessing.h, src/lioOptimization.cpp), just enough surface for those
// files to be compiled where they lie (oracle/Makefile: refpath).  Nothing here does anything: parameters keep their
// defaults, publishers drop their messages.  The scan-matching path never calls into ROS.  Test infrastructure only.
#pragma once
#include <cstdint>
#include <memory>
#include <string>
// what a real ROS / PCL / Boost install brings in transitively and the reference relies on without including it itself
#include <algorithm>
#include <map>
#include <mutex>
#include <queue>
#include <random>
#include <sstream>
#include <type_traits>
#include <unordered_map>
#include <vector>

namespace boost {
// boost::mt19937_64 (src/lioOptimization.cpp:840) and std::mt19937_64 are the same generator by definition
// (same parameters, default seed 5489; C++11 [rand.predef] check value pinned in tests/test_oracle.py)
typedef std::mt19937_64 mt19937_64;
}  // namespace boost

namespace ros {
struct Time {
    double t = 0.0;
    double toSec() const { return t; }
    Time &fromSec(double s) { t = s; return *this; }
    static Time now() { return Time(); }
    static void init() {}
};
struct Rate { explicit Rate(double) {} void sleep() {} };
class Publisher {
public:
    template <class M> void publish(const M &) const {}
};
class Subscriber {};
// a process-wide stand-in for the parameter server: what the test harness puts here is what the reference's own
// readParameters() (src/lioOptimization.cpp:249-349) reads; everything else keeps the default the caller passes
namespace standin {
inline std::map<std::string, std::vector<double>> &num_params() { static std::map<std::string, std::vector<double>> m; return m; }
inline std::map<std::string, std::string> &str_params() { static std::map<std::string, std::string> m; return m; }
inline bool lookup(const std::string &name, std::string &var) {
    auto it = str_params().find(name);
    if (it == str_params().end()) return false;
    var = it->second;
    return true;
}
inline bool lookup(const std::string &name, std::vector<double> &var) {
    auto it = num_params().find(name);
    if (it == num_params().end()) return false;
    var = it->second;
    return true;
}
template <class T> typename std::enable_if<std::is_arithmetic<T>::value, bool>::type lookup(const std::string &name, T &var) {
    auto it = num_params().find(name);
    if (it == num_params().end() || it->second.empty()) return false;
    var = static_cast<T>(it->second[0]);
    return true;
}
}  // namespace standin
class NodeHandle {
public:
    template <class T, class D> bool param(const std::string &name, T &var, const D &def) const {
        if (standin::lookup(name, var)) return true;
        var = T(def);
        return false;
    }
    template <class M> Publisher advertise(const std::string &, int) { return Publisher(); }
    template <class... A> Subscriber subscribe(A &&...) { return Subscriber(); }
    template <class M, class... A> Subscriber subscribe(A &&...) { return Subscriber(); }
};
inline void init(int &, char **, const std::string &) {}
inline bool ok() { return false; }
inline void spinOnce() {}
}  // namespace ros

namespace std_msgs {
struct Header { ros::Time stamp; std::string frame_id; unsigned seq = 0; };
}  // namespace std_msgs

#define ROS_INFO(...) ((void)0)
#define ROS_WARN(...) ((void)0)
#define ROS_ERROR(...) ((void)0)
#define ROS_INFO_STREAM(x) ((void)0)
#define ROS_WARN_STREAM(x) ((void)0)
#define ROS_ERROR_STREAM(x) ((void)0)
