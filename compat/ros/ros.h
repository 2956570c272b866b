// ros/ros.h -- minimal stand-in (oracle/ref_build only): a parameter map and a clock.
#ifndef ROS_LITE_H_
#define ROS_LITE_H_
#include <chrono>
#include <iostream>
#include <cstdio>
#include <map>
#include <string>
namespace ros {
struct Duration { double s; Duration(double s_ = 0) : s(s_) {} double toSec() const { return s; } };
struct Time {
  double t = 0;
  static Time now() { Time r; r.t = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); return r; }
  double toSec() const { return t; }
};
inline Duration operator-(const Time& a, const Time& b) { return Duration{a.t - b.t}; }
inline bool operator>(const Time& a, const Time& b) { return a.t > b.t; }
inline bool operator<(const Time& a, const Time& b) { return a.t < b.t; }
inline bool ok() { return true; }
struct TimerEvent { Time current_real, last_real; };
struct Timer {};
struct Publisher { template <typename M> void publish(const M&) const {} };
class NodeHandle {
public:
  template <typename M> Publisher advertise(const std::string&, int) { return Publisher(); }
  template <typename F, typename O> Timer createTimer(Duration, F, O*) { return Timer(); }
  std::map<std::string, double> num;
  std::map<std::string, std::string> str;
  template <typename T> bool param(const std::string& k, T& v, const T& def) const {
    auto it = num.find(k);
    if (it == num.end()) { v = def; return false; }
    v = (T)it->second; return true;
  }
  bool param(const std::string& k, std::string& v, const std::string& def) const {
    auto it = str.find(k);
    if (it == str.end()) { v = def; return false; }
    v = it->second; return true;
  }
};
}  // namespace ros
#define ROS_ERROR(...) do { std::fprintf(stderr, __VA_ARGS__); std::fprintf(stderr, "\n"); } while (0)
#define ROS_WARN(...) do { } while (0)
#define ROS_INFO(...) do { } while (0)
#define ROS_ERROR_COND(c, ...) do { } while (0)
#define ROS_WARN_THROTTLE(p, ...) do { } while (0)
#define ROS_INFO_STREAM(x) do { } while (0)
#define ROS_WARN_STREAM(x) do { } while (0)
#define ROS_ERROR_STREAM(x) do { } while (0)
#endif
