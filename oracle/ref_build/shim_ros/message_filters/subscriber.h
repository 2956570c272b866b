// message_filters stand-ins (oracle/ref_build only): constructible, never deliver anything.
#ifndef MF_LITE_SUBSCRIBER_H_
#define MF_LITE_SUBSCRIBER_H_
#include <ros/ros.h>
#include <string>
namespace message_filters {
template <typename M> struct Subscriber {
  Subscriber(ros::NodeHandle&, const std::string&, int) {}
};
namespace sync_policies {
template <typename A, typename B> struct ApproximateTime { explicit ApproximateTime(int) {} };
template <typename A, typename B> struct ExactTime { explicit ExactTime(int) {} };
}
template <typename P> struct Synchronizer {
  template <typename A, typename B> Synchronizer(const P&, A&, B&) {}
  template <typename F> void registerCallback(const F&) {}
};
}
#endif
