// bspline/Bspline.h -- stand-in (tests/dropin only) for the generated ROS message the callers' headers name
#ifndef DROPIN_BSPLINE_MSG_H_
#define DROPIN_BSPLINE_MSG_H_
#include <ros/ros.h>
#include <geometry_msgs/PoseStamped.h>
#include <vector>
namespace bspline {
struct Bspline {
  int order = 0;
  long traj_id = 0;
  ros::Time start_time;
  std::vector<double> knots;
  std::vector<geometry_msgs::Point> pos_pts;
  std::vector<double> yaw_pts;
  double yaw_dt = 0;
};
}  // namespace bspline
#endif
