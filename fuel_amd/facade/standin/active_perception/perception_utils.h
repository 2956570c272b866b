// active_perception/perception_utils.h -- STAND-IN for this repository's own facade build only (see
// graph_node.h beside it).  FrontierFinder owns a PerceptionUtils for its callers
// (fast_exploration_manager.cpp:172-173 draws the field of view through frontier_finder_->percep_utils_);
// the class itself (active_perception/src/perception_utils.cpp) stays the reference's own code.  The
// facade only constructs it.
#ifndef _PERCEPTION_UTILS_STANDIN_H_
#define _PERCEPTION_UTILS_STANDIN_H_
#include <ros/ros.h>
namespace fast_planner {
class PerceptionUtils {
public:
  PerceptionUtils(ros::NodeHandle& nh);
};
}  // namespace fast_planner
#endif
