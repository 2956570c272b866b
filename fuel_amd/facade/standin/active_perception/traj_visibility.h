// active_perception/traj_visibility.h -- STAND-IN for this repository's own facade build only (see
// graph_node.h beside it).  BsplineOptimizer's header takes struct ViewConstraint from the visibility
// module like the reference's does (bspline_optimizer.h:5); in a FUEL workspace the package's real header
// (active_perception/include/active_perception/traj_visibility.h:18-24) is found instead of this file.
#ifndef _TRAJ_VISIBILITY_STANDIN_H_
#define _TRAJ_VISIBILITY_STANDIN_H_
#include <Eigen/Eigen>
namespace fast_planner {
struct ViewConstraint {
  Eigen::Vector3d pt_, pc_, dir_, pcons_;
  int idx_;
};
}  // namespace fast_planner
#endif
