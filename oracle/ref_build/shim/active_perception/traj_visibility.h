// active_perception/traj_visibility.h -- stand-in carrying only struct ViewConstraint
// (reference: active_perception/include/active_perception/traj_visibility.h:18-24); VisibilityUtil
// itself is off the exploration path.
#ifndef TRAJ_VISIBILITY_LITE_H_
#define TRAJ_VISIBILITY_LITE_H_
#include <plan_env/edt_environment.h>
#include <string>
using std::string;
namespace fast_planner {
struct ViewConstraint {
  Eigen::Vector3d pt_;
  Eigen::Vector3d pc_;
  Eigen::Vector3d dir_;
  Eigen::Vector3d pcons_;
  int idx_;
};
}
#endif
