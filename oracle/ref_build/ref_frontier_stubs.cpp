// ref_frontier_stubs.cpp -- definitions for the out-of-scope collaborator that frontier_finder.cpp
// links against (ViewNode: A*/yaw path costs for the TSP cost matrix).  It is not reached by
// searchFrontiers / splitLargeFrontiers / computeFrontiersToVisit / sampleViewpoints /
// isFrontierCovered, the parts oracle/_ref exists to pin.  PerceptionUtils is the REAL
// active_perception/src/perception_utils.cpp.
#include <cmath>
#include <active_perception/graph_node.h>
#include <active_perception/perception_utils.h>
namespace fast_planner {
double ViewNode::vm_ = 0, ViewNode::am_ = 0, ViewNode::yd_ = 0, ViewNode::ydd_ = 0, ViewNode::w_dir_ = 0;
shared_ptr<Astar> ViewNode::astar_;
shared_ptr<RayCaster> ViewNode::caster_;
shared_ptr<SDFMap> ViewNode::map_;
ViewNode::ViewNode(const Vector3d& p, const double& y) { pos_ = p; yaw_ = y; }
double ViewNode::costTo(const ViewNode::Ptr&) { return 0; }
// deterministic stand-ins (straight flight + yaw term, path = the two end points): enough to pin the
// BOOKKEEPING of updateFrontierCostMatrix / getFullCostMatrix / getPathForTour, which is what the facade
// re-implements; the A* behind the real functions is not replaced by anything in this repository
double ViewNode::computeCost(const Vector3d& p1, const Vector3d& p2, const double& y1, const double& y2, const Vector3d&,
                             const double&, vector<Vector3d>& path) {
  path = {p1, p2};
  return (p2 - p1).norm() + 0.1 * std::fabs(y2 - y1);
}
double ViewNode::searchPath(const Vector3d& p1, const Vector3d& p2, vector<Vector3d>& path) {
  path = {p1, p2};
  return (p2 - p1).norm();
}
}
