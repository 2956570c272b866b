// active_perception/graph_node.h -- STAND-IN for this repository's own facade build only.
// In a FUEL workspace this directory is not on the include path and the package's real header
// (fuel_planner/active_perception/include/active_perception/graph_node.h:49-84) is found instead; the
// facade uses nothing of ViewNode but the two static path-cost calls below, which stay the reference's
// own code (A* / straight-line search through the map, graph_node.cpp).
#ifndef _GRAPH_NODE_STANDIN_H_
#define _GRAPH_NODE_STANDIN_H_
#include <Eigen/Eigen>
#include <vector>
namespace fast_planner {
class ViewNode {
public:
  // time to fly p1 -> p2 (path returned) and to turn y1 -> y2, the larger of the two
  static double computeCost(const Eigen::Vector3d& p1, const Eigen::Vector3d& p2, const double& y1, const double& y2,
                            const Eigen::Vector3d& v1, const double& yd1, std::vector<Eigen::Vector3d>& path);
  static double searchPath(const Eigen::Vector3d& p1, const Eigen::Vector3d& p2, std::vector<Eigen::Vector3d>& path);
};
}  // namespace fast_planner
#endif
