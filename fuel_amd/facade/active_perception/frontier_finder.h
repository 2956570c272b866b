// active_perception/frontier_finder.h -- drop-in replacement of the grid part of the reference's
// FrontierFinder (fuel_planner/active_perception/include/active_perception/frontier_finder.h:25-80).
// searchFrontiers() runs the scan + clustering on the GPU (libfuelmi) and fills the same
// Frontier records (cells_, average_, box_min_, box_max_).  Viewpoint sampling, the cost matrix and
// splitLargeFrontiers are SURVEY 8(f) "next" rows: computeFrontiersToVisit() here promotes every
// new cluster to frontiers_ without sampling viewpoints, and the viewpoint/tour queries are not
// provided by this header.
#ifndef _FRONTIER_FINDER_H_
#define _FRONTIER_FINDER_H_

#include <ros/ros.h>
#include <Eigen/Eigen>
#include <list>
#include <memory>
#include <utility>
#include <vector>

#include "fuelmi.h"

using Eigen::Vector3d;
using std::list;
using std::pair;
using std::shared_ptr;
using std::unique_ptr;
using std::vector;

namespace fast_planner {
class EDTEnvironment;

struct Viewpoint {
  Vector3d pos_;
  double yaw_;
  int visib_num_;
};

struct Frontier {
  vector<Vector3d> cells_;           // voxel centres, ascending voxel address (reference: BFS order)
  vector<Vector3d> filtered_cells_;  // VoxelGrid centroids (filled when frontier/cluster_size_xy and down_sample are set)
  Vector3d average_;
  int id_;
  vector<Viewpoint> viewpoints_;
  Vector3d box_min_, box_max_;
  list<vector<Vector3d>> paths_;
  list<double> costs_;
};

class FrontierFinder {
public:
  FrontierFinder(const shared_ptr<EDTEnvironment>& edt, ros::NodeHandle& nh);
  ~FrontierFinder();

  void searchFrontiers();
  void computeFrontiersToVisit();

  void getFrontiers(vector<vector<Vector3d>>& clusters);
  void getDormantFrontiers(vector<vector<Vector3d>>& clusters);
  void getFrontierBoxes(vector<pair<Vector3d, Vector3d>>& boxes);
  // Get viewpoint with highest coverage for each frontier
  void getTopViewpointsInfo(const Vector3d& cur_pos, vector<Vector3d>& points, vector<double>& yaws,
                            vector<Vector3d>& averages);
  // Get several viewpoints for a subset of frontiers
  void getViewpointsInfo(const Vector3d& cur_pos, const vector<int>& ids, const int& view_num,
                         const double& max_decay, vector<vector<Vector3d>>& points,
                         vector<vector<double>>& yaws);
  bool isFrontierCovered();
  void wrapYaw(double& yaw);

  // additions: clusters found by the last searchFrontiers() and ids removed by it
  const list<Frontier>& newFrontiers() const { return tmp_frontiers_; }
  const vector<int>& removedIds() const { return removed_ids_; }

private:
  void pull(int which, list<Frontier>& out);

  shared_ptr<EDTEnvironment> edt_env_;
  fuelmi_frontier* dev_;
  list<Frontier> frontiers_, dormant_frontiers_, tmp_frontiers_;
  vector<int> removed_ids_;
  int cluster_min_;
  double resolution_;
  double min_candidate_dist_;
  bool have_viewpoints_;  // frontier/candidate_* and perception_utils/* were all given
};
}  // namespace fast_planner
#endif
