// active_perception/frontier_finder.h -- drop-in for the reference header of the same name
// (fuel_planner/active_perception/include/active_perception/frontier_finder.h:25-80).
//
// Behind the same records (Frontier, Viewpoint) and calls: searchFrontiers() runs scan, clustering,
// splitLargeFrontiers and the down-sampling on the device; computeFrontiersToVisit() samples and scores
// the viewpoints there; isFrontierCovered() checks coverage there.  The cost-matrix / tour group
// (updateFrontierCostMatrix, getFullCostMatrix, getPathForTour, setNextFrontier: A* searches through
// ViewNode) is not part of this library -- a maintainer keeps the reference's code for it on top of
// frontiers_ / viewpoints_.
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
class PerceptionUtils;  // active_perception/perception_utils.h (the package's own class, not replaced)

// one sampled viewpoint of a frontier cluster: where to hover, where to look, how many of the cluster's
// down-sampled cells it sees
struct Viewpoint {
  Vector3d pos_;
  double yaw_;
  int visib_num_;
};

// a frontier cluster as the exploration planner consumes it
struct Frontier {
  int id_;
  Vector3d average_, box_min_, box_max_;  // mean and AABB of the cells (voxel centres)
  vector<Vector3d> cells_;                // voxel centres, ascending voxel address (reference: BFS order)
  vector<Vector3d> filtered_cells_;       // VoxelGrid centroids (needs frontier/cluster_size_xy + down_sample)
  vector<Viewpoint> viewpoints_;          // best coverage first (needs the candidate_* / perception_utils params)
  list<vector<Vector3d>> paths_;          // to every frontier of frontiers_, in list order (updateFrontierCostMatrix)
  list<double> costs_;
};

class FrontierFinder {
public:
  FrontierFinder(const shared_ptr<EDTEnvironment>& edt, ros::NodeHandle& node);
  ~FrontierFinder();

  // per plan cycle: find / update the clusters, then sample viewpoints for the new ones
  void searchFrontiers();
  void computeFrontiersToVisit();
  bool isFrontierCovered();

  // cluster queries
  void getFrontiers(vector<vector<Vector3d>>& clusters);
  void getDormantFrontiers(vector<vector<Vector3d>>& clusters);
  void getFrontierBoxes(vector<pair<Vector3d, Vector3d>>& boxes);

  // viewpoint queries: the best viewpoint of every active cluster that is not too close to cur_pos, and
  // the few best of selected clusters
  void getTopViewpointsInfo(const Vector3d& cur_pos, vector<Vector3d>& points, vector<double>& yaws,
                            vector<Vector3d>& averages);
  void getViewpointsInfo(const Vector3d& cur_pos, const vector<int>& ids, const int& view_num,
                         const double& max_decay, vector<vector<Vector3d>>& points,
                         vector<vector<double>>& yaws);
  void wrapYaw(double& yaw);

  // tour planning (TSP input): pairwise costs between the best viewpoints of the active clusters, kept
  // incrementally across searches; the full matrix with the current state in row 0; the stored paths
  // along a tour
  void updateFrontierCostMatrix();
  void getFullCostMatrix(const Vector3d& cur_pos, const Vector3d& cur_vel, const Vector3d cur_yaw,
                         Eigen::MatrixXd& mat);
  void getPathForTour(const Vector3d& pos, const vector<int>& frontier_ids, vector<Vector3d>& path);
  void setNextFrontier(const int& id);

  // camera model for the callers (field-of-view drawing); the device samples viewpoints with its own copy
  // of the same perception_utils/* parameters
  shared_ptr<PerceptionUtils> percep_utils_;

  // additions: clusters found by the last searchFrontiers() and the list positions it removed
  const list<Frontier>& newFrontiers() const { return tmp_frontiers_; }
  const vector<int>& removedIds() const { return removed_ids_; }
  fuelmi_frontier* device() const { return dev_; }  // the C-ABI object behind the finder (like SDFMap::device())

private:
  void pull(int which, list<Frontier>& out, int from = 0);

  fuelmi_frontier* dev_;
  shared_ptr<EDTEnvironment> edt_env_;
  int cluster_min_;
  double resolution_, min_candidate_dist_;
  bool have_viewpoints_;  // frontier/candidate_* and perception_utils/* were all given
  bool order_fallback_logged_ = false;  // the first address-order fallback of reference_order = 2 has been reported
  vector<int> removed_ids_;
  list<Frontier> frontiers_, dormant_frontiers_, tmp_frontiers_;
  // storage of the large cells_ vectors a pull() replaces, kept for the next one: a fresh 3.4 MB vector<Vector3d> is a
  // new mapping whose first touch faults 830 pages -- as long as decoding the cells into it
  vector<vector<Vector3d>> cells_spare_;
  list<Frontier>::iterator first_new_ftr_;  // first cluster appended by the last computeFrontiersToVisit()
  Frontier next_frontier_;
};
}  // namespace fast_planner
#endif
