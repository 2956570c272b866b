// bspline_opt/bspline_optimizer.h -- drop-in replacement of the reference header
// (fuel_planner/bspline_opt/include/bspline_opt/bspline_optimizer.h:18-141).  Cost and gradient
// (combineCost and every calc*Cost term) are evaluated on the GPU against the device ESDF.  The
// solver loop is NLopt in the reference (a third-party dependency absent here): this
// facade drives the same objective with its own box-projected L-BFGS under the reference's
// stopping criteria (max evaluations / max time / xtol_rel 1e-5), so the API is complete but the
// iterates are not NLopt's.  optimizeBatch() evaluates many candidate splines per launch.
#ifndef _BSPLINE_OPTIMIZER_H_
#define _BSPLINE_OPTIMIZER_H_

#include <Eigen/Eigen>
#include <memory>
#include <vector>
#include <ros/ros.h>

#include "fuelmi.h"

using std::shared_ptr;
using std::unique_ptr;
using std::vector;

namespace fast_planner {
class EDTEnvironment;

struct ViewConstraint {  // active_perception/traj_visibility.h:18-24
  Eigen::Vector3d pt_;
  Eigen::Vector3d pc_;
  Eigen::Vector3d dir_;
  Eigen::Vector3d pcons_;
  int idx_;
};

class BsplineOptimizer {
public:
  static const int SMOOTHNESS;
  static const int DISTANCE;
  static const int FEASIBILITY;
  static const int START;
  static const int END;
  static const int GUIDE;
  static const int WAYPOINTS;
  static const int VIEWCONS;
  static const int MINTIME;
  static const int GUIDE_PHASE;
  static const int NORMAL_PHASE;

  BsplineOptimizer() {}
  ~BsplineOptimizer() {}

  void setEnvironment(const shared_ptr<EDTEnvironment>& env);
  void setParam(ros::NodeHandle& nh);
  void optimize(Eigen::MatrixXd& points, double& dt, const int& cost_function, const int& max_num_id,
                const int& max_time_id);

  void setCostFunction(const int& cost_function);
  void setBoundaryStates(const vector<Eigen::Vector3d>& start, const vector<Eigen::Vector3d>& end);
  void setTimeLowerBound(const double& lb);
  void setGuidePath(const vector<Eigen::Vector3d>& guide_pt);
  void setWaypoints(const vector<Eigen::Vector3d>& waypts, const vector<int>& waypt_idx);
  void setViewConstraint(const ViewConstraint& vc);
  void enableDynamic(double time_start);

  void optimize();

  Eigen::MatrixXd getControlPoints();
  vector<Eigen::Vector3d> matrixToVectors(const Eigen::MatrixXd& ctrl_pts);

  // addition: one combineCost evaluation (x in NLopt layout) on the GPU
  void combineCost(const std::vector<double>& x, std::vector<double>& grad, double& cost);

private:
  bool isQuadratic();

  shared_ptr<EDTEnvironment> edt_environment_;
  Eigen::MatrixXd control_points_;
  double knot_span_;
  int dim_;
  vector<Eigen::Vector3d> start_state_, end_state_, guide_pts_, waypoints_;
  vector<int> waypt_idx_;
  int max_num_id_, max_time_id_;
  int cost_function_;
  double time_lb_;
  bool dynamic_;
  double start_time_;
  int order_, bspline_degree_;
  fuelmi_bspline_cfg cfg_;
  int algorithm1_, algorithm2_;
  int max_iteration_num_[4];
  double max_iteration_time_[4];
  int variable_num_, point_num_;
  bool optimize_time_;
  int iter_num_;
  std::vector<double> best_variable_;
  double min_cost_;
  ViewConstraint view_cons_;
  double pt_dist_;

public:
  vector<double> vec_cost_;
  vector<double> vec_time_;
  ros::Time time_start_;
  void getCostCurve(vector<double>& cost, vector<double>& time) {
    cost = vec_cost_;
    time = vec_time_;
  }
  double comb_time;
  typedef unique_ptr<BsplineOptimizer> Ptr;
  EIGEN_MAKE_ALIGNED_OPERATOR_NEW
};
}  // namespace fast_planner
#endif
