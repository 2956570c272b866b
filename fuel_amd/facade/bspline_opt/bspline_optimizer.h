// bspline_opt/bspline_optimizer.h -- drop-in for the reference header of the same name
// (fuel_planner/bspline_opt/include/bspline_opt/bspline_optimizer.h:18-141).
//
// What is behind the same API here: combineCost and every calc*Cost term are one HIP kernel against the
// device ESDF; optimize() is one kernel launch that runs the whole solve (start clamping, bounds,
// best-variable tracking, evaluation cap and xtol_rel as the reference configures NLopt; the iteration
// is a box-projected L-BFGS because NLopt is a third-party library that is absent here -- final costs
// are comparable with NLopt's, iterates are not).
#ifndef _BSPLINE_OPTIMIZER_H_
#define _BSPLINE_OPTIMIZER_H_

#include <Eigen/Eigen>

#include <memory>
#include <vector>

#include <ros/ros.h>

// struct ViewConstraint comes from the visibility module, as in the reference header (:5); in this
// repository's own build the stand-in under facade/standin/ supplies it
#include <active_perception/traj_visibility.h>

#include "fuelmi.h"

using std::shared_ptr;
using std::unique_ptr;
using std::vector;

namespace fast_planner {
class EDTEnvironment;

class BsplineOptimizer {
public:
  typedef unique_ptr<BsplineOptimizer> Ptr;

  // bit masks of the cost terms and the two usual combinations (values: bspline_optimizer.cpp:10-23)
  static const int SMOOTHNESS, DISTANCE, FEASIBILITY, START, END, GUIDE, WAYPOINTS, VIEWCONS, MINTIME;
  static const int GUIDE_PHASE, NORMAL_PHASE;

  BsplineOptimizer() {}
  ~BsplineOptimizer() {}

  // configuration
  void setParam(ros::NodeHandle& node);
  void setEnvironment(const shared_ptr<EDTEnvironment>& env);

  // per-solve inputs
  void setCostFunction(const int& cost_mask);
  void setBoundaryStates(const vector<Eigen::Vector3d>& start, const vector<Eigen::Vector3d>& end);
  void setTimeLowerBound(const double& lower);
  void setGuidePath(const vector<Eigen::Vector3d>& guide);
  void setWaypoints(const vector<Eigen::Vector3d>& points, const vector<int>& indices);
  void setViewConstraint(const ViewConstraint& constraint);
  void enableDynamic(double time_start);

  // the solve: control points (rows) and knot span in, optimised values out
  void optimize(Eigen::MatrixXd& ctrl_pts, double& knot_span, const int& cost_mask, const int& max_num_id,
                const int& max_time_id);
  void optimize();

  Eigen::MatrixXd getControlPoints();
  vector<Eigen::Vector3d> matrixToVectors(const Eigen::MatrixXd& ctrl_pts);

  // addition: one combineCost evaluation (variables in NLopt layout) on the device
  void combineCost(const std::vector<double>& x, std::vector<double>& grad, double& cost);

  // diagnostics the reference exposes as public members
  double comb_time;
  ros::Time time_start_;
  vector<double> vec_cost_, vec_time_;
  void getCostCurve(vector<double>& cost, vector<double>& time) {
    cost = vec_cost_;
    time = vec_time_;
  }

  EIGEN_MAKE_ALIGNED_OPERATOR_NEW

private:
  bool isQuadratic();

  // environment and parameters
  shared_ptr<EDTEnvironment> edt_environment_;
  fuelmi_bspline_cfg cfg_;
  int bspline_degree_, algorithm1_, algorithm2_;
  int max_iteration_num_[4];
  double max_iteration_time_[4];
  bool dynamic_;
  double start_time_;

  // the problem being solved
  Eigen::MatrixXd control_points_;
  double knot_span_, time_lb_, pt_dist_;
  int cost_function_, dim_, order_, point_num_, variable_num_, max_num_id_, max_time_id_;
  bool optimize_time_;
  vector<Eigen::Vector3d> start_state_, end_state_, guide_pts_, waypoints_;
  vector<int> waypt_idx_;
  ViewConstraint view_cons_;

  // result of the last solve
  std::vector<double> best_variable_;
  double min_cost_;
  int iter_num_;
};
}  // namespace fast_planner
#endif
