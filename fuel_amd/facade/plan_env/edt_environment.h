// plan_env/edt_environment.h -- drop-in replacement of the reference header
// (fuel_planner/plan_env/include/plan_env/edt_environment.h:16-44): distance/gradient query facade
// over SDFMap.  Moving-obstacle prediction (setObjPrediction/setObjScale) is out of scope
// (exploration runs with dynamic_environment = 0); the setters are kept as no-ops.
#ifndef _EDT_ENVIRONMENT_H_
#define _EDT_ENVIRONMENT_H_

#include <Eigen/Eigen>
#include <iostream>
#include <list>
#include <memory>
#include <utility>
#include <vector>

using std::cout;
using std::endl;
using std::list;
using std::pair;
using std::shared_ptr;
using std::unique_ptr;
using std::vector;

namespace fast_planner {
class SDFMap;
class PolynomialPrediction;
typedef shared_ptr<vector<PolynomialPrediction>> ObjPrediction;
typedef shared_ptr<vector<Eigen::Vector3d>> ObjScale;

class EDTEnvironment {
private:
  ObjPrediction obj_prediction_;
  ObjScale obj_scale_;
  double resolution_inv_;

public:
  EDTEnvironment() {}
  ~EDTEnvironment() {}

  shared_ptr<SDFMap> sdf_map_;

  void init();
  void setMap(shared_ptr<SDFMap>& map);
  void setObjPrediction(ObjPrediction prediction);
  void setObjScale(ObjScale scale);
  void evaluateEDTWithGrad(const Eigen::Vector3d& pos, double time, double& dist, Eigen::Vector3d& grad);
  double evaluateCoarseEDT(Eigen::Vector3d& pos, double time);

  typedef shared_ptr<EDTEnvironment> Ptr;
};
}  // namespace fast_planner
#endif
