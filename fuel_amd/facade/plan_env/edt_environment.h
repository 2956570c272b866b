// plan_env/edt_environment.h -- drop-in for the reference header of the same name
// (fuel_planner/plan_env/include/plan_env/edt_environment.h:16-44).  EDTEnvironment is the thin query
// object the planners hold: it forwards distance / gradient look-ups to the SDFMap, which answers them
// on the device.  Moving-obstacle prediction is not part of the exploration path
// (dynamic_environment = 0): setObjPrediction / setObjScale are accepted and ignored.
#ifndef _EDT_ENVIRONMENT_H_
#define _EDT_ENVIRONMENT_H_

#include <Eigen/Eigen>

#include <iostream>
#include <list>
#include <memory>
#include <utility>
#include <vector>

// the reference header exports these names into the global namespace; callers rely on it
using std::cout;
using std::endl;
using std::list;
using std::pair;
using std::shared_ptr;
using std::unique_ptr;
using std::vector;

namespace fast_planner {
class PolynomialPrediction;
class SDFMap;
typedef shared_ptr<vector<Eigen::Vector3d>> ObjScale;
typedef shared_ptr<vector<PolynomialPrediction>> ObjPrediction;

class EDTEnvironment {
public:
  typedef shared_ptr<EDTEnvironment> Ptr;

  // the map is a public member in the reference and is reached directly by FrontierFinder, A*, ...
  shared_ptr<SDFMap> sdf_map_;

  EDTEnvironment() {}
  ~EDTEnvironment() {}
  void init();
  void setMap(shared_ptr<SDFMap>& map);

  // trilinear distance + gradient at `where` (the time argument only matters with moving obstacles)
  void evaluateEDTWithGrad(const Eigen::Vector3d& where, double time, double& distance, Eigen::Vector3d& gradient);
  // nearest-voxel distance
  double evaluateCoarseEDT(Eigen::Vector3d& where, double time);

  void setObjScale(ObjScale scale);
  void setObjPrediction(ObjPrediction prediction);

private:
  double resolution_inv_;
  ObjScale obj_scale_;
  ObjPrediction obj_prediction_;
};
}  // namespace fast_planner
#endif
