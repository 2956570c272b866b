// plan_env/obj_predictor.h -- stand-in: moving-obstacle prediction is out of scope
// (dynamic_environment = 0); EDTEnvironment only needs the two typedefs to compile.
#ifndef OBJ_PREDICTOR_LITE_H_
#define OBJ_PREDICTOR_LITE_H_
#include <Eigen/Eigen>
#include <algorithm>
#include <cmath>
#include <iostream>
#include <list>
#include <memory>
#include <vector>
#include <ros/ros.h>
using std::cout; using std::endl; using std::list; using std::shared_ptr; using std::unique_ptr; using std::vector;
using std::min; using std::max;
namespace fast_planner {
class PolynomialPrediction {
public:
  Eigen::Vector3d evaluateConstVel(double) { return Eigen::Vector3d(0, 0, 0); }
};
typedef shared_ptr<vector<PolynomialPrediction>> ObjPrediction;
typedef shared_ptr<vector<Eigen::Vector3d>> ObjScale;
}
#endif
