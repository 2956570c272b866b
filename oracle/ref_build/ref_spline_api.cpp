// ref_spline_api.cpp -- C entry points into the REAL NonUniformBspline (bspline/src/non_uniform_bspline.cpp,
// compiled unmodified with the Eigen stand-in): parameterizeToBspline and getBoundaryStates.  Test
// infrastructure only (pins oracle/fuel_oracle.cpp's restatement of the same two functions).
#include <vector>

#include "bspline/non_uniform_bspline.h"

using fast_planner::NonUniformBspline;

extern "C" {
// ctrl: (K + degree - 1) x 3, row-major
void ref_spline_parameterize(double ts, const double* pts, int K, const double* derivs4, int degree, double* ctrl) {
  std::vector<Eigen::Vector3d> point_set, der;
  for (int i = 0; i < K; ++i) point_set.push_back(Eigen::Vector3d(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]));
  for (int i = 0; i < 4; ++i) der.push_back(Eigen::Vector3d(derivs4[3 * i], derivs4[3 * i + 1], derivs4[3 * i + 2]));
  Eigen::MatrixXd cp;
  NonUniformBspline::parameterizeToBspline(ts, point_set, der, degree, cp);
  for (int i = 0; i < cp.rows(); ++i)
    for (int j = 0; j < 3; ++j) ctrl[3 * i + j] = cp(i, j);
}
// start: (ks + 1) x 3, end: (ke + 1) x 3
void ref_spline_boundary_states(const double* ctrl, int n, int degree, double ts, int ks, int ke, double* start,
                                double* end) {
  Eigen::MatrixXd cp(n, 3);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < 3; ++j) cp(i, j) = ctrl[3 * i + j];
  NonUniformBspline sp(cp, degree, ts);
  std::vector<Eigen::Vector3d> s, e;
  sp.getBoundaryStates(ks, ke, s, e);
  for (size_t i = 0; i < s.size(); ++i)
    for (int j = 0; j < 3; ++j) start[3 * i + j] = s[i](j);
  for (size_t i = 0; i < e.size(); ++i)
    for (int j = 0; j < 3; ++j) end[3 * i + j] = e[i](j);
}
// positions (deriv = 0) or the deriv-th derivative at nt times in [0, duration]
void ref_spline_evaluate(const double* ctrl, int n, int degree, double ts, int deriv, const double* t, int nt,
                         double* out) {
  Eigen::MatrixXd cp(n, 3);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < 3; ++j) cp(i, j) = ctrl[3 * i + j];
  NonUniformBspline sp(cp, degree, ts);
  for (int d = 0; d < deriv; ++d) sp = sp.getDerivative();
  for (int k = 0; k < nt; ++k) {
    Eigen::VectorXd p = sp.evaluateDeBoorT(t[k]);
    for (int j = 0; j < 3; ++j) out[3 * k + j] = p(j);
  }
}
double ref_spline_duration(int n, int degree, double ts) {
  Eigen::MatrixXd cp(n, 3);
  NonUniformBspline sp(cp, degree, ts);
  return sp.getTimeSum();
}
}
