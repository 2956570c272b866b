// nlopt.hpp -- stand-in for NLopt 2.7.1 (not installed).  The solver loop is outside the parity
// contract; optimize() evaluates the objective once so BsplineOptimizer::optimize() still runs.
#ifndef NLOPT_LITE_HPP_
#define NLOPT_LITE_HPP_
#include <vector>
namespace nlopt {
typedef int algorithm;
typedef int result;
typedef double (*vfunc)(const std::vector<double>& x, std::vector<double>& grad, void* data);
class opt {
  vfunc f_ = nullptr; void* d_ = nullptr;
public:
  opt(algorithm, unsigned) {}
  void set_min_objective(vfunc f, void* d) { f_ = f; d_ = d; }
  void set_maxeval(int) {} void set_maxtime(double) {} void set_xtol_rel(double) {}
  void set_lower_bounds(const std::vector<double>&) {} void set_upper_bounds(const std::vector<double>&) {}
  result optimize(std::vector<double>& x, double& fval) { std::vector<double> g(x.size()); fval = f_(x, g, d_); return 1; }
};
}
#endif
