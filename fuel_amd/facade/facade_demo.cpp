// facade_demo.cpp -- drives the facade exactly the way the reference's callers drive the original
// classes (MapROS::depthPoseCallback -> inputPointCloud -> clearAndInflateLocalMap;
// updateESDFCallback -> updateESDF3d; planExploreMotion -> searchFrontiers;
// planExploreTraj -> BsplineOptimizer::optimize) and dumps results for tests/test_facade_gpu.py.
//   facade_demo <scenario.bin> <result.bin>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include <plan_env/sdf_map.h>
#include <plan_env/edt_environment.h>
#include <active_perception/frontier_finder.h>
#include <bspline_opt/bspline_optimizer.h>
#include <active_perception/graph_node.h>
#include <active_perception/perception_utils.h>

namespace fast_planner {
// ViewNode belongs to the part of active_perception the facade does not replace (graph_node.cpp: A*
// through the map).  For this demo: straight flight plus a yaw term, the path is its two end points
// (tests/test_facade_gpu.py rebuilds the expected cost matrix with the same formula).
double ViewNode::computeCost(const Eigen::Vector3d& p1, const Eigen::Vector3d& p2, const double& y1, const double& y2,
                             const Eigen::Vector3d&, const double&, std::vector<Eigen::Vector3d>& path) {
  path = {p1, p2};
  return (p2 - p1).norm() + 0.1 * std::fabs(y2 - y1);
}
PerceptionUtils::PerceptionUtils(ros::NodeHandle&) {}  // (the package's perception_utils.cpp in a FUEL workspace)
double ViewNode::searchPath(const Eigen::Vector3d& p1, const Eigen::Vector3d& p2, std::vector<Eigen::Vector3d>& path) {
  path = {p1, p2};
  return (p2 - p1).norm();
}
// the reference's MapROS is a friend of SDFMap and calls its private clearAndInflateLocalMap
// (plan_env/src/map_ros.cpp:142,170); this stand-in does the same
class MapROS {
public:
  static void inflate(SDFMap& m) { m.clearAndInflateLocalMap(); }
};
}  // namespace fast_planner

using namespace fast_planner;

template <typename T>
static void rd(FILE* f, T* p, size_t n) {
  if (fread(p, sizeof(T), n, f) != n) {
    std::fprintf(stderr, "short read\n");
    std::exit(2);
  }
}
template <typename T>
static void wr(FILE* f, const T* p, size_t n) { fwrite(p, sizeof(T), n, f); }

int main(int argc, char** argv) {
  if (argc < 3) return 1;
  FILE* in = fopen(argv[1], "rb");
  FILE* out = fopen(argv[2], "wb");
  if (!in || !out) return 1;
  double hdr[10];  // map_size[3], box_min[3], box_max[3], cluster_min
  rd(in, hdr, 10);
  ros::NodeHandle nh;
  auto& P = nh.num;
  P["sdf_map/resolution"] = 0.1;
  P["sdf_map/map_size_x"] = hdr[0], P["sdf_map/map_size_y"] = hdr[1], P["sdf_map/map_size_z"] = hdr[2];
  P["sdf_map/obstacles_inflation"] = 0.199, P["sdf_map/local_bound_inflate"] = 0.5, P["sdf_map/ground_height"] = -1.0;
  P["sdf_map/default_dist"] = 0.0, P["sdf_map/optimistic"] = 0, P["sdf_map/signed_dist"] = 0;
  P["sdf_map/p_hit"] = 0.65, P["sdf_map/p_miss"] = 0.35, P["sdf_map/p_min"] = 0.12, P["sdf_map/p_max"] = 0.90;
  P["sdf_map/p_occ"] = 0.80, P["sdf_map/max_ray_length"] = 4.5, P["sdf_map/virtual_ceil_height"] = -10;
  const char* ax[3] = {"x", "y", "z"};
  for (int i = 0; i < 3; ++i) {
    P[std::string("sdf_map/box_min_") + ax[i]] = hdr[3 + i];
    P[std::string("sdf_map/box_max_") + ax[i]] = hdr[6 + i];
  }
  P["frontier/cluster_min"] = hdr[9];
  P["optimization/ld_smooth"] = 20.0, P["optimization/ld_dist"] = 10.0, P["optimization/ld_feasi"] = 2.0;
  P["optimization/ld_start"] = 100.0, P["optimization/ld_end"] = 0.5, P["optimization/ld_guide"] = 1.5;
  P["optimization/ld_waypt"] = 0.3, P["optimization/ld_view"] = 0.0, P["optimization/ld_time"] = 1.0;
  P["optimization/dist0"] = 0.7, P["optimization/max_vel"] = 2.0, P["optimization/max_acc"] = 2.0;
  P["optimization/max_iteration_num2"] = 300, P["optimization/max_iteration_time2"] = 5.0;
  P["optimization/algorithm1"] = 15, P["optimization/algorithm2"] = 11;

  SDFMap::Ptr map(new SDFMap);
  map->initMap(nh);
  EDTEnvironment::Ptr edt(new EDTEnvironment);
  edt->setMap(map);

  int n_frames;
  rd(in, &n_frames, 1);
  for (int k = 0; k < n_frames; ++k) {
    int n;
    double cam[3];
    rd(in, &n, 1);
    rd(in, cam, 3);
    std::vector<float> xyz((size_t)n * 3);
    rd(in, xyz.data(), xyz.size());
    pcl::PointCloud<pcl::PointXYZ> cloud;
    cloud.points.resize(n);
    for (int i = 0; i < n; ++i) cloud.points[i] = pcl::PointXYZ(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
    map->inputPointCloud(cloud, n, Eigen::Vector3d(cam[0], cam[1], cam[2]));
    MapROS::inflate(*map);
    map->updateESDF3d();
  }
  // host-side inline getters (compiled into THIS translation unit, reading the host mirrors)
  Eigen::Vector3d ori, size;
  map->getRegion(ori, size);
  const int N = map->getVoxelNum();
  Eigen::Vector3i id;
  Eigen::Vector3d far(ori(0) + size(0) - 0.05, ori(1) + size(1) - 0.05, ori(2) + size(2) - 0.05);
  map->posToIndex(far, id);
  const int nx = id(0) + 1, ny = id(1) + 1, nz = id(2) + 1;
  std::vector<signed char> occ(N), infl(N);
  std::vector<double> dist(N);
  for (int x = 0; x < nx; ++x)
    for (int y = 0; y < ny; ++y)
      for (int z = 0; z < nz; ++z) {
        Eigen::Vector3i v(x, y, z);
        const int a = map->toAddress(v);
        occ[a] = (signed char)map->getOccupancy(v);
        infl[a] = (signed char)map->getInflateOccupancy(v);
        dist[a] = map->getDistance(v);
      }
  wr(out, &N, 1);
  wr(out, occ.data(), N);
  wr(out, infl.data(), N);
  wr(out, dist.data(), N);

  Eigen::Vector3d ub_min, ub_max;
  map->getUpdatedBox(ub_min, ub_max, false);  // kept for the second finder below
  FrontierFinder ff(edt, nh);
  ff.searchFrontiers();
  ff.computeFrontiersToVisit();
  std::vector<std::vector<Eigen::Vector3d>> clusters;
  ff.getFrontiers(clusters);
  int nc = (int)clusters.size();
  wr(out, &nc, 1);
  for (auto& c : clusters) {
    int sz = (int)c.size();
    wr(out, &sz, 1);
    for (auto& p : c) {
      double q[3] = {p(0), p(1), p(2)};
      wr(out, q, 3);
    }
  }

  // trajectory optimisation as planExploreTraj does (planner_manager.cpp:304-312)
  int npts;
  double dt;
  rd(in, &npts, 1);
  rd(in, &dt, 1);
  Eigen::MatrixXd ctrl(npts, 3);
  for (int i = 0; i < npts; ++i) {
    double p[3];
    rd(in, p, 3);
    for (int j = 0; j < 3; ++j) ctrl(i, j) = p[j];
  }
  double st[9], en[9];
  rd(in, st, 9);
  rd(in, en, 9);
  std::vector<Eigen::Vector3d> start, end;
  for (int i = 0; i < 3; ++i) start.push_back(Eigen::Vector3d(st[3 * i], st[3 * i + 1], st[3 * i + 2]));
  for (int i = 0; i < 3; ++i) end.push_back(Eigen::Vector3d(en[3 * i], en[3 * i + 1], en[3 * i + 2]));
  BsplineOptimizer opt;
  opt.setParam(nh);
  opt.setEnvironment(edt);
  const int cf = BsplineOptimizer::NORMAL_PHASE | BsplineOptimizer::MINTIME;
  // cost/gradient at the initial guess (checked against the oracle)
  {
    BsplineOptimizer probe;
    probe.setParam(nh);
    probe.setEnvironment(edt);
    probe.setBoundaryStates(start, end);
    Eigen::MatrixXd c0 = ctrl;
    double d0 = dt;
    nh.num["optimization/max_iteration_num1"] = 1;
    probe.setParam(nh);
    // one evaluation: max_num_id 0 -> max_iteration_num1 = 1
    probe.optimize(c0, d0, cf, 0, 1);
  }
  opt.setBoundaryStates(start, end);
  std::vector<double> x0((size_t)npts * 3 + 1), g0;
  for (int i = 0; i < npts; ++i)
    for (int j = 0; j < 3; ++j) x0[3 * i + j] = ctrl(i, j);
  x0.back() = dt;
  Eigen::MatrixXd cpts = ctrl;
  double dtt = dt;
  opt.optimize(cpts, dtt, cf, 1, 1);
  // evaluate initial and final cost with a fresh optimiser (optimize() clears the start state)
  BsplineOptimizer ev;
  ev.setParam(nh);
  ev.setEnvironment(edt);
  ev.setBoundaryStates(start, end);
  Eigen::MatrixXd tmpc = ctrl;
  double tmpd = dt;
  nh.num["optimization/max_iteration_num1"] = 1;
  ev.setParam(nh);
  ev.optimize(tmpc, tmpd, cf, 0, 1);  // sets dim/point_num/pt_dist from the INITIAL points
  ev.setBoundaryStates(start, end);
  double f0, f1;
  ev.combineCost(x0, g0, f0);
  std::vector<double> x1(x0.size()), g1;
  for (int i = 0; i < npts; ++i)
    for (int j = 0; j < 3; ++j) x1[3 * i + j] = cpts(i, j);
  x1.back() = dtt;
  ev.combineCost(x1, g1, f1);
  wr(out, &f0, 1);
  wr(out, &f1, 1);
  int ng = (int)g0.size();
  wr(out, &ng, 1);
  wr(out, g0.data(), g0.size());
  // the complete exploration front end as FastExplorationManager::planExploreMotion drives it
  // (fast_exploration_manager.cpp:97-118): searchFrontiers (with splitLargeFrontiers) ->
  // computeFrontiersToVisit -> getTopViewpointsInfo, plus the FSM's isFrontierCovered check
  {
    ros::NodeHandle nh2 = nh;
    auto& Q = nh2.num;
    Q["frontier/cluster_size_xy"] = 1.0;
    Q["frontier/down_sample"] = 3;
    Q["frontier/candidate_rmin"] = 1.5;
    Q["frontier/candidate_rmax"] = 2.5;
    Q["frontier/candidate_rnum"] = 3;
    Q["frontier/candidate_dphi"] = 15 * 3.1415926 / 180.0;
    Q["frontier/min_candidate_clearance"] = 0.21;
    Q["frontier/min_visib_num"] = 3;
    Q["frontier/min_candidate_dist"] = 0.75;
    Q["frontier/min_view_finish_fraction"] = 0.2;
    Q["perception_utils/top_angle"] = 0.56125;
    Q["perception_utils/left_angle"] = 0.69222;
    Q["perception_utils/right_angle"] = 0.68901;
    Q["perception_utils/max_dist"] = 4.5;
    double lo[3] = {ub_min(0), ub_min(1), ub_min(2)}, hi[3] = {ub_max(0), ub_max(1), ub_max(2)};
    fuelmi_map_set_updated_box(map->device(), lo, hi);
    FrontierFinder ff2(edt, nh2);
    ff2.searchFrontiers();
    ff2.computeFrontiersToVisit();
    std::vector<std::vector<Eigen::Vector3d>> act, dor;
    ff2.getFrontiers(act);
    ff2.getDormantFrontiers(dor);
    std::vector<Eigen::Vector3d> pts, avgs;
    std::vector<double> yaws;
    ff2.getTopViewpointsInfo(Eigen::Vector3d(0.0, 0.0, 1.0), pts, yaws, avgs);
    int hdr2[4] = {(int)act.size(), (int)dor.size(), (int)pts.size(), ff2.isFrontierCovered() ? 1 : 0};
    wr(out, hdr2, 4);
    for (size_t i = 0; i < pts.size(); ++i) {
      double q[7] = {pts[i](0), pts[i](1), pts[i](2), yaws[i], avgs[i](0), avgs[i](1), avgs[i](2)};
      wr(out, q, 7);
    }
    // tour planning over two rounds (fast_exploration_manager.cpp:187,334-335,423): the cost matrix is
    // kept incrementally -- after more frames some clusters are dropped (removed_ids_) and new ones linked
    ff2.updateFrontierCostMatrix();
    int n_extra = 0;
    if (fread(&n_extra, sizeof(int), 1, in) != 1) n_extra = 0;
    for (int k = 0; k < n_extra; ++k) {
      int n;
      double cam[3];
      rd(in, &n, 1);
      rd(in, cam, 3);
      std::vector<float> xyz((size_t)n * 3);
      rd(in, xyz.data(), xyz.size());
      pcl::PointCloud<pcl::PointXYZ> cloud;
      cloud.points.resize(n);
      for (int i = 0; i < n; ++i) cloud.points[i] = pcl::PointXYZ(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
      map->inputPointCloud(cloud, n, Eigen::Vector3d(cam[0], cam[1], cam[2]));
      MapROS::inflate(*map);
    }
    ff2.searchFrontiers();
    const int n_removed = (int)ff2.removedIds().size();
    ff2.computeFrontiersToVisit();
    ff2.updateFrontierCostMatrix();
    Eigen::MatrixXd mat;
    const Eigen::Vector3d cur(0.0, 0.0, 1.0);
    ff2.getFullCostMatrix(cur, Eigen::Vector3d(0, 0, 0), Eigen::Vector3d(0.3, 0, 0), mat);
    std::vector<int> tour;
    for (int k = 0; k < mat.rows() - 1 && k < 4; ++k) tour.push_back((k * 3) % (mat.rows() - 1));
    std::vector<Eigen::Vector3d> tpath;
    ff2.getPathForTour(cur, tour, tpath);
    int hdr3[4] = {n_extra, n_removed, (int)mat.rows(), (int)tpath.size()};
    wr(out, hdr3, 4);
    for (int i = 0; i < mat.rows(); ++i)
      for (int j = 0; j < mat.cols(); ++j) {
        const double v = mat(i, j);
        wr(out, &v, 1);
      }
    for (auto& q : tpath) {
      double v[3] = {q(0), q(1), q(2)};
      wr(out, v, 3);
    }
  }
  fclose(in);
  fclose(out);
  std::printf("facade_demo ok: %d voxels, %d frontier clusters, cost %.6f -> %.6f\n", N, nc, f0, f1);
  return 0;
}
