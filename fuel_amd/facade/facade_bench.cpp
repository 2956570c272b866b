// facade_bench.cpp -- what a FUEL maintainer gets: the streaming plan cycle timed THROUGH the C++ facade
// (the reference's class interfaces: MapROS::depthPoseCallback -> inputPointCloud -> clearAndInflateLocalMap,
// updateESDFCallback -> updateESDF3d, planExploreMotion -> searchFrontiers + computeFrontiersToVisit), with
// the host mirrors the callers' inline getters read switched on and off, beside the same sequence issued
// straight at the C-ABI.  Scenario file as facade_demo's: header + point-cloud frames.
//   facade_bench <scenario.bin> [repeat]
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include <plan_env/sdf_map.h>
#include <plan_env/edt_environment.h>
#include <active_perception/frontier_finder.h>
#include <active_perception/graph_node.h>
#include <active_perception/perception_utils.h>
#include <bspline_opt/bspline_optimizer.h>

#include <algorithm>
#include <thread>

namespace fast_planner {
double ViewNode::computeCost(const Eigen::Vector3d& p1, const Eigen::Vector3d& p2, const double& y1, const double& y2,
                             const Eigen::Vector3d&, const double&, std::vector<Eigen::Vector3d>& path) {
  path = {p1, p2};
  return (p2 - p1).norm() + 0.1 * std::fabs(y2 - y1);
}
double ViewNode::searchPath(const Eigen::Vector3d& p1, const Eigen::Vector3d& p2, std::vector<Eigen::Vector3d>& path) {
  path = {p1, p2};
  return (p2 - p1).norm();
}
PerceptionUtils::PerceptionUtils(ros::NodeHandle&) {}
class MapROS {
public:
  static void inflate(SDFMap& m) { m.clearAndInflateLocalMap(); }
};
}  // namespace fast_planner
using namespace fast_planner;

struct Frame {
  pcl::PointCloud<pcl::PointXYZ> cloud;
  std::vector<float> xyz;
  double cam[3];
};
static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
template <typename T>
static void rd(FILE* f, T* p, size_t n) {
  if (fread(p, sizeof(T), n, f) != n) {
    std::fprintf(stderr, "short read\n");
    std::exit(2);
  }
}

static void params(ros::NodeHandle& nh, const double* hdr) {
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
}

// one pass over the frames through the facade; returns seconds per cycle
// (ref_order < 0: the facade's default for frontier/reference_order)
static double run_facade(const double* hdr, const std::vector<Frame>& frames, bool mirrors, int ref_order, int* n_clusters) {
  ros::NodeHandle nh;
  params(nh, hdr);
  if (ref_order >= 0) nh.num["frontier/reference_order"] = ref_order;
  SDFMap::Ptr map(new SDFMap);
  map->initMap(nh);
  map->setHostMirror(mirrors, mirrors, mirrors);
  EDTEnvironment::Ptr edt(new EDTEnvironment);
  edt->setMap(map);
  FrontierFinder ff(edt, nh);
  fuelmi_map_synchronize(map->device());
  const double t0 = now_s();
  for (const Frame& f : frames) {
    map->inputPointCloud(f.cloud, (int)f.cloud.points.size(), Eigen::Vector3d(f.cam[0], f.cam[1], f.cam[2]));
    MapROS::inflate(*map);
    map->updateESDF3d();
    ff.searchFrontiers();
    ff.computeFrontiersToVisit();
  }
  fuelmi_map_synchronize(map->device());
  const double dt = now_s() - t0;
  std::vector<std::vector<Eigen::Vector3d>> cl;
  ff.getFrontiers(cl);
  *n_clusters = (int)cl.size();
  return dt / (double)frames.size();
}

// the same sequence at the C-ABI (no host lists, no mirrors)
static double run_cabi(const double* hdr, const std::vector<Frame>& frames, int* n_clusters) {
  fuelmi_map_cfg c;
  c.resolution = 0.1;
  for (int i = 0; i < 3; ++i) c.map_size[i] = hdr[i], c.box_min[i] = hdr[3 + i], c.box_max[i] = hdr[6 + i];
  c.obstacles_inflation = 0.199, c.local_bound_inflate = 0.5, c.ground_height = -1.0, c.default_dist = 0.0;
  c.optimistic = 0, c.signed_dist = 0;
  c.p_hit = 0.65, c.p_miss = 0.35, c.p_min = 0.12, c.p_max = 0.90, c.p_occ = 0.80;
  c.max_ray_length = 4.5, c.virtual_ceil_height = -10, c.device = 0;
  fuelmi_map* m = nullptr;
  if (fuelmi_map_create(&c, &m) != FUELMI_OK) return -1.0;
  fuelmi_frontier_cfg fc;
  fc.cluster_min = (int)hdr[9], fc.min_z = 0.4, fc.cluster_size_xy = -1.0, fc.down_sample = -1, fc.split = 0, fc.reference_order = 0;
  fuelmi_frontier* f = nullptr;
  if (fuelmi_frontier_create(m, &fc, &f) != FUELMI_OK) return -1.0;
  fuelmi_map_synchronize(m);
  const double t0 = now_s();
  for (const Frame& fr : frames) {
    fuelmi_map_input_points(m, fr.xyz.data(), 12, (int)(fr.xyz.size() / 3), fr.cam);
    fuelmi_frontier_search_begin(f);
    fuelmi_map_inflate_local(m);
    fuelmi_map_update_esdf(m);
    int n_new = 0;
    fuelmi_frontier_search_end(f, &n_new);
    fuelmi_frontier_commit(f, 0);
  }
  fuelmi_map_synchronize(m);
  const double dt = now_s() - t0;
  *n_clusters = fuelmi_frontier_count(f, 1);
  fuelmi_frontier_destroy(f);
  fuelmi_map_destroy(m);
  return dt / (double)frames.size();
}

// BASELINE's headline cycle through the facade (the drop-in's own figure): the whole 400x400x100 map as local bound
// and as updated box -- clearAndInflateLocalMap, updateESDF3d, searchFrontiers from fresh flags over the whole
// exploration box, and the clusters handed to the caller's containers (getFrontiers: vector<vector<Vector3d>>, what
// fast_exploration_manager.cpp:99-114 reads).  The occupancy state is uploaded once through the C-ABI (the reference
// has no such call: it only ever fuses frames); fuelmi_frontier_reset stands for "a fresh finder" each cycle.
static double run_fullbox(const double* hdr, const std::vector<double>& occ, bool mirrors, int cycles, int* n_clusters,
                          size_t* n_cells) {
  ros::NodeHandle nh;
  params(nh, hdr);
  SDFMap::Ptr map(new SDFMap);
  map->initMap(nh);
  map->setHostMirror(mirrors, mirrors, mirrors);
  EDTEnvironment::Ptr edt(new EDTEnvironment);
  edt->setMap(map);
  FrontierFinder ff(edt, nh);
  fuelmi_map* dev = map->device();
  if (fuelmi_map_upload_occupancy(dev, occ.data()) != FUELMI_OK) return -1.0;
  fuelmi_map_info inf;
  fuelmi_map_get_info(dev, &inf);
  const int lo[3] = {0, 0, 0}, hi[3] = {inf.voxel_num[0] - 1, inf.voxel_num[1] - 1, inf.voxel_num[2] - 1};
  fuelmi_map_set_local_bound(dev, lo, hi);
  std::vector<std::vector<Eigen::Vector3d>> cl;
  double t0 = 0.0;
  for (int k = -3; k < cycles; ++k) {  // three untimed cycles first
    if (k == 0) {
      fuelmi_map_synchronize(dev);
      t0 = now_s();
    }
    fuelmi_frontier_reset(ff.device());
    fuelmi_map_set_updated_box(dev, hdr + 3, hdr + 6);
    MapROS::inflate(*map);
    map->updateESDF3d();
    ff.searchFrontiers();
    ff.getFrontiers(cl);  // (empty: nothing was committed) ...
    cl.clear();
    for (const auto& fr : ff.newFrontiers()) cl.push_back(fr.cells_);  // ... the new clusters' cell lists, copied out
  }
  fuelmi_map_synchronize(dev);
  const double dt = now_s() - t0;
  *n_clusters = (int)cl.size();
  *n_cells = 0;
  for (const auto& c : cl) *n_cells += c.size();
  return dt / (double)cycles;
}

// The per-candidate path the reference actually calls (plan_manage/src/planner_manager.cpp:296-314: optimize() of ONE
// trajectory; bspline_optimizer.cpp:693-707 -> combineCost per evaluation): median latency of BsplineOptimizer::optimize
// and of one combineCost through the facade on the map of the full-box run, and the same solve from ten threads at
// once on ONE map (topoReplan, planner_manager.cpp:446-453).  out: [0] optimize ms, [1] combineCost us,
// [2] wall ms for 10 concurrent solves, [3] evaluations of the solve.
static void run_candidate(const double* hdr, const std::vector<double>& occ, double out[4]) {
  ros::NodeHandle nh;
  params(nh, hdr);
  auto& P = nh.num;
  P["optimization/ld_smooth"] = 20.0, P["optimization/ld_dist"] = 10.0, P["optimization/ld_feasi"] = 2.0;
  P["optimization/ld_start"] = 100.0, P["optimization/ld_end"] = 0.5, P["optimization/ld_guide"] = 1.5;
  P["optimization/ld_waypt"] = 0.3, P["optimization/ld_view"] = 0.0, P["optimization/ld_time"] = 1.0;
  P["optimization/dist0"] = 0.7, P["optimization/max_vel"] = 2.0, P["optimization/max_acc"] = 2.0;
  P["optimization/dlmin"] = 0.0, P["optimization/wnl"] = 1.0;
  for (int i = 1; i <= 4; ++i) {
    P["optimization/max_iteration_num" + std::to_string(i)] = 300;   // algorithm.xml:184-191
    P["optimization/max_iteration_time" + std::to_string(i)] = 0.005;
  }
  P["manager/bspline_degree"] = 3;
  SDFMap::Ptr map(new SDFMap);
  map->initMap(nh);
  EDTEnvironment::Ptr edt(new EDTEnvironment);
  edt->setMap(map);
  fuelmi_map* dev = map->device();
  if (fuelmi_map_upload_occupancy(dev, occ.data()) != FUELMI_OK) return;
  fuelmi_map_info inf;
  fuelmi_map_get_info(dev, &inf);
  const int lo[3] = {0, 0, 0}, hi[3] = {inf.voxel_num[0] - 1, inf.voxel_num[1] - 1, inf.voxel_num[2] - 1};
  fuelmi_map_set_local_bound(dev, lo, hi);
  MapROS::inflate(*map);
  map->updateESDF3d();
  fuelmi_map_synchronize(dev);
  const int N = 32;
  auto make = [&](int seed, Eigen::MatrixXd& ctrl, std::vector<Eigen::Vector3d>& st) {
    ctrl = Eigen::MatrixXd(N, 3);
    const double x0 = hdr[3] + 2.0 + 0.37 * seed, y0 = hdr[4] + 3.0 + 0.61 * seed;
    for (int i = 0; i < N; ++i) {
      ctrl(i, 0) = x0 + 6.0 * i / (N - 1.0), ctrl(i, 1) = y0 + 0.3 * std::sin(0.7 * i + seed), ctrl(i, 2) = 1.0 + 0.1 * std::cos(0.5 * i);
    }
    st = {Eigen::Vector3d(ctrl(0, 0), ctrl(0, 1), ctrl(0, 2)), Eigen::Vector3d(0.5, 0, 0), Eigen::Vector3d(0, 0, 0)};
  };
  const int cf = BsplineOptimizer::NORMAL_PHASE | BsplineOptimizer::MINTIME;
  BsplineOptimizer opt;
  opt.setParam(nh);
  opt.setEnvironment(edt);
  std::vector<double> t_opt, t_cc;
  int evals = 0;
  for (int r = 0; r < 24; ++r) {
    Eigen::MatrixXd ctrl;
    std::vector<Eigen::Vector3d> st, en = {Eigen::Vector3d(0, 0, 0)};
    make(r % 6, ctrl, st);
    en[0] = Eigen::Vector3d(ctrl(N - 1, 0), ctrl(N - 1, 1), ctrl(N - 1, 2));
    double dt = 0.175;
    opt.setBoundaryStates(st, en);
    const double a = now_s();
    opt.optimize(ctrl, dt, cf, 1, 1);
    if (r >= 4) t_opt.push_back(now_s() - a);
    (void)evals;
  }
  {
    Eigen::MatrixXd ctrl;
    std::vector<Eigen::Vector3d> st, en = {Eigen::Vector3d(0, 0, 0)};
    make(1, ctrl, st);
    double dt = 0.175;
    opt.setBoundaryStates(st, en);
    opt.optimize(ctrl, dt, cf, 1, 1);  // (leaves dim_/point_num_/pt_dist_ set for combineCost)
    std::vector<double> x(3 * N + 1), g;
    for (int i = 0; i < N; ++i)
      for (int k = 0; k < 3; ++k) x[3 * i + k] = ctrl(i, k);
    x[3 * N] = dt;
    opt.setBoundaryStates(st, en);
    double c = 0;
    for (int r = 0; r < 60; ++r) {
      const double a = now_s();
      opt.combineCost(x, g, c);
      if (r >= 10) t_cc.push_back(now_s() - a);
    }
  }
  std::sort(t_opt.begin(), t_opt.end());
  std::sort(t_cc.begin(), t_cc.end());
  out[0] = 1e3 * t_opt[t_opt.size() / 2];
  out[1] = 1e6 * t_cc[t_cc.size() / 2];
  // ten optimisers, ten threads, one map
  std::vector<std::unique_ptr<BsplineOptimizer>> opts;
  for (int k = 0; k < 10; ++k) {
    opts.emplace_back(new BsplineOptimizer);
    opts.back()->setParam(nh);
    opts.back()->setEnvironment(edt);
  }
  double best = 1e30;
  for (int r = 0; r < 5; ++r) {
    std::vector<std::thread> th;
    const double a = now_s();
    for (int k = 0; k < 10; ++k)
      th.emplace_back([&, k] {
        Eigen::MatrixXd ctrl;
        std::vector<Eigen::Vector3d> st, en = {Eigen::Vector3d(0, 0, 0)};
        make(k % 6, ctrl, st);
        en[0] = Eigen::Vector3d(ctrl(N - 1, 0), ctrl(N - 1, 1), ctrl(N - 1, 2));
        double dt = 0.175;
        opts[k]->setBoundaryStates(st, en);
        opts[k]->optimize(ctrl, dt, cf, 1, 1);
      });
    for (auto& t : th) t.join();
    best = std::min(best, now_s() - a);
  }
  out[2] = 1e3 * best;
  out[3] = 0.0;
}

int main(int argc, char** argv) {
  if (argc >= 5 && std::string(argv[3]) == "fullbox") {
    FILE* in = fopen(argv[1], "rb");
    if (!in) return 1;
    double hdr[10];
    rd(in, hdr, 10);
    fclose(in);
    FILE* fo = fopen(argv[4], "rb");
    if (!fo) return 1;
    const size_t n = (size_t)std::llround(hdr[0] * 10) * (size_t)std::llround(hdr[1] * 10) * (size_t)std::llround(hdr[2] * 10);
    std::vector<double> occ(n);
    rd(fo, occ.data(), n);
    fclose(fo);
    const int repeat = std::atoi(argv[2]);
    double best[2] = {1e30, 1e30};
    int ncl[2] = {0, 0};
    size_t ncell[2] = {0, 0};
    for (int r = 0; r < repeat; ++r) {
      best[0] = std::min(best[0], run_fullbox(hdr, occ, true, 10, &ncl[0], &ncell[0]));
      best[1] = std::min(best[1], run_fullbox(hdr, occ, false, 20, &ncl[1], &ncell[1]));
    }
    double cand[4] = {0, 0, 0, 0};
    run_candidate(hdr, occ, cand);
    std::printf("{\"workload\": \"full-box plan cycle through the facade on a %.0fx%.0fx%.0f m map: clearAndInflateLocalMap + "
                "updateESDF3d + searchFrontiers (fresh flags, whole exploration box) + the cell lists of the new clusters as "
                "vector<Vector3d>\", \"facade_mirrors_on_ms\": %.4f, \"facade_mirrors_off_ms\": %.4f, "
                "\"cycles_per_s_mirrors_off\": %.1f, \"clusters\": %d, \"cells\": %zu, "
                "\"per_candidate\": {\"what\": \"BsplineOptimizer::optimize (one 32-point trajectory, NORMAL_PHASE|MINTIME, "
                "max 300 evaluations / 5 ms) and one combineCost through the facade, medians; ten optimisers on ten threads on "
                "ONE map, wall time for the ten solves\", \"optimize_ms\": %.4f, \"combine_cost_us\": %.2f, "
                "\"ten_threads_ten_solves_ms\": %.4f}}\n",
                hdr[0], hdr[1], hdr[2], 1e3 * best[0], 1e3 * best[1], 1.0 / best[1], ncl[1], ncell[1], cand[0], cand[1], cand[2]);
    return 0;
  }

  if (argc < 2) return 1;
  FILE* in = fopen(argv[1], "rb");
  if (!in) return 1;
  const int repeat = argc > 2 ? std::atoi(argv[2]) : 3;
  double hdr[10];
  rd(in, hdr, 10);
  int n_frames;
  rd(in, &n_frames, 1);
  std::vector<Frame> frames(n_frames);
  for (Frame& f : frames) {
    int n;
    rd(in, &n, 1);
    rd(in, f.cam, 3);
    f.xyz.resize((size_t)n * 3);
    rd(in, f.xyz.data(), f.xyz.size());
    f.cloud.points.resize(n);
    for (int i = 0; i < n; ++i) f.cloud.points[i] = pcl::PointXYZ(f.xyz[3 * i], f.xyz[3 * i + 1], f.xyz[3 * i + 2]);
  }
  fclose(in);
  double best[4] = {1e30, 1e30, 1e30, 1e30};
  int ncl[4] = {0, 0, 0, 0};
  for (int r = 0; r < repeat; ++r) {  // fresh map every pass: the same frames, the same work
    best[0] = std::min(best[0], run_facade(hdr, frames, true, -1, &ncl[0]));
    best[1] = std::min(best[1], run_facade(hdr, frames, false, -1, &ncl[1]));
    best[2] = std::min(best[2], run_cabi(hdr, frames, &ncl[2]));
    best[3] = std::min(best[3], run_facade(hdr, frames, false, 0, &ncl[3]));
  }
  std::printf("{\"workload\": \"streaming cycle: %d point-cloud frames on a %.0fx%.0fx%.0f m map (fusion, local inflation, "
              "local ESDF, incremental frontier search, commit)\", \"facade_mirrors_on_ms\": %.4f, "
              "\"facade_mirrors_off_ms\": %.4f, \"facade_mirrors_off_address_order_ms\": %.4f, \"c_abi_ms\": %.4f, "
              "\"mirrors_on_over_off\": %.3f, \"note\": \"facade rows at its default frontier/reference_order = 2 (the "
              "reference's BFS cell order for searches of this size) unless named; the C-ABI row uses the address order\", "
              "\"clusters\": [%d, %d, %d, %d]}\n",
              n_frames, hdr[0], hdr[1], hdr[2], 1e3 * best[0], 1e3 * best[1], 1e3 * best[3], 1e3 * best[2],
              best[0] / best[1], ncl[0], ncl[1], ncl[2], ncl[3]);
  return 0;
}
