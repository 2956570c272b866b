// ref_api.cpp -- C entry points over the REAL reference classes (fast_planner::SDFMap,
// EDTEnvironment, BsplineOptimizer), compiled from the sources where they lie under
// /root/reference with the header stand-ins in shim/ (Eigen/ROS/PCL/NLopt are not installed).
// TEST INFRASTRUCTURE: used by oracle/ref_build/check_ref.py and tests/ to pin the oracle.
// No reference source is copied into this repository; only its public/private class members are
// driven from here.
#include <cmath>
#include <cstring>
#include <iostream>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
#include <ros/ros.h>

#define private public
#define protected public
#include <plan_env/sdf_map.h>
#include <plan_env/map_ros.h>
#include <plan_env/raycast.h>
#include <plan_env/edt_environment.h>
#include <bspline_opt/bspline_optimizer.h>
#include <active_perception/frontier_finder.h>
#undef private
#undef protected

using fast_planner::BsplineOptimizer;
using fast_planner::EDTEnvironment;
using fast_planner::SDFMap;

extern "C" {

typedef struct {  // identical to fo_map_cfg (oracle/fuel_oracle.h)
  double resolution;
  double map_size[3];
  double ground_height;
  double obstacles_inflation;
  double local_bound_inflate;
  double default_dist;
  int optimistic;
  int signed_dist;
  double p_hit, p_miss, p_min, p_max, p_occ;
  double max_ray_length;
  double virtual_ceil_height;
  double box_min[3], box_max[3];
} ref_map_cfg;

struct ref_map {
  std::shared_ptr<SDFMap> map;
  std::shared_ptr<EDTEnvironment> edt;
};

ref_map* ref_map_create(const ref_map_cfg* c) {
  ros::NodeHandle nh;
  auto& p = nh.num;
  p["sdf_map/resolution"] = c->resolution;
  p["sdf_map/map_size_x"] = c->map_size[0];
  p["sdf_map/map_size_y"] = c->map_size[1];
  p["sdf_map/map_size_z"] = c->map_size[2];
  p["sdf_map/obstacles_inflation"] = c->obstacles_inflation;
  p["sdf_map/local_bound_inflate"] = c->local_bound_inflate;
  p["sdf_map/ground_height"] = c->ground_height;
  p["sdf_map/default_dist"] = c->default_dist;
  p["sdf_map/optimistic"] = c->optimistic;
  p["sdf_map/signed_dist"] = c->signed_dist;
  p["sdf_map/p_hit"] = c->p_hit;
  p["sdf_map/p_miss"] = c->p_miss;
  p["sdf_map/p_min"] = c->p_min;
  p["sdf_map/p_max"] = c->p_max;
  p["sdf_map/p_occ"] = c->p_occ;
  p["sdf_map/max_ray_length"] = c->max_ray_length;
  p["sdf_map/virtual_ceil_height"] = c->virtual_ceil_height;
  const char* ax[3] = {"x", "y", "z"};
  for (int i = 0; i < 3; ++i) {
    p[std::string("sdf_map/box_min_") + ax[i]] = c->box_min[i];
    p[std::string("sdf_map/box_max_") + ax[i]] = c->box_max[i];
  }
  std::streambuf* old = std::cout.rdbuf(nullptr);  // initMap prints the logit constants
  ref_map* r = new ref_map;
  r->map.reset(new SDFMap);
  r->map->initMap(nh);
  std::cout.rdbuf(old);
  r->edt.reset(new EDTEnvironment);
  r->edt->setMap(r->map);
  return r;
}
void ref_map_destroy(ref_map* r) { delete r; }

void ref_map_voxel_num(ref_map* r, int out[3]) {
  for (int i = 0; i < 3; ++i) out[i] = r->map->mp_->map_voxel_num_(i);
}
double* ref_map_occupancy(ref_map* r) { return r->map->md_->occupancy_buffer_.data(); }
char* ref_map_inflate_buf(ref_map* r) { return r->map->md_->occupancy_buffer_inflate_.data(); }
double* ref_map_distance(ref_map* r) { return r->map->md_->distance_buffer_.data(); }

void ref_map_input_points(ref_map* r, const float* xyz, int n, const double cam[3]) {
  pcl::PointCloud<pcl::PointXYZ> cloud;
  cloud.points.resize(n);
  for (int i = 0; i < n; ++i) {
    cloud.points[i].x = xyz[3 * i];
    cloud.points[i].y = xyz[3 * i + 1];
    cloud.points[i].z = xyz[3 * i + 2];
  }
  r->map->inputPointCloud(cloud, n, Eigen::Vector3d(cam[0], cam[1], cam[2]));
}
void ref_map_inflate_local(ref_map* r) { r->map->clearAndInflateLocalMap(); }
void ref_map_update_esdf(ref_map* r) { r->map->updateESDF3d(); }
void ref_map_get_local_bound(ref_map* r, int lo[3], int hi[3]) {
  for (int i = 0; i < 3; ++i) lo[i] = r->map->md_->local_bound_min_(i), hi[i] = r->map->md_->local_bound_max_(i);
}
void ref_map_set_local_bound(ref_map* r, const int lo[3], const int hi[3]) {
  for (int i = 0; i < 3; ++i) r->map->md_->local_bound_min_(i) = lo[i], r->map->md_->local_bound_max_(i) = hi[i];
}
void ref_map_get_updated_box(ref_map* r, double lo[3], double hi[3], int reset) {
  Eigen::Vector3d a, b;
  r->map->getUpdatedBox(a, b, reset != 0);
  for (int i = 0; i < 3; ++i) lo[i] = a(i), hi[i] = b(i);
}
void ref_map_reset_buffer_all(ref_map* r) { r->map->resetBuffer(); }
void ref_map_set_occupied(ref_map* r, const double pos[3], int occ) {
  r->map->setOccupied(Eigen::Vector3d(pos[0], pos[1], pos[2]), occ);
}
int ref_map_get_occupancy_idx(ref_map* r, const int id[3]) {
  return r->map->getOccupancy(Eigen::Vector3i(id[0], id[1], id[2]));
}
void ref_map_dist_grad(ref_map* r, const double* pos, int n, double* dist, double* grad) {
  for (int i = 0; i < n; ++i) {
    Eigen::Vector3d g;
    double d;
    r->edt->evaluateEDTWithGrad(Eigen::Vector3d(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]), -1.0, d, g);
    dist[i] = d;
    for (int k = 0; k < 3; ++k) grad[3 * i + k] = g(k);
  }
}
// RayCaster exactly as inputPointCloud drives it (first nextId discarded)
int ref_raycast_cells(ref_map* r, const double start[3], const double end[3], int* out, int cap) {
  RayCaster rc;
  rc.setParams(r->map->mp_->resolution_, r->map->mp_->map_origin_);
  rc.input(Eigen::Vector3d(start[0], start[1], start[2]), Eigen::Vector3d(end[0], end[1], end[2]));
  Eigen::Vector3i idx;
  rc.nextId(idx);
  int n = 0;
  while (rc.nextId(idx) && n < 100000) {
    if (n < cap) out[3 * n] = idx(0), out[3 * n + 1] = idx(1), out[3 * n + 2] = idx(2);
    ++n;
  }
  return n;
}

// ---- FrontierFinder: the reference's own searchFrontiers()/expandFrontier() -----------------------
// cluster_size_xy < 0: a huge value is used, so splitHorizontally never splits and tmp_frontiers_ holds
// the region-grown clusters (the F1-F4 contract); otherwise splitLargeFrontiers runs with it (the
// down-sampled cells come from the pcl::VoxelGrid restatement in compat/, the principal direction
// from the Eigen::EigenSolver stand-in: both third-party libraries that are absent here).
struct ref_frontier {
  std::unique_ptr<fast_planner::FrontierFinder> ff;
  ref_map* map;
};
ref_frontier* ref_frontier_create(ref_map* r, int cluster_min, double cluster_size_xy) {
  ros::NodeHandle nh;
  nh.num["frontier/cluster_min"] = cluster_min;
  nh.num["frontier/cluster_size_xy"] = cluster_size_xy < 0 ? 1e18 : cluster_size_xy;
  nh.num["frontier/cluster_size_z"] = 10.0;
  nh.num["frontier/down_sample"] = 3;
  ref_frontier* f = new ref_frontier;
  f->map = r;
  f->ff.reset(new fast_planner::FrontierFinder(r->edt, nh));
  return f;
}
// full parameter set: vp = {candidate_rmin, candidate_rmax, candidate_rnum, candidate_dphi,
// min_candidate_clearance, min_visib_num, min_candidate_dist, min_view_finish_fraction,
// top_angle, left_angle, right_angle, max_dist}
ref_frontier* ref_frontier_create_full(ref_map* r, int cluster_min, double cluster_size_xy, const double* vp) {
  ros::NodeHandle nh;
  nh.num["frontier/cluster_min"] = cluster_min;
  nh.num["frontier/cluster_size_xy"] = cluster_size_xy < 0 ? 1e18 : cluster_size_xy;
  nh.num["frontier/cluster_size_z"] = 10.0;
  nh.num["frontier/down_sample"] = 3;
  nh.num["frontier/candidate_rmin"] = vp[0];
  nh.num["frontier/candidate_rmax"] = vp[1];
  nh.num["frontier/candidate_rnum"] = vp[2];
  nh.num["frontier/candidate_dphi"] = vp[3];
  nh.num["frontier/min_candidate_clearance"] = vp[4];
  nh.num["frontier/min_visib_num"] = vp[5];
  nh.num["frontier/min_candidate_dist"] = vp[6];
  nh.num["frontier/min_view_finish_fraction"] = vp[7];
  nh.num["perception_utils/top_angle"] = vp[8];
  nh.num["perception_utils/left_angle"] = vp[9];
  nh.num["perception_utils/right_angle"] = vp[10];
  nh.num["perception_utils/max_dist"] = vp[11];
  nh.num["perception_utils/vis_dist"] = 1.0;
  ref_frontier* f = new ref_frontier;
  f->map = r;
  f->ff.reset(new fast_planner::FrontierFinder(r->edt, nh));
  return f;
}
// computeFrontiersToVisit (:392-423): samples viewpoints for tmp_frontiers_, moves them to
// frontiers_ (viewpoints sorted by coverage) or dormant_frontiers_
void ref_frontier_compute_to_visit(ref_frontier* f) {
  std::streambuf* old = std::cout.rdbuf(nullptr);
  f->ff->computeFrontiersToVisit();
  std::cout.rdbuf(old);
}
int ref_frontier_is_covered(ref_frontier* f) { return f->ff->isFrontierCovered() ? 1 : 0; }
void ref_frontier_destroy(ref_frontier* f) { delete f; }
char* ref_frontier_flags(ref_frontier* f) { return f->ff->frontier_flag_.data(); }
int ref_frontier_search(ref_frontier* f) {
  std::streambuf* old = std::cout.rdbuf(nullptr);  // "Before remove: ..." chatter
  f->ff->searchFrontiers();
  std::cout.rdbuf(old);
  return (int)f->ff->tmp_frontiers_.size();
}
void ref_frontier_commit(ref_frontier* f, int dormant) {
  auto& dst = dormant ? f->ff->dormant_frontiers_ : f->ff->frontiers_;
  dst.insert(dst.end(), f->ff->tmp_frontiers_.begin(), f->ff->tmp_frontiers_.end());
  f->ff->tmp_frontiers_.clear();
}
static std::list<fast_planner::Frontier>& ref_pick(ref_frontier* f, int which) {
  return which == 0 ? f->ff->tmp_frontiers_ : (which == 1 ? f->ff->frontiers_ : f->ff->dormant_frontiers_);
}
int ref_frontier_count(ref_frontier* f, int which) { return (int)ref_pick(f, which).size(); }
int ref_frontier_cluster_size(ref_frontier* f, int which, int k) {
  auto it = ref_pick(f, which).begin();
  std::advance(it, k);
  return (int)it->cells_.size();
}
void ref_frontier_cluster_cells(ref_frontier* f, int which, int k, int* adr) {
  auto it = ref_pick(f, which).begin();
  std::advance(it, k);
  Eigen::Vector3i idx;
  for (size_t i = 0; i < it->cells_.size(); ++i) {
    f->map->map->posToIndex(it->cells_[i], idx);
    adr[i] = f->map->map->toAddress(idx);
  }
}
void ref_frontier_cluster_info(ref_frontier* f, int which, int k, double* out9) {
  auto it = ref_pick(f, which).begin();
  std::advance(it, k);
  for (int i = 0; i < 3; ++i) out9[i] = it->average_(i), out9[3 + i] = it->box_min_(i), out9[6 + i] = it->box_max_(i);
}
int ref_frontier_cluster_filtered_size(ref_frontier* f, int which, int k) {
  auto it = ref_pick(f, which).begin();
  std::advance(it, k);
  return (int)it->filtered_cells_.size();
}
void ref_frontier_cluster_filtered(ref_frontier* f, int which, int k, double* xyz) {
  auto it = ref_pick(f, which).begin();
  std::advance(it, k);
  for (size_t i = 0; i < it->filtered_cells_.size(); ++i)
    for (int q = 0; q < 3; ++q) xyz[3 * i + q] = it->filtered_cells_[i](q);
}
int ref_frontier_viewpoint_count(ref_frontier* f, int which, int k) {
  auto it = ref_pick(f, which).begin();
  std::advance(it, k);
  return (int)it->viewpoints_.size();
}
void ref_frontier_viewpoints(ref_frontier* f, int which, int k, double* pos_yaw, int* visib) {
  auto it = ref_pick(f, which).begin();
  std::advance(it, k);
  for (size_t i = 0; i < it->viewpoints_.size(); ++i) {
    for (int q = 0; q < 3; ++q) pos_yaw[4 * i + q] = it->viewpoints_[i].pos_(q);
    pos_yaw[4 * i + 3] = it->viewpoints_[i].yaw_;
    visib[i] = it->viewpoints_[i].visib_num_;
  }
}
// tour planning (frontier_finder.cpp:258-324, 507-589) with the deterministic ViewNode of ref_frontier_stubs.cpp
void ref_frontier_update_cost_matrix(ref_frontier* f) {
  std::streambuf* keep = std::cout.rdbuf(nullptr);  // the function narrates every pair on stdout
  f->ff->updateFrontierCostMatrix();
  std::cout.rdbuf(keep);
}
int ref_frontier_full_cost_matrix(ref_frontier* f, const double pos[3], const double vel[3], const double yaw[3],
                                  double* mat, int cap) {
  Eigen::MatrixXd m;
  f->ff->getFullCostMatrix(Eigen::Vector3d(pos[0], pos[1], pos[2]), Eigen::Vector3d(vel[0], vel[1], vel[2]),
                           Eigen::Vector3d(yaw[0], yaw[1], yaw[2]), m);
  const int d = m.rows();
  if (d * d > cap) return -d;
  for (int i = 0; i < d; ++i)
    for (int j = 0; j < d; ++j) mat[i * d + j] = m(i, j);
  return d;
}
int ref_frontier_path_for_tour(ref_frontier* f, const double pos[3], const int* ids, int n, double* xyz, int cap) {
  std::vector<int> v(ids, ids + n);
  std::vector<Eigen::Vector3d> path;
  f->ff->getPathForTour(Eigen::Vector3d(pos[0], pos[1], pos[2]), v, path);
  if ((int)path.size() > cap) return -(int)path.size();
  for (size_t i = 0; i < path.size(); ++i)
    for (int k = 0; k < 3; ++k) xyz[3 * i + k] = path[i](k);
  return (int)path.size();
}
int ref_frontier_removed_count(ref_frontier* f) { return (int)f->ff->removed_ids_.size(); }
void ref_frontier_removed_ids(ref_frontier* f, int* ids) {
  for (size_t i = 0; i < f->ff->removed_ids_.size(); ++i) ids[i] = f->ff->removed_ids_[i];
}
void ref_map_set_updated_box(ref_map* r, const double lo[3], const double hi[3]) {
  for (int i = 0; i < 3; ++i) r->map->md_->update_min_(i) = lo[i], r->map->md_->update_max_(i) = hi[i];
  r->map->md_->reset_updated_box_ = false;
}

typedef struct {  // identical to fo_bspline_cfg
  double ld_smooth, ld_dist, ld_feasi, ld_start, ld_end, ld_guide, ld_waypt, ld_view, ld_time;
  double dist0, max_vel, max_acc, wnl, dlmin;
  int bspline_degree;
} ref_bspline_cfg;
typedef struct {  // identical to fo_bspline_problem
  int cost_function, dim, point_num;
  double knot_span, pt_dist, time_lb;
  const double* start_state;
  const double* end_state;
  int end_n;
  const double* guide_pts;
  const double* waypoints;
  const int* waypt_idx;
  int n_waypt;
  const double* view_pt;
  const double* view_dir;
  int view_idx;
} ref_bspline_problem;

// BsplineOptimizer::combineCost with the state optimize() would have prepared (:110-163)
void ref_bspline_cost_grad(ref_map* r, const ref_bspline_cfg* c, const ref_bspline_problem* pb, const double* x,
                           double* cost, double* grad) {
  BsplineOptimizer o;
  o.setEnvironment(r->edt);
  o.ld_smooth_ = c->ld_smooth, o.ld_dist_ = c->ld_dist, o.ld_feasi_ = c->ld_feasi, o.ld_start_ = c->ld_start;
  o.ld_end_ = c->ld_end, o.ld_guide_ = c->ld_guide, o.ld_waypt_ = c->ld_waypt, o.ld_view_ = c->ld_view;
  o.ld_time_ = c->ld_time, o.dist0_ = c->dist0, o.max_vel_ = c->max_vel, o.max_acc_ = c->max_acc;
  o.wnl_ = c->wnl, o.dlmin_ = c->dlmin, o.bspline_degree_ = c->bspline_degree;
  o.cost_function_ = pb->cost_function;
  o.dim_ = pb->dim;
  o.order_ = (pb->dim == 1) ? 3 : c->bspline_degree;
  o.point_num_ = pb->point_num;
  o.optimize_time_ = pb->cost_function & BsplineOptimizer::MINTIME;
  o.variable_num_ = o.optimize_time_ ? pb->dim * pb->point_num + 1 : pb->dim * pb->point_num;
  o.knot_span_ = pb->knot_span;
  o.pt_dist_ = pb->pt_dist;
  o.time_lb_ = pb->time_lb;
  auto v3 = [](const double* p) { return Eigen::Vector3d(p[0], p[1], p[2]); };
  if (pb->start_state)
    for (int i = 0; i < 3; ++i) o.start_state_.push_back(v3(pb->start_state + 3 * i));
  if (pb->end_state)
    for (int i = 0; i < pb->end_n; ++i) o.end_state_.push_back(v3(pb->end_state + 3 * i));
  if (pb->guide_pts)
    for (int i = 0; i < pb->point_num - 2 * o.order_; ++i) o.guide_pts_.push_back(v3(pb->guide_pts + 3 * i));
  for (int i = 0; i < pb->n_waypt; ++i) {
    o.waypoints_.push_back(v3(pb->waypoints + 3 * i));
    o.waypt_idx_.push_back(pb->waypt_idx[i]);
  }
  if (pb->view_pt) {
    o.view_cons_.pt_ = v3(pb->view_pt);
    o.view_cons_.dir_ = v3(pb->view_dir);
    o.view_cons_.idx_ = pb->view_idx;
  }
  const int N = pb->point_num;
  o.g_q_.resize(N), o.g_smoothness_.resize(N), o.g_distance_.resize(N), o.g_feasibility_.resize(N);
  o.g_start_.resize(N), o.g_end_.resize(N), o.g_guide_.resize(N), o.g_waypoints_.resize(N);
  o.g_view_.resize(N), o.g_time_.resize(N);
  o.comb_time = 0.0;
  std::vector<double> xv(x, x + o.variable_num_), gv;
  double f;
  o.combineCost(xv, gv, f);
  *cost = f;
  for (int i = 0; i < o.variable_num_; ++i) grad[i] = gv[i];
}

}  // extern "C"
