// facade.cpp -- the reference's C++ class interfaces over the libfuelmi C-ABI (host side only;
// every computation is a HIP kernel behind include/fuelmi.h).  See INTEGRATION.md.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <deque>
#include <limits>

#include "plan_env/sdf_map.h"
#include "plan_env/edt_environment.h"
#include "active_perception/frontier_finder.h"
#include "active_perception/graph_node.h"
#include "active_perception/perception_utils.h"
#include "bspline_opt/bspline_optimizer.h"

namespace fast_planner {

static void warn(const char* what, int rc) {
  // error convention of the reference: void returns, log and continue
  if (rc != FUELMI_OK) std::fprintf(stderr, "[fuelmi] %s failed (%d): %s\n", what, rc, fuelmi_last_error());
}

// ------------------------------------------------------------------------------------------------
// SDFMap
// ------------------------------------------------------------------------------------------------
SDFMap::SDFMap() : ext_(new Ext{nullptr, true, true, true}) {}
SDFMap::~SDFMap() {
  if (ext_->dev) fuelmi_map_destroy(ext_->dev);
  delete ext_;
}
// (never called: this library does not create either object -- see the data members in plan_env/sdf_map.h)
void SDFMap::MapROSDelete::operator()(MapROS*) const {}
void SDFMap::RayCasterDelete::operator()(RayCaster*) const {}

void SDFMap::initMap(ros::NodeHandle& nh) {
  // parameter names and defaults of the reference (plan_env/src/sdf_map.cpp:19-47,78-82)
  mp_.reset(new MapParam);
  md_.reset(new MapData);
  fuelmi_map_cfg c;
  double x_size, y_size, z_size;
  nh.param("sdf_map/resolution", c.resolution, -1.0);
  nh.param("sdf_map/map_size_x", x_size, -1.0);
  nh.param("sdf_map/map_size_y", y_size, -1.0);
  nh.param("sdf_map/map_size_z", z_size, -1.0);
  nh.param("sdf_map/obstacles_inflation", c.obstacles_inflation, -1.0);
  nh.param("sdf_map/local_bound_inflate", c.local_bound_inflate, 1.0);
  nh.param("sdf_map/local_map_margin", mp_->local_map_margin_, 1);
  nh.param("sdf_map/ground_height", c.ground_height, 1.0);
  nh.param("sdf_map/default_dist", c.default_dist, 5.0);
  bool optimistic, signed_dist;
  nh.param("sdf_map/optimistic", optimistic, true);
  nh.param("sdf_map/signed_dist", signed_dist, false);
  c.optimistic = optimistic;
  c.signed_dist = signed_dist;
  nh.param("sdf_map/p_hit", c.p_hit, 0.70);
  nh.param("sdf_map/p_miss", c.p_miss, 0.35);
  nh.param("sdf_map/p_min", c.p_min, 0.12);
  nh.param("sdf_map/p_max", c.p_max, 0.97);
  nh.param("sdf_map/p_occ", c.p_occ, 0.80);
  nh.param("sdf_map/max_ray_length", c.max_ray_length, -0.1);
  nh.param("sdf_map/virtual_ceil_height", c.virtual_ceil_height, -0.1);
  int device = 0;
  nh.param("sdf_map/hip_device", device, 0);  // addition: which GPU hosts this map
  c.device = device;
  c.map_size[0] = x_size, c.map_size[1] = y_size, c.map_size[2] = z_size;
  const double org[3] = {-x_size / 2.0, -y_size / 2.0, c.ground_height};
  const char* axis[3] = {"x", "y", "z"};
  for (int i = 0; i < 3; ++i) {
    nh.param(std::string("sdf_map/box_min_") + axis[i], c.box_min[i], org[i]);
    nh.param(std::string("sdf_map/box_max_") + axis[i], c.box_max[i], org[i] + c.map_size[i]);
  }
  // process set-up (include/fuelmi.h): hardware queues for the streams of the map, its finder and the optimiser threads.
  // In time only if nothing in the node has initialised HIP yet -- fuelmi_hw_queues_state() tells; exporting
  // GPU_MAX_HW_QUEUES=16 in the launch file is the way that does not depend on the order
  fuelmi_init(0);
  int rc = fuelmi_map_create(&c, &ext_->dev);
  warn("fuelmi_map_create", rc);
  if (rc != FUELMI_OK) return;
  fuelmi_map_info I;
  fuelmi_map_get_info(ext_->dev, &I);
  mp_->resolution_ = c.resolution;
  mp_->resolution_inv_ = I.resolution_inv;
  mp_->obstacles_inflation_ = c.obstacles_inflation;
  mp_->local_bound_inflate_ = std::max(c.resolution, c.local_bound_inflate);
  mp_->ground_height_ = c.ground_height;
  mp_->virtual_ceil_height_ = c.virtual_ceil_height;
  mp_->default_dist_ = c.default_dist;
  mp_->optimistic_ = optimistic;
  mp_->signed_dist_ = signed_dist;
  mp_->p_hit_ = c.p_hit, mp_->p_miss_ = c.p_miss, mp_->p_min_ = c.p_min, mp_->p_max_ = c.p_max, mp_->p_occ_ = c.p_occ;
  mp_->prob_hit_log_ = I.prob_hit_log, mp_->prob_miss_log_ = I.prob_miss_log;
  mp_->clamp_min_log_ = I.clamp_min_log, mp_->clamp_max_log_ = I.clamp_max_log;
  mp_->min_occupancy_log_ = I.min_occupancy_log;
  mp_->max_ray_length_ = c.max_ray_length;
  mp_->unknown_flag_ = 0.01;
  for (int i = 0; i < 3; ++i) {
    mp_->map_origin_(i) = I.origin[i];
    mp_->map_size_(i) = c.map_size[i];
    mp_->map_min_boundary_(i) = I.min_boundary[i];
    mp_->map_max_boundary_(i) = I.max_boundary[i];
    mp_->map_voxel_num_(i) = I.voxel_num[i];
    mp_->box_min_(i) = I.box_min[i];
    mp_->box_max_(i) = I.box_max[i];
    mp_->box_mind_(i) = c.box_min[i];
    mp_->box_maxd_(i) = c.box_max[i];
  }
  const size_t n = (size_t)getVoxelNum();
  md_->occupancy_buffer_.assign(n, mp_->clamp_min_log_ - mp_->unknown_flag_);
  md_->occupancy_buffer_inflate_.assign(n, 0);
  md_->distance_buffer_.assign(n, mp_->default_dist_);
  md_->reset_updated_box_ = true;
  for (int i = 0; i < 3; ++i) {
    md_->update_min_(i) = md_->update_max_(i) = 0.0;
    md_->local_bound_min_(i) = md_->local_bound_max_(i) = 0;
  }
  // the mirrors never move again: pin and map them once, every refresh is then one kernel storing the box
  // voxels straight into them (no staging copy, one synchronisation)
  warn("fuelmi_map_register_mirrors",
       fuelmi_map_register_mirrors(ext_->dev, md_->occupancy_buffer_.data(), md_->occupancy_buffer_inflate_.data(),
                                   md_->distance_buffer_.data()));
}

void SDFMap::setHostMirror(bool occupancy, bool inflate, bool distance) {
  ext_->mirror_occ = occupancy, ext_->mirror_infl = inflate, ext_->mirror_dist = distance;
}

void SDFMap::pullBounds() {
  int lo[3], hi[3];
  double a[3], b[3];
  fuelmi_map_get_local_bound(ext_->dev, lo, hi);
  fuelmi_map_get_updated_box(ext_->dev, a, b, 0);
  for (int i = 0; i < 3; ++i) {
    md_->local_bound_min_(i) = lo[i], md_->local_bound_max_(i) = hi[i];
    md_->update_min_(i) = a[i], md_->update_max_(i) = b[i];
  }
}

void SDFMap::syncMirrors(const Eigen::Vector3i& bmin, const Eigen::Vector3i& bmax, bool occ, bool infl, bool dist) {
  if (!(occ || infl || dist)) return;
  const int lo[3] = {bmin(0), bmin(1), bmin(2)}, hi[3] = {bmax(0), bmax(1), bmax(2)};
  warn("fuelmi_map_sync_host",
       fuelmi_map_sync_host(ext_->dev, lo, hi, occ ? md_->occupancy_buffer_.data() : nullptr,
                            infl ? md_->occupancy_buffer_inflate_.data() : nullptr,
                            dist ? md_->distance_buffer_.data() : nullptr));
}

void SDFMap::inputPointCloud(const pcl::PointCloud<pcl::PointXYZ>& points, const int& point_num,
                             const Eigen::Vector3d& camera_pos) {
  if (point_num == 0) return;
  const double cam[3] = {camera_pos(0), camera_pos(1), camera_pos(2)};
  warn("fuelmi_map_input_points",
       fuelmi_map_input_points(ext_->dev, &points.points[0].x, (int)sizeof(pcl::PointXYZ), point_num, cam));
  pullBounds();
  md_->reset_updated_box_ = false;
  // fused voxels lie in the index box of camera + end points, inside the inflated local bound
  syncMirrors(md_->local_bound_min_, md_->local_bound_max_, ext_->mirror_occ, false, false);
}

void SDFMap::clearAndInflateLocalMap() {
  warn("fuelmi_map_inflate_local", fuelmi_map_inflate_local(ext_->dev));
  // stamps spill up to inflate_step voxels outside the box; a stamp that leaves the map in z lands in the
  // neighbouring y row at the other end of z, one that leaves it in y in the neighbouring x slab (the
  // reference only tests the linear address, sdf_map.cpp:453-458): widen the refreshed box accordingly
  Eigen::Vector3i lo = md_->local_bound_min_, hi = md_->local_bound_max_;
  const int s = (int)std::ceil(mp_->obstacles_inflation_ / mp_->resolution_);
  for (int k = 0; k < 3; ++k) lo(k) -= s, hi(k) += s;
  for (int k = 2; k >= 1; --k)
    if (lo(k) < 0 || hi(k) > mp_->map_voxel_num_(k) - 1) {
      lo(k) = 0, hi(k) = mp_->map_voxel_num_(k) - 1;
      lo(k - 1) -= 1, hi(k - 1) += 1;
    }
  boundIndex(lo);
  boundIndex(hi);
  syncMirrors(lo, hi, false, ext_->mirror_infl, false);
  // the virtual ceiling (sdf_map.cpp:464-471) rewrites occupancy_buffer_ in one z row over the x,y extent of
  // the local bound: the callers' inline getOccupancy() must see it right away, as in the reference
  if (ext_->mirror_occ && mp_->virtual_ceil_height_ > -0.5) {
    const int ceil_id = (int)floor((mp_->virtual_ceil_height_ - mp_->map_origin_(2)) * mp_->resolution_inv_);
    if (ceil_id >= 0 && ceil_id < mp_->map_voxel_num_(2)) {
      Eigen::Vector3i clo = md_->local_bound_min_, chi = md_->local_bound_max_;
      clo(2) = chi(2) = ceil_id;
      syncMirrors(clo, chi, true, false, false);
    }
  }
}

void SDFMap::updateESDF3d() {
  warn("fuelmi_map_update_esdf", fuelmi_map_update_esdf(ext_->dev));
  syncMirrors(md_->local_bound_min_, md_->local_bound_max_, false, false, ext_->mirror_dist);
}

void SDFMap::resetBuffer() {
  warn("fuelmi_map_reset_buffer_all", fuelmi_map_reset_buffer_all(ext_->dev));
  pullBounds();
  syncMirrors(md_->local_bound_min_, md_->local_bound_max_, false, ext_->mirror_infl, ext_->mirror_dist);
}

void SDFMap::resetBuffer(const Eigen::Vector3d& min_pos, const Eigen::Vector3d& max_pos) {
  const double a[3] = {min_pos(0), min_pos(1), min_pos(2)}, b[3] = {max_pos(0), max_pos(1), max_pos(2)};
  warn("fuelmi_map_reset_buffer", fuelmi_map_reset_buffer(ext_->dev, a, b));
  Eigen::Vector3i lo, hi;
  posToIndex(min_pos, lo);
  posToIndex(max_pos, hi);
  boundIndex(lo);
  boundIndex(hi);
  syncMirrors(lo, hi, false, ext_->mirror_infl, ext_->mirror_dist);
}

void SDFMap::setOccupied(const Eigen::Vector3d& pos, const int& occ) {
  if (!isInMap(pos)) return;
  const double p[3] = {pos(0), pos(1), pos(2)};
  const int rc = fuelmi_map_set_occupied(ext_->dev, p, 1, occ);
  warn("fuelmi_map_set_occupied", rc);
  if (rc != FUELMI_OK) return;  // the device refused the value (only 0 / 1 exist there): the mirror must not diverge
  Eigen::Vector3i id;
  posToIndex(pos, id);
  md_->occupancy_buffer_inflate_[toAddress(id)] = (char)occ;
}

// Trilinear distance + gradient (sdf_map.cpp:497-536).  The reference's callers ask point by point inside host
// loops (traj_visibility.cpp:85,114,146, topo_prm.cpp:490): with the distance mirror on, the answer comes from
// the host copy -- eight loads like the reference, the device's own arithmetic on the same f64 values, hence
// the same doubles -- instead of a GPU round trip per point.  Batches go to the device (getDistWithGradBatch).
double SDFMap::getDistWithGrad(const Eigen::Vector3d& pos, Eigen::Vector3d& grad) {
  if (ext_->mirror_dist) {
    if (!isInMap(pos)) {
      grad = Eigen::Vector3d(0, 0, 0);
      return 0;
    }
    Eigen::Vector3i idx;
    double diff[3];
    for (int k = 0; k < 3; ++k) {
      const double pm = pos(k) - 0.5 * mp_->resolution_ * 1.0;
      idx(k) = (int)floor((pm - mp_->map_origin_(k)) * mp_->resolution_inv_);
      const double centre = (idx(k) + 0.5) * mp_->resolution_ + mp_->map_origin_(k);
      diff[k] = (pos(k) - centre) * mp_->resolution_inv_;
    }
    double v[2][2][2];
    for (int x = 0; x < 2; ++x)
      for (int y = 0; y < 2; ++y)
        for (int z = 0; z < 2; ++z) v[x][y][z] = getDistance(Eigen::Vector3i(idx(0) + x, idx(1) + y, idx(2) + z));
    const double ri = mp_->resolution_inv_;
    const double v00 = (1 - diff[0]) * v[0][0][0] + diff[0] * v[1][0][0];
    const double v01 = (1 - diff[0]) * v[0][0][1] + diff[0] * v[1][0][1];
    const double v10 = (1 - diff[0]) * v[0][1][0] + diff[0] * v[1][1][0];
    const double v11 = (1 - diff[0]) * v[0][1][1] + diff[0] * v[1][1][1];
    const double v0 = (1 - diff[1]) * v00 + diff[1] * v10;
    const double v1 = (1 - diff[1]) * v01 + diff[1] * v11;
    const double dist = (1 - diff[2]) * v0 + diff[2] * v1;
    grad(2) = (v1 - v0) * ri;
    grad(1) = ((1 - diff[2]) * (v10 - v00) + diff[2] * (v11 - v01)) * ri;
    double g0 = (1 - diff[2]) * (1 - diff[1]) * (v[1][0][0] - v[0][0][0]);
    g0 += (1 - diff[2]) * diff[1] * (v[1][1][0] - v[0][1][0]);
    g0 += diff[2] * (1 - diff[1]) * (v[1][0][1] - v[0][0][1]);
    g0 += diff[2] * diff[1] * (v[1][1][1] - v[0][1][1]);
    grad(0) = g0 * ri;
    return dist;
  }
  const double p[3] = {pos(0), pos(1), pos(2)};
  double d = 0.0, g[3] = {0, 0, 0};
  warn("fuelmi_map_dist_grad", fuelmi_map_dist_grad(ext_->dev, p, 1, &d, g));
  for (int k = 0; k < 3; ++k) grad(k) = g[k];
  return d;
}
void SDFMap::getDistWithGradBatch(const double* pos_xyz, int n, double* dist, double* grad_xyz) {
  warn("fuelmi_map_dist_grad", fuelmi_map_dist_grad(ext_->dev, pos_xyz, n, dist, grad_xyz));
}

void SDFMap::getRegion(Eigen::Vector3d& ori, Eigen::Vector3d& size) { ori = mp_->map_origin_, size = mp_->map_size_; }
void SDFMap::getBox(Eigen::Vector3d& bmin, Eigen::Vector3d& bmax) { bmin = mp_->box_mind_, bmax = mp_->box_maxd_; }
void SDFMap::getUpdatedBox(Eigen::Vector3d& bmin, Eigen::Vector3d& bmax, bool reset) {
  double a[3], b[3];
  fuelmi_map_get_updated_box(ext_->dev, a, b, reset ? 1 : 0);
  for (int k = 0; k < 3; ++k) bmin(k) = a[k], bmax(k) = b[k];
  if (reset) md_->reset_updated_box_ = true;
}
double SDFMap::getResolution() { return mp_->resolution_; }
int SDFMap::getVoxelNum() { return mp_->map_voxel_num_(0) * mp_->map_voxel_num_(1) * mp_->map_voxel_num_(2); }

// ------------------------------------------------------------------------------------------------
// EDTEnvironment (plan_env/src/edt_environment.cpp:7-20,78-97)
// ------------------------------------------------------------------------------------------------
void EDTEnvironment::init() {}
void EDTEnvironment::setMap(shared_ptr<SDFMap>& map) {
  sdf_map_ = map;
  resolution_inv_ = 1 / sdf_map_->getResolution();
}
void EDTEnvironment::setObjPrediction(ObjPrediction prediction) { obj_prediction_ = prediction; }
void EDTEnvironment::setObjScale(ObjScale scale) { obj_scale_ = scale; }
void EDTEnvironment::evaluateEDTWithGrad(const Eigen::Vector3d& pos, double, double& dist, Eigen::Vector3d& grad) {
  dist = sdf_map_->getDistWithGrad(pos, grad);
}
double EDTEnvironment::evaluateCoarseEDT(Eigen::Vector3d& pos, double) { return sdf_map_->getDistance(pos); }

// ------------------------------------------------------------------------------------------------
// FrontierFinder (grid part)
// ------------------------------------------------------------------------------------------------
FrontierFinder::FrontierFinder(const shared_ptr<EDTEnvironment>& edt, ros::NodeHandle& nh) : dev_(nullptr), edt_env_(edt) {
  nh.param("frontier/cluster_min", cluster_min_, -1);
  percep_utils_.reset(new PerceptionUtils(nh));  // for the callers (:44); the device has its own copy of the parameters
  resolution_ = edt_env_->sdf_map_->getResolution();
  double cluster_size_xy = -1.0;
  int down_sample = -1;
  nh.param("frontier/cluster_size_xy", cluster_size_xy, -1.0);
  nh.param("frontier/down_sample", down_sample, -1);
  fuelmi_frontier_cfg c;
  c.cluster_min = cluster_min_;
  c.min_z = 0.4;  // literal in frontier_finder.cpp:151
  c.cluster_size_xy = cluster_size_xy;
  c.down_sample = down_sample;
  // searchFrontiers ends with splitLargeFrontiers (:120); without its two parameters the search stops
  // at the region-grown clusters
  c.split = (cluster_size_xy > 0.0 && down_sample > 0) ? 1 : 0;
  // addition: frontier/reference_order.  1 lists cells_ in the reference's BFS order and sums average_ / the
  // VoxelGrid centroids in that order (bit-identical means, filtered_cells_, viewpoints and visib_num_) at the price
  // of a level-by-level sweep per search.  2 (default here): order 1 for every search whose clusters hold at most
  // 26624 cells each -- every incremental search of an exploration run, +0.15 ... 0.6 ms as measured on the
  // streaming workload -- and the address order for giant ones (a fresh full-box search: the sweep would cost
  // 24 ms on the 400x400x100 map).  0 is the address order throughout: the fastest path, means equal to the last
  // bits of a double, visib_num_ within a few cells (tests/test_golden_pillar.py asserts the bound).
  int ref_order = 2;
  nh.param("frontier/reference_order", ref_order, 2);
  c.reference_order = ref_order;
  warn("fuelmi_frontier_create", fuelmi_frontier_create(edt_env_->sdf_map_->device(), &c, &dev_));
  // viewpoint sampling parameters (frontier_finder.cpp:32-43, perception_utils.cpp:7-11)
  fuelmi_viewpoint_cfg v;
  nh.param("frontier/candidate_rmin", v.candidate_rmin, -1.0);
  nh.param("frontier/candidate_rmax", v.candidate_rmax, -1.0);
  nh.param("frontier/candidate_rnum", v.candidate_rnum, -1);
  nh.param("frontier/candidate_dphi", v.candidate_dphi, -1.0);
  nh.param("frontier/min_candidate_clearance", v.min_candidate_clearance, -1.0);
  nh.param("frontier/min_visib_num", v.min_visib_num, -1);
  nh.param("frontier/min_candidate_dist", v.min_candidate_dist, -1.0);
  nh.param("frontier/min_view_finish_fraction", v.min_view_finish_fraction, -1.0);
  nh.param("perception_utils/top_angle", v.top_angle, -1.0);
  nh.param("perception_utils/left_angle", v.left_angle, -1.0);
  nh.param("perception_utils/right_angle", v.right_angle, -1.0);
  nh.param("perception_utils/max_dist", v.max_dist, -1.0);
  min_candidate_dist_ = v.min_candidate_dist;
  have_viewpoints_ = c.split && dev_ && v.candidate_rnum > 0 && v.candidate_dphi > 0.0 && v.max_dist > 0.0;
  if (have_viewpoints_) warn("fuelmi_frontier_set_viewpoint_cfg", fuelmi_frontier_set_viewpoint_cfg(dev_, &v));
}
FrontierFinder::~FrontierFinder() {
  if (dev_) fuelmi_frontier_destroy(dev_);
}

// host copies of the device list `which`, from position `from` on (from = 0 replaces `out`)
void FrontierFinder::pull(int which, list<Frontier>& out, int from) {
  if (from == 0) {
    for (Frontier& old : out)  // (large cell buffers stay mapped for the clusters about to be built)
      if (old.cells_.capacity() >= 4096 && cells_spare_.size() < 8) cells_spare_.emplace_back(std::move(old.cells_));
    out.clear();
  }
  const int n = fuelmi_frontier_count(dev_, which);
  static_assert(sizeof(Vector3d) == 3 * sizeof(double), "cells_ is written as packed doubles");
  for (int k = from; k < n; ++k) {
    out.emplace_back();
    Frontier& f = out.back();  // (built in place: a 140 k-cell cluster is 3.4 MB, a copy of it a third of a plan cycle)
    const int sz = fuelmi_frontier_cluster_size(dev_, which, k);
    if (sz >= 4096) {  // the smallest spare buffer that holds the cluster
      int pick = -1;
      for (int i = 0; i < (int)cells_spare_.size(); ++i)
        if ((int)cells_spare_[i].capacity() >= sz && (pick < 0 || cells_spare_[i].capacity() < cells_spare_[pick].capacity())) pick = i;
      if (pick >= 0) {
        f.cells_.swap(cells_spare_[pick]);
        cells_spare_.erase(cells_spare_.begin() + pick);
      }
    }
    f.cells_.resize(sz);
    // voxel centres straight into the vector's storage (the library decodes them from its pinned result block)
    if (sz > 0) warn("fuelmi_frontier_cluster_centres", fuelmi_frontier_cluster_centres(dev_, which, k, f.cells_[0].data()));
    double info[9];
    fuelmi_frontier_cluster_info(dev_, which, k, info);
    for (int i = 0; i < 3; ++i) f.average_(i) = info[i], f.box_min_(i) = info[3 + i], f.box_max_(i) = info[6 + i];
    const int nf = fuelmi_frontier_cluster_filtered_size(dev_, which, k);
    if (nf > 0) {
      std::vector<float> xyz(3 * (size_t)nf);
      fuelmi_frontier_cluster_filtered(dev_, which, k, xyz.data());
      f.filtered_cells_.resize(nf);
      for (int i = 0; i < nf; ++i) f.filtered_cells_[i] = Vector3d(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
    }
    const int nvp = fuelmi_frontier_viewpoint_count(dev_, which, k);
    if (nvp > 0) {
      std::vector<double> py(4 * (size_t)nvp);
      std::vector<int> vis(nvp);
      fuelmi_frontier_viewpoints(dev_, which, k, py.data(), vis.data());
      f.viewpoints_.resize(nvp);
      for (int i = 0; i < nvp; ++i) {
        f.viewpoints_[i].pos_ = Vector3d(py[4 * i], py[4 * i + 1], py[4 * i + 2]);
        f.viewpoints_[i].yaw_ = py[4 * i + 3];
        f.viewpoints_[i].visib_num_ = vis[i];
      }
    }
    f.id_ = k;
  }
}

void FrontierFinder::searchFrontiers() {
  int n_new = 0;
  warn("fuelmi_frontier_search", fuelmi_frontier_search(dev_, &n_new));
  {
    // frontier/reference_order = 2 answers per search: say so the first time a search is delivered in the address
    // order instead of the reference's (a cluster too large for the in-LDS level sweep) -- never silently
    int os[4] = {0, 0, 0, 0};
    if (fuelmi_frontier_order_stats(dev_, os) == 0 && os[2] > 0 && !order_fallback_logged_) {
      order_fallback_logged_ = true;
      std::fprintf(stderr, "[fuelmi facade] FrontierFinder: search delivered in ascending-address cell order, not the "
                   "reference's BFS order: a cluster of %d cells exceeds the in-LDS level sweep (frontier/reference_order = 2; "
                   "set it to 1 to pay for the order on every search).  Further fallbacks are counted, not logged: "
                   "fuelmi_frontier_order_stats.\n", os[3]);
    }
  }
  pull(0, tmp_frontiers_);
  removed_ids_.resize(fuelmi_frontier_removed_count(dev_));
  if (!removed_ids_.empty()) fuelmi_frontier_removed_ids(dev_, removed_ids_.data());
  // clusters dropped as "changed" leave the persistent list; the survivors keep their records (and with
  // them costs_ / paths_).  removed_ids_ are positions in the list as it shrinks (:74-85).
  for (int id : removed_ids_) {
    if (id < 0 || id >= (int)frontiers_.size()) break;
    auto it = frontiers_.begin();
    std::advance(it, id);
    frontiers_.erase(it);
  }
  if ((int)frontiers_.size() != fuelmi_frontier_count(dev_, 1)) pull(1, frontiers_);  // lists out of step: start over
  first_new_ftr_ = frontiers_.end();
  pull(2, dormant_frontiers_);
}

void FrontierFinder::computeFrontiersToVisit() {
  const int before = (int)frontiers_.size();
  if (have_viewpoints_) {
    // reference (:392-423): sampleViewpoints per new cluster on the device; clusters with viewpoints
    // join frontiers_ (viewpoints sorted by coverage), the others dormant_frontiers_
    int na = 0, nd = 0;
    warn("fuelmi_frontier_compute_to_visit", fuelmi_frontier_compute_to_visit(dev_, &na, &nd));
    pull(1, frontiers_, before);
    pull(2, dormant_frontiers_);
  } else {
    // without the viewpoint parameters every new cluster becomes an active frontier
    warn("fuelmi_frontier_commit", fuelmi_frontier_commit(dev_, 0));
    frontiers_.insert(frontiers_.end(), tmp_frontiers_.begin(), tmp_frontiers_.end());
  }
  first_new_ftr_ = frontiers_.begin();
  std::advance(first_new_ftr_, std::min(before, (int)frontiers_.size()));
  int id = 0;
  for (auto& f : frontiers_) f.id_ = id++;  // :417-419
}

// ---- tour planning: the reference's bookkeeping (frontier_finder.cpp:258-324, 507-589) on the host lists;
// path costs come from the package's own ViewNode (A* through the map)
void FrontierFinder::updateFrontierCostMatrix() {
  if (!removed_ids_.empty()) {
    // every surviving old cluster forgets its entries towards the removed ones
    for (auto it = frontiers_.begin(); it != first_new_ftr_; ++it)
      for (int id : removed_ids_) {
        if (id < 0 || id >= (int)it->costs_.size()) break;
        auto c = it->costs_.begin();
        auto p = it->paths_.begin();
        std::advance(c, id);
        std::advance(p, id);
        it->costs_.erase(c);
        it->paths_.erase(p);
      }
    removed_ids_.clear();
  }
  // best viewpoint to best viewpoint, stored in both records (the way back is the same path reversed)
  auto link = [](Frontier& a, Frontier& b) {
    const Viewpoint& va = a.viewpoints_.front();
    const Viewpoint& vb = b.viewpoints_.front();
    vector<Vector3d> path;
    const double cost = ViewNode::computeCost(va.pos_, vb.pos_, va.yaw_, vb.yaw_, Vector3d(0, 0, 0), 0, path);
    a.costs_.push_back(cost);
    a.paths_.push_back(path);
    std::reverse(path.begin(), path.end());
    b.costs_.push_back(cost);
    b.paths_.push_back(path);
  };
  for (auto old = frontiers_.begin(); old != first_new_ftr_; ++old)
    for (auto fresh = first_new_ftr_; fresh != frontiers_.end(); ++fresh) link(*old, *fresh);
  for (auto a = first_new_ftr_; a != frontiers_.end(); ++a) {
    a->costs_.push_back(0);  // itself
    a->paths_.push_back({});
    auto b = a;
    for (++b; b != frontiers_.end(); ++b) link(*a, *b);
  }
  first_new_ftr_ = frontiers_.end();  // everything is linked now
}

void FrontierFinder::getFullCostMatrix(const Vector3d& cur_pos, const Vector3d& cur_vel, const Vector3d cur_yaw,
                                       Eigen::MatrixXd& mat) {
  // asymmetric TSP (:561-588): row/column 0 = the current state, nothing leads back to it
  const int n = (int)frontiers_.size();
  mat.resize(n + 1, n + 1);
  int i = 1;
  for (auto& ftr : frontiers_) {
    int j = 1;
    for (double cs : ftr.costs_) mat(i, j++) = cs;
    ++i;
  }
  mat.leftCols<1>().setZero();
  int j = 1;
  for (auto& ftr : frontiers_) {
    const Viewpoint& v = ftr.viewpoints_.front();
    vector<Vector3d> path;
    mat(0, j++) = ViewNode::computeCost(cur_pos, v.pos_, cur_yaw[0], v.yaw_, cur_vel, cur_yaw[1], path);
  }
}

void FrontierFinder::getPathForTour(const Vector3d& pos, const vector<int>& frontier_ids, vector<Vector3d>& path) {
  if (frontier_ids.empty()) return;
  vector<list<Frontier>::iterator> at;
  for (auto it = frontiers_.begin(); it != frontiers_.end(); ++it) at.push_back(it);
  vector<Vector3d> segment;
  ViewNode::searchPath(pos, at[frontier_ids[0]]->viewpoints_.front().pos_, segment);
  path.insert(path.end(), segment.begin(), segment.end());
  for (size_t k = 0; k + 1 < frontier_ids.size(); ++k) {  // stored path from tour stop k to stop k+1
    auto p = at[frontier_ids[k]]->paths_.begin();
    std::advance(p, frontier_ids[k + 1]);
    path.insert(path.end(), p->begin(), p->end());
  }
}

void FrontierFinder::setNextFrontier(const int& id) {  // declared by the reference, never defined there
  for (auto& ftr : frontiers_)
    if (ftr.id_ == id) next_frontier_ = ftr;
}

void FrontierFinder::getTopViewpointsInfo(const Vector3d& cur_pos, vector<Vector3d>& points, vector<double>& yaws,
                                          vector<Vector3d>& averages) {  // :425-450
  points.clear();
  yaws.clear();
  averages.clear();
  for (auto& frontier : frontiers_) {
    if (frontier.viewpoints_.empty()) continue;
    const Viewpoint* pick = &frontier.viewpoints_.front();  // all close: the best one
    for (auto& view : frontier.viewpoints_) {
      if ((view.pos_ - cur_pos).norm() < min_candidate_dist_) continue;
      pick = &view;
      break;
    }
    points.push_back(pick->pos_);
    yaws.push_back(pick->yaw_);
    averages.push_back(frontier.average_);
  }
}

void FrontierFinder::getViewpointsInfo(const Vector3d& cur_pos, const vector<int>& ids, const int& view_num,
                                       const double& max_decay, vector<vector<Vector3d>>& points,
                                       vector<vector<double>>& yaws) {  // :452-487
  points.clear();
  yaws.clear();
  for (auto id : ids) {
    for (auto& frontier : frontiers_) {
      if (frontier.id_ != id || frontier.viewpoints_.empty()) continue;
      vector<Vector3d> pts;
      vector<double> ys;
      const int visib_thresh = frontier.viewpoints_.front().visib_num_ * max_decay;
      for (auto& view : frontier.viewpoints_) {
        if ((int)pts.size() >= view_num || view.visib_num_ <= visib_thresh) break;
        if ((view.pos_ - cur_pos).norm() < min_candidate_dist_) continue;
        pts.push_back(view.pos_);
        ys.push_back(view.yaw_);
      }
      if (pts.empty()) {  // all viewpoints are very close: take them regardless of the distance
        for (auto& view : frontier.viewpoints_) {
          if ((int)pts.size() >= view_num || view.visib_num_ <= visib_thresh) break;
          pts.push_back(view.pos_);
          ys.push_back(view.yaw_);
        }
      }
      points.push_back(pts);
      yaws.push_back(ys);
    }
  }
}

bool FrontierFinder::isFrontierCovered() {  // :697-719
  int covered = 0;
  warn("fuelmi_frontier_is_covered", fuelmi_frontier_is_covered(dev_, &covered));
  return covered != 0;
}

void FrontierFinder::getFrontiers(vector<vector<Vector3d>>& clusters) {
  clusters.clear();
  for (auto& f : frontiers_) clusters.push_back(f.cells_);
}
void FrontierFinder::getDormantFrontiers(vector<vector<Vector3d>>& clusters) {
  clusters.clear();
  for (auto& f : dormant_frontiers_) clusters.push_back(f.cells_);
}
void FrontierFinder::getFrontierBoxes(vector<pair<Vector3d, Vector3d>>& boxes) {
  boxes.clear();
  for (auto& f : frontiers_) boxes.push_back(std::make_pair((f.box_max_ + f.box_min_) * 0.5, f.box_max_ - f.box_min_));
}
void FrontierFinder::wrapYaw(double& yaw) {
  while (yaw < -M_PI) yaw += 2 * M_PI;
  while (yaw > M_PI) yaw -= 2 * M_PI;
}

// ------------------------------------------------------------------------------------------------
// BsplineOptimizer
// ------------------------------------------------------------------------------------------------
const int BsplineOptimizer::SMOOTHNESS = FUELMI_COST_SMOOTHNESS;
const int BsplineOptimizer::DISTANCE = FUELMI_COST_DISTANCE;
const int BsplineOptimizer::FEASIBILITY = FUELMI_COST_FEASIBILITY;
const int BsplineOptimizer::START = FUELMI_COST_START;
const int BsplineOptimizer::END = FUELMI_COST_END;
const int BsplineOptimizer::GUIDE = FUELMI_COST_GUIDE;
const int BsplineOptimizer::WAYPOINTS = FUELMI_COST_WAYPOINTS;
const int BsplineOptimizer::VIEWCONS = FUELMI_COST_VIEWCONS;
const int BsplineOptimizer::MINTIME = FUELMI_COST_MINTIME;
const int BsplineOptimizer::GUIDE_PHASE =
    BsplineOptimizer::SMOOTHNESS | BsplineOptimizer::GUIDE | BsplineOptimizer::START | BsplineOptimizer::END;
const int BsplineOptimizer::NORMAL_PHASE = BsplineOptimizer::SMOOTHNESS | BsplineOptimizer::DISTANCE |
    BsplineOptimizer::FEASIBILITY | BsplineOptimizer::START | BsplineOptimizer::END;

void BsplineOptimizer::setParam(ros::NodeHandle& nh) {
  nh.param("optimization/ld_smooth", cfg_.ld_smooth, -1.0);
  nh.param("optimization/ld_dist", cfg_.ld_dist, -1.0);
  nh.param("optimization/ld_feasi", cfg_.ld_feasi, -1.0);
  nh.param("optimization/ld_start", cfg_.ld_start, -1.0);
  nh.param("optimization/ld_end", cfg_.ld_end, -1.0);
  nh.param("optimization/ld_guide", cfg_.ld_guide, -1.0);
  nh.param("optimization/ld_waypt", cfg_.ld_waypt, -1.0);
  nh.param("optimization/ld_view", cfg_.ld_view, -1.0);
  nh.param("optimization/ld_time", cfg_.ld_time, -1.0);
  nh.param("optimization/dist0", cfg_.dist0, -1.0);
  nh.param("optimization/max_vel", cfg_.max_vel, -1.0);
  nh.param("optimization/max_acc", cfg_.max_acc, -1.0);
  nh.param("optimization/dlmin", cfg_.dlmin, -1.0);
  nh.param("optimization/wnl", cfg_.wnl, -1.0);
  const char* nums[4] = {"optimization/max_iteration_num1", "optimization/max_iteration_num2",
                         "optimization/max_iteration_num3", "optimization/max_iteration_num4"};
  const char* times[4] = {"optimization/max_iteration_time1", "optimization/max_iteration_time2",
                          "optimization/max_iteration_time3", "optimization/max_iteration_time4"};
  for (int i = 0; i < 4; ++i) {
    nh.param(nums[i], max_iteration_num_[i], -1);
    nh.param(times[i], max_iteration_time_[i], -1.0);
  }
  nh.param("optimization/algorithm1", algorithm1_, -1);
  nh.param("optimization/algorithm2", algorithm2_, -1);
  nh.param("manager/bspline_degree", bspline_degree_, 3);
  cfg_.bspline_degree = bspline_degree_;
  time_lb_ = -1;
  static bool told = false;
  if (!told && (algorithm1_ >= 0 || algorithm2_ >= 0)) {
    told = true;
    std::fprintf(stderr,
                 "[fuelmi] BsplineOptimizer: optimization/algorithm1,2 (NLopt ids %d, %d) are not used -- every solve is "
                 "one device launch of a box-projected L-BFGS under max_iteration_num* and max_iteration_time*\n",
                 algorithm1_, algorithm2_);
  }
}

void BsplineOptimizer::setEnvironment(const shared_ptr<EDTEnvironment>& env) {
  edt_environment_ = env;
  dynamic_ = false;
}
void BsplineOptimizer::setCostFunction(const int& cost_code) { cost_function_ = cost_code; }
void BsplineOptimizer::setGuidePath(const vector<Eigen::Vector3d>& guide_pt) { guide_pts_ = guide_pt; }
void BsplineOptimizer::setWaypoints(const vector<Eigen::Vector3d>& waypts, const vector<int>& waypt_idx) {
  waypoints_ = waypts;
  waypt_idx_ = waypt_idx;
}
void BsplineOptimizer::setViewConstraint(const ViewConstraint& vc) { view_cons_ = vc; }
void BsplineOptimizer::enableDynamic(double time_start) {
  dynamic_ = true;  // moving obstacles are out of scope: the static ESDF is used regardless
  start_time_ = time_start;
}
void BsplineOptimizer::setBoundaryStates(const vector<Eigen::Vector3d>& start, const vector<Eigen::Vector3d>& end) {
  start_state_ = start;
  end_state_ = end;
}
void BsplineOptimizer::setTimeLowerBound(const double& lb) { time_lb_ = lb; }

namespace {
struct BatchStore {  // keeps the arrays a one-candidate fuelmi_bspline_batch points to
  std::vector<double> st, en, guide, wp, vpt, vdir;
  double tlb;
};
}  // namespace
// one-candidate batch from the optimiser's members (what combineCost reads, :518-691)
#define FUELMI_FILL_BATCH(b, S, XV)                                                                      \
  S.st.assign(9, 0.0), S.en.assign(9, 0.0), S.vpt.assign(3, 0.0), S.vdir.assign(3, 0.0);                 \
  for (size_t i = 0; i < start_state_.size() && i < 3; ++i)                                              \
    for (int k = 0; k < 3; ++k) S.st[3 * i + k] = start_state_[i](k);                                    \
  for (size_t i = 0; i < end_state_.size() && i < 3; ++i)                                                \
    for (int k = 0; k < 3; ++k) S.en[3 * i + k] = end_state_[i](k);                                      \
  for (auto& g : guide_pts_)                                                                             \
    for (int k = 0; k < 3; ++k) S.guide.push_back(g(k));                                                 \
  for (auto& w : waypoints_)                                                                             \
    for (int k = 0; k < 3; ++k) S.wp.push_back(w(k));                                                    \
  for (int k = 0; k < 3; ++k) S.vpt[k] = view_cons_.pt_(k), S.vdir[k] = view_cons_.dir_(k);              \
  S.tlb = time_lb_;                                                                                      \
  b.cost_function = cost_function_;                                                                      \
  b.dim = dim_;                                                                                          \
  b.point_num = point_num_;                                                                              \
  b.n_traj = 1;                                                                                          \
  b.x = (XV).data();                                                                                     \
  b.pt_dist = &pt_dist_;                                                                                 \
  b.knot_span = &knot_span_;                                                                             \
  b.time_lb = &S.tlb;                                                                                    \
  b.start_state = S.st.data();                                                                           \
  b.end_state = S.en.data();                                                                             \
  b.end_n = (int)std::max<size_t>(1, std::min<size_t>(3, end_state_.size()));                            \
  b.guide_pts = S.guide.empty() ? nullptr : S.guide.data();                                              \
  b.waypoints = S.wp.empty() ? nullptr : S.wp.data();                                                    \
  b.waypt_idx = waypt_idx_.empty() ? nullptr : waypt_idx_.data();                                        \
  b.n_waypt = (int)waypoints_.size();                                                                    \
  b.view_pt = S.vpt.data();                                                                              \
  b.view_dir = S.vdir.data();                                                                            \
  b.view_idx = &view_cons_.idx_;

void BsplineOptimizer::combineCost(const std::vector<double>& x, std::vector<double>& grad, double& cost) {
  auto t1 = std::chrono::steady_clock::now();
  fuelmi_bspline_batch b;
  BatchStore S;
  FUELMI_FILL_BATCH(b, S, x)
  grad.assign(variable_num_, 0.0);
  warn("fuelmi_bspline_cost_grad",
       fuelmi_bspline_cost_grad(edt_environment_->sdf_map_->device(), &cfg_, &b, &cost, grad.data()));
  comb_time += std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
}

void BsplineOptimizer::optimize(Eigen::MatrixXd& points, double& dt, const int& cost_function,
                                const int& max_num_id, const int& max_time_id) {
  if (start_state_.empty()) {
    ROS_ERROR("Initial state undefined!");
    return;
  }
  control_points_ = points;
  knot_span_ = dt;
  max_num_id_ = max_num_id;
  max_time_id_ = max_time_id;
  setCostFunction(cost_function);
  dim_ = control_points_.cols();
  order_ = (dim_ == 1) ? 3 : bspline_degree_;
  point_num_ = control_points_.rows();
  optimize_time_ = cost_function_ & MINTIME;
  variable_num_ = optimize_time_ ? dim_ * point_num_ + 1 : dim_ * point_num_;
  if (variable_num_ <= 0) {
    ROS_ERROR("Empty varibale to optimization solver.");
    return;
  }
  pt_dist_ = 0.0;
  for (int i = 0; i < point_num_ - 1; ++i) {
    double s = 0.0;
    for (int j = 0; j < dim_; ++j) {
      const double e = control_points_(i + 1, j) - control_points_(i, j);
      s += e * e;
    }
    pt_dist_ += std::sqrt(s);
  }
  pt_dist_ /= double(point_num_);
  iter_num_ = 0;
  min_cost_ = std::numeric_limits<double>::max();
  comb_time = 0.0;
  optimize();
  points = control_points_;
  dt = knot_span_;
  start_state_.clear();
  time_lb_ = -1;
}

// Box-projected L-BFGS (memory 8, Armijo backtracking) under the reference's stopping criteria.
// Replaces nlopt::opt (bspline_optimizer.cpp:165-229); iterates are NOT NLopt's.
void BsplineOptimizer::optimize() {
  // The reference hands the variables to NLopt (:165-253).  Here the whole solve runs in one kernel
  // launch on the device (fuelmi_bspline_dev_optimize: start clamping, bounds, best-variable tracking,
  // evaluation cap and xtol_rel as the reference configures them; box-projected L-BFGS in place of
  // NLopt's LD_LBFGS / LD_TNEWTON).  max_iteration_time_ is the solver's wall-clock cap (set_maxtime, :170-172):
  // the device checks its clock between evaluations and returns the best variables seen when it runs out.
  const int n = variable_num_;
  std::vector<double> q(n);
  for (int i = 0; i < point_num_; ++i)
    for (int j = 0; j < dim_; ++j) q[dim_ * i + j] = control_points_(i, j);
  if (optimize_time_) q[n - 1] = knot_span_;
  auto t1 = std::chrono::steady_clock::now();
  fuelmi_bspline_batch b;
  BatchStore S;
  FUELMI_FILL_BATCH(b, S, q)
  // one call on a query slot of the map: no device allocation, own side stream -- the ten optimiser threads of
  // topoReplan (planner_manager.cpp:446-453) solve side by side
  best_variable_.assign(n, 0.0);
  double cost = 0.0;
  int evals = 0;
  const int rc = fuelmi_bspline_optimize(edt_environment_->sdf_map_->device(), &cfg_, &b, std::max(1, max_iteration_num_[max_num_id_]),
                                         max_iteration_time_[max_time_id_], best_variable_.data(), &cost, &evals);
  warn("fuelmi_bspline_optimize", rc);
  comb_time += std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
  if (rc) return;
  iter_num_ = evals;
  min_cost_ = cost;
  for (int i = 0; i < point_num_; ++i)
    for (int j = 0; j < dim_; ++j) control_points_(i, j) = best_variable_[dim_ * i + j];
  if (optimize_time_) knot_span_ = best_variable_[n - 1];
}

vector<Eigen::Vector3d> BsplineOptimizer::matrixToVectors(const Eigen::MatrixXd& ctrl_pts) {
  vector<Eigen::Vector3d> out;
  for (int i = 0; i < ctrl_pts.rows(); ++i) {
    Eigen::Vector3d p(0, 0, 0);
    for (int j = 0; j < ctrl_pts.cols() && j < 3; ++j) p(j) = ctrl_pts(i, j);
    out.push_back(p);
  }
  return out;
}
Eigen::MatrixXd BsplineOptimizer::getControlPoints() { return control_points_; }
bool BsplineOptimizer::isQuadratic() {
  return cost_function_ == GUIDE_PHASE || cost_function_ == SMOOTHNESS || cost_function_ == (SMOOTHNESS | WAYPOINTS);
}

}  // namespace fast_planner
