// plan_env/sdf_map.h -- drop-in replacement of the reference header
// (fuel_planner/plan_env/include/plan_env/sdf_map.h:27-266): same namespace, class name, public
// API and inline getters, but the grid lives on an MI355X behind libfuelmi (include/fuelmi.h).
//
// The reference's inline getters read host std::vectors (sdf_map.h:196-237) and are compiled into
// the callers' objects (A*, kino-A*, ViewNode, FSM ...).  This header keeps that contract: MapData
// still owns occupancy_buffer_ / occupancy_buffer_inflate_ / distance_buffer_ as HOST MIRRORS that
// every mutator refreshes from the device for the box it touched (fuelmi_map_sync_host).  Callers
// that only use the GPU path (FrontierFinder / BsplineOptimizer facades) can switch the mirrors off
// with setHostMirror() and save the PCIe traffic.
#ifndef _SDF_MAP_H
#define _SDF_MAP_H

#include <Eigen/Eigen>
#include <Eigen/StdVector>

#include <cmath>
#include <memory>
#include <queue>
#include <tuple>
#include <vector>

#include <ros/ros.h>

#include <pcl/point_cloud.h>
#include <pcl/point_types.h>

#include "fuelmi.h"

using namespace std;

namespace cv {
class Mat;
}
class RayCaster;

namespace fast_planner {
struct MapParam;
struct MapData;
class MapROS;

class SDFMap {
public:
  SDFMap();
  ~SDFMap();

  enum OCCUPANCY { UNKNOWN, FREE, OCCUPIED };

  void initMap(ros::NodeHandle& nh);
  void inputPointCloud(const pcl::PointCloud<pcl::PointXYZ>& points, const int& point_num,
                       const Eigen::Vector3d& camera_pos);

  void posToIndex(const Eigen::Vector3d& pos, Eigen::Vector3i& id);
  void indexToPos(const Eigen::Vector3i& id, Eigen::Vector3d& pos);
  void boundIndex(Eigen::Vector3i& id);
  int toAddress(const Eigen::Vector3i& id);
  int toAddress(const int& x, const int& y, const int& z);
  bool isInMap(const Eigen::Vector3d& pos);
  bool isInMap(const Eigen::Vector3i& idx);
  bool isInBox(const Eigen::Vector3i& id);
  bool isInBox(const Eigen::Vector3d& pos);
  void boundBox(Eigen::Vector3d& low, Eigen::Vector3d& up);
  int getOccupancy(const Eigen::Vector3d& pos);
  int getOccupancy(const Eigen::Vector3i& id);
  void setOccupied(const Eigen::Vector3d& pos, const int& occ = 1);
  int getInflateOccupancy(const Eigen::Vector3d& pos);
  int getInflateOccupancy(const Eigen::Vector3i& id);
  double getDistance(const Eigen::Vector3d& pos);
  double getDistance(const Eigen::Vector3i& id);
  double getDistWithGrad(const Eigen::Vector3d& pos, Eigen::Vector3d& grad);
  void updateESDF3d();
  void resetBuffer();
  void resetBuffer(const Eigen::Vector3d& min, const Eigen::Vector3d& max);

  void getRegion(Eigen::Vector3d& ori, Eigen::Vector3d& size);
  void getBox(Eigen::Vector3d& bmin, Eigen::Vector3d& bmax);
  void getUpdatedBox(Eigen::Vector3d& bmin, Eigen::Vector3d& bmax, bool reset = false);
  double getResolution();
  int getVoxelNum();

  // ---- additions of this implementation (not in the reference) ----
  // device handle for the FrontierFinder / BsplineOptimizer facades
  fuelmi_map* device() const { return dev_; }
  // which host mirrors the mutators keep coherent (default: all, as the reference's getters need)
  void setHostMirror(bool occupancy, bool inflate, bool distance);
  // batched SDFMap::getDistWithGrad for n positions (xyz packed), one kernel launch
  void getDistWithGradBatch(const double* pos_xyz, int n, double* dist, double* grad_xyz);

private:
  void clearAndInflateLocalMap();
  void syncMirrors(const Eigen::Vector3i& bmin, const Eigen::Vector3i& bmax, bool occ, bool infl, bool dist);
  void pullBounds();

  unique_ptr<MapParam> mp_;
  unique_ptr<MapData> md_;
  fuelmi_map* dev_;
  bool mirror_occ_, mirror_infl_, mirror_dist_;

  friend MapROS;

public:
  typedef std::shared_ptr<SDFMap> Ptr;
  EIGEN_MAKE_ALIGNED_OPERATOR_NEW
};

struct MapParam {
  // map properties
  Eigen::Vector3d map_origin_, map_size_;
  Eigen::Vector3d map_min_boundary_, map_max_boundary_;
  Eigen::Vector3i map_voxel_num_;
  double resolution_, resolution_inv_;
  double obstacles_inflation_;
  double virtual_ceil_height_, ground_height_;
  Eigen::Vector3i box_min_, box_max_;
  Eigen::Vector3d box_mind_, box_maxd_;
  double default_dist_;
  bool optimistic_, signed_dist_;
  // map fusion
  double p_hit_, p_miss_, p_min_, p_max_, p_occ_;
  double prob_hit_log_, prob_miss_log_, clamp_min_log_, clamp_max_log_, min_occupancy_log_;
  double max_ray_length_;
  double local_bound_inflate_;
  int local_map_margin_;
  double unknown_flag_;
};

struct MapData {
  // host mirrors of the device grid (same names and element types as the reference)
  std::vector<double> occupancy_buffer_;
  std::vector<char> occupancy_buffer_inflate_;
  std::vector<double> distance_buffer_;
  Eigen::Vector3i local_bound_min_, local_bound_max_;
  Eigen::Vector3d update_min_, update_max_;
  bool reset_updated_box_;

  EIGEN_MAKE_ALIGNED_OPERATOR_NEW
};

// ---- inline helpers: same arithmetic as the reference (sdf_map.h:127-237), own wording ----
inline void SDFMap::posToIndex(const Eigen::Vector3d& pos, Eigen::Vector3i& id) {
  for (int k = 0; k < 3; ++k) id(k) = (int)floor((pos(k) - mp_->map_origin_(k)) * mp_->resolution_inv_);
}
inline void SDFMap::indexToPos(const Eigen::Vector3i& id, Eigen::Vector3d& pos) {
  for (int k = 0; k < 3; ++k) pos(k) = (id(k) + 0.5) * mp_->resolution_ + mp_->map_origin_(k);
}
inline void SDFMap::boundIndex(Eigen::Vector3i& id) {
  for (int k = 0; k < 3; ++k) id(k) = std::max(std::min(id(k), mp_->map_voxel_num_(k) - 1), 0);
}
inline int SDFMap::toAddress(const int& x, const int& y, const int& z) {
  return (x * mp_->map_voxel_num_(1) + y) * mp_->map_voxel_num_(2) + z;
}
inline int SDFMap::toAddress(const Eigen::Vector3i& id) { return toAddress(id(0), id(1), id(2)); }
inline bool SDFMap::isInMap(const Eigen::Vector3d& pos) {
  for (int k = 0; k < 3; ++k)
    if (pos(k) < mp_->map_min_boundary_(k) + 1e-4 || pos(k) > mp_->map_max_boundary_(k) - 1e-4) return false;
  return true;
}
inline bool SDFMap::isInMap(const Eigen::Vector3i& idx) {
  for (int k = 0; k < 3; ++k)
    if (idx(k) < 0 || idx(k) > mp_->map_voxel_num_(k) - 1) return false;
  return true;
}
inline bool SDFMap::isInBox(const Eigen::Vector3i& id) {
  for (int k = 0; k < 3; ++k)
    if (id(k) < mp_->box_min_(k) || id(k) >= mp_->box_max_(k)) return false;
  return true;
}
inline bool SDFMap::isInBox(const Eigen::Vector3d& pos) {
  for (int k = 0; k < 3; ++k)
    if (pos(k) <= mp_->box_mind_(k) || pos(k) >= mp_->box_maxd_(k)) return false;
  return true;
}
inline void SDFMap::boundBox(Eigen::Vector3d& low, Eigen::Vector3d& up) {
  for (int k = 0; k < 3; ++k) {
    low(k) = std::max(low(k), mp_->box_mind_(k));
    up(k) = std::min(up(k), mp_->box_maxd_(k));
  }
}
inline int SDFMap::getOccupancy(const Eigen::Vector3i& id) {
  if (!isInMap(id)) return -1;
  const double o = md_->occupancy_buffer_[toAddress(id)];
  if (o < mp_->clamp_min_log_ - 1e-3) return UNKNOWN;
  return o > mp_->min_occupancy_log_ ? OCCUPIED : FREE;
}
inline int SDFMap::getOccupancy(const Eigen::Vector3d& pos) {
  Eigen::Vector3i id;
  posToIndex(pos, id);
  return getOccupancy(id);
}
inline int SDFMap::getInflateOccupancy(const Eigen::Vector3i& id) {
  if (!isInMap(id)) return -1;
  return int(md_->occupancy_buffer_inflate_[toAddress(id)]);
}
inline int SDFMap::getInflateOccupancy(const Eigen::Vector3d& pos) {
  Eigen::Vector3i id;
  posToIndex(pos, id);
  return getInflateOccupancy(id);
}
inline double SDFMap::getDistance(const Eigen::Vector3i& id) {
  if (!isInMap(id)) return -1;
  return md_->distance_buffer_[toAddress(id)];
}
inline double SDFMap::getDistance(const Eigen::Vector3d& pos) {
  Eigen::Vector3i id;
  posToIndex(pos, id);
  return getDistance(id);
}
}  // namespace fast_planner
#endif
