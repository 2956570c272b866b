// plan_env/sdf_map.h -- drop-in replacement of the reference header
// (fuel_planner/plan_env/include/plan_env/sdf_map.h:27-266): same namespace, class name, public
// API and inline getters, but the grid lives on an MI355X behind libfuelmi (include/fuelmi.h).
//
// The reference's inline getters read host std::vectors (sdf_map.h:196-237) and are compiled into
// the callers' objects (A*, kino-A*, ViewNode, FSM ...).  This header keeps that contract down to the
// layout: SDFMap's data members, MapParam and MapData have the reference's fields in the reference's order
// (round 5; tests/test_abi_cpu.py compares every offset with the reference's own header).  MapData
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
struct MapData;
struct MapParam;
class MapROS;

// shorthand for the Eigen types of the reference's signatures (aliases: the mangled names are unchanged)
typedef Eigen::Vector3d V3d;
typedef Eigen::Vector3i V3i;

class SDFMap {
public:
  typedef std::shared_ptr<SDFMap> Ptr;
  enum OCCUPANCY { UNKNOWN, FREE, OCCUPIED };

  SDFMap();
  ~SDFMap();

  // -- life cycle and the mutators MapROS / the planners call (device work) --
  void initMap(ros::NodeHandle& node);
  void inputPointCloud(const pcl::PointCloud<pcl::PointXYZ>& cloud, const int& n_points, const V3d& cam);
  void updateESDF3d();
  void resetBuffer();
  void resetBuffer(const V3d& lo, const V3d& hi);
  void setOccupied(const V3d& where, const int& value = 1);

  // -- queries answered by the device --
  double getDistWithGrad(const V3d& where, V3d& gradient);

  // -- map geometry --
  double getResolution();
  int getVoxelNum();
  void getRegion(V3d& origin, V3d& extent);
  void getBox(V3d& lo, V3d& hi);
  void getUpdatedBox(V3d& lo, V3d& hi, bool reset = false);

  // -- inline helpers compiled into the callers (they read the host mirrors in MapData) --
  int toAddress(const int& ix, const int& iy, const int& iz);
  int toAddress(const V3i& cell);
  void posToIndex(const V3d& where, V3i& cell);
  void indexToPos(const V3i& cell, V3d& centre);
  void boundIndex(V3i& cell);
  void boundBox(V3d& lo, V3d& hi);
  bool isInMap(const V3i& cell);
  bool isInMap(const V3d& where);
  bool isInBox(const V3i& cell);
  bool isInBox(const V3d& where);
  int getOccupancy(const V3i& cell);
  int getOccupancy(const V3d& where);
  int getInflateOccupancy(const V3i& cell);
  int getInflateOccupancy(const V3d& where);
  double getDistance(const V3i& cell);
  double getDistance(const V3d& where);

  // -- additions of this implementation (not in the reference) --
  // device handle for the FrontierFinder / BsplineOptimizer facades
  fuelmi_map* device() const { return ext_ ? ext_->dev : nullptr; }
  // which host mirrors the mutators keep coherent (default: all, as the reference's getters need)
  void setHostMirror(bool occupancy, bool inflate, bool distance);
  // batched getDistWithGrad for n positions (xyz packed), one kernel launch
  void getDistWithGradBatch(const double* pos_xyz, int n, double* dist, double* grad_xyz);

  EIGEN_MAKE_ALIGNED_OPERATOR_NEW

private:
  friend MapROS;  // calls clearAndInflateLocalMap() and reads mp_ / md_ like in the reference

  void clearAndInflateLocalMap();
  void pullBounds();
  void syncMirrors(const V3i& lo, const V3i& hi, bool occ, bool infl, bool dist);

  // Data members: the reference's four, in its order and of pointer size each (sdf_map.h:74-77) -- the inline getters
  // below are compiled into the callers and reach mp_ / md_ at these offsets -- then ONE pointer to this
  // implementation's state ("new state behind an opaque pointer appended at the end").  mr_ / caster_ are
  // never set by this library (MapROS is the caller's, the ray caster lives on the device); their deleters are declared
  // here and defined in the library so that the header needs neither class.
  struct MapROSDelete {
    void operator()(MapROS*) const;
  };
  struct RayCasterDelete {
    void operator()(RayCaster*) const;
  };
  struct Ext {
    fuelmi_map* dev;
    bool mirror_occ, mirror_infl, mirror_dist;
  };
  unique_ptr<MapParam> mp_;
  unique_ptr<MapData> md_;
  unique_ptr<MapROS, MapROSDelete> mr_;
  unique_ptr<RayCaster, RayCasterDelete> caster_;
  Ext* ext_;
};

// The reference's MapParam / MapData, field for field: order, names and types of plan_env/sdf_map.h:86-125
// (tests/test_abi_cpu.py compiles one translation unit against the reference's header and one against this one and
// compares sizeof / offsetof of every field).  What the callers' inline getters and MapROS read -- occupancy_buffer_,
// occupancy_buffer_inflate_, distance_buffer_, the bounds, the updated box -- are HOST MIRRORS the mutators refresh
// from the device; the buffers only the reference's CPU algorithms used (distance_buffer_neg_, tmp_buffer*, count_*,
// flag_*, cache_voxel_) stay empty: that state lives on the device.
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
  // main map data (host mirrors of the device planes)
  std::vector<double> occupancy_buffer_;
  std::vector<char> occupancy_buffer_inflate_;
  std::vector<double> distance_buffer_neg_;  // (empty: signed distances are merged on the device)
  std::vector<double> distance_buffer_;
  std::vector<double> tmp_buffer1_;  // (empty)
  std::vector<double> tmp_buffer2_;  // (empty)
  // data for updating (empty: the fusion's bookkeeping lives on the device)
  vector<short> count_hit_, count_miss_, count_hit_and_miss_;
  vector<char> flag_rayend_, flag_visited_;
  char raycast_num_;
  queue<int> cache_voxel_;
  Eigen::Vector3i local_bound_min_, local_bound_max_;
  Eigen::Vector3d update_min_, update_max_;
  bool reset_updated_box_;

  EIGEN_MAKE_ALIGNED_OPERATOR_NEW
};

// ---- inline helpers.  The arithmetic is the reference's (sdf_map.h:127-237: floor((p - origin) *
// res_inv), centre = (i + 0.5) * res + origin, 1e-4 margins of isInMap, half-open index box, open
// position box, the three-way occupancy thresholds); they read the host mirrors above. ----
inline int SDFMap::toAddress(const int& ix, const int& iy, const int& iz) {
  const V3i& n = mp_->map_voxel_num_;
  return (ix * n(1) + iy) * n(2) + iz;
}
inline int SDFMap::toAddress(const V3i& cell) { return toAddress(cell(0), cell(1), cell(2)); }
inline void SDFMap::posToIndex(const V3d& where, V3i& cell) {
  for (int a = 0; a < 3; ++a) cell(a) = (int)floor((where(a) - mp_->map_origin_(a)) * mp_->resolution_inv_);
}
inline void SDFMap::indexToPos(const V3i& cell, V3d& centre) {
  for (int a = 0; a < 3; ++a) centre(a) = (cell(a) + 0.5) * mp_->resolution_ + mp_->map_origin_(a);
}
inline void SDFMap::boundIndex(V3i& cell) {
  for (int a = 0; a < 3; ++a) cell(a) = std::max(std::min(cell(a), mp_->map_voxel_num_(a) - 1), 0);
}
inline void SDFMap::boundBox(V3d& lo, V3d& hi) {
  for (int a = 0; a < 3; ++a) {
    lo(a) = std::max(lo(a), mp_->box_mind_(a));
    hi(a) = std::min(hi(a), mp_->box_maxd_(a));
  }
}
inline bool SDFMap::isInMap(const V3i& cell) {
  bool inside = true;
  for (int a = 0; a < 3; ++a) inside = inside && cell(a) >= 0 && cell(a) <= mp_->map_voxel_num_(a) - 1;
  return inside;
}
inline bool SDFMap::isInMap(const V3d& where) {
  bool inside = true;
  for (int a = 0; a < 3; ++a)
    inside = inside && !(where(a) < mp_->map_min_boundary_(a) + 1e-4) && !(where(a) > mp_->map_max_boundary_(a) - 1e-4);
  return inside;
}
inline bool SDFMap::isInBox(const V3i& cell) {
  bool inside = true;
  for (int a = 0; a < 3; ++a) inside = inside && cell(a) >= mp_->box_min_(a) && cell(a) < mp_->box_max_(a);
  return inside;
}
inline bool SDFMap::isInBox(const V3d& where) {
  bool inside = true;
  for (int a = 0; a < 3; ++a) inside = inside && where(a) > mp_->box_mind_(a) && where(a) < mp_->box_maxd_(a);
  return inside;
}
inline int SDFMap::getOccupancy(const V3i& cell) {
  if (!isInMap(cell)) return -1;
  const double logodds = md_->occupancy_buffer_[toAddress(cell)];
  if (logodds < mp_->clamp_min_log_ - 1e-3) return UNKNOWN;
  return logodds > mp_->min_occupancy_log_ ? OCCUPIED : FREE;
}
inline int SDFMap::getInflateOccupancy(const V3i& cell) {
  return isInMap(cell) ? int(md_->occupancy_buffer_inflate_[toAddress(cell)]) : -1;
}
inline double SDFMap::getDistance(const V3i& cell) {
  return isInMap(cell) ? md_->distance_buffer_[toAddress(cell)] : -1.0;
}
// position overloads: index of the voxel containing the position, then the cell version
#define FUELMI_BY_POSITION(RET, NAME)          \
  inline RET SDFMap::NAME(const V3d& where) {  \
    V3i cell;                                  \
    posToIndex(where, cell);                   \
    return NAME(cell);                         \
  }
FUELMI_BY_POSITION(int, getOccupancy)
FUELMI_BY_POSITION(int, getInflateOccupancy)
FUELMI_BY_POSITION(double, getDistance)
#undef FUELMI_BY_POSITION
}  // namespace fast_planner
#endif
