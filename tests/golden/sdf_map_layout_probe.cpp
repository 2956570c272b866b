// Layout probe of the drop-in's header-level ABI: prints sizeof / offsetof of every field of
// fast_planner::MapParam, fast_planner::MapData and of SDFMap's data members.  Compiled twice by
// tests/test_abi_cpu.py -- against the reference's plan_env/sdf_map.h and against fuel_amd/facade/plan_env/sdf_map.h,
// with the same Eigen / ROS / PCL header stand-ins -- and the two outputs must agree line for line
// (tests/golden/sdf_map_layout.txt is the reference's output, written by the same test when the reference is present).
#include <cstddef>
#include <cstdio>
#include <memory>
#include <queue>
#include <vector>

#include <Eigen/Eigen>
#include <Eigen/StdVector>
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
#include <ros/ros.h>

#define private public
#define protected public
#include <plan_env/sdf_map.h>
#undef private
#undef protected

#pragma GCC diagnostic ignored "-Winvalid-offsetof"
using namespace fast_planner;
#define F(S, f) std::printf(#S "." #f " offset %zu size %zu\n", offsetof(S, f), sizeof(((S*)0)->f))

int main() {
  std::printf("MapParam size %zu align %zu\n", sizeof(MapParam), alignof(MapParam));
  F(MapParam, map_origin_); F(MapParam, map_size_); F(MapParam, map_min_boundary_); F(MapParam, map_max_boundary_);
  F(MapParam, map_voxel_num_); F(MapParam, resolution_); F(MapParam, resolution_inv_); F(MapParam, obstacles_inflation_);
  F(MapParam, virtual_ceil_height_); F(MapParam, ground_height_); F(MapParam, box_min_); F(MapParam, box_max_);
  F(MapParam, box_mind_); F(MapParam, box_maxd_); F(MapParam, default_dist_); F(MapParam, optimistic_); F(MapParam, signed_dist_);
  F(MapParam, p_hit_); F(MapParam, p_miss_); F(MapParam, p_min_); F(MapParam, p_max_); F(MapParam, p_occ_);
  F(MapParam, prob_hit_log_); F(MapParam, prob_miss_log_); F(MapParam, clamp_min_log_); F(MapParam, clamp_max_log_);
  F(MapParam, min_occupancy_log_); F(MapParam, max_ray_length_); F(MapParam, local_bound_inflate_);
  F(MapParam, local_map_margin_); F(MapParam, unknown_flag_);
  std::printf("MapData size %zu align %zu\n", sizeof(MapData), alignof(MapData));
  F(MapData, occupancy_buffer_); F(MapData, occupancy_buffer_inflate_); F(MapData, distance_buffer_neg_);
  F(MapData, distance_buffer_); F(MapData, tmp_buffer1_); F(MapData, tmp_buffer2_); F(MapData, count_hit_);
  F(MapData, count_miss_); F(MapData, count_hit_and_miss_); F(MapData, flag_rayend_); F(MapData, flag_visited_);
  F(MapData, raycast_num_); F(MapData, cache_voxel_); F(MapData, local_bound_min_); F(MapData, local_bound_max_);
  F(MapData, update_min_); F(MapData, update_max_); F(MapData, reset_updated_box_);
  // SDFMap's data members: the reference's four at the reference's offsets (its object ends there; the drop-in appends
  // one pointer, printed on a line of its own that the comparison skips)
  F(SDFMap, mp_); F(SDFMap, md_); F(SDFMap, mr_); F(SDFMap, caster_);
  std::printf("SDFMap size %zu\n", sizeof(SDFMap));
  return 0;
}
