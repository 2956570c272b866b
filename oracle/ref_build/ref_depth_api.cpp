// ref_depth_api.cpp -- C entry point around the REAL MapROS::proessDepthImage
// (plan_env/src/map_ros.cpp:176-215), compiled with the real map_ros.h and the ROS/OpenCV
// stand-ins of shim_ros/.  Lives in its own shared object (libfuel_ref_mapros.so) because
// libfuel_ref.so builds sdf_map.cpp against the reduced MapROS of shim/.
// Test infrastructure only.
#define private public  // reach MapROS's private members (no layout change: same access for all)
#include <plan_env/sdf_map.h>
#include <plan_env/map_ros.h>
#undef private
#include <cstring>

using namespace fast_planner;

struct ref_depth_cfg {
  double fx, fy, cx, cy, maxdist, mindist, k_depth_scaling_factor;
  int margin, skip_pixel;
};

// returns the number of projected points; xyz receives proj_points_cnt float triples
extern "C" int ref_process_depth(const uint16_t* depth, int rows, int cols, const ref_depth_cfg* c,
                                 const double pos[3], const double quat_wxyz[4], float* xyz, int cap) {
  MapROS mr;
  mr.fx_ = c->fx, mr.fy_ = c->fy, mr.cx_ = c->cx, mr.cy_ = c->cy;
  mr.depth_filter_maxdist_ = c->maxdist, mr.depth_filter_mindist_ = c->mindist;
  mr.depth_filter_margin_ = c->margin, mr.k_depth_scaling_factor_ = c->k_depth_scaling_factor;
  mr.skip_pixel_ = c->skip_pixel;
  mr.frame_id_ = "world";
  // init() sizes the cloud for 640x480; size it for this image instead (same code path afterwards)
  mr.point_cloud_.points.resize((size_t)rows * cols / (c->skip_pixel * c->skip_pixel) + 16);
  mr.depth_image_.reset(new cv::Mat(rows, cols));
  std::memcpy(mr.depth_image_->px.data(), depth, (size_t)rows * cols * sizeof(uint16_t));
  mr.camera_pos_ = Eigen::Vector3d(pos[0], pos[1], pos[2]);
  mr.camera_q_ = Eigen::Quaterniond(quat_wxyz[0], quat_wxyz[1], quat_wxyz[2], quat_wxyz[3]);
  mr.proessDepthImage();
  const int n = mr.proj_points_cnt;
  for (int i = 0; i < n && i < cap; ++i) {
    xyz[3 * i] = mr.point_cloud_.points[i].x;
    xyz[3 * i + 1] = mr.point_cloud_.points[i].y;
    xyz[3 * i + 2] = mr.point_cloud_.points[i].z;
  }
  return n;
}
