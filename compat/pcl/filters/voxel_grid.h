// pcl/filters/voxel_grid.h -- stand-in.  FrontierFinder::downsample (frontier_finder.cpp:757-774)
// feeds only splitLargeFrontiers and viewpoint sampling ("next" rows, PCL parity unpinned).
// filter() returns an EMPTY cloud so that splitHorizontally never splits and searchFrontiers
// leaves the region-grown clusters (the pinned quantity) untouched in tmp_frontiers_.
#ifndef PCL_LITE_VOXEL_GRID_H_
#define PCL_LITE_VOXEL_GRID_H_
#include <pcl/point_cloud.h>
namespace pcl {
template <typename T> class VoxelGrid {
public:
  void setInputCloud(const typename PointCloud<T>::Ptr&) {}
  void setLeafSize(float, float, float) {}
  void filter(PointCloud<T>& out) { out.points.clear(); }
};
}
#endif
