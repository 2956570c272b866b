// pcl/filters/voxel_grid.h -- stand-in for pcl::VoxelGrid<PointT> (PCL is not installed here).
// A restatement of applyFilter() as published in PCL 1.8-1.12
// (filters/include/pcl/filters/impl/voxel_grid.hpp): float arithmetic, leaves aligned to global
// multiples of the leaf size, bounding box from the cloud's float min/max, leaf index
// i + j*div_x + k*div_x*div_y, one float centroid per occupied leaf, output ascending in leaf index.
// PCL orders the points of a leaf with std::sort (unspecified for equal keys; affects only the float
// summation order); this stand-in keeps input order.  Used by oracle/ref_build and the facade build.
#ifndef PCL_LITE_VOXEL_GRID_H_
#define PCL_LITE_VOXEL_GRID_H_
#include <pcl/point_cloud.h>
#include <algorithm>
#include <cmath>
#include <utility>
#include <vector>
namespace pcl {
template <typename T> class VoxelGrid {
  typename PointCloud<T>::Ptr in_;
  float leaf_[3] = {0.f, 0.f, 0.f};
public:
  void setInputCloud(const typename PointCloud<T>::Ptr& c) { in_ = c; }
  void setLeafSize(float lx, float ly, float lz) { leaf_[0] = lx; leaf_[1] = ly; leaf_[2] = lz; }
  void filter(PointCloud<T>& out) {
    out.points.clear();
    if (!in_ || in_->points.empty()) return;
    const std::vector<T>& p = in_->points;
    const float inv[3] = {1.0f / leaf_[0], 1.0f / leaf_[1], 1.0f / leaf_[2]};
    float mn[3] = {p[0].x, p[0].y, p[0].z}, mx[3] = {p[0].x, p[0].y, p[0].z};
    for (const T& q : p) {
      const float v[3] = {q.x, q.y, q.z};
      for (int i = 0; i < 3; ++i) { mn[i] = std::min(mn[i], v[i]); mx[i] = std::max(mx[i], v[i]); }
    }
    int min_b[3], div_b[3];
    for (int i = 0; i < 3; ++i) {
      min_b[i] = (int)std::floor(mn[i] * inv[i]);
      div_b[i] = (int)std::floor(mx[i] * inv[i]) - min_b[i] + 1;
    }
    const int mul[3] = {1, div_b[0], div_b[0] * div_b[1]};
    std::vector<std::pair<unsigned, unsigned>> iv;
    iv.reserve(p.size());
    for (unsigned k = 0; k < p.size(); ++k) {
      const int i0 = (int)(std::floor(p[k].x * inv[0]) - (float)min_b[0]);
      const int i1 = (int)(std::floor(p[k].y * inv[1]) - (float)min_b[1]);
      const int i2 = (int)(std::floor(p[k].z * inv[2]) - (float)min_b[2]);
      iv.emplace_back((unsigned)(i0 * mul[0] + i1 * mul[1] + i2 * mul[2]), k);
    }
    std::stable_sort(iv.begin(), iv.end(), [](const std::pair<unsigned, unsigned>& a,
                                              const std::pair<unsigned, unsigned>& b) { return a.first < b.first; });
    size_t i = 0;
    while (i < iv.size()) {
      size_t j = i;
      float sx = 0.f, sy = 0.f, sz = 0.f;
      while (j < iv.size() && iv[j].first == iv[i].first) {
        sx += p[iv[j].second].x; sy += p[iv[j].second].y; sz += p[iv[j].second].z;
        ++j;
      }
      const float n = (float)(j - i);
      out.points.push_back(T(sx / n, sy / n, sz / n));
      i = j;
    }
  }
};
}
#endif
