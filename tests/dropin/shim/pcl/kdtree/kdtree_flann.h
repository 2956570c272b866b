// stand-in (tests/dropin only): the declaration HeadingPlanner's header needs
#ifndef DROPIN_KDTREE_FLANN_H_
#define DROPIN_KDTREE_FLANN_H_
#include <pcl/point_cloud.h>
#include <vector>
namespace pcl {
template <typename P> class KdTreeFLANN {
public:
  void setInputCloud(const typename PointCloud<P>::Ptr&) {}
  int nearestKSearch(const P&, int, std::vector<int>&, std::vector<float>&) const { return 0; }
  int radiusSearch(const P&, double, std::vector<int>&, std::vector<float>&) const { return 0; }
};
}  // namespace pcl
#endif
