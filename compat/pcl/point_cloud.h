#ifndef PCL_LITE_CLOUD_H_
#define PCL_LITE_CLOUD_H_
#include <memory>
#include <string>
#include <vector>
namespace pcl {
template <typename T> struct PointCloud {
  std::vector<T> points;
  unsigned width = 0, height = 1; bool is_dense = true;
  struct { std::string frame_id; } header;
  typedef std::shared_ptr<PointCloud<T>> Ptr;
  void push_back(const T& p) { points.push_back(p); }
  size_t size() const { return points.size(); }
  typename std::vector<T>::iterator begin() { return points.begin(); }
  typename std::vector<T>::iterator end() { return points.end(); }
};
}
#endif
