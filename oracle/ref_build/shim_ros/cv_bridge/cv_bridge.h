// cv::Mat / cv_bridge stand-ins: a row-major 16-bit image with the members map_ros.cpp uses.
#ifndef CV_BRIDGE_LITE_H_
#define CV_BRIDGE_LITE_H_
#include <sensor_msgs/msgs.h>
#include <Eigen/Eigen>
#include <boost/bind.hpp>
#include <cstdint>
#include <cstring>
#define CV_16UC1 2
namespace cv {
struct Mat {
  int rows = 0, cols = 0;
  std::vector<uint16_t> px;
  Mat() {}
  Mat(int r, int c) : rows(r), cols(c), px((size_t)r * c, 0) {}
  template <typename T> T* ptr(int v) { return reinterpret_cast<T*>(px.data() + (size_t)v * cols); }
  void convertTo(Mat& dst, int, double) const { dst = *this; }  // 32FC1 input is not exercised
  void copyTo(Mat& dst) const { dst = *this; }
};
}
namespace cv_bridge {
struct CvImage { cv::Mat image; };
typedef std::shared_ptr<CvImage> CvImagePtr;
inline CvImagePtr toCvCopy(const sensor_msgs::ImageConstPtr& img, const std::string&) {
  CvImagePtr p(new CvImage);
  p->image = cv::Mat(img->height, img->width);
  std::memcpy(p->image.px.data(), img->data.data(), std::min(img->data.size(), p->image.px.size() * 2));
  return p;
}
}
#endif
