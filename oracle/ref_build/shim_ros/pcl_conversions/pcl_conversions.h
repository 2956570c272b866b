#ifndef PCL_CONV_LITE_H_
#define PCL_CONV_LITE_H_
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
#include <sensor_msgs/msgs.h>
namespace pcl {
template <typename C> void toROSMsg(const C&, sensor_msgs::PointCloud2&) {}
template <typename C> void fromROSMsg(const sensor_msgs::PointCloud2&, C&) {}
}
#endif
