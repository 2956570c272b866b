// sensor_msgs / geometry_msgs / visualization_msgs stand-ins: plain structs with the fields
// map_ros.cpp touches.
#ifndef MSGS_LITE_H_
#define MSGS_LITE_H_
#include <ros/ros.h>
#include <memory>
#include <string>
#include <vector>
namespace std_msgs { struct Header { std::string frame_id; ros::Time stamp; }; }
namespace sensor_msgs {
struct Image { std_msgs::Header header; std::string encoding; int height = 0, width = 0; std::vector<unsigned char> data; };
typedef std::shared_ptr<const Image> ImageConstPtr;
struct PointCloud2 { std_msgs::Header header; };
typedef std::shared_ptr<const PointCloud2> PointCloud2ConstPtr;
namespace image_encodings { static const std::string TYPE_32FC1 = "32FC1"; static const std::string TYPE_16UC1 = "16UC1"; }
}
namespace geometry_msgs {
struct Point { double x = 0, y = 0, z = 0; };
struct Quaternion { double x = 0, y = 0, z = 0, w = 1; };
struct Pose { Point position; Quaternion orientation; };
struct PoseStamped { std_msgs::Header header; Pose pose; };
typedef std::shared_ptr<const PoseStamped> PoseStampedConstPtr;
}
namespace visualization_msgs {
struct Marker {
  enum { CUBE = 1, ADD = 0 };
  std_msgs::Header header; int type = 0, action = 0, id = 0;
  geometry_msgs::Pose pose; struct { double x, y, z; } scale; struct { double a, r, g, b; } color;
};
}
namespace nav_msgs { struct Odometry {}; }
#endif
