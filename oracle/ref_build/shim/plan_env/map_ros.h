// plan_env/map_ros.h -- replacement of the reference's ROS I/O shell (subscribers, timers,
// publishers: out of scope).  Only what SDFMap::initMap / inputPointCloud touch, plus accessors
// that use the `friend MapROS` declaration of SDFMap to reach its private state from ref_api.cpp.
#ifndef MAP_ROS_LITE_H_
#define MAP_ROS_LITE_H_
#include <ros/ros.h>
namespace fast_planner {
class SDFMap;
struct MapParam;
struct MapData;
class MapROS {
public:
  void setMap(SDFMap* m) { map_ = m; }
  void init() {}
  ros::NodeHandle node_;
  bool local_updated_ = false;
  SDFMap* map_ = nullptr;
  // friend-access hooks (defined in ref_api.cpp)
  static MapParam* params(SDFMap* m);
  static MapData* data(SDFMap* m);
  static void inflate(SDFMap* m);
};
}
#endif
