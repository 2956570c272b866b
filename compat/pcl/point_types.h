#ifndef PCL_LITE_TYPES_H_
#define PCL_LITE_TYPES_H_
namespace pcl {
struct PointXYZ { float x, y, z, pad_; PointXYZ() : x(0), y(0), z(0), pad_(1) {} PointXYZ(float a, float b, float c) : x(a), y(b), z(c), pad_(1) {} };
struct PointXYZI { float x, y, z, pad_, intensity; PointXYZI() : x(0), y(0), z(0), pad_(1), intensity(0) {} };
}
#endif
