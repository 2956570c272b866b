#include <sensor_msgs/msgs.h>
