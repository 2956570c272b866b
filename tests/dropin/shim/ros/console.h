#include <ros/ros.h>
