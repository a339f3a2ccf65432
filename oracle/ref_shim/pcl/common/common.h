// NOT PCL: see pcl/point_types.h in this directory tree.
#pragma once
#include <pcl/point_types.h>
