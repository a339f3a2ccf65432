// NOT Ceres / glog: the reference only reaches glog's LOG(...) through this include (src/eskfEstimator.cpp:53,59).
#pragma once
#include <iostream>
#ifndef LOG
#define LOG(severity) std::cerr
#endif
