// NOT OpenCV: see opencv2/opencv.hpp in this directory tree.
#pragma once
#include <opencv2/opencv.hpp>
