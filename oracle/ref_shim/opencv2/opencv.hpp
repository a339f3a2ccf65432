// NOT OpenCV: empty stand-ins for the cv types the reference's headers name in declarations (include/cloudMap.h:35,80,
// include/lioOptimization.h:76,113-114,217).  The vision stage is out of scope; nothing on the
// scan-matching path touches an image.  Test infrastructure only.
#pragma once
namespace cv {
class Mat {};
class RNG { public: RNG() {} explicit RNG(unsigned long long) {} };
struct Scalar { double v[4]; };
}  // namespace cv
