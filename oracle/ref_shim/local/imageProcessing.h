// oracle/ref_shim/local/imageProcessing.h -- shadows include/imageProcessing.h of the reference in the build-time include
// mirror (oracle/Makefile: refpath).  lioOptimization only holds an `imageProcessing *` (include/lioOptimization.h:60,195);
// the real header pulls in the whole vision stage (opticalFlowTracker, lkpyramid, OpenCV intrinsics), which is out of
// scope and not compilable without OpenCV.
#pragma once
class imageProcessing;
