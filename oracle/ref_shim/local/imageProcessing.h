// oracle/ref_shim/local/imageProcessing.h -- shadows include/imageProcessing.h of the reference in the build-time include
// mirror (oracle/Makefile: refpath).  The real header is the entry to the vision stage (opticalFlowTracker, lkpyramid,
// rgbMapTracker: OpenCV internals), which is out of scope and not compilable without OpenCV.  The LiDAR side
// only holds an `imageProcessing *` and touches the members declared below (src/lioOptimization.cpp:330-397,430-441,
// 462-476,538-545,1040-1048); oracle/ref_harness.cpp defines them as no-ops.
#pragma once
#include <memory>
#include <mutex>
#include <vector>
#include <Eigen/Core>
#include "cloudMap.h"

class cloudFrame;

class rgbMapTracker {                       // include/rgbMapTracker.h:34-50, the members the LiDAR side touches
public:
    std::vector<voxelId> voxels_recent_visited;
    std::vector<rgbPoint *> rgb_points_vec;
    std::shared_ptr<std::mutex> mutex_rgb_points_vec = std::make_shared<std::mutex>();
    int number_of_new_visited_voxel = 0;
    int updated_frame_index = 0;
    std::shared_ptr<std::mutex> mutex_frame_index = std::make_shared<std::mutex>();
};

class imageProcessing {                     // include/imageProcessing.h:28-104, likewise
public:
    double time_last_process = 0.0;
    rgbMapTracker *map_tracker = nullptr;
    imageProcessing();
    void setImageWidth(int &para);
    void setImageHeight(int &para);
    void setCameraIntrinsic(std::vector<double> &v_camera_intrinsic);
    void setCameraDistCoeffs(std::vector<double> &v_camera_dist_coeffs);
    void setExtrinR(Eigen::Matrix3d &R);
    void setExtrinT(Eigen::Vector3d &t);
    Eigen::Matrix3d getCameraIntrinsic();
    void process(voxelHashMap &voxel_map, cloudFrame *p_frame);
    void printParameter();
};
