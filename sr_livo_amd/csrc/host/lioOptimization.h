// lioOptimization.h (host mirror) -- the reference's class surface for the LIO scan-matching path
// (include/lioOptimization.h:334-357: optimize / buildPlaneResiduals / updateIEKF /
// computeNeighborhoodDistribution / searchNeighbors / addPointToMap / addPointsToMap / mapSize),
// forwarding through the C-ABI of include/srlivo_hip.h to the gfx950 kernels.  ROS I/O, sensor
// decoding and the vision stage of the reference class are out of scope.
#pragma once
#include "../../../include/srlivo_hip.h"
#include "cloudMap.h"
#include "eskfEstimator.h"
#include "parameters.h"
#include "state.h"
#include "utility.h"

#include <string>
#include <vector>

// When Eigen is on the include path (the reference's own build: CMakeLists.txt:51) the two members whose signatures carry
// Eigen types -- computeNeighborhoodDistribution and searchNeighbors, include/lioOptimization.h:340-343 -- are ALSO
// declared with exactly those types, so call sites written against the reference header compile unchanged.  They convert
// at the boundary and forward to the srl:: versions (the image this was built in has no Eigen: tests/stub_eigen holds the
// minimal stand-in the signature test compiles against).  -DSRL_NO_EIGEN switches the block off.
#if defined(__has_include) && !defined(SRL_NO_EIGEN)
#if __has_include(<Eigen/Core>)
#include <Eigen/Core>
#include <Eigen/StdVector>
#define SRL_HAVE_EIGEN 1
#endif
#endif

namespace srlivo {

extern srl::Vec3 G;        // include/utility.h:43 (written by updateIEKF, optimize.cpp:260)
extern double G_norm;      // include/utility.h:44

class cloudFrame {         // include/lioOptimization.h:80-92 (LIO members only)
public:
    double time_sweep_begin = 0, time_sweep_end = 0;
    double time_frame_begin = 0, time_frame_end = 0;
    int id = 0;            // the index in all_cloud_frame
    int sub_id = 0;
    int frame_id = 0;
    double offset_begin = 0, offset_end = 0, dt_offset = 0;
    state *p_state = nullptr;
    std::vector<point3D> point_frame;
    bool success = true;
    cloudFrame(std::vector<point3D> &point_frame_, state *p_state_) : p_state(p_state_), point_frame(point_frame_) {}
};

struct Neighborhood {      // include/lioOptimization.h:127-137
    srl::Vec3 center = srl::Vec3::Zero();
    srl::Vec3 normal = srl::Vec3::Zero();
    srl::Mat3 covariance = srl::Mat3::Identity();
    double a2D = 1.0;
};

#ifdef SRL_HAVE_EIGEN
struct NeighborhoodEigen {  // include/lioOptimization.h:127-137 with the reference's member types
    Eigen::Vector3d center = Eigen::Vector3d::Zero();
    Eigen::Vector3d normal = Eigen::Vector3d::Zero();
    Eigen::Matrix3d covariance = Eigen::Matrix3d::Identity();
    double a2D = 1.0;
};
#endif

struct optimizeSummary {   // include/lioOptimization.h:181-188
    bool success = false;
    int num_residuals_used = 0;
    std::string error_log;
};

struct iterationLog {      // ours: per-ESIKF-iteration tap for the parity tests
    srl_normal_eq neq;
    srl::Vec17 d_x;
};

// source of the per-iteration normal equations: the HIP backend by default; an external provider lets
// the identical host update run on equations reduced elsewhere (multi-process CPU tests).
typedef int (*normal_eq_provider)(const srl_frame *frame, const srl_icp_opts *opts, srl_normal_eq *out, void *user);

void subSampleFrame(std::vector<point3D> &frame, double size_voxel);                                     // utility.cpp:167-186
void gridSampling(const std::vector<point3D> &frame, std::vector<point3D> &keypoints, double size_voxel_subsampling);  // utility.cpp:188-201

class lioOptimization {
public:
    // device >= 0: creates the HIP context (throws std::runtime_error when no GPU -- there is no CPU path).
    // device < 0 : host-only object, usable solely with setNormalEqProvider().
    explicit lioOptimization(int device);
    ~lioOptimization();
    lioOptimization(const lioOptimization &) = delete;
    lioOptimization &operator=(const lioOptimization &) = delete;

    // ---- reference surface (include/lioOptimization.h:334-357) ----
    optimizeSummary optimize(cloudFrame *p_frame, const icpOptions &cur_icp_options, double sample_voxel_size);
    optimizeSummary buildPlaneResiduals(const icpOptions &cur_icp_options, voxelHashMap &voxel_map_temp,
                                        std::vector<point3D> &keypoints, std::vector<planeParam> &plane_residuals,
                                        cloudFrame *p_frame, double &loss_sum);
    optimizeSummary updateIEKF(const icpOptions &cur_icp_options, voxelHashMap &voxel_map_temp,
                               std::vector<point3D> &keypoints, cloudFrame *p_frame);
    Neighborhood computeNeighborhoodDistribution(const std::vector<srl::Vec3> &points);
    std::vector<srl::Vec3> searchNeighbors(voxelHashMap &map, const srl::Vec3 &point, int nb_voxels_visited,
                                           double size_voxel_map, int max_num_neighbors, int threshold_voxel_capacity = 1,
                                           std::vector<voxel> *voxels = nullptr);
#ifdef SRL_HAVE_EIGEN
    // the reference's own signatures (include/lioOptimization.h:340-343), forwarding to the two members above
    NeighborhoodEigen computeNeighborhoodDistribution(const std::vector<Eigen::Vector3d, Eigen::aligned_allocator<Eigen::Vector3d>> &points) {
        std::vector<srl::Vec3> p(points.size());
        for (size_t i = 0; i < points.size(); i++) p[i] = srl::vec3(points[i][0], points[i][1], points[i][2]);
        const Neighborhood nb = computeNeighborhoodDistribution(p);
        NeighborhoodEigen out;
        for (int i = 0; i < 3; i++) { out.center[i] = nb.center[i]; out.normal[i] = nb.normal[i]; for (int j = 0; j < 3; j++) out.covariance(i, j) = nb.covariance(i, j); }
        out.a2D = nb.a2D;
        return out;
    }
    std::vector<Eigen::Vector3d, Eigen::aligned_allocator<Eigen::Vector3d>> searchNeighbors(voxelHashMap &map, const Eigen::Vector3d &point,
            int nb_voxels_visited, double size_voxel_map, int max_num_neighbors, int threshold_voxel_capacity = 1, std::vector<voxel> *voxels = nullptr) {
        const std::vector<srl::Vec3> r = searchNeighbors(map, srl::vec3(point[0], point[1], point[2]), nb_voxels_visited, size_voxel_map,
                                                         max_num_neighbors, threshold_voxel_capacity, voxels);
        std::vector<Eigen::Vector3d, Eigen::aligned_allocator<Eigen::Vector3d>> out(r.size());
        for (size_t i = 0; i < r.size(); i++) out[i] = Eigen::Vector3d(r[i][0], r[i][1], r[i][2]);
        return out;
    }
#endif
    void addPointToMap(voxelHashMap &map, const srl::Vec3 &point, double voxel_size, int max_num_points_in_voxel,
                       double min_distance_points, int min_num_points, cloudFrame *p_frame);
    void addPointsToMap(voxelHashMap &map, cloudFrame *p_frame, double voxel_size, int max_num_points_in_voxel,
                        double min_distance_points, int min_num_points = 0, bool to_rendering = false);
    size_t mapSize(const voxelHashMap &map);

    // stateInitialization (lioOptimization.cpp:895-990): pose prior of the next frame from the last two frames of
    // all_cloud_frame (constant velocity), the filter (INIT_IMU once initial_flag is set) or the last pose.
    void stateInitialization(state *cur_state);
    int index_frame = 1;                                                 // lioOptimization.cpp:355
    enum StateInitialization { INIT_IMU = 0, INIT_CONSTANT_VELOCITY = 1 };   // include/utility.h:88-92
    int initialization = INIT_IMU;                                      // odometry_options.initialization (both yaml: "imu")

    // sweep reconstruction (lioOptimization.cpp:786-893).  The per-point math of distortFrameByConstant/-ByImu and
    // transformAllImuPoint runs on the device over the whole cut sweep (srl_frame_undistort); the order decisions
    // (two std::shuffle with one default-seeded mt19937_64, subSampleFrame's tr1 iteration order) stay on the host;
    // the surviving points become the resident frame (srl_frame_take), so optimizeResident(p_frame, ...) and
    // commitFrame() need no further upload.  point_frame of the returned frame is filled from the device results.
    void makePointTimestamp(std::vector<point3D> &sweep, double time_begin, double time_end);
    cloudFrame *buildFrame(std::vector<point3D> &cut_sweep, state *cur_state, double timestamp_begin, double timestamp_offset);
    // optimize() on the frame buildFrame left resident in HBM
    optimizeSummary optimizeBuiltFrame(cloudFrame *p_frame, const icpOptions &cur_icp_options, double sample_voxel_size,
                                       std::vector<int> *keypoint_index = nullptr);
    std::vector<imuState> imu_states;                                    // lioOptimization.h:263
    bool point_time_enable = true;                                       // cloud_pro->isPointTimeEnable()
    enum MotionCompensation { IMU = 0, CONSTANT_VELOCITY = 1 };          // include/utility.h:82-86
    int motion_compensation = CONSTANT_VELOCITY;                         // odometry_options (parameters.h:84)
    double init_voxel_size = 0.2, voxel_size = 0.5;                      // parameters.h:62,70
    int init_num_frames = 20;                                            // parameters.h:66

    // ---- ROS-free replay driver: the LIO part of run() / process() / stateEstimation() ----
    // (lioOptimization.cpp:1427-1584, :1036-1133, :991-1034).  Image / rendering / publishing / file output are out
    // of scope; recordSinglePose's rows are kept in `trajectory` instead of pose.txt.
    struct imuSample { double time; srl::Vec3 acc, gyr; };
    struct Measurement {
        double time_frame = 0;                         // measurement.time_image
        std::vector<imuSample> imu;                    // measurement.imu_measurements
        std::vector<point3D> lidar_points;             // raw_point, point (= raw_point), timestamp
        double time_sweep_begin = 0, time_sweep_offset = 0;   // measurement.time_sweep
    };
    struct poseRecord { double time; srl::Vec3 translation; srl::Quat rotation; };
    // returns false while the filter is still initialising (no frame processed)
    bool runMeasurement(Measurement &measurement, optimizeSummary *summary = nullptr);
    void process(std::vector<point3D> &cut_sweep, double timestamp_begin, double timestamp_offset, optimizeSummary *summary = nullptr);
    optimizeSummary stateEstimation(cloudFrame *p_frame);
    void releaseFrames();
    icpOptions optimize_options;                                         // odometry_options.optimize_options
    double init_sample_voxel_size = 1.0, sample_voxel_size = 1.5;        // parameters.h:64,72
    int num_for_initialization = 10;                                     // parameters.h:68
    int max_num_points_in_voxel = 20;                                    // parameters.h:76
    double min_distance_points = 0.1;                                    // parameters.h:78
    bool download_frame_points = true;       // fill point3D::point of the frame after the device commit
    std::vector<poseRecord> trajectory;
    std::vector<imuMeas> imu_meas;
    double current_time = -1, last_time_frame = -1, dt_sum = 0;         // lioOptimization.cpp:353-357
    int last_frame_keypoints = 0, last_frame_points = 0, last_points_added = 0;

    // ---- ours ----
    // pin a sweep in HBM (bench: inputs resident before the timed region): solveIEKF() then runs on it.  updateIEKF()
    // with a keypoint vector always uploads that vector and releases the pin.
    int residentSweep(const double *raw_xyz, int n);
    // upload the NEXT sweep on the copy stream while solveIEKF() works on the current one; swapSweep() makes it the pinned one
    int prefetchSweep(const double *raw_xyz, int n);
    int swapSweep();
    // ... the same upload issued by the next solveIEKF() beside the kernel of its first pass (srl_lio_prefetch_sweep_during_solve)
    void prefetchSweepDuringSolve(const double *raw_xyz, int n) { pending_prefetch_raw = raw_xyz; pending_prefetch_n = n; pending_prefetch_rc = 0; }
    bool sweepPinned(int n) const { return sweep_pinned && resident_n == n; }
    // updateIEKF on the sweep already resident in HBM (no keypoint vector needed)
    optimizeSummary solveIEKF(const icpOptions &cur_icp_options, cloudFrame *p_frame);
    void releaseSweep() { sweep_pinned = false; resident_n = -1; }
    // frame-resident form of optimize(): the raw frame is uploaded once, keypoints are selected on the device with
    // the prior pose of p_frame->p_state (same set and order as gridSampling on transformPoint-ed points), the
    // ESIKF runs on them in place.  Returns the keypoints' frame indices through keypoint_index (optional).
    optimizeSummary optimizeResident(cloudFrame *p_frame, const double *frame_raw, int n, const icpOptions &cur_icp_options,
                                     double sample_voxel_size, std::vector<int> *keypoint_index = nullptr);
    // transformPoint over the uploaded frame with p_state's pose + addPointsToMap, all on the device
    // (want_added = false: the insertion is only enqueued -- srl_frame_commit with num_added = NULL -- and -1 is returned)
    int commitFrame(const state *p_state, double voxel_size, int max_num_points_in_voxel, double min_distance_points,
                    int min_num_points, double *world_out = nullptr, bool want_added = true);
    void setNormalEqProvider(normal_eq_provider fn, void *user) { provider = fn; provider_user = user; }
    srl_ctx *context() { return voxel_map.ctx; }

    // members the path reads (lioOptimization.h:216-228,249,274); public for replay drivers
    eskfEstimator *eskf_pro = nullptr;
    voxelHashMap voxel_map;
    srl::Mat3 R_imu_lidar = srl::Mat3::Identity();
    srl::Vec3 t_imu_lidar = srl::Vec3::Zero();
    double laser_point_cov = 0.001;                    // lioOptimization.cpp:364
    std::vector<cloudFrame *> all_cloud_frame;

    bool record_iterations = false;
    std::vector<iterationLog> iteration_log;
    int last_num_iterations = 0;
    int last_num_observed = 0;         // eskf_pro->observe() calls of the last solve (optimize.cpp:253): 0 = p_state / G were never written (:255-261)
    int last_solve_launches = 0;       // passes (= association kernel launches) of the last solve

private:
    int normalEquations(const icpOptions &o, cloudFrame *p_frame, srl_normal_eq &neq, void (*while_running)(void *) = nullptr, void *user = nullptr);
    void fillFrame(cloudFrame *p_frame, srl_frame &f) const;
    normal_eq_provider provider = nullptr;
    void *provider_user = nullptr;
    int resident_n = -1;
    int prefetched_n = -1;
    const double *pending_prefetch_raw = nullptr;      // prefetchSweepDuringSolve: issued by the next solve's first pass
    int pending_prefetch_n = -1;
    int pending_prefetch_rc = 0;
    bool sweep_pinned = false;
};

}  // namespace srlivo
