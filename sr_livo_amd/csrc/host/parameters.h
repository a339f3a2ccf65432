// parameters.h (host mirror) -- icpOptions with the reference's member names and defaults
// (include/parameters.h:8-56).  toAbi() flattens it into the C-ABI struct.
#pragma once
#include "../../../include/srlivo_hip.h"

namespace srlivo {

class icpOptions {
public:
    int threshold_voxel_occupancy = 1;
    int init_num_frames = 20;
    double size_voxel_map = 1.0;
    int num_iters_icp = 5;
    int min_number_neighbors = 20;
    int voxel_neighborhood = 1;
    double power_planarity = 2.0;
    bool estimate_normal_from_neighborhood = true;
    int max_number_neighbors = 20;
    double max_dist_to_plane_icp = 0.3;
    double threshold_orientation_norm = 0.0001;
    double threshold_translation_norm = 0.001;
    bool point_to_plane_with_distortion = true;
    int max_num_residuals = -1;
    int min_num_residuals = 100;
    int num_closest_neighbors = 1;
    double weight_alpha = 0.9;
    double weight_neighborhood = 0.1;
    bool debug_print = true;
    bool debug_viz = false;

    srl_icp_opts toAbi() const {
        srl_icp_opts o;
        o.threshold_voxel_occupancy = threshold_voxel_occupancy;
        o.init_num_frames = init_num_frames;
        o.size_voxel_map = size_voxel_map;
        o.num_iters_icp = num_iters_icp;
        o.min_number_neighbors = min_number_neighbors;
        o.voxel_neighborhood = voxel_neighborhood;
        o.power_planarity = power_planarity;
        o.max_number_neighbors = max_number_neighbors;
        o.max_dist_to_plane_icp = max_dist_to_plane_icp;
        o.threshold_orientation_norm = threshold_orientation_norm;
        o.threshold_translation_norm = threshold_translation_norm;
        o.max_num_residuals = max_num_residuals;
        o.weight_alpha = weight_alpha;
        o.weight_neighborhood = weight_neighborhood;
        return o;
    }
    static icpOptions fromAbi(const srl_icp_opts &o) {
        icpOptions r;
        r.threshold_voxel_occupancy = o.threshold_voxel_occupancy;
        r.init_num_frames = o.init_num_frames;
        r.size_voxel_map = o.size_voxel_map;
        r.num_iters_icp = o.num_iters_icp;
        r.min_number_neighbors = o.min_number_neighbors;
        r.voxel_neighborhood = o.voxel_neighborhood;
        r.power_planarity = o.power_planarity;
        r.max_number_neighbors = o.max_number_neighbors;
        r.max_dist_to_plane_icp = o.max_dist_to_plane_icp;
        r.threshold_orientation_norm = o.threshold_orientation_norm;
        r.threshold_translation_norm = o.threshold_translation_norm;
        r.max_num_residuals = o.max_num_residuals;
        r.weight_alpha = o.weight_alpha;
        r.weight_neighborhood = o.weight_neighborhood;
        return r;
    }
};

}  // namespace srlivo
