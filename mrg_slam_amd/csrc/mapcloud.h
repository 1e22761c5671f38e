// csrc/mapcloud.h — per-point passes next to the hot path, launchers of mapcloud.hip.
#pragma once
#include "common.h"

namespace mrgfe {

// MapCloudGenerator::generate (/root/reference/src/mrg_slam/map_cloud_generator.cpp:14-86): transform every keyframe
// cloud by its pose, optional far-distance cut, concatenate, pcl::ApproximateMeanVoxelGrid
// (include/pcl/filters/ApproximateMeanVoxelGrid.hpp:63-126).  d_cat: the keyframe clouds back to back (packed float4),
// kf_off[K + 1]: first point of every keyframe (host), poses_f[K][16]: column-major float 4x4 (host).
// d_out needs room for kf_off[K] points.  resolution <= 0: no voxel filter.
int map_cloud_device(mrgfe_ctx* ctx, const float4* d_cat, const uint32_t* kf_off, const float* poses_f, int K, float resolution, int min_points_per_voxel,
                     float distance_far_thresh, float4* d_out, size_t* out_n, size_t* n_unfiltered /* points after the distance cut */,
                     const float4* const* kf_ptrs = nullptr /* host array of K device pointers: read keyframe k there instead of in d_cat */);

// other-robot point removal (apps/mrg_slam_component.cpp:396-429): order-preserving split into kept / removed
int remove_points_near_device(mrgfe_ctx* ctx, const float4* d_in, size_t n, const float* centres_xyz, int n_centres, float radius_sqr, float4* d_kept, size_t* n_kept,
                              float4* d_removed, size_t* n_removed);

// PrefilteringComponent::deskewing (apps/prefiltering_component.cpp:231-292)
int deskew_device(mrgfe_ctx* ctx, const float4* d_in, size_t n, const float ang_v[3], double scan_period, float4* d_out);
// pcl::transformPointCloud(in, out, Matrix4f): the float arithmetic of dev_float.h's transform_point, intensity copied
int transform_cloud_device(mrgfe_ctx* ctx, const float4* d_in, size_t n, const float T_rowmajor[16], float4* d_out);

}  // namespace mrgfe
