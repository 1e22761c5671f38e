// csrc/filters.h — prefilter chain launchers (filters.hip). Device-pointer forms chain without leaving HBM; the
// host-pointer forms back the C ABI (include/mrgfe.h).
#pragma once
#include "common.h"

namespace mrgfe {

int filter_distance_device(mrgfe_ctx* ctx, const float4* d_in, size_t n, double near_t, double far_t, float4* d_out, size_t* out_n);
int filter_voxelgrid_device(mrgfe_ctx* ctx, const float4* d_in, size_t n, float leaf, int min_pts, float4* d_out, size_t* out_n, int* overflow);
int filter_radius_outlier_device(mrgfe_ctx* ctx, const float4* d_in, size_t n, double radius, int min_neighbors, float4* d_out, size_t* out_n);
// pcl::ApproximateVoxelGrid: the sequential 512-entry history loop decomposed into independent per-entry sequences (filters.hip)
int filter_approx_voxelgrid_device(mrgfe_ctx* ctx, const float4* d_in, size_t n, float leaf, float4* d_out, size_t* out_n);
int filter_statistical_outlier_device(mrgfe_ctx* ctx, const float4* d_in, size_t n, int mean_k, double stddev_mul, float4* d_out, size_t* out_n);

// order-preserving compaction of the points whose flag is 1 (exclusive scan of the flags + scatter); synchronises and
// returns the kept count.  Uses ctx scratch slots 0, 4 and 8.
// per-run float centroids (sums in ascending sorted position) and the count-threshold flags: the tail of the voxel-grid passes
int launch_voxel_centroids(mrgfe_ctx* ctx, const float4* d_pts, const uint32_t* d_sorted_vals, const uint32_t* d_seg_start, uint32_t n_seg, int min_pts, float4* d_centroids,
                           uint32_t* d_keep);
int compact_by_flags(mrgfe_ctx* ctx, const float4* d_in, uint32_t n, uint32_t* d_flags, float4* d_out, uint32_t* h_total);

struct PrefilterChain {
    bool   distance = true;
    double near_t = 0.1, far_t = 35.0;
    bool   voxelgrid = true;
    bool   approx_voxelgrid = false;  // downsample_method APPROX_VOXELGRID (instead of voxelgrid; same leaf)
    float  leaf = 0.1f;
    int    min_pts = 1;
    int    outlier = 1;  // 0 none, 1 radius, 2 statistical
    double radius = 0.5;
    int    radius_min_neighbors = 2;
    int    mean_k = 30;
    double stddev_mul = 1.2;
};
// 1: the usual chain (voxel grid + radius filter) keeps its point counts on the device and waits once (default); 0: host-driven stages; < 0: query
int prefilter_set_device_driven(int mode);
// the three passes back to back on the device (one upload, one download)
int filter_chain(mrgfe_ctx* ctx, const PrefilterChain& chain, const float* xyzi, size_t n, size_t stride, void* out, size_t* out_n, bool out_on_device = false);

int filter_distance(mrgfe_ctx* ctx, const float* xyzi, size_t n, size_t stride, double near_t, double far_t, float* out, size_t* out_n);
int filter_voxelgrid(mrgfe_ctx* ctx, const float* xyzi, size_t n, size_t stride, float leaf, int min_pts, float* out, size_t* out_n, int* overflow);
int filter_approx_voxelgrid(mrgfe_ctx* ctx, const float* xyzi, size_t n, size_t stride, float leaf, float* out, size_t* out_n);
int filter_radius_outlier(mrgfe_ctx* ctx, const float* xyzi, size_t n, size_t stride, double radius, int min_neighbors, float* out, size_t* out_n);
int filter_statistical_outlier(mrgfe_ctx* ctx, const float* xyzi, size_t n, size_t stride, int mean_k, double stddev_mul, float* out, size_t* out_n);

}  // namespace mrgfe
