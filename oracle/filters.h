// oracle/filters.h — TEST INFRASTRUCTURE (CPU oracle). Not part of the shipped product path.
// Prefilter chain of /root/reference/apps/prefiltering_component.cpp:149-229 (see filters.cpp).
#pragma once

namespace orc {

// all clouds are packed xyzi float4; outputs need capacity n points; return value = number of output points
int distance_filter(const float* in, int n, double near_thresh, double far_thresh, float* out);
// returns 0 ok, 1 = index overflow (output == input, PCL behaviour)
int voxelgrid(const float* in, int n, float leaf, int min_points_per_voxel, int order_mode, float* out, int* out_n);
// pcl::ApproximateVoxelGrid<PointXYZI>::applyFilter (downsample_method APPROX_VOXELGRID, prefiltering_component.cpp:172-175,
// scan_matching_odometry_component.cpp:180-183); returns the number of output points (capacity n)
int approx_voxelgrid(const float* in, int n, float leaf, float* out);
int radius_outlier(const float* in, int n, double radius, int min_neighbors, float* out, unsigned char* keep_mask);
int statistical_outlier(const float* in, int n, int mean_k, double stddev_mul, float* out, unsigned char* keep_mask);

}  // namespace orc
