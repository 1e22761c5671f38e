// oracle/mapcloud.h — TEST INFRASTRUCTURE (CPU oracle). Not part of the shipped product path.
// map-cloud generation, other-robot point removal, IMU deskewing (see mapcloud.cpp).
#pragma once

namespace orc {

// MapCloudGenerator::generate.  clouds[k]: packed xyzi float4, n[k] points; poses: K column-major 4x4 doubles
// (Eigen::Isometry3d::matrix()); first_keyframe: K flags or null.  out needs capacity sum(n).  Voxels are emitted in
// ascending (iz, iy, ix) order.  Returns 0 ok, -1 keyframes empty, -2 cloud empty after processing (K > 1).
int map_cloud_generate(int K, const float* const* clouds, const int* n, const double* poses, const unsigned char* first_keyframe, float resolution,
                       int min_points_per_voxel, float distance_far_thresh, int skip_first_cloud, float* out, int* out_n);

// mrg_slam_component.cpp:396-429: drop every point closer than sqrt(radius_sqr) to one of the K centres (sensor frame).
// out / removed need capacity n; returns the number kept, *n_removed the number dropped (both order-preserving).
int remove_points_near(const float* in, int n, const float* centres_xyz, int K, float radius_sqr, float* out, float* removed, int* n_removed);

// PrefilteringComponent::deskewing (prefiltering_component.cpp:231-292): per-point small-angle rotation by the negated
// angular velocity; ang_v is the IMU angular velocity as received (the negation happens inside).
void deskew(const float* in, int n, const float ang_v_xyz[3], double scan_period, float* out);

}  // namespace orc
