// oracle/mapcloud.cpp — TEST INFRASTRUCTURE (CPU oracle). Not part of the shipped product path.
//
// Restates three in-tree per-point passes next to the hot path:
//   MapCloudGenerator::generate              /root/reference/src/mrg_slam/map_cloud_generator.cpp:14-86
//   pcl::ApproximateMeanVoxelGrid::applyFilter  /root/reference/include/pcl/filters/ApproximateMeanVoxelGrid.hpp:63-126
//   other-robot point removal                /root/reference/apps/mrg_slam_component.cpp:396-429
//   PrefilteringComponent::deskewing         /root/reference/apps/prefiltering_component.cpp:231-292
// The control flow is in the reference tree; the float expression order inside Eigen's fixed-size products,
// squaredNorm and Quaternion::_transformVector is Eigen's (not vendored) and compiler dependent: PARITY UNPINNED for
// that part.  The order chosen here (and executed identically by the HIP kernels): sums left to right, no FMA.
#include "mapcloud.h"

#include <cmath>
#include <cstring>
#include <map>
#include <tuple>
#include <vector>

namespace orc {

namespace {
struct Centroid {
    int   count = 0;
    float s[4] = {0, 0, 0, 0};
};
}  // namespace

int map_cloud_generate(int K, const float* const* clouds, const int* n, const double* poses, const unsigned char* first_keyframe, float resolution,
                       int min_points_per_voxel, float distance_far_thresh, int skip_first_cloud, float* out, int* out_n)
{
    *out_n = 0;
    if (K <= 0) return -1;  // "Keyframes are empty, cannot generate map cloud." (:19-22)
    const bool  use_distance_filter = distance_far_thresh > 0;                          // :27
    const float far_sq = distance_far_thresh * distance_far_thresh;                     // :28
    std::vector<float> cloud;
    for (int k = 0; k < K; ++k) {
        if (first_keyframe && first_keyframe[k] && skip_first_cloud) continue;          // :32-34
        float P[16];                                                                    // pose.matrix().cast<float>(), column-major
        for (int t = 0; t < 16; ++t) P[t] = static_cast<float>(poses[16 * k + t]);
        for (int i = 0; i < n[k]; ++i) {
            const float* p = clouds[k] + 4 * i;
            if (use_distance_filter) {
                float s = p[0] * p[0] + p[1] * p[1];                                    // getVector3fMap().squaredNorm()
                s = s + p[2] * p[2];
                if (s > far_sq) continue;                                               // :39-41
            }
            float q[4];
            for (int r = 0; r < 3; ++r) {                                               // pose * (x, y, z, 1): column by column
                float s = P[0 * 4 + r] * p[0];
                s = s + P[1 * 4 + r] * p[1];
                s = s + P[2 * 4 + r] * p[2];
                q[r] = s + P[3 * 4 + r] * 1.0f;
            }
            q[3] = p[3];                                                                // dst_pt.intensity = src_pt.intensity
            cloud.insert(cloud.end(), q, q + 4);
        }
    }
    const int total = static_cast<int>(cloud.size() / 4);
    if (total == 0 && K > 1) return -2;                                                 // :57-60
    if (resolution <= 0.0f) {                                                           // :66-70: full resolution
        if (total) std::memcpy(out, cloud.data(), sizeof(float) * 4 * total);
        *out_n = total;
        return 0;
    }
    // ApproximateMeanVoxelGrid: ijk = floor(p * inverse_leaf_size), float sums of every field in input order, divided
    // by float(count); voxels with count >= count_threshold are kept.  The reference walks a boost::unordered_map
    // (unspecified order); the oracle and the HIP path emit ascending (iz, iy, ix).
    const float inv = 1.0f / resolution;
    std::map<std::tuple<int, int, int>, Centroid> history;
    for (int i = 0; i < total; ++i) {
        const float* p = &cloud[4 * i];
        const int ix = static_cast<int>(std::floor(p[0] * inv)), iy = static_cast<int>(std::floor(p[1] * inv)), iz = static_cast<int>(std::floor(p[2] * inv));
        Centroid& c = history[std::make_tuple(iz, iy, ix)];
        c.count++;
        for (int f = 0; f < 4; ++f) c.s[f] += p[f];
    }
    int m = 0;
    for (auto& kv : history) {
        const Centroid& c = kv.second;
        if (c.count && c.count >= min_points_per_voxel) {
            const float cnt = static_cast<float>(c.count);
            for (int f = 0; f < 4; ++f) out[4 * m + f] = c.s[f] / cnt;
            ++m;
        }
    }
    *out_n = m;
    return 0;
}

int remove_points_near(const float* in, int n, const float* centres, int K, float radius_sqr, float* out, float* removed, int* n_removed)
{
    int kept = 0, gone = 0;
    for (int i = 0; i < n; ++i) {
        const float* p = in + 4 * i;
        bool drop = false;
        for (int k = 0; k < K && !drop; ++k) {
            const float dx = p[0] - centres[3 * k], dy = p[1] - centres[3 * k + 1], dz = p[2] - centres[3 * k + 2];
            float s = dx * dx + dy * dy;  // (point - other).squaredNorm()
            s = s + dz * dz;
            drop = s < radius_sqr;        // :413
        }
        if (drop) { if (removed) std::memcpy(removed + 4 * gone, p, 16); ++gone; }
        else      { std::memcpy(out + 4 * kept, p, 16); ++kept; }
    }
    if (n_removed) *n_removed = gone;
    return kept;
}

void deskew(const float* in, int n, const float ang_v_xyz[3], double scan_period, float* out)
{
    const float av[3] = {ang_v_xyz[0] * -1.0f, ang_v_xyz[1] * -1.0f, ang_v_xyz[2] * -1.0f};  // ang_v *= -1 (:275)
    for (int i = 0; i < n; ++i) {
        const float* p = in + 4 * i;
        const double delta_t = scan_period * static_cast<double>(i) / n;                      // :289 (double; cloud->size() converts to double)
        // Quaternionf delta_q(1, delta_t/2*ang_v[0], ...): double products narrowed to float by the constructor
        const float qw = 1.0f;
        const float qx = static_cast<float>(delta_t / 2.0 * av[0]), qy = static_cast<float>(delta_t / 2.0 * av[1]), qz = static_cast<float>(delta_t / 2.0 * av[2]);
        // delta_q.inverse(): conjugate / squaredNorm (coeff order x, y, z, w)
        float n2 = qx * qx + qy * qy;
        n2 = n2 + qz * qz;
        n2 = n2 + qw * qw;
        const float ix = -qx / n2, iy = -qy / n2, iz = -qz / n2, iw = qw / n2;
        // Quaternion * Vector3 (Eigen _transformVector): uv = 2 * vec x v; v + w * uv + vec x uv
        float uvx = iy * p[2] - iz * p[1], uvy = iz * p[0] - ix * p[2], uvz = ix * p[1] - iy * p[0];
        uvx = uvx + uvx; uvy = uvy + uvy; uvz = uvz + uvz;
        const float cx = iy * uvz - iz * uvy, cy = iz * uvx - ix * uvz, cz = ix * uvy - iy * uvx;
        float* o = out + 4 * i;
        o[0] = (p[0] + iw * uvx) + cx;
        o[1] = (p[1] + iw * uvy) + cy;
        o[2] = (p[2] + iw * uvz) + cz;
        o[3] = p[3];
    }
}

}  // namespace orc
