// oracle/filters.cpp — TEST INFRASTRUCTURE (CPU oracle). Not part of the shipped product path.
//
// Restates the prefilter chain the reference runs per raw scan
// (/root/reference/apps/prefiltering_component.cpp:149-151):
//   distance_filter  :206-229  (in-tree lambda: float norm promoted to double, strict compares)
//   downsample       :158-180  -> pcl::VoxelGrid<PointXYZI>::applyFilter           (PCL 1.12)
//   outlier_removal  :182-204  -> pcl::RadiusOutlierRemoval / StatisticalOutlierRemoval
// PCL is not installed here: PARITY UNPINNED for the PCL parts (see quirks.h).
#include "filters.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

#include "nn.h"
#include "quirks.h"

namespace orc {

int distance_filter(const float* in, int n, double near_thresh, double far_thresh, float* out)
{
    int m = 0;
    for (int i = 0; i < n; ++i) {
        const float* p = in + 4 * i;
        // p.getVector3fMap().norm(): float sqrt((x*x + y*y) + z*z), then promoted to double
        float  xx = p[0] * p[0], yy = p[1] * p[1], zz = p[2] * p[2];
        float  s  = xx + yy;
        s         = s + zz;
        double d  = std::sqrt(s);
        if (d > near_thresh && d < far_thresh) { std::memcpy(out + 4 * m, p, 16); ++m; }
    }
    return m;
}

// pcl::ApproximateVoxelGrid<PointXYZI>::applyFilter (PCL 1.12 filters/impl/approximate_voxel_grid.hpp; [UPSTREAM-RECALL]):
// a direct-mapped history of histsize_ = 512 entries keyed by (ix * 7171 + iy * 3079 + iz * 4231) & 511 with ijk = floor(p * inverse_leaf_size)
// (float, inverse_leaf_size = 1 / leaf as float).  A point whose entry holds ANOTHER cell flushes that entry — its centroid (float sums of
// x, y, z, intensity in arrival order, divided by the float count) becomes the next output point — and takes it over; at the end the
// occupied entries are flushed in entry order.  Order dependent; a cell may be emitted several times.  There is no min-points threshold and
// no bounding box (a non-finite coordinate goes through floor and the int cast like any other: x86's cvttss2si gives INT_MIN for it).
int approx_voxelgrid(const float* in, int n, float leaf, float* out)
{
    constexpr int kHist = 512;
    struct He { int ix, iy, iz, count; float c[4]; };
    std::vector<He> hist(kHist);
    for (auto& h : hist) { h.ix = h.iy = h.iz = 0; h.count = 0; h.c[0] = h.c[1] = h.c[2] = h.c[3] = 0.0f; }
    const float inv = 1.0f / leaf;
    auto cell = [](float v) {
        const float f = std::floor(v);
        if (!(f >= -2147483648.0f && f < 2147483648.0f)) return std::numeric_limits<int>::min();  // what the x86 conversion returns out of range / for NaN
        return static_cast<int>(f);
    };
    auto flush = [&](He& h, int op) {
        const float cnt = static_cast<float>(h.count);
        for (int k = 0; k < 4; ++k) out[4 * op + k] = h.c[k] / cnt;
    };
    int op = 0;
    for (int i = 0; i < n; ++i) {
        const float* p = in + 4 * i;
        const int ix = cell(p[0] * inv), iy = cell(p[1] * inv), iz = cell(p[2] * inv);
        const unsigned hash = (static_cast<unsigned>(ix) * 7171u + static_cast<unsigned>(iy) * 3079u + static_cast<unsigned>(iz) * 4231u) & (kHist - 1);
        He& h = hist[hash];
        if (h.count && (ix != h.ix || iy != h.iy || iz != h.iz)) {
            flush(h, op++);
            h.count = 0;
            h.c[0] = h.c[1] = h.c[2] = h.c[3] = 0.0f;
        }
        h.ix = ix; h.iy = iy; h.iz = iz;
        h.count++;
        for (int k = 0; k < 4; ++k) h.c[k] += p[k];
    }
    for (auto& h : hist)
        if (h.count) flush(h, op++);
    return op;
}

int voxelgrid(const float* in, int n, float leaf, int min_points_per_voxel, int order_mode, float* out, int* out_n)
{
    *out_n = 0;
    const float inv_leaf = 1.0f / leaf;
    float min_p[3] = {std::numeric_limits<float>::max(), std::numeric_limits<float>::max(), std::numeric_limits<float>::max()};
    float max_p[3] = {-min_p[0], -min_p[1], -min_p[2]};
    int   finite   = 0;
    for (int i = 0; i < n; ++i) {
        const float* p = in + 4 * i;
        if (!std::isfinite(p[0]) || !std::isfinite(p[1]) || !std::isfinite(p[2])) continue;
        ++finite;
        for (int a = 0; a < 3; ++a) { min_p[a] = std::min(min_p[a], p[a]); max_p[a] = std::max(max_p[a], p[a]); }
    }
    if (finite == 0) return 0;
    int64_t dx = static_cast<int64_t>((max_p[0] - min_p[0]) * inv_leaf) + 1;
    int64_t dy = static_cast<int64_t>((max_p[1] - min_p[1]) * inv_leaf) + 1;
    int64_t dz = static_cast<int64_t>((max_p[2] - min_p[2]) * inv_leaf) + 1;
    if (dx * dy * dz > static_cast<int64_t>(std::numeric_limits<int32_t>::max())) {
        // "Leaf size is too small for the input dataset. Integer indices would overflow." -> output = input
        std::memcpy(out, in, static_cast<size_t>(n) * 16);
        *out_n = n;
        return 1;
    }
    int min_b[3], max_b[3], div_b[3], divb_mul[3];
    for (int a = 0; a < 3; ++a) {
        min_b[a] = static_cast<int>(std::floor(min_p[a] * inv_leaf));
        max_b[a] = static_cast<int>(std::floor(max_p[a] * inv_leaf));
        div_b[a] = max_b[a] - min_b[a] + 1;
    }
    divb_mul[0] = 1; divb_mul[1] = div_b[0]; divb_mul[2] = div_b[0] * div_b[1];
    struct Entry { unsigned int idx; unsigned int cloud_point_index; };
    std::vector<Entry> index_vector;
    index_vector.reserve(n);
    for (int i = 0; i < n; ++i) {
        const float* p = in + 4 * i;
        if (!std::isfinite(p[0]) || !std::isfinite(p[1]) || !std::isfinite(p[2])) continue;
        int ijk0 = static_cast<int>(std::floor(p[0] * inv_leaf) - static_cast<float>(min_b[0]));
        int ijk1 = static_cast<int>(std::floor(p[1] * inv_leaf) - static_cast<float>(min_b[1]));
        int ijk2 = static_cast<int>(std::floor(p[2] * inv_leaf) - static_cast<float>(min_b[2]));
        int idx  = ijk0 * divb_mul[0] + ijk1 * divb_mul[1] + ijk2 * divb_mul[2];
        index_vector.push_back({static_cast<unsigned int>(idx), static_cast<unsigned int>(i)});
    }
    auto less_idx = [](const Entry& a, const Entry& b) { return a.idx < b.idx; };
    if (order_mode == quirks::ORDER_STD_SORT) std::sort(index_vector.begin(), index_vector.end(), less_idx);
    else                                      std::stable_sort(index_vector.begin(), index_vector.end(), less_idx);
    size_t index = 0;
    int    m     = 0;
    while (index < index_vector.size()) {
        size_t i = index + 1;
        while (i < index_vector.size() && index_vector[i].idx == index_vector[index].idx) ++i;
        if (i - index >= static_cast<size_t>(min_points_per_voxel)) {
            // CentroidPoint<PointXYZI>: float running sums of xyz and intensity, divided by the count
            float sx = 0, sy = 0, sz = 0, si = 0;
            for (size_t li = index; li < i; ++li) {
                const float* p = in + 4 * static_cast<size_t>(index_vector[li].cloud_point_index);
                sx += p[0]; sy += p[1]; sz += p[2]; si += p[3];
            }
            float cnt = static_cast<float>(i - index);
            out[4 * m + 0] = sx / cnt; out[4 * m + 1] = sy / cnt; out[4 * m + 2] = sz / cnt; out[4 * m + 3] = si / cnt;
            ++m;
        }
        index = i;
    }
    *out_n = m;
    return 0;
}

int radius_outlier(const float* in, int n, double radius, int min_neighbors, float* out, unsigned char* keep_mask)
{
    NnGrid grid;
    grid.build(in, n, static_cast<float>(radius));
    const double r2 = radius * radius;
    int m = 0;
    for (int i = 0; i < n; ++i) {
        const float* p = in + 4 * i;
        bool keep = false;
        if (std::isfinite(p[0]) && std::isfinite(p[1]) && std::isfinite(p[2])) {
            // dense path: nearestKSearch(min_pts + 1); inlier iff that many neighbours (self included) exist
            // and the farthest has sqdist <= r^2  <=>  #{q : sqdist(q,p) <= r^2} >= min_pts + 1
            int c[3];
            grid.coords(p[0], p[1], p[2], c);
            int count = 0;
            int rings = static_cast<int>(std::ceil(radius / grid.cell)) + 1;
            for (int r = 0; r <= rings; ++r)
                grid.ring(c, r, [&](int j) {
                    const float* q = in + 4 * j;
                    if (static_cast<double>(sqdist_f(q[0], q[1], q[2], p[0], p[1], p[2])) <= r2) ++count;
                });
            keep = count >= min_neighbors + 1;
        }
        if (keep_mask) keep_mask[i] = keep ? 1 : 0;
        if (keep) { std::memcpy(out + 4 * m, p, 16); ++m; }
    }
    return m;
}

int statistical_outlier(const float* in, int n, int mean_k, double stddev_mul, float* out, unsigned char* keep_mask)
{
    NnGrid grid;
    grid.build(in, n, 0.5f);
    std::vector<float> distances(n, 0.0f);
    std::vector<char>  valid(n, 0);
    std::vector<int>   idx(mean_k + 1);
    std::vector<float> sqd(mean_k + 1);
    int valid_distances = 0;
    for (int i = 0; i < n; ++i) {
        const float* p = in + 4 * i;
        if (!std::isfinite(p[0]) || !std::isfinite(p[1]) || !std::isfinite(p[2])) continue;
        if (grid.knn(p[0], p[1], p[2], mean_k + 1, idx.data(), sqd.data()) != mean_k + 1) continue;  // "not enough neighbours": distance 0, skipped
        double dist_sum = 0;
        for (int k = 1; k < mean_k + 1; ++k) dist_sum += std::sqrt(sqd[k]);  // float sqrt, double accumulate
        distances[i] = static_cast<float>(dist_sum / mean_k);
        valid[i] = 1;
        ++valid_distances;
    }
    double sum = 0, sq_sum = 0;
    for (int i = 0; i < n; ++i) { float d2 = distances[i] * distances[i]; sum += distances[i]; sq_sum += d2; }  // float square, double sums (PCL)
    double mean = sum / static_cast<double>(valid_distances);
    double variance = (sq_sum - sum * sum / static_cast<double>(valid_distances)) / (static_cast<double>(valid_distances) - 1);
    double stddev = std::sqrt(variance);
    double distance_threshold = mean + stddev_mul * stddev;
    int m = 0;
    for (int i = 0; i < n; ++i) {
        // PCL: points whose search failed (or non-finite points) keep distance 0 and therefore pass
        const float* p = in + 4 * i;
        bool keep = !(distances[i] > distance_threshold);
        if (keep_mask) keep_mask[i] = keep ? 1 : 0;
        if (keep) { std::memcpy(out + 4 * m, p, 16); ++m; }
    }
    return m;
}

}  // namespace orc
