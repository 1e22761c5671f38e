// oracle/nn.h — TEST INFRASTRUCTURE (CPU oracle). Not part of the shipped product path.
//
// Exact nearest-neighbour queries standing in for pcl::search::KdTree / pcl::KdTreeFLANN (FLANN
// KDTreeSingleIndex, exact search, L2_Simple<float>) as used through
//  * pcl::Registration::getFitnessScore                  (reference call: src/mrg_slam/loop_detector.cpp:137)
//  * InformationMatrixCalculator::calc_fitness_score      (src/mrg_slam/information_matrix_calculator.cpp:46-81)
//  * publish_scan_matching_status inlier loop             (apps/scan_matching_odometry_component.cpp:405-417)
//  * pcl::RadiusOutlierRemoval / StatisticalOutlierRemoval (apps/prefiltering_component.cpp:182-204)
// A kd-tree and this uniform grid return the same neighbours (exact search); only the order among points at
// exactly equal distance may differ (here: lowest index first).  Squared distances are float,
// accumulated ((dx*dx + dy*dy) + dz*dz) like FLANN's L2_Simple.  PARITY UNPINNED — see quirks.h.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <vector>

namespace orc {

inline float sqdist_f(float ax, float ay, float az, float bx, float by, float bz)
{
    float dx = ax - bx, dy = ay - by, dz = az - bz;
    float xx = dx * dx, yy = dy * dy, zz = dz * dz;
    float s = xx + yy;
    return s + zz;
}

struct NnGrid {
    const float* pts = nullptr;  // xyzi, stride 4
    int          n = 0;
    float        cell = 1.0f;
    float        origin[3] = {0, 0, 0};
    int          dim[3] = {1, 1, 1};
    std::vector<int> cell_start;  // dim product + 1
    std::vector<int> order;       // point ids grouped by cell, ascending id inside a cell

    void build(const float* xyzi, int count, float cell_size)
    {
        pts = xyzi; n = count;
        float mn[3] = {std::numeric_limits<float>::max(), std::numeric_limits<float>::max(), std::numeric_limits<float>::max()};
        float mx[3] = {-mn[0], -mn[1], -mn[2]};
        for (int i = 0; i < n; ++i) {
            const float* p = pts + 4 * i;
            if (!std::isfinite(p[0]) || !std::isfinite(p[1]) || !std::isfinite(p[2])) continue;
            for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], p[a]); mx[a] = std::max(mx[a], p[a]); }
        }
        if (mn[0] > mx[0]) { for (int a = 0; a < 3; ++a) { mn[a] = 0; mx[a] = 0; } }
        cell = cell_size;
        for (;;) {
            double prod = 1;
            for (int a = 0; a < 3; ++a) { dim[a] = static_cast<int>(std::floor((mx[a] - mn[a]) / cell)) + 1; prod *= dim[a]; }
            if (prod <= 16.0e6) break;
            cell *= 2.0f;
        }
        for (int a = 0; a < 3; ++a) origin[a] = mn[a];
        size_t cells = static_cast<size_t>(dim[0]) * dim[1] * dim[2];
        cell_start.assign(cells + 1, 0);
        std::vector<int> cid(n, -1);
        for (int i = 0; i < n; ++i) {
            const float* p = pts + 4 * i;
            if (!std::isfinite(p[0]) || !std::isfinite(p[1]) || !std::isfinite(p[2])) continue;
            int c[3];
            coords(p[0], p[1], p[2], c);
            cid[i] = (c[2] * dim[1] + c[1]) * dim[0] + c[0];
            cell_start[cid[i] + 1]++;
        }
        for (size_t k = 0; k < cells; ++k) cell_start[k + 1] += cell_start[k];
        order.assign(cell_start[cells], 0);
        std::vector<int> cursor(cell_start.begin(), cell_start.end() - 1);
        for (int i = 0; i < n; ++i) if (cid[i] >= 0) order[cursor[cid[i]]++] = i;
    }
    void coords(float x, float y, float z, int c[3]) const
    {
        float q[3] = {x, y, z};
        for (int a = 0; a < 3; ++a) {
            double f = std::floor((static_cast<double>(q[a]) - origin[a]) / cell);
            c[a] = static_cast<int>(std::min<double>(std::max<double>(f, 0.0), dim[a] - 1));
        }
    }
    // visit all points of the cells at Chebyshev ring r around c; returns false if the ring is entirely outside
    template <class F>
    bool ring(const int c[3], int r, F&& f) const
    {
        bool any = false;
        int z0 = std::max(c[2] - r, 0), z1 = std::min(c[2] + r, dim[2] - 1);
        int y0 = std::max(c[1] - r, 0), y1 = std::min(c[1] + r, dim[1] - 1);
        int x0 = std::max(c[0] - r, 0), x1 = std::min(c[0] + r, dim[0] - 1);
        for (int z = z0; z <= z1; ++z)
            for (int y = y0; y <= y1; ++y)
                for (int x = x0; x <= x1; ++x) {
                    int cheb = std::max(std::abs(x - c[0]), std::max(std::abs(y - c[1]), std::abs(z - c[2])));
                    if (cheb != r) continue;
                    any = true;
                    int id = (z * dim[1] + y) * dim[0] + x;
                    for (int k = cell_start[id]; k < cell_start[id + 1]; ++k) f(order[k]);
                }
        return any;
    }
    int max_ring(const int c[3]) const
    {
        int m = 0;
        for (int a = 0; a < 3; ++a) m = std::max(m, std::max(c[a], dim[a] - 1 - c[a]));
        return m;
    }
    // k nearest neighbours sorted by (sqdist, index). returns count found (< k only if n < k).
    int knn(float x, float y, float z, int k, int* idx, float* sqd) const
    {
        std::vector<std::pair<float, int>> best;  // kept sorted ascending, size <= k
        int c[3];
        coords(x, y, z, c);
        int rmax = max_ring(c);
        for (int r = 0; r <= rmax; ++r) {
            if (static_cast<int>(best.size()) == k) {
                double bound = static_cast<double>(r - 1) * cell;  // unvisited points are farther than (r-1)*cell from q
                if (r >= 1 && static_cast<double>(best.back().first) < bound * bound * (1.0 - 1e-6)) break;
            }
            ring(c, r, [&](int i) {
                const float* p = pts + 4 * i;
                float d = sqdist_f(p[0], p[1], p[2], x, y, z);
                std::pair<float, int> e(d, i);
                if (static_cast<int>(best.size()) < k) {
                    best.insert(std::upper_bound(best.begin(), best.end(), e), e);
                } else if (e < best.back()) {
                    best.pop_back();
                    best.insert(std::upper_bound(best.begin(), best.end(), e), e);
                }
            });
        }
        for (size_t i = 0; i < best.size(); ++i) { idx[i] = best[i].second; sqd[i] = best[i].first; }
        return static_cast<int>(best.size());
    }
    int nearest(float x, float y, float z, float& sqd) const
    {
        int id = -1;
        float d = std::numeric_limits<float>::max();
        int got = knn(x, y, z, 1, &id, &d);
        sqd = d;
        return got ? id : -1;
    }
};

}  // namespace orc
