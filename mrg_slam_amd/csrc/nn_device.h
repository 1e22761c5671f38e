// csrc/nn_device.h — device-side walk of the radix-sorted uniform grid (see nn_grid.h): shared by the query kernels
// of nn_grid.hip and by the GICP correspondence kernel.
#pragma once
#include <hip/hip_runtime.h>

#include "dev_float.h"
#include "dev_utils.h"
#include "nn_grid.h"

namespace mrgfe {

__device__ __forceinline__ bool nn_cell_of(const NnGridDev& g, float x, float y, float z, int c[3])
{
    if (!finite3(x, y, z)) return false;
    const float q[3] = {x, y, z};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float f = floorf((q[a] - g.origin[a]) / g.cell);
        f = fminf(fmaxf(f, 0.0f), static_cast<float>(g.dim[a] - 1));
        c[a] = static_cast<int>(f);
    }
    return true;
}

// Distance from the query to the nearest face of its own cell (0 when the query was clamped into the grid), minus the
// grid's binning slack: everything outside the Chebyshev ball of r cells around c is at least (r-1)*cell + this away.
__device__ __forceinline__ double nn_face_margin(const NnGridDev& g, const int c[3], float x, float y, float z)
{
    const float q[3] = {x, y, z};
    double      m = 1e300;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double lo = static_cast<double>(g.origin[a]) + static_cast<double>(c[a]) * static_cast<double>(g.cell);
        const double d0 = static_cast<double>(q[a]) - lo, d1 = lo + static_cast<double>(g.cell) - static_cast<double>(q[a]);
        m = fmin(m, fmin(d0, d1));
    }
    m -= static_cast<double>(g.slack);
    return m > 0.0 ? m : 0.0;
}

// Walks the grid ring by ring around cell c. `range(b, e)` receives every contiguous run [b, e) of g.sorted that
// belongs to the ring; `stop(lower_bound_sq)` is asked before each ring r >= 1 with a lower bound on the squared
// distance of everything not yet visited, ((r-1)*cell + margin)^2 shrunk by 1e-5 against float rounding.
template <class Range, class Stop>
__device__ __forceinline__ void nn_walk_ranges(const NnGridDev& g, const int c[3], double margin, int max_rings, Range&& range, Stop&& stop)
{
    int rmax = 0;
#pragma unroll
    for (int a = 0; a < 3; ++a) rmax = max(rmax, max(c[a], g.dim[a] - 1 - c[a]));
    if (max_rings >= 0) rmax = min(rmax, max_rings);
    for (int r = 0; r <= rmax; ++r) {
        if (r >= 1) {
            const double b = static_cast<double>(r - 1) * static_cast<double>(g.cell) + margin;
            if (stop(b * b * (1.0 - 1e-5))) break;
        }
        const int z0 = max(c[2] - r, 0), z1 = min(c[2] + r, g.dim[2] - 1);
        const int y0 = max(c[1] - r, 0), y1 = min(c[1] + r, g.dim[1] - 1);
        for (int z = z0; z <= z1; ++z) {
            const bool zface = (z - c[2] == r) || (c[2] - z == r);
            for (int y = y0; y <= y1; ++y) {
                const bool     face = zface || (y - c[1] == r) || (c[1] - y == r);
                const uint32_t row = (static_cast<uint32_t>(z) * g.dim[1] + y) * g.dim[0];
                if (face) {
                    const int x0 = max(c[0] - r, 0), x1 = min(c[0] + r, g.dim[0] - 1);
                    range(g.cell_start[row + x0], g.cell_start[row + x1 + 1]);
                } else {
                    const int xa = c[0] - r, xb = c[0] + r;
                    if (xa >= 0) range(g.cell_start[row + xa], g.cell_start[row + xa + 1]);
                    if (xb < g.dim[0]) range(g.cell_start[row + xb], g.cell_start[row + xb + 1]);
                }
            }
        }
    }
}

// Point-wise walk: `visit(p)` sees every candidate (xyz + index bits in w).
template <class Visit, class Stop>
__device__ __forceinline__ void nn_walk(const NnGridDev& g, const int c[3], double margin, int max_rings, Visit&& visit, Stop&& stop)
{
    nn_walk_ranges(
        g, c, margin, max_rings,
        [&](uint32_t b, uint32_t e) {
            for (uint32_t k = b; k < e; ++k) visit(g.sorted[k]);
        },
        stop);
}

// Group-cooperative exact 1-NN: the G consecutive lanes of a group (G = 2, 4, 8 or 16; group-aligned lane ids) call
// with the SAME query and different `sub` = lane % G, split the x-rows of each ring between them and meet in a
// (distance, index) min-reduction, so every lane returns the group's result.  Rings 0 and 1 are taken together as the
// nine x-rows of the 3x3x3 block: the common case costs one or two row scans per lane instead of ten dependent ones.
template <int G>
__device__ __forceinline__ void nn_group_min(float& d, int32_t& i)
{
#pragma unroll
    for (int m = 1; m < G; m <<= 1) {
        const float   od = __shfl_xor(d, m);
        const int32_t oi = __shfl_xor(i, m);
        if (oi >= 0 && (i < 0 || od < d || (od == d && oi < i))) { d = od; i = oi; }
    }
}

// One level of the group search.  Order of work, each step followed by a (distance, index) min over the group:
//   1. the query's own cell, candidates strided over the G lanes (coalesced);
//   2. the other 26 cells of the 3x3x3 block as x-rows, one row per lane, skipping rows and end cells whose box is
//      farther than the best so far — on dense clouds the nearest neighbour is centimetres away and almost all go;
//   3. rings 2 .. max_ring, rows pruned the same way.
// Returns true when the search is finished: everything not visited is farther than the best candidate, or farther
// than sqrt(max_sq), or the level is exhausted.  (best_d, best_i) carry over from an earlier level.
template <int G>
__device__ __forceinline__ bool nn_level_search(const NnGridDev& g, float x, float y, float z, int sub, int max_ring, double max_sq, int32_t& best_i, float& best_d)
{
    int c[3];
    nn_cell_of(g, x, y, z, c);  // the caller checked that the query is finite
    // distance from the query to the low / high face of its cell per axis, never negative (clamped queries), less the
    // binning slack: a point in a cell k > 0 cells away along an axis is at least face + (k - 1) * cell away along it
    const float q[3] = {x, y, z};
    double      flo[3], fhi[3], margin = 1e300;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double lo = static_cast<double>(g.origin[a]) + static_cast<double>(c[a]) * static_cast<double>(g.cell);
        flo[a] = fmax(static_cast<double>(q[a]) - lo - static_cast<double>(g.slack), 0.0);
        fhi[a] = fmax(lo + static_cast<double>(g.cell) - static_cast<double>(q[a]) - static_cast<double>(g.slack), 0.0);
        margin = fmin(margin, fmin(flo[a], fhi[a]));
    }
    auto consider = [&](const float4& p) {
        const float   d = sqdist3f(p.x, p.y, p.z, x, y, z);
        const int32_t i = __float_as_int(p.w);
        if (best_i < 0 || d < best_d || (d == best_d && i < best_i)) { best_d = d; best_i = i; }
    };
    auto scan = [&](uint32_t b, uint32_t e) {
        uint32_t k = b;
        for (; k + 2 <= e; k += 2) {  // two independent loads in flight
            const float4 p0 = g.sorted[k], p1 = g.sorted[k + 1];
            consider(p0);
            consider(p1);
        }
        if (k < e) consider(g.sorted[k]);
    };
    auto axis_lb = [&](int a, int d) {  // lower bound on the distance along axis a to a cell d cells away
        if (d == 0) return 0.0;
        return (d < 0 ? flo[a] : fhi[a]) + static_cast<double>((d < 0 ? -d : d) - 1) * static_cast<double>(g.cell);
    };
    auto finished = [&](int r) {  // before ring r: lower bound on everything outside the Chebyshev ball of r - 1 cells
        const double b = static_cast<double>(r - 1) * static_cast<double>(g.cell) + margin;
        const double b2 = b * b * (1.0 - 1e-5);
        return (best_i >= 0 && static_cast<double>(best_d) < b2) || b2 > max_sq;
    };
    // squared distance beyond which a cell cannot matter to this lane any more: its best so far (any candidate bounds
    // the answer from above) or the caller's cut-off
    auto cur_lim = [&]() { return fmin(best_i >= 0 ? static_cast<double>(best_d) : 1e300, max_sq); };
    int rmax = 0;
#pragma unroll
    for (int a = 0; a < 3; ++a) rmax = max(rmax, max(c[a], g.dim[a] - 1 - c[a]));

    {  // 1. own cell
        const uint32_t at = (static_cast<uint32_t>(c[2]) * g.dim[1] + c[1]) * g.dim[0] + c[0];
        const uint32_t b = g.cell_start[at], e = g.cell_start[at + 1];
        for (uint32_t k = b + sub; k < e; k += G) consider(g.sorted[k]);
        nn_group_min<G>(best_d, best_i);
    }
    if (rmax == 0) return true;
    if (!finished(1)) {  // 2. rest of the 3x3x3 block
        const double fx0 = flo[0] * flo[0], fx1 = fhi[0] * fhi[0];
        for (int j = sub; j < 9; j += G) {
            const double lim = cur_lim();
            const int dz = j / 3 - 1, dy = j % 3 - 1;
            const int zz = c[2] + dz, yy = c[1] + dy;
            if (zz < 0 || zz >= g.dim[2] || yy < 0 || yy >= g.dim[1]) continue;
            const double ly = axis_lb(1, dy), lz = axis_lb(2, dz);
            const double lyz = ly * ly + lz * lz;
            if (lyz * (1.0 - 1e-5) > lim) continue;
            const bool     left = c[0] > 0 && (lyz + fx0) * (1.0 - 1e-5) <= lim;
            const bool     right = c[0] + 1 < g.dim[0] && (lyz + fx1) * (1.0 - 1e-5) <= lim;
            const uint32_t row = (static_cast<uint32_t>(zz) * g.dim[1] + yy) * g.dim[0];
            if (j == 4) {  // own row: the own cell is done
                if (left) scan(g.cell_start[row + c[0] - 1], g.cell_start[row + c[0]]);
                if (right) scan(g.cell_start[row + c[0] + 1], g.cell_start[row + c[0] + 2]);
            } else {
                scan(g.cell_start[row + (left ? c[0] - 1 : c[0])], g.cell_start[row + (right ? c[0] + 1 : c[0]) + 1]);
            }
        }
        nn_group_min<G>(best_d, best_i);
    }
    const int rlast = min(rmax, max_ring);
    for (int r = 2; r <= rlast; ++r) {  // 3. rings
        if (finished(r)) return true;
        const int w = 2 * r + 1;
        // each x-row of the ring is one run (rows on a y/z face) or two single cells (x faces), cut down to the chord of
        // the sphere of the best distance so far; the run bounds of kRows rows are fetched together so their latencies
        // overlap (far rings are mostly empty rows) and the group shares its best after every such batch
        constexpr int kRows = 4;
        for (int j0 = 0; j0 < w * w; j0 += kRows * G) {
            uint32_t rb[kRows][2], re[kRows][2];
            const double lim = cur_lim();
#pragma unroll
            for (int u = 0; u < kRows; ++u) {
                rb[u][0] = re[u][0] = rb[u][1] = re[u][1] = 0u;
                const int j = j0 + sub + u * G;
                if (j >= w * w) continue;
                const int dz = j / w - r, dy = j % w - r;
                const int zz = c[2] + dz, yy = c[1] + dy;
                if (zz < 0 || zz >= g.dim[2] || yy < 0 || yy >= g.dim[1]) continue;
                const double ly = axis_lb(1, dy), lz = axis_lb(2, dz);
                const double lyz = (ly * ly + lz * lz) * (1.0 - 1e-5);
                if (lyz > lim) continue;
                // cells k >= 1 to the left / right can matter while face + (k - 1) * cell <= hx
                int kl = r, kr = r;
                if (lim < 1e299) {
                    const double hx = sqrt(lim - lyz) * (1.0 + 1e-5);
                    kl = hx >= flo[0] ? static_cast<int>(fmin((hx - flo[0]) / static_cast<double>(g.cell), 1e9)) + 1 : 0;
                    kr = hx >= fhi[0] ? static_cast<int>(fmin((hx - fhi[0]) / static_cast<double>(g.cell), 1e9)) + 1 : 0;
                }
                const uint32_t row = (static_cast<uint32_t>(zz) * g.dim[1] + yy) * g.dim[0];
                if (dz == r || dz == -r || dy == r || dy == -r) {
                    const int x0 = max(c[0] - min(r, kl), 0), x1 = min(c[0] + min(r, kr), g.dim[0] - 1);
                    rb[u][0] = g.cell_start[row + x0];
                    re[u][0] = g.cell_start[row + x1 + 1];
                } else {
                    const int xa = c[0] - r, xb = c[0] + r;
                    if (xa >= 0 && kl >= r) { rb[u][0] = g.cell_start[row + xa]; re[u][0] = g.cell_start[row + xa + 1]; }
                    if (xb < g.dim[0] && kr >= r) { rb[u][1] = g.cell_start[row + xb]; re[u][1] = g.cell_start[row + xb + 1]; }
                }
            }
#pragma unroll
            for (int u = 0; u < kRows; ++u) {
                scan(rb[u][0], re[u][0]);
                scan(rb[u][1], re[u][1]);
            }
            nn_group_min<G>(best_d, best_i);
        }
    }
    return rlast == rmax || finished(rlast + 1);
}

// Exact nearest neighbour among the points within sqrt(max_sq) of the query (max_sq = +inf: of all points); a best
// candidate farther than that may be returned and is the caller's to reject.
template <int G>
__device__ __forceinline__ void nn_nearest_group(const NnGrid2Dev& g, float x, float y, float z, int sub, double max_sq, int32_t& best_i, float& best_d)
{
    best_i = -1;
    best_d = INFINITY;
    if (g.level[0].n == 0 || !finite3(x, y, z)) return;  // uniform within the group
    for (int l = 0; l < g.n_levels; ++l)
        if (nn_level_search<G>(g.level[l], x, y, z, sub, l + 1 < g.n_levels ? g.fine_rings : 0x7fffffff, max_sq, best_i, best_d)) return;
}

}  // namespace mrgfe
