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

// Walks the grid ring by ring around the query's cell. `visit(p)` sees every candidate (xyz + index bits in w);
// `stop(lower_bound_sq)` is asked before each ring r >= 1 with a lower bound on the squared distance of everything
// not yet visited, ((r-1)*cell)^2 shrunk by 1e-5 against float rounding in the binning.
template <class Visit, class Stop>
__device__ __forceinline__ void nn_walk(const NnGridDev& g, const int c[3], int max_rings, Visit&& visit, Stop&& stop)
{
    int rmax = 0;
#pragma unroll
    for (int a = 0; a < 3; ++a) rmax = max(rmax, max(c[a], g.dim[a] - 1 - c[a]));
    if (max_rings >= 0) rmax = min(rmax, max_rings);
    for (int r = 0; r <= rmax; ++r) {
        if (r >= 1) {
            const double b = static_cast<double>(r - 1) * static_cast<double>(g.cell);
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
                    const int      x0 = max(c[0] - r, 0), x1 = min(c[0] + r, g.dim[0] - 1);
                    const uint32_t b = g.cell_start[row + x0], e = g.cell_start[row + x1 + 1];
                    for (uint32_t k = b; k < e; ++k) visit(g.sorted[k]);
                } else {
                    const int xa = c[0] - r, xb = c[0] + r;
                    if (xa >= 0) {
                        const uint32_t b = g.cell_start[row + xa], e = g.cell_start[row + xa + 1];
                        for (uint32_t k = b; k < e; ++k) visit(g.sorted[k]);
                    }
                    if (xb < g.dim[0]) {
                        const uint32_t b = g.cell_start[row + xb], e = g.cell_start[row + xb + 1];
                        for (uint32_t k = b; k < e; ++k) visit(g.sorted[k]);
                    }
                }
            }
        }
    }
}

__device__ __forceinline__ void nn_nearest(const NnGridDev& g, float x, float y, float z, int32_t& best_i, float& best_d)
{
    best_i = -1;
    best_d = INFINITY;
    int c[3];
    if (g.n == 0 || !nn_cell_of(g, x, y, z, c)) return;
    nn_walk(
        g, c, -1,
        [&](const float4& p) {
            const float   d = sqdist3f(p.x, p.y, p.z, x, y, z);
            const int32_t i = __float_as_int(p.w);
            if (d < best_d || (d == best_d && i < best_i)) { best_d = d; best_i = i; }
        },
        [&](double bound_sq) { return best_i >= 0 && static_cast<double>(best_d) < bound_sq; });
}

}  // namespace mrgfe
