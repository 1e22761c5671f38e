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
__device__ __forceinline__ void nn_walk_ranges(const NnGridDev& g, const int c[3], double margin, int max_rings, Range&& range, Stop&& stop, int first_ring = 0)
{
    int rmax = 0;
#pragma unroll
    for (int a = 0; a < 3; ++a) rmax = max(rmax, max(c[a], g.dim[a] - 1 - c[a]));
    if (max_rings >= 0) rmax = min(rmax, max_rings);
    for (int r = first_ring; r <= rmax; ++r) {
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
                    range(as_global(g.cell_start)[row + x0], as_global(g.cell_start)[row + x1 + 1]);
                } else {
                    const int xa = c[0] - r, xb = c[0] + r;
                    if (xa >= 0) range(as_global(g.cell_start)[row + xa], as_global(g.cell_start)[row + xa + 1]);
                    if (xb < g.dim[0]) range(as_global(g.cell_start)[row + xb], as_global(g.cell_start)[row + xb + 1]);
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
            for (uint32_t k = b; k < e; ++k) visit(load_point(g.sorted + k));
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

// ---- occupancy pyramid ------------------------------------------------------------------------------------------------
// Nodes: bricks (4^3 cells), super-bricks (4^3 bricks), blocks (4^3 super-bricks); one 64-bit word each, bit x + 4 y + 16 z
// set iff the child holds a point.  An occupied node bounds the answer from BOTH sides without touching a point: nothing
// in it is nearer than its box (lb) and something in it is no farther than the far corner of its box (ub).  A query with
// an empty neighbourhood therefore first narrows `lim` — the squared radius that can still matter — from the words alone,
// and opens only cells with lb <= lim.  (Walking rings of cells instead cost ~400 dependent row probes per such query,
// and walking coarser point grids meant scanning 200-point cells.)
// Bounds are float with a margin of four binning slacks; comparisons carry another 2e-5 relative.
struct NnPyramidQuery {
    float t[3];  // query - origin
    float m;     // margin, metres
    __device__ __forceinline__ void box(float E, int nx, int ny, int nz, float& lb2, float& ub2) const
    {
        const int n[3] = {nx, ny, nz};
        lb2 = 0.0f;
        ub2 = 0.0f;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float lo = static_cast<float>(n[a]) * E, hi = lo + E;
            const float l = fmaxf(fmaxf(lo - t[a], t[a] - hi) - m, 0.0f);
            const float u = fmaxf(t[a] - lo, hi - t[a]) + m;
            lb2 += l * l;
            ub2 += u * u;
        }
    }
};
constexpr float kNnPrune = 1.0f + 2e-5f;
constexpr int   kNnBrickShells = 2;  // shells of nodes walked on a pyramid level before the next coarser one takes over

template <int G>
__device__ __forceinline__ float nn_group_fmin(float v)
{
#pragma unroll
    for (int k = 1; k < G; k <<= 1) v = fminf(v, __shfl_xor(v, k));
    return v;
}

// offsets (dx + 2) | (dy + 2) << 3 | (dz + 2) << 6 of the 5 x 5 x 5 nodes around a centre, shell 0 first, then the 26 of
// shell 1 (the six that share a face with the centre, the twelve that share an edge, the eight corners: nearest kind first, round 3),
// then the 98 of shell 2 by the number of their coordinates that
// are +-2 — 54 face nodes, 36 edge nodes, 8 corners: a node with k such coordinates is at least sqrt(k) (E + nm) away, so the walk
// of the shell can stop at a class boundary (24.4 -> 23.9 ms)
__device__ __forceinline__ void nn_small_shell_pos(int idx, int d[3])
{
    static constexpr uint16_t kTab[125] = {146, 82, 210, 138, 154, 145, 147, 74, 81, 83, 90, 202, 209, 211, 218, 137, 139, 153, 155, 73, 75, 89, 91, 201, 203, 217, 219, 18, 274, 130, 162, 144, 148, 10, 17, 19, 26, 266, 273, 275, 282, 66, 98, 80, 84, 129, 131, 161, 163, 136, 140, 152, 156, 194, 226, 208, 212, 9, 11, 25, 27, 265, 267, 281, 283, 65, 67, 97, 99, 72, 76, 88, 92, 193, 195, 225, 227, 200, 204, 216, 220, 2, 16, 20, 34, 258, 272, 276, 290, 128, 132, 160, 164, 1, 3, 8, 12, 24, 28, 33, 35, 257, 259, 264, 268, 280, 284, 289, 291, 64, 68, 96, 100, 192, 196, 224, 228, 0, 4, 32, 36, 256, 260, 288, 292};
    const int v = kTab[idx];
    d[0] = (v & 7) - 2;
    d[1] = ((v >> 3) & 7) - 2;
    d[2] = (v >> 6) - 2;
}

// Shells s_first .. s_last of the nodes around `ctr` on one pyramid level.  Each lane fetches the words of two nodes of
// the shell (those whose box can still matter); the group then calls proc(word, nx, ny, nz) for every non-empty one, all
// lanes together, and proc leaves `lim` agreed within the group.  Returns true when the search is over: the next shell
// lies beyond lim, or the level has no more nodes.
template <int G, bool kSmall, class Proc>
__device__ __forceinline__ bool nn_shell_walk(const unsigned long long* __restrict__ words, const int dims[3], const int ctr[3], float E, const NnPyramidQuery& pq, int s_first,
                                              int s_last, int sub, float& lim, Proc&& proc)
{
    constexpr int U = G >= 8 ? 2 : (G == 4 ? 4 : 8);  // words a lane fetches per trip: sixteen to the group (eight for one or two lanes: registers)
    int   smax = 0;
    float nm = INFINITY;  // distance to the nearest face of the centre node, >= 0
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        smax = max(smax, max(ctr[a], dims[a] - 1 - ctr[a]));
        const float lo = static_cast<float>(ctr[a]) * E;
        nm = fminf(nm, fminf(pq.t[a] - lo, lo + E - pq.t[a]));
    }
    nm = fmaxf(nm - pq.m, 0.0f);
    const int base = static_cast<int>(lane_id()) & ~(G - 1);
    const int send = min(s_last, smax);
    for (int s = s_first; s <= send; ++s) {
        if (s >= 1) {  // everything outside the Chebyshev ball of s - 1 nodes
            const float b = static_cast<float>(s - 1) * E + nm;
            if (b * b > lim * kNnPrune) return true;
        }
        const int  w = 2 * s + 1, per_z = 4 * w - 4;
        const bool merged = kSmall && s == 0 && send >= 1;  // the centre node and shell 1 in one go: one round trip less
        const int  total = merged ? 27 : s == 0 ? 1 : 2 * w * w + (w - 2) * per_z;
        const int  first = s == 0 ? 0 : s == 1 ? 1 : 27;  // kSmall: where the shell starts in the table
        auto shell_pos = [&](int tt, int d[3]) {  // tt-th node of the shell: the two z faces, then per middle z the y faces and the x ends
            if (kSmall) { nn_small_shell_pos(first + tt, d); return; }
            if (s == 0) { d[0] = d[1] = d[2] = 0; return; }
            if (tt < 2 * w * w) {
                const int u = tt < w * w ? tt : tt - w * w;
                const int qy = u / w;
                d[2] = tt < w * w ? -s : s;
                d[1] = qy - s;
                d[0] = u - qy * w - s;
                return;
            }
            const int u = tt - 2 * w * w, qz = u / per_z, v = u - qz * per_z;
            d[2] = qz - s + 1;
            if (v < 2 * w) {
                d[1] = v < w ? -s : s;
                d[0] = (v < w ? v : v - w) - s;
            } else {
                const int k = v - 2 * w;
                d[1] = (k >> 1) - s + 1;
                d[0] = (k & 1) ? s : -s;
            }
        };
        for (int t0 = 0; t0 < total; t0 += U * G) {
            if (kSmall && s == 2) {  // the table lists shell 2 nearest class first: faces [0, 54), edges [54, 90), corners [90, 98)
                const float kmin = t0 >= 90 ? 3.0f : (t0 >= 54 ? 2.0f : 1.0f), b = E + nm;
                if (kmin * b * b > lim * kNnPrune) break;
            }
            unsigned long long word[U];
            float              wlb[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                word[u] = 0ull;
                wlb[u] = INFINITY;
                const int tt = t0 + u * G + sub;
                if (tt >= total) continue;
                int d[3];
                shell_pos(tt, d);
                const int nx = ctr[0] + d[0], ny = ctr[1] + d[1], nz = ctr[2] + d[2];
                if (nx < 0 || nx >= dims[0] || ny < 0 || ny >= dims[1] || nz < 0 || nz >= dims[2]) continue;
                float lb2, ub2;
                pq.box(E, nx, ny, nz, lb2, ub2);
                if (lb2 <= lim * kNnPrune) {
                    word[u] = as_global(words)[(static_cast<uint32_t>(nz) * dims[1] + ny) * dims[0] + nx];
                    wlb[u] = word[u] != 0ull ? lb2 : INFINITY;
                }
            }
            // non-empty nodes nearest first: the first ones usually pull lim in far enough to drop the rest unopened
            for (;;) {
                float nb = wlb[0];
                int   ub = 0;
#pragma unroll
                for (int u = 1; u < U; ++u)
                    if (wlb[u] < nb) { nb = wlb[u]; ub = u; }
                int code = ub * G + sub;  // u * G + lane of the group
#pragma unroll
                for (int k = 1; k < G; k <<= 1) {
                    const float ob = __shfl_xor(nb, k);
                    const int   oc = __shfl_xor(code, k);
                    if (ob < nb || (ob == nb && oc < code)) { nb = ob; code = oc; }
                }
                if (nb == INFINITY || !(nb <= lim * kNnPrune)) break;  // nothing left / nothing near enough (uniform within the group)
                const int u = code / G, b = code - u * G;
                unsigned long long wsel = word[0];
#pragma unroll
                for (int v = 1; v < U; ++v)
                    if (u == v) wsel = word[v];
#pragma unroll
                for (int v = 0; v < U; ++v)
                    if (b == sub && u == v) wlb[v] = INFINITY;
                {
                    unsigned long long wb = wsel;
                    if (G > 1) {
                        const uint32_t wlo = __shfl(static_cast<uint32_t>(wsel), base + b), whi = __shfl(static_cast<uint32_t>(wsel >> 32), base + b);
                        wb = (static_cast<unsigned long long>(whi) << 32) | wlo;
                    }
                    int d[3];
                    shell_pos(t0 + u * G + b, d);
                    proc(wb, ctr[0] + d[0], ctr[1] + d[1], ctr[2] + d[2]);
                }
            }
        }
        if (merged) ++s;
    }
    if (send == smax) return true;
    const float b = static_cast<float>(send) * E + nm;
    return b * b > lim * kNnPrune;
}

// The lane's share of the set bits of a node's word: the bit POSITIONS dealt to it (one AND), not every G-th set bit (a loop over
// the word's population; the far pass is bound by VALU issue, 0.76 busy: 25.5 -> 25.0 ms for config[3]'s 256 pairs).  For groups of eight, position p = x + 4 y + 16 z (two bits per
// axis) goes to lane (x0 ^ y1) | (x1 ^ z0) << 1 | (y0 ^ z1) << 2: every axis-aligned plane of the 4 x 4 x 4 node — a wall, the
// ground — spreads over all eight lanes, two cells each.  (The plain pattern p % 8 hands a wall to one lane: 26.6 ms.)
__device__ __forceinline__ unsigned long long nn_deal_mask(int G, int sub)
{
    if (G == 8) {
        static constexpr unsigned long long kDeal8[8] = {0x8040201008040201ull, 0x4080102004080102ull, 0x2010804002010804ull, 0x1020408001020408ull,
                                                        0x0804020180402010ull, 0x0408010240801020ull, 0x0201080420108040ull, 0x0102040810204080ull};
        return kDeal8[sub & 7];
    }
    if (G == 4) {  // the same mixing, two of the eight shares per lane
        static constexpr unsigned long long kDeal4[4] = {0x8040201008040201ull | 0x0804020180402010ull, 0x4080102004080102ull | 0x0408010240801020ull,
                                                        0x2010804002010804ull | 0x0201080420108040ull, 0x1020408001020408ull | 0x0102040810204080ull};
        return kDeal4[sub & 3];
    }
    if (G == 1) return ~0ull;
    if (G == 2) return (sub & 1) ? 0x55aa55aa55aa55aaull : 0xaa55aa55aa55aa55ull;  // by the parity of x0 ^ y1
    return 0x0001000100010001ull << sub;
}

// Exhaustive remainder of a 1-NN search after the query's 3x3x3 block of cells: bricks within two shells of the query's
// brick, then super-bricks within two shells (descending through their bricks), then all blocks.
// kByPos: (best_i, best_d) name the candidate by its position in g.sorted instead of its original index (ties then go to
// the lowest position; callers that only want the distance, or the point itself, save the index indirection).
// `bound`: a squared distance already known to be attained by some point (INFINITY: none) that has no (best_i, best_d).
template <int G, bool kByPos = false>
__device__ __forceinline__ void nn_pyramid_walk(const NnGridDev& g, float x, float y, float z, const int c[3], int sub, double max_sq, int32_t& best_i, float& best_d,
                                                float bound = INFINITY)
{
    NnPyramidQuery pq;
    pq.t[0] = x - g.origin[0];
    pq.t[1] = y - g.origin[1];
    pq.t[2] = z - g.origin[2];
    pq.m = 4.0f * g.slack;
    const float E0 = g.cell, E1 = 4.0f * g.cell, E2 = 16.0f * g.cell, E3 = 64.0f * g.cell;
    float lim = max_sq >= 3.0e38 ? INFINITY : static_cast<float>(max_sq) * (1.0f + 1e-6f);
    lim = fminf(lim, bound);
    if (best_i >= 0) lim = fminf(lim, best_d);
    constexpr int kShells = kNnBrickShells;
    const unsigned long long deal = nn_deal_mask(G, sub);
    const int b1[3] = {c[0] >> 2, c[1] >> 2, c[2] >> 2}, b2[3] = {c[0] >> 4, c[1] >> 4, c[2] >> 4}, b3[3] = {c[0] >> 6, c[1] >> 6, c[2] >> 6};
    const int d1[3] = {g.bdim[0], g.bdim[1], g.bdim[2]};
    const int d2[3] = {(d1[0] + 3) >> 2, (d1[1] + 3) >> 2, (d1[2] + 3) >> 2};
    const int d3[3] = {(d2[0] + 3) >> 2, (d2[1] + 3) >> 2, (d2[2] + 3) >> 2};

    // the lane's share `mine` of the cells of brick (bx, by, bz): narrow lim from the occupancy alone and open the cells that can still matter,
    // in one pass (bounding all of the brick's cells before opening any cost a second box evaluation per cell: far pass 25.0 -> 24.4 ms)
    auto cells_open = [&](unsigned long long mine, int bx, int by, int bz) {
        for (unsigned long long w = mine; w != 0ull; w &= w - 1ull) {
            const int bit = __ffsll(w) - 1;
            const int cx = bx * 4 + (bit & 3), cy = by * 4 + ((bit >> 2) & 3), cz = bz * 4 + (bit >> 4);
            float lb2, ub2;
            pq.box(E0, cx, cy, cz, lb2, ub2);
            lim = fminf(lim, ub2 * kNnPrune);  // an occupied cell: something is no farther than its far corner
            if (cx - c[0] >= -1 && cx - c[0] <= 1 && cy - c[1] >= -1 && cy - c[1] <= 1 && cz - c[2] >= -1 && cz - c[2] <= 1) continue;  // the block is done
            if (lb2 > lim * kNnPrune) continue;
            const uint32_t at = (static_cast<uint32_t>(cz) * g.dim[1] + cy) * g.dim[0] + cx;
            const uint32_t kb = as_global(g.cell_start)[at], ke = as_global(g.cell_start)[at + 1];
            for (uint32_t k = kb; k < ke; ++k) {
                const float4  p = load_point(g.sorted + k);
                const float   d = sqdist3f(p.x, p.y, p.z, x, y, z);
                const int32_t i = kByPos ? static_cast<int32_t>(k) : __float_as_int(p.w);
                if (best_i < 0 || d < best_d || (d == best_d && i < best_i)) { best_d = d; best_i = i; }
            }
            if (best_i >= 0) lim = fminf(lim, best_d);
        }
    };
    auto agree = [&]() {
        nn_group_min<G>(best_d, best_i);
        lim = nn_group_fmin<G>(lim);
        if (best_i >= 0) lim = fminf(lim, best_d);
    };
    // the lane's share of the bricks of super-brick (sx, sy, sz), except those within `done` bricks of the query's
    auto bricks = [&](unsigned long long mine, int sx, int sy, int sz, int done) {
        for (unsigned long long w = mine; w != 0ull; w &= w - 1ull) {
            const int bit = __ffsll(w) - 1;
            const int bx = sx * 4 + (bit & 3), by = sy * 4 + ((bit >> 2) & 3), bz = sz * 4 + (bit >> 4);
            float     lb2, ub2;
            pq.box(E1, bx, by, bz, lb2, ub2);
            lim = fminf(lim, ub2 * kNnPrune);
            if (lb2 > lim * kNnPrune) continue;
            if (bx - b1[0] >= -done && bx - b1[0] <= done && by - b1[1] >= -done && by - b1[1] <= done && bz - b1[2] >= -done && bz - b1[2] <= done) continue;
            const unsigned long long w0 = as_global(g.occ)[(static_cast<uint32_t>(bz) * d1[1] + by) * d1[0] + bx];
            cells_open(w0, bx, by, bz);
        }
    };

    // A. bricks
    if (nn_shell_walk<G, true>(g.occ, d1, b1, E1, pq, 0, kShells, sub, lim, [&](unsigned long long wb, int bx, int by, int bz) {
            cells_open(wb & deal, bx, by, bz);
            agree();
        }))
        return;
    // B. super-bricks
    if (nn_shell_walk<G, true>(g.occ1, d2, b2, E2, pq, 0, kShells, sub, lim, [&](unsigned long long wb, int sx, int sy, int sz) {
            bricks(wb & deal, sx, sy, sz, kShells);
            agree();
        }))
        return;
    // C. blocks, to the end of the grid
    nn_shell_walk<G, false>(g.occ2, d3, b3, E3, pq, 0, 0x3fffffff, sub, lim, [&](unsigned long long wb, int kx, int ky, int kz) {
        const unsigned long long mine = wb & deal;
        for (unsigned long long w = mine; w != 0ull; w &= w - 1ull) {
            const int bit = __ffsll(w) - 1;
            const int sx = kx * 4 + (bit & 3), sy = ky * 4 + ((bit >> 2) & 3), sz = kz * 4 + (bit >> 4);
            float     lb2, ub2;
            pq.box(E2, sx, sy, sz, lb2, ub2);
            lim = fminf(lim, ub2 * kNnPrune);
            if (lb2 > lim * kNnPrune) continue;
            if (sx - b2[0] >= -kShells && sx - b2[0] <= kShells && sy - b2[1] >= -kShells && sy - b2[1] <= kShells && sz - b2[2] >= -kShells && sz - b2[2] <= kShells) continue;
            bricks(as_global(g.occ1)[(static_cast<uint32_t>(sz) * d2[1] + sy) * d2[0] + sx], sx, sy, sz, -1);
        }
        agree();
    });
}

// One level of the group search.  Order of work, each step followed by a (distance, index) min over the group:
//   1. the query's own cell, candidates strided over the G lanes (coalesced);
//   2. the other 26 cells of the 3x3x3 block as x-rows, one row per lane, skipping rows and end cells whose box is
//      farther than the best so far — on dense clouds the nearest neighbour is centimetres away and almost all go;
//   3. (max_ring > 1) the rest of the grid through the occupancy pyramid, see nn_pyramid_walk.
// Returns true when the search is finished: everything not visited is farther than the best candidate or farther
// than sqrt(max_sq); false only for max_ring <= 1 when the block was not conclusive.
template <int G>
__device__ __forceinline__ bool nn_level_search(const NnGridDev& g, float x, float y, float z, int sub, int max_ring, double max_sq, int32_t& best_i, float& best_d)
{
    int c[3];
    nn_cell_of(g, x, y, z, c);  // the caller checked that the query is finite
    // distance from the query to the low / high face of its cell per axis, never negative (clamped queries), less a
    // margin of four binning slacks (float bounds, see NnPyramidQuery): a point in a cell k > 0 cells away along an axis is
    // at least face + (k - 1) * cell away along it
    const float m = 4.0f * g.slack;
    const float t[3] = {x - g.origin[0], y - g.origin[1], z - g.origin[2]};
    const float max_sq_f = max_sq >= 3.0e38 ? INFINITY : static_cast<float>(max_sq) * (1.0f + 1e-6f);
    float       flo[3], fhi[3], margin = INFINITY;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float lo = static_cast<float>(c[a]) * g.cell;
        flo[a] = fmaxf(t[a] - lo - m, 0.0f);
        fhi[a] = fmaxf(lo + g.cell - t[a] - m, 0.0f);
        margin = fminf(margin, fminf(flo[a], fhi[a]));
    }
    auto consider = [&](const float4& p) {
        const float   d = sqdist3f(p.x, p.y, p.z, x, y, z);
        const int32_t i = __float_as_int(p.w);
        if (best_i < 0 || d < best_d || (d == best_d && i < best_i)) { best_d = d; best_i = i; }
    };
    auto scan = [&](uint32_t b, uint32_t e) {
        uint32_t k = b;
        for (; k + 2 <= e; k += 2) {  // two independent loads in flight
            const float4 p0 = load_point(g.sorted + k), p1 = load_point(g.sorted + k + 1);
            consider(p0);
            consider(p1);
        }
        if (k < e) consider(load_point(g.sorted + k));
    };
    auto axis_lb = [&](int a, int d) {  // lower bound on the distance along axis a to a cell d cells away
        if (d == 0) return 0.0f;
        return (d < 0 ? flo[a] : fhi[a]) + static_cast<float>((d < 0 ? -d : d) - 1) * g.cell;
    };
    auto finished = [&](int r) {  // before ring r: lower bound on everything outside the Chebyshev ball of r - 1 cells
        const float b = static_cast<float>(r - 1) * g.cell + margin;
        const float b2 = b * b;
        return (best_i >= 0 && best_d * kNnPrune < b2) || b2 > max_sq_f * kNnPrune;
    };
    // squared distance beyond which a cell cannot matter to this lane any more: its best so far (any candidate bounds
    // the answer from above) or the caller's cut-off, with the comparison slack
    auto cur_lim = [&]() { return fminf(best_i >= 0 ? best_d : INFINITY, max_sq_f) * kNnPrune; };
    int rmax = 0;
#pragma unroll
    for (int a = 0; a < 3; ++a) rmax = max(rmax, max(c[a], g.dim[a] - 1 - c[a]));

    {  // 1. own cell
        const uint32_t at = (static_cast<uint32_t>(c[2]) * g.dim[1] + c[1]) * g.dim[0] + c[0];
        const uint32_t b = as_global(g.cell_start)[at], e = as_global(g.cell_start)[at + 1];
        for (uint32_t k = b + sub; k < e; k += G) consider(load_point(g.sorted + k));
        nn_group_min<G>(best_d, best_i);
    }
    if (rmax == 0) return true;
    if (!finished(1)) {  // 2. rest of the 3x3x3 block
        const float fx0 = flo[0] * flo[0], fx1 = fhi[0] * fhi[0];
        for (int j = sub; j < 9; j += G) {
            const float lim = cur_lim();
            const int dz = j / 3 - 1, dy = j % 3 - 1;
            const int zz = c[2] + dz, yy = c[1] + dy;
            if (zz < 0 || zz >= g.dim[2] || yy < 0 || yy >= g.dim[1]) continue;
            const float ly = axis_lb(1, dy), lz = axis_lb(2, dz);
            const float lyz = ly * ly + lz * lz;
            if (lyz > lim) continue;
            const bool left = c[0] > 0 && lyz + fx0 <= lim;
            const bool right = c[0] + 1 < g.dim[0] && lyz + fx1 <= lim;
            const uint32_t row = (static_cast<uint32_t>(zz) * g.dim[1] + yy) * g.dim[0];
            if (j == 4) {  // own row: the own cell is done
                if (left) scan(as_global(g.cell_start)[row + c[0] - 1], as_global(g.cell_start)[row + c[0]]);
                if (right) scan(as_global(g.cell_start)[row + c[0] + 1], as_global(g.cell_start)[row + c[0] + 2]);
            } else {
                scan(as_global(g.cell_start)[row + (left ? c[0] - 1 : c[0])], as_global(g.cell_start)[row + (right ? c[0] + 1 : c[0]) + 1]);
            }
        }
        nn_group_min<G>(best_d, best_i);
    }
    if (finished(2) || rmax <= 1) return true;
    if (max_ring <= 1) return false;  // block-only pass (nn_fit_block_kernel)
    nn_pyramid_walk<G>(g, x, y, z, c, sub, max_sq, best_i, best_d);  // 3. everything else
    return true;
}

// Exact nearest neighbour among the points within sqrt(max_sq) of the query (max_sq = +inf: of all points); a best
// candidate farther than that may be returned and is the caller's to reject.
template <int G>
__device__ __forceinline__ void nn_nearest_group(const NnGrid2Dev& g, float x, float y, float z, int sub, double max_sq, int32_t& best_i, float& best_d)
{
    best_i = -1;
    best_d = INFINITY;
    if (g.level[0].n == 0 || !finite3(x, y, z)) return;  // uniform within the group
    nn_level_search<G>(g.level[0], x, y, z, sub, 0x7fffffff, max_sq, best_i, best_d);
}

// The part of a 1-NN search that follows an inconclusive look at the query's 3x3x3 block (nn_level_search with
// max_ring <= 1), for callers that want the nearest POINT rather than its index: `bound` is the best squared distance the
// block gave (INFINITY: none) or any other distance known to be attained; on return best_pos >= 0 names the position in
// level[0].sorted of the nearest point found outside the block with its squared distance best_d, if one is within the
// bound; the answer is the smaller of bound and best_d.
template <int G>
__device__ __forceinline__ void nn_far_search(const NnGrid2Dev& g, float x, float y, float z, int sub, double max_sq, float bound, int32_t& best_pos, float& best_d)
{
    best_pos = -1;
    best_d = INFINITY;
    int c[3];
    nn_cell_of(g.level[0], x, y, z, c);
    nn_pyramid_walk<G, true>(g.level[0], x, y, z, c, sub, max_sq, best_pos, best_d, bound);
}

}  // namespace mrgfe
