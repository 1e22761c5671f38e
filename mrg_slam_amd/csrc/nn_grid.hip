// csrc/nn_grid.hip — exact nearest-neighbour search on a radix-sorted uniform grid (see nn_grid.h).
//
// Memory behaviour: the query kernels read one query (16 B, coalesced) and then a handful of contiguous candidate
// ranges (16 B per candidate) out of a cloud that was reordered cell-major at build time, so neighbouring lanes —
// which hold spatially neighbouring queries for a LiDAR scan — hit the same cache lines.  HBM/L2 bound, no MFMA.
#include <atomic>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "dev_float.h"
#include "dev_utils.h"
#include "nn_device.h"
#include "nn_grid.h"
#include "bbox_device.h"
#include "bbox_device.h"

namespace mrgfe {

// ---- build ---------------------------------------------------------------------------------------------------
constexpr uint32_t kCrowdSlots = 32;
__device__ __forceinline__ void nn_cellkey_body(const float4* __restrict__ pts, uint32_t n, const NnGridDev& g, uint32_t n_cells, uint32_t* __restrict__ keys,
                                                uint32_t* __restrict__ vals, uint32_t* __restrict__ counts, unsigned long long* __restrict__ crowd)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    uint32_t       key = n_cells;  // non-finite points: behind every cell
    if (i < n) {
        const float4 p = pts[i];
        int          c[3];
        if (nn_cell_of(g, p.x, p.y, p.z, c)) key = (static_cast<uint32_t>(c[2]) * g.dim[1] + c[1]) * g.dim[0] + c[0];
        keys[i] = key;  // (no index array: the sort's first scatter makes the indices up)
    }
    if (counts == nullptr) return;  // uniform: full builds take the cell table from the SORTED keys (nn_fill_body), only the adaptive counting passes count here
    // count per cell, one atomic per distinct cell of the wavefront (consecutive points of a scan crowd into few cells);
    // `before` = points counted into this point's cell ahead of it: summed over the cloud it is sum_c n_c (n_c - 1) / 2
    uint32_t before = 0;
    bool     todo = i < n && key < n_cells;
    // ... when at least half of the lanes share their cell with the lane before them; otherwise plain per-lane atomics
    if (__popcll(__ballot(todo && key == __shfl_up(key, 1) && lane_id() > 0)) < 32) {
        if (todo) before = atomicAdd(&counts[key], 1u);
        todo = false;
    }
    while (__ballot(todo)) {
        const uint64_t pending = __ballot(todo);
        const int      leader = __ffsll(static_cast<unsigned long long>(pending)) - 1;
        const uint32_t lkey = wave_read(key, leader);
        const bool     mine = todo && key == lkey;
        const uint64_t grp = __ballot(mine);
        if (mine) {
            uint32_t old = 0;
            if (lane_id() == leader) old = atomicAdd(&counts[lkey], static_cast<uint32_t>(__popcll(grp)));
            old = wave_read(old, leader);
            before = old + static_cast<uint32_t>(__popcll(grp & ((1ull << lane_id()) - 1ull)));
            todo = false;
        }
    }
    if (crowd == nullptr) return;  // uniform: only the adaptive passes ask for the crowding figure
    __shared__ uint32_t s_w[4];
    const uint32_t w = wave_sum(before);
    if (lane_id() == 0) s_w[wave_id()] = w;
    __syncthreads();
    // one atomic per workgroup, spread over kCrowdSlots addresses (two thousand wavefronts adding to ONE address took 20+ us)
    if (threadIdx.x == 0) {
        const uint32_t t = s_w[0] + s_w[1] + s_w[2] + s_w[3];
        if (t) atomicAdd(&crowd[blockIdx.x % kCrowdSlots], static_cast<unsigned long long>(t));
    }
}

__global__ __launch_bounds__(256) void nn_cellkey_kernel(const float4* __restrict__ pts, uint32_t n, NnGridDev g, uint32_t n_cells, uint32_t* __restrict__ keys,
                                                          uint32_t* __restrict__ vals, uint32_t* __restrict__ counts, unsigned long long* __restrict__ crowd)
{
    nn_cellkey_body(pts, n, g, n_cells, keys, vals, counts, crowd);
}

// next pyramid level: bit of a child node set iff its word is non-zero (dims = child grid, pdim = parent grid)
__device__ __forceinline__ void nn_occupancy_up_body(const unsigned long long* __restrict__ child, int dx, int dy, int dz, int px, int py, unsigned long long* __restrict__ parent)
{
    const uint32_t at = blockIdx.x * 256u + threadIdx.x;
    if (at >= static_cast<uint32_t>(dx) * dy * dz || child[at] == 0ull) return;
    const uint32_t x = at % dx, y = (at / dx) % dy, z = at / (static_cast<uint32_t>(dx) * dy);
    atomicOr(&parent[((z >> 2) * py + (y >> 2)) * px + (x >> 2)], 1ull << ((x & 3u) | ((y & 3u) << 2) | ((z & 3u) << 4)));
}
__global__ __launch_bounds__(256) void nn_occupancy_up_kernel(const unsigned long long* __restrict__ child, int dx, int dy, int dz, int px, int py, unsigned long long* __restrict__ parent)
{
    nn_occupancy_up_body(child, dx, dy, dz, px, py, parent);
}

// The cell table straight from the sorted keys (round 4).  After the stable sort, cell_start[c] is the position of the first sorted point
// whose key is >= c: a point whose key differs from its predecessor's (a run head, at position i) owns the entries (previous key, own key]
// and writes i into them — its own entry itself, the gap of empty cells before it through a loop of its wavefront (64 consecutive entries
// per store instruction; a 2^24-cell table over a 130k-point scan is 99.6 % gaps).  Thread n, one past the last point, closes the table
// with n.  Every entry is written exactly once: no zeroing of the table, no atomic counting, no scan over 2^24 + 1 entries, and the
// occupancy bits of the bricks come from the run heads (one atomic per OCCUPIED cell) instead of a kernel that reads the whole table —
// the dense-table passes were 13 x the target points in traffic and ~9 ms of co-running kernels per config[3] step.  The same launch
// gathers the points into sorted order and — for builds that measure their own crowding — adds up rank-in-run = (position - position of
// the run's head), which is the figure the counting atomics used to produce: sum over cells of n (n - 1) / 2.
// Long gaps.  A scan's cells are x-fastest, so the gap in front of the first occupied cell of a row is a few cells, of a plane a few rows — and
// of the first occupied plane, or behind the last one, MILLIONS of entries: one wavefront writing such a gap alone (a 256-byte store at a
// time) was the whole kernel's tail (63 us for a 2^24-entry table whose 64 MB go out in 20 us at the chip's rate).  Gaps of kLongGap entries
// or more are therefore cut into chunks of kGapChunk entries and queued; nn_fill_long_kernel writes the chunks, a workgroup each.
constexpr uint32_t kLongGap = 4096, kGapChunk = 16384, kGapCap = 8192;
struct NnGapQueue {  // zeroed with the pyramid words before every build
    uint32_t count, pad[3];
    uint4    e[kGapCap];  // first entry, one past the last, value, -
};

__device__ __forceinline__ void nn_fill_body(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals, const float4* __restrict__ pts, uint32_t n, const NnGridDev& g,
                                             uint32_t n_cells, uint32_t* __restrict__ cell_start, unsigned long long* __restrict__ occ, float4* __restrict__ sorted,
                                             unsigned long long* __restrict__ crowd, NnGapQueue* __restrict__ gapq)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const int      lane = lane_id();
    uint32_t       k = n_cells, kp = n_cells;
    bool           head = false;
    if (i < n) {
        k = keys[i];
        kp = i ? keys[i - 1] : 0xffffffffu;  // (kp + 1 wraps to entry 0 for the first point)
        head = i == 0 || k != kp;
    } else if (i == n) {  // closes the table: entries (last key, n_cells] = n
        kp = keys[n - 1];
        head = kp != n_cells;
    }
    if (head) cell_start[k] = i;
    const uint32_t c0 = kp + 1u;
    uint64_t       gaps = __ballot(head && c0 != k);
    while (gaps) {
        const int l = __ffsll(static_cast<unsigned long long>(gaps)) - 1;
        gaps &= gaps - 1;
        const uint32_t a = wave_read(c0, l), b = wave_read(k, l), v = wave_read(i, l);
        if (gapq != nullptr && b - a >= kLongGap) {  // uniform
            const uint32_t nch = (b - a + kGapChunk - 1) / kGapChunk;
            uint32_t       base = 0;
            if (lane == 0) base = atomicAdd(&gapq->count, nch);
            base = wave_read(base, 0);
            if (base + nch <= kGapCap) {
                for (uint32_t c = static_cast<uint32_t>(lane); c < nch; c += 64u) {
                    const uint32_t lo = a + c * kGapChunk, hi = (c + 1 == nch) ? b : lo + kGapChunk;
                    gapq->e[base + c] = make_uint4(lo, hi, v, 0u);
                }
                continue;
            }  // (a full queue: the wavefront writes the gap itself; the reserved slots stay zero = empty chunks)
        }
        for (uint32_t c = a + static_cast<uint32_t>(lane); c < b; c += 64u) cell_start[c] = v;
    }
    if (head && i < n && k < n_cells && occ != nullptr) {
        const uint32_t d0 = static_cast<uint32_t>(g.dim[0]), d1 = static_cast<uint32_t>(g.dim[1]);
        const uint32_t t = k / d0, x = k - t * d0, z = t / d1, y = t - z * d1;
        atomicOr(&occ[((z >> 2) * g.bdim[1] + (y >> 2)) * g.bdim[0] + (x >> 2)], 1ull << ((x & 3u) | ((y & 3u) << 2) | ((z & 3u) << 4)));
    }
    if (i < g.n) {  // the finite points: sorted order, original index in w
        const uint32_t v = vals[i];
        float4 p = pts[v];
        p.w = __int_as_float(static_cast<int>(v));
        sorted[i] = p;
    }
    if (crowd == nullptr) return;  // uniform
    uint32_t before = 0;
    {
        const uint64_t hm = __ballot(head && i < n);
        // the run of a lane without a head at or below it began before this wavefront: where, says a bisection of the sorted keys (uniform)
        uint32_t run0 = 0;
        const uint32_t i0 = i - static_cast<uint32_t>(lane);
        if (!(hm & 1ull) && i0 < n) {
            const uint32_t k0 = keys[i0];
            uint32_t lo = 0, hi = i0;  // first position whose key is >= k0
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (keys[mid] < k0) lo = mid + 1; else hi = mid;
            }
            run0 = lo;
        }
        if (i < n && k < n_cells) {
            const uint64_t below = hm & ((2ull << lane) - 1ull);
            const uint32_t hp = below ? i0 + static_cast<uint32_t>(63 - __clzll(static_cast<long long>(below))) : run0;
            before = i - hp;
        }
    }
    __shared__ uint32_t s_w[4];
    const uint32_t w = wave_sum(before);
    if (lane == 0) s_w[wave_id()] = w;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t t = s_w[0] + s_w[1] + s_w[2] + s_w[3];
        if (t) atomicAdd(&crowd[blockIdx.x % kCrowdSlots], static_cast<unsigned long long>(t));
    }
}

__device__ __forceinline__ void nn_fill_long_body(const NnGapQueue* __restrict__ q, uint32_t* __restrict__ cell_start)
{
    const uint32_t cnt = min(q->count, kGapCap);
    for (uint32_t e = blockIdx.x; e < cnt; e += gridDim.x) {
        const uint4 r = q->e[e];
        for (uint32_t c = r.x + threadIdx.x; c < r.y; c += 256u) cell_start[c] = r.z;
    }
}
constexpr uint32_t kFillLongBlocks = 1024;

// ---- the same steps for the members of an NnGridSet: blockIdx.y = member --------------------------------------
struct NnBuildDev {
    NnGridDev           lv;       // geometry and device arrays of the level being built
    const float4*       pts;
    uint32_t*           counts;   // == lv.cell_start, writable
    unsigned long long* crowd;    // kCrowdSlots counters, or null
    NnGapQueue*         gapq;     // long gaps of the cell table (nn_fill_body), or null
    unsigned long long* occ[3];   // == lv.occ, occ1, occ2, writable
    float4*             sorted;   // == lv.sorted, writable
    uint32_t            n;        // points of the cloud
    uint32_t            off;      // first element of the member in the packed key / value arrays
    uint32_t            n_cells;
    uint32_t            active;   // 0: the launch skips this member
    int32_t             pd[3][3]; // node grids of the pyramid: bricks, super-bricks, blocks
};

__global__ __launch_bounds__(256) void nn_cellkey_many_kernel(const NnBuildDev* __restrict__ d, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals, int counting)
{
    const NnBuildDev& b = d[blockIdx.y];
    if (!b.active || blockIdx.x * 256u >= b.n) return;  // uniform per workgroup
    nn_cellkey_body(b.pts, b.n, b.lv, b.n_cells, keys + b.off, vals + b.off, counting ? b.counts : nullptr, counting ? b.crowd : nullptr);
}
__global__ __launch_bounds__(256) void nn_fill_many_kernel(const NnBuildDev* __restrict__ d, const uint32_t* __restrict__ sorted_keys, const uint32_t* __restrict__ sorted_vals, int with_occ)
{
    const NnBuildDev& b = d[blockIdx.y];
    if (!b.active || blockIdx.x * 256u > b.n) return;  // (thread n closes the table)
    nn_fill_body(sorted_keys + b.off, sorted_vals + b.off, b.pts, b.n, b.lv, b.n_cells, b.counts, with_occ ? b.occ[0] : nullptr, b.sorted, b.crowd, b.gapq);
}
__global__ __launch_bounds__(256) void nn_fill_long_many_kernel(const NnBuildDev* __restrict__ d)
{
    const NnBuildDev& b = d[blockIdx.y];
    if (!b.active || b.gapq == nullptr) return;
    nn_fill_long_body(b.gapq, b.counts);
}
__global__ __launch_bounds__(256) void nn_occupancy_up_many_kernel(const NnBuildDev* __restrict__ d, int from)
{
    const NnBuildDev& b = d[blockIdx.y];
    if (!b.active) return;
    nn_occupancy_up_body(b.occ[from], b.pd[from][0], b.pd[from][1], b.pd[from][2], b.pd[from + 1][0], b.pd[from + 1][1], b.occ[from + 1]);
}

// one COUNTING pass of the adaptive cell search: bins the points at edge `cell` with one atomic per distinct cell of a wavefront and reads back
// the crowding figure (population of the cell an average point sits in).  Full builds are build_levels_together's.
int NnGrid::count_level(mrgfe_ctx* ctx, const float4* d_pts, uint32_t nn, const BBox& bb, float cell, DevBuf& d_cells, double* crowding)
{
    hipStream_t st = ctx->stream;
    DevBuf &dk = ctx->scratch[2], &dv = ctx->scratch[3];
    NnGridDev lv;
    std::memset(&lv, 0, sizeof(lv));
    float extent = 0.0f;
    for (int a = 0; a < 3; ++a) { lv.origin[a] = bb.mn[a]; extent = std::max(extent, bb.mx[a] - bb.mn[a]); }
    lv.cell = cell;
    lv.slack = 1e-6f * (extent + cell);
    lv.n = bb.n_finite;
    for (int a = 0; a < 3; ++a) lv.dim[a] = static_cast<int>(std::floor((bb.mx[a] - bb.mn[a]) / cell)) + 1;
    const uint32_t n_cells = static_cast<uint32_t>(lv.dim[0]) * lv.dim[1] * lv.dim[2];
    // [counts: n_cells + 1][crowd counters], zeroed together
    const size_t crowd_word = (size_t(n_cells) + 2) & ~size_t(1);
    const size_t all_words = crowd_word + 2 * kCrowdSlots;
    MRGFE_TRY(d_cells.ensure(sizeof(uint32_t) * all_words));
    MRGFE_HIP_CHECK(hipMemsetAsync(d_cells.p, 0, sizeof(uint32_t) * all_words, st));
    unsigned long long* d_crowd = reinterpret_cast<unsigned long long*>(d_cells.as<uint32_t>() + crowd_word);
    hipLaunchKernelGGL(nn_cellkey_kernel, dim3((nn + 255) / 256), dim3(256), 0, st, d_pts, nn, lv, n_cells, dk.as<uint32_t>(), dv.as<uint32_t>(), d_cells.as<uint32_t>(), d_crowd);
    MRGFE_HIP_CHECK(hipGetLastError());
    unsigned long long slots[kCrowdSlots];
    MRGFE_HIP_CHECK(hipMemcpyAsync(slots, d_crowd, sizeof(slots), hipMemcpyDeviceToHost, st));
    MRGFE_HIP_CHECK(hipStreamSynchronize(st));
    unsigned long long crowd = 0;
    for (unsigned long long v : slots) crowd += v;
    *crowding = 1.0 + 2.0 * double(crowd) / double(bb.n_finite);  // queries are distributed like the points, not like the cells
    return MRGFE_OK;
}

// All levels of one grid in ONE pass of launches (round 4).  The levels of a k-NN grid (cell edges c, 4 c, 16 c over the same points) used to be
// built one after the other — bin, sort, fill: a dozen launches each — and a 130k-point GICP frame spent 0.38 ms in them for 0.2 ms of kernels.
// Here every level is a member of the batched kernels (blockIdx.y = level: its own geometry, cell table and sorted copy, the same points), so
// the three levels cost the launches of one; only level 0 carries the pyramid and the crowding counters.
int NnGrid::build_levels_together(mrgfe_ctx* ctx, const float4* d_pts, uint32_t nn, const BBox& bb, const float* cell, int L, double* crowding, unsigned long long* h_slots_async)
{
    hipStream_t st = ctx->stream;
    std::vector<uint32_t> sizes(L, nn);
    SliceTable tab;
    tab.build(sizes.data(), L);
    DevBuf &ds = ctx->scratch[0], &dk = ctx->scratch[2], &dv = ctx->scratch[3], &dkt = ctx->scratch[4], &dvt = ctx->scratch[5], &dh = ctx->scratch[6];
    const size_t ne = std::max<size_t>(tab.total_elems, 4);
    MRGFE_TRY(dk.ensure(ne * 4)); MRGFE_TRY(dv.ensure(ne * 4)); MRGFE_TRY(dkt.ensure(ne * 4)); MRGFE_TRY(dvt.ensure(ne * 4));
    MRGFE_TRY(dh.ensure(sizeof(uint32_t) * 256 * (tab.total_blks + L)));
    const size_t at_dev = (sizeof(Slice) * L + 15) & ~size_t(15);
    MRGFE_TRY(ds.ensure(at_dev + sizeof(NnBuildDev) * L));
    std::vector<NnBuildDev> dev(L);
    unsigned long long* d_crowd = nullptr;
    uint32_t max_cells = 1, max_bricks = 1, max_super = 1;
    bool any_long = false;
    for (int l = 0; l < L; ++l) {
        NnBuildDev& b = dev[l];
        std::memset(&b, 0, sizeof(b));
        NnGridDev& lv = b.lv;
        float extent = 0.0f;
        for (int a = 0; a < 3; ++a) { lv.origin[a] = bb.mn[a]; extent = std::max(extent, bb.mx[a] - bb.mn[a]); }
        lv.cell = cell[l];
        lv.slack = 1e-6f * (extent + cell[l]);
        lv.n = bb.n_finite;
        for (int a = 0; a < 3; ++a) lv.dim[a] = static_cast<int>(std::floor((bb.mx[a] - bb.mn[a]) / cell[l])) + 1;
        const uint32_t n_cells = static_cast<uint32_t>(lv.dim[0]) * lv.dim[1] * lv.dim[2];
        size_t pn[3] = {1, 1, 1};
        for (int a = 0; a < 3; ++a) {
            lv.bdim[a] = b.pd[0][a] = (lv.dim[a] + 3) / 4;
            b.pd[1][a] = (b.pd[0][a] + 3) / 4;
            b.pd[2][a] = (b.pd[1][a] + 3) / 4;
            for (int k = 0; k < 3; ++k) pn[k] *= static_cast<size_t>(b.pd[k][a]);
        }
        // [cell table: n_cells + 1][crowd counters][occupancy words of the three pyramid levels][queue of long gaps]; the table is written entry by
        // entry (nn_fill_body), the rest is cleared
        const size_t crowd_word = (size_t(n_cells) + 2) & ~size_t(1);
        const size_t head_words = size_t(n_cells) + 4 + 2 * kCrowdSlots, occ_at = (head_words + 1) & ~size_t(1);
        const bool   long_gaps = n_cells >= 64u * kLongGap;
        const size_t gapq_at = (occ_at + 2 * (pn[0] + pn[1] + pn[2]) + 3) & ~size_t(3);
        const size_t all_words = gapq_at + (long_gaps ? sizeof(NnGapQueue) / 4 : 0);
        DevBuf& d_cells = d_cell_start_[l];
        MRGFE_TRY(d_cells.ensure(sizeof(uint32_t) * all_words));
        MRGFE_HIP_CHECK(hipMemsetAsync(d_cells.as<uint32_t>() + crowd_word, 0, sizeof(uint32_t) * (all_words - crowd_word), st));
        MRGFE_TRY(d_sorted_[l].ensure(sizeof(float4) * std::max<size_t>(nn, 1)));
        lv.cell_start = d_cells.as<uint32_t>();
        lv.sorted = d_sorted_[l].as<float4>();
        lv.occ = reinterpret_cast<const unsigned long long*>(d_cells.as<uint32_t>() + occ_at);
        lv.occ1 = lv.occ + pn[0];
        lv.occ2 = lv.occ1 + pn[1];
        b.pts = d_pts;
        b.counts = d_cells.as<uint32_t>();
        b.sorted = d_sorted_[l].as<float4>();
        b.n = nn;
        b.off = tab.h[l].off;
        b.n_cells = n_cells;
        b.active = 1;
        b.gapq = long_gaps ? reinterpret_cast<NnGapQueue*>(d_cells.as<uint32_t>() + gapq_at) : nullptr;
        any_long = any_long || long_gaps;
        if (l == 0) {  // only the finest level is searched through the pyramid and measures its crowding
            b.occ[0] = const_cast<unsigned long long*>(lv.occ);
            b.occ[1] = const_cast<unsigned long long*>(lv.occ1);
            b.occ[2] = const_cast<unsigned long long*>(lv.occ2);
            d_crowd = reinterpret_cast<unsigned long long*>(d_cells.as<uint32_t>() + crowd_word);
            b.crowd = crowding ? d_crowd : nullptr;
            max_bricks = static_cast<uint32_t>(pn[0]);
            max_super = static_cast<uint32_t>(pn[1]);
        }
        max_cells = std::max(max_cells, n_cells);
        h_.level[l] = lv;
    }
    h_.n_levels = L;
    const Slice*      d_slices = ds.as<Slice>();
    const NnBuildDev* d_dev = reinterpret_cast<const NnBuildDev*>(ds.as<char>() + at_dev);
    MRGFE_TRY(ctx->stage_h2d(ds.p, tab.h.data(), sizeof(Slice) * L, st));
    MRGFE_TRY(ctx->stage_h2d(ds.as<char>() + at_dev, dev.data(), sizeof(NnBuildDev) * L, st));
    const dim3 grid(tab.max_blks * (kTile / 256), L);
    hipLaunchKernelGGL(nn_cellkey_many_kernel, grid, dim3(256), 0, st, d_dev, dk.as<uint32_t>(), dv.as<uint32_t>(), 0);
    int key_bits = 1;
    while (key_bits < 32 && (uint64_t(1) << key_bits) <= max_cells) ++key_bits;
    uint32_t *sk = nullptr, *sv = nullptr;
    MRGFE_TRY(radix_sort_pairs(ctx, dk.as<uint32_t>(), dv.as<uint32_t>(), dkt.as<uint32_t>(), dvt.as<uint32_t>(), d_slices, tab, key_bits, dh.as<uint32_t>(), &sk, &sv, true));
    hipLaunchKernelGGL(nn_fill_many_kernel, dim3(tab.max_blks * (kTile / 256) + 1, L), dim3(256), 0, st, d_dev, sk, sv, 1);
    if (any_long) hipLaunchKernelGGL(nn_fill_long_many_kernel, dim3(kFillLongBlocks, L), dim3(256), 0, st, d_dev);
    hipLaunchKernelGGL(nn_occupancy_up_many_kernel, dim3((max_bricks + 255) / 256, 1), dim3(256), 0, st, d_dev, 0);  // (member 0 = the finest level)
    hipLaunchKernelGGL(nn_occupancy_up_many_kernel, dim3((max_super + 255) / 256, 1), dim3(256), 0, st, d_dev, 1);
    MRGFE_HIP_CHECK(hipGetLastError());
    if (crowding && h_slots_async) {  // (pinned destination: the copy is a command on the stream, nobody waits for it here)
        MRGFE_HIP_CHECK(hipMemcpyAsync(h_slots_async, d_crowd, sizeof(unsigned long long) * kCrowdSlots, hipMemcpyDeviceToHost, st));
    } else if (crowding) {
        unsigned long long slots[kCrowdSlots];
        MRGFE_HIP_CHECK(hipMemcpyAsync(slots, d_crowd, sizeof(slots), hipMemcpyDeviceToHost, st));
        MRGFE_HIP_CHECK(hipStreamSynchronize(st));
        unsigned long long crowd = 0;
        for (unsigned long long v : slots) crowd += v;
        *crowding = 1.0 + 2.0 * double(crowd) / double(bb.n_finite);
    }
    return MRGFE_OK;
}

int NnGrid::build(mrgfe_ctx* ctx, const float4* d_pts, size_t n, float cell_size, double crowding_target, int max_levels, const float* known_box)
{
    built_ = false;
    n_ = n;
    if (const char* e = std::getenv("MRGFE_NN_CELL")) { cell_size = static_cast<float>(std::atof(e)); crowding_target = 0; }  // tuning hook
    std::memset(&h_, 0, sizeof(h_));
    h_.n_levels = 1;
    for (auto& lv : h_.level) { lv.cell = cell_size; lv.dim[0] = lv.dim[1] = lv.dim[2] = 1; lv.bdim[0] = lv.bdim[1] = lv.bdim[2] = 1; }
    if (n > 0x7fffffffu) { set_error("NnGrid: cloud too large"); return MRGFE_ERR_INVALID; }
    hipStream_t st = ctx->stream;
    uint32_t    nn = static_cast<uint32_t>(n);
    SliceTable  tab;
    tab.build(&nn, 1);
    BBox bb;
    if (known_box && n > 0) {
        for (int a = 0; a < 3; ++a) { bb.mn[a] = known_box[a]; bb.mx[a] = known_box[3 + a]; }
        bb.n_finite = nn;
        bb.pad = 0;
    } else {
        // descriptor: slice + cloud pointer
        DevBuf& ds = ctx->scratch[0];
        MRGFE_TRY(ds.ensure(sizeof(Slice) * 2 + sizeof(void*)));
        const void* cp = d_pts;
        MRGFE_HIP_CHECK(hipMemcpyAsync(ds.p, tab.h.data(), sizeof(Slice), hipMemcpyHostToDevice, st));
        MRGFE_HIP_CHECK(hipMemcpyAsync(ds.as<char>() + 2 * sizeof(Slice), &cp, sizeof(void*), hipMemcpyHostToDevice, st));
        DevBuf& dbb = ctx->scratch[1];
        MRGFE_TRY(dbb.ensure(sizeof(BBox) * (tab.total_blks + 1)));
        BBox* d_part = dbb.as<BBox>();
        BBox* d_out = d_part + tab.total_blks;
        MRGFE_TRY(bounding_boxes(ctx, reinterpret_cast<const float4* const*>(ds.as<char>() + 2 * sizeof(Slice)), ds.as<Slice>(), tab, d_part, d_out));
        MRGFE_HIP_CHECK(hipMemcpyAsync(&bb, d_out, sizeof(BBox), hipMemcpyDeviceToHost, st));
        MRGFE_HIP_CHECK(hipStreamSynchronize(st));
    }
    auto cells_of = [&](float c) {
        double prod = 1;
        for (int a = 0; a < 3; ++a) prod *= std::floor((bb.mx[a] - bb.mn[a]) / c) + 1;
        return prod;
    };
    if (crowd_pending_) {
        // what the previous build on the hinted edge measured (the event behind its copy has normally passed long ago): a hint that has become
        // too crowded or too fine is dropped, the adaptive passes below choose a new one
        MRGFE_HIP_CHECK(hipEventSynchronize(crowd_event_));
        crowd_pending_ = false;
        unsigned long long crowd = 0;
        for (int k = 0; k < kCrowdSlots; ++k) crowd += crowd_box_.as<unsigned long long>()[k];
        const double crowding = 1.0 + 2.0 * double(crowd) / double(std::max(crowd_n_finite_, 1u));
        if (hint_cell_ > 0 && hint_target_ > 0) {
            const bool too_crowded = crowding > 1.5 * hint_target_ && cells_of(hint_cell_ * 0.5f) <= double(1u << 24) && hint_cell_ * 16.0f > hint_cell_size_ * 0.999f;
            const bool too_fine = crowding * 6.0 < hint_target_ && hint_cell_ < hint_cell_size_;
            if (too_crowded || too_fine) hint_cell_ = 0;
        }
    }
    if (bb.n_finite == 0) {  // empty grid: one cell, no points
        MRGFE_TRY(d_cell_start_[0].ensure(32));
        MRGFE_HIP_CHECK(hipMemsetAsync(d_cell_start_[0].p, 0, 32, st));
        h_.level[0].occ = reinterpret_cast<const unsigned long long*>(d_cell_start_[0].as<uint32_t>() + 2);
        h_.level[0].occ1 = h_.level[0].occ + 1;
        h_.level[0].occ2 = h_.level[0].occ + 2;
        MRGFE_TRY(d_sorted_[0].ensure(16));
        h_.level[0].cell_start = d_cell_start_[0].as<uint32_t>();
        h_.level[0].sorted = d_sorted_[0].as<float4>();
        built_ = true;
        return MRGFE_OK;
    }
    DevBuf &dk = ctx->scratch[2], &dv = ctx->scratch[3], &dkt = ctx->scratch[4], &dvt = ctx->scratch[5], &dh = ctx->scratch[6];
    MRGFE_TRY(dk.ensure(n * 4)); MRGFE_TRY(dv.ensure(n * 4)); MRGFE_TRY(dkt.ensure(n * 4)); MRGFE_TRY(dvt.ensure(n * 4));
    MRGFE_TRY(dh.ensure(sizeof(uint32_t) * 256 * (tab.total_blks + 1)));
    auto cells_at = [&](float c) {
        double prod = 1;
        for (int a = 0; a < 3; ++a) prod *= std::floor((bb.mx[a] - bb.mn[a]) / c) + 1;
        return prod;
    };
    float cell = cell_size;
    while (cells_at(cell) > double(1u << 24)) cell *= 2.0f;
    // The edge the adaptive passes below chose for the previous cloud of this grid object (a registration's source cloud frame after frame, the
    // k-th target of a batch call after call: clouds of one sensor) is tried first, as a FULL build that measures its own crowding: when the
    // figure is still in range the build is done — no counting pass, no extra host round trip (each is a launch, a read-back and a wait: a third
    // of a 130k-point build).  The search results do not depend on the edge, only the time does.
    // the edges of all levels once the finest is known: kLevelRatio x the edge each, same origin, as long as the level below has more than a
    // handful of cells per axis
    float ratio = kLevelRatio;
    if (const char* e = std::getenv("MRGFE_NN_COARSE_RATIO")) ratio = std::max(2.0f, static_cast<float>(std::atof(e)));  // tuning hook
    auto level_edges = [&](float c0, float* edges) {
        int L = 1;
        edges[0] = c0;
        while (L < std::min(max_levels, kNnMaxLevels)) {
            int dmax = 0;
            for (int a = 0; a < 3; ++a) dmax = std::max(dmax, static_cast<int>(std::floor((bb.mx[a] - bb.mn[a]) / edges[L - 1])) + 1);
            if (dmax <= 4) break;
            edges[L] = edges[L - 1] * ratio;
            ++L;
        }
        return L;
    };
    float edges[kNnMaxLevels];
    if (crowding_target > 0 && hint_cell_ > 0 && hint_target_ == crowding_target && hint_cell_size_ == cell_size && hint_cell_ <= cell && cells_at(hint_cell_) <= double(1u << 24)) {
        double crowding = 0;
        const int L = level_edges(hint_cell_, edges);
        static const bool defer = [] { const char* e = std::getenv("MRGFE_NN_DEFER_CROWDING"); return e == nullptr || std::atoi(e) != 0; }();
        if (defer) {
            MRGFE_TRY(crowd_box_.ensure(sizeof(unsigned long long) * kCrowdSlots));
            MRGFE_TRY(build_levels_together(ctx, d_pts, nn, bb, edges, L, &crowding, crowd_box_.as<unsigned long long>()));
            if (!crowd_event_) MRGFE_HIP_CHECK(hipEventCreateWithFlags(&crowd_event_, hipEventDisableTiming));
            MRGFE_HIP_CHECK(hipEventRecord(crowd_event_, st));
            crowd_pending_ = true;
            crowd_n_finite_ = bb.n_finite;
            built_ = true;
            return MRGFE_OK;
        }
        MRGFE_TRY(build_levels_together(ctx, d_pts, nn, bb, edges, L, &crowding));
        const bool too_crowded = crowding > 1.5 * crowding_target && cells_at(hint_cell_ * 0.5f) <= double(1u << 24) && hint_cell_ * 16.0f > cell_size * 0.999f;
        const bool too_fine = crowding * 6.0 < crowding_target && hint_cell_ < cell;
        if (!too_crowded) {
            if (too_fine) hint_cell_ = 0;  // this build stands; the next one adapts from the top again
            built_ = true;
            return MRGFE_OK;
        }
        hint_cell_ = 0;
    }
    if (crowding_target > 0) {
        // halve the edge while the cell an average point sits in is more crowded than the target (at most four times).
        // LiDAR returns lie on surfaces, so the crowding falls about 4x per halving: jump by the predicted number of
        // halvings, then correct by single steps.
        int halvings = 0;
        for (int pass = 0; pass < 3 && halvings < 4 && cells_at(cell * 0.5f) <= double(1u << 24); ++pass) {
            double crowding = 0;
            MRGFE_TRY(count_level(ctx, d_pts, nn, bb, cell, d_cell_start_[0], &crowding));
            if (crowding <= crowding_target) break;
            int step = std::max(1, static_cast<int>(std::ceil(std::log(crowding / crowding_target) / std::log(4.0))));
            step = std::min(step, 4 - halvings);
            while (step > 1 && cells_at(cell * std::ldexp(1.0f, -step)) > double(1u << 24)) --step;
            cell *= std::ldexp(1.0f, -step);
            halvings += step;
        }
    }
    {
        const int L = level_edges(cell, edges);
        MRGFE_TRY(build_levels_together(ctx, d_pts, nn, bb, edges, L, nullptr));
    }
    if (crowding_target > 0) { hint_cell_ = cell; hint_target_ = crowding_target; hint_cell_size_ = cell_size; }
    built_ = true;
    return MRGFE_OK;
}

NnGrid& ctx_tmp_grid(mrgfe_ctx* ctx)
{
    if (!ctx->tmp_grid) ctx->tmp_grid = new NnGrid();
    return *ctx->tmp_grid;
}
void ctx_tmp_grid_free(mrgfe_ctx* ctx)
{
    if (ctx->pf_grid) {
        NnDeviceDrivenGrid* g = static_cast<NnDeviceDrivenGrid*>(ctx->pf_grid);
        g->cells.release(); g->sorted.release(); g->desc.release();
        delete g;
        ctx->pf_grid = nullptr;
    }
    if (!ctx->tmp_grid) return;
    ctx->tmp_grid->release();
    delete ctx->tmp_grid;
    ctx->tmp_grid = nullptr;
}

void NnGrid::release()
{
    if (crowd_pending_) { (void)hipEventSynchronize(crowd_event_); crowd_pending_ = false; }  // the copy into crowd_box_ must have landed before the box goes
    if (crowd_event_) { (void)hipEventDestroy(crowd_event_); crowd_event_ = nullptr; }
    crowd_box_.release();
    for (auto& b : d_cell_start_) b.release();
    for (auto& b : d_sorted_) b.release();
    built_ = false;
}

// ---- NnGridSet ---------------------------------------------------------------------------------------------------
void NnGridSet::release()
{
    for (auto& b : d_cells_) b.release();
    for (auto& b : d_sorted_) b.release();
    hint_cell_.clear();
}

int NnGridSet::build(mrgfe_ctx* ctx, const float4* const* d_clouds, const uint32_t* n, int count, float cell_size, double crowding_target, int max_levels, NnGrid* const* out)
{
    if (count <= 0) return MRGFE_OK;
    if (const char* e = std::getenv("MRGFE_NN_CELL")) { cell_size = static_cast<float>(std::atof(e)); crowding_target = 0; }  // tuning hook (as NnGrid::build)
    hipStream_t st = ctx->stream;
    const size_t M = static_cast<size_t>(count);
    uint64_t total = 0;
    for (size_t m = 0; m < M; ++m) total += (uint64_t(n[m]) + 3u) & ~uint64_t(3);
    if (total > 0x7fffffffu) { set_error("NnGridSet: %llu points in one set", static_cast<unsigned long long>(total)); return MRGFE_ERR_INVALID; }
    SliceTable tab;
    tab.build(n, count);
    // descriptors: [point slices][cloud pointers][NnBuildDev]
    DevBuf& ds = ctx->scratch[0];
    const size_t at_ptr = sizeof(Slice) * M, at_dev = (at_ptr + sizeof(void*) * M + 15) & ~size_t(15);
    MRGFE_TRY(ds.ensure(at_dev + sizeof(NnBuildDev) * M));
    const Slice*      d_slices = ds.as<Slice>();
    const NnBuildDev* d_dev = reinterpret_cast<const NnBuildDev*>(ds.as<char>() + at_dev);
    MRGFE_HIP_CHECK(hipMemcpyAsync(ds.p, tab.h.data(), sizeof(Slice) * M, hipMemcpyHostToDevice, st));
    MRGFE_HIP_CHECK(hipMemcpyAsync(ds.as<char>() + at_ptr, d_clouds, sizeof(void*) * M, hipMemcpyHostToDevice, st));
    DevBuf& dbb = ctx->scratch[1];
    MRGFE_TRY(dbb.ensure(sizeof(BBox) * (tab.total_blks + M)));
    BBox* d_part = dbb.as<BBox>();
    BBox* d_out = d_part + tab.total_blks;
    MRGFE_TRY(bounding_boxes(ctx, reinterpret_cast<const float4* const*>(ds.as<char>() + at_ptr), d_slices, tab, d_part, d_out));
    std::vector<BBox> bb(M);
    MRGFE_HIP_CHECK(hipMemcpyAsync(bb.data(), d_out, sizeof(BBox) * M, hipMemcpyDeviceToHost, st));
    MRGFE_HIP_CHECK(hipStreamSynchronize(st));
    DevBuf &dk = ctx->scratch[2], &dv = ctx->scratch[3], &dkt = ctx->scratch[4], &dvt = ctx->scratch[5], &dh = ctx->scratch[6];
    const size_t ne = std::max<size_t>(tab.total_elems, 4);
    MRGFE_TRY(dk.ensure(ne * 4)); MRGFE_TRY(dv.ensure(ne * 4)); MRGFE_TRY(dkt.ensure(ne * 4)); MRGFE_TRY(dvt.ensure(ne * 4));
    MRGFE_TRY(dh.ensure(sizeof(uint32_t) * 256 * (tab.total_blks + M)));

    auto cells_at = [&](size_t m, float c) {
        double prod = 1;
        for (int a = 0; a < 3; ++a) prod *= std::floor((bb[m].mx[a] - bb[m].mn[a]) / c) + 1;
        return prod;
    };
    std::vector<NnGrid2Dev> h(M);
    for (auto& g : h) {
        std::memset(&g, 0, sizeof(g));
        g.n_levels = 1;
        for (auto& lv : g.level) { lv.cell = cell_size; lv.dim[0] = lv.dim[1] = lv.dim[2] = 1; lv.bdim[0] = lv.bdim[1] = lv.bdim[2] = 1; }
    }
    std::vector<NnBuildDev> dev(M);
    // one level of every active member at its own edge: counting passes stop after the binning, full passes leave the level complete
    auto build_level = [&](int level, const std::vector<float>& cell, const std::vector<char>& active, bool counts_only, std::vector<double>* crowding) -> int {
        std::vector<uint32_t> nc1(M);
        std::vector<size_t>   pn(M * 3, 1);
        for (size_t m = 0; m < M; ++m) {
            NnBuildDev& b = dev[m];
            std::memset(&b, 0, sizeof(b));
            b.active = active[m] && bb[m].n_finite > 0 ? 1u : 0u;
            nc1[m] = 8;  // members left out of this launch: eight zero words (an empty cloud's whole grid, see below)
            if (!b.active) continue;
            NnGridDev& lv = b.lv;
            float extent = 0.0f;
            for (int a = 0; a < 3; ++a) { lv.origin[a] = bb[m].mn[a]; extent = std::max(extent, bb[m].mx[a] - bb[m].mn[a]); }
            lv.cell = cell[m];
            lv.slack = 1e-6f * (extent + cell[m]);
            lv.n = bb[m].n_finite;
            for (int a = 0; a < 3; ++a) lv.dim[a] = static_cast<int>(std::floor((bb[m].mx[a] - bb[m].mn[a]) / cell[m])) + 1;
            b.n_cells = static_cast<uint32_t>(lv.dim[0]) * lv.dim[1] * lv.dim[2];
            for (int a = 0; a < 3; ++a) {
                lv.bdim[a] = b.pd[0][a] = (lv.dim[a] + 3) / 4;
                b.pd[1][a] = (b.pd[0][a] + 3) / 4;
                b.pd[2][a] = (b.pd[1][a] + 3) / 4;
                for (int k = 0; k < 3; ++k) pn[m * 3 + k] *= static_cast<size_t>(b.pd[k][a]);
            }
            b.pts = d_clouds[m];
            b.n = n[m];
            b.off = tab.h[m].off;
            nc1[m] = b.n_cells + 1;
        }
        // [cell counts / starts of every member, laid out by the scan's slice table][crowd counters][pyramid words], zeroed together
        SliceTable ctab;
        ctab.build(nc1.data(), count);
        uint64_t words = (uint64_t(ctab.total_elems) + 1) & ~uint64_t(1);
        const uint64_t crowd_at = words;
        words += 2ull * kCrowdSlots * M;
        std::vector<uint64_t> occ_at(M);
        for (size_t m = 0; m < M; ++m) { occ_at[m] = words; if (dev[m].active) words += 2ull * (pn[m * 3] + pn[m * 3 + 1] + pn[m * 3 + 2]); }
        // queues of long gaps (nn_fill_body) for the members whose tables are large enough to have any worth a second launch
        std::vector<uint64_t> gapq_at(M, 0);
        bool any_long = false;
        for (size_t m = 0; m < M && !counts_only; ++m) {
            if (!dev[m].active || dev[m].n_cells < 64u * kLongGap) continue;
            words = (words + 3) & ~uint64_t(3);
            gapq_at[m] = words;
            words += sizeof(NnGapQueue) / 4;
            any_long = true;
        }
        if (words > 0xffffffffull) { set_error("NnGridSet: cell tables of %llu words", static_cast<unsigned long long>(words)); return MRGFE_ERR_INVALID; }
        DevBuf& dc = d_cells_[level];
        MRGFE_TRY(dc.ensure(sizeof(uint32_t) * words));
        // counting passes zero the count tables; a full build writes every entry of the cell tables itself (nn_fill_many_kernel) and needs only the
        // crowd counters and pyramid words behind them cleared — and the eight zero words that stand for the grid of a member without a finite point
        if (counts_only) {
            MRGFE_HIP_CHECK(hipMemsetAsync(dc.p, 0, sizeof(uint32_t) * words, st));
        } else {
            MRGFE_HIP_CHECK(hipMemsetAsync(dc.as<uint32_t>() + crowd_at, 0, sizeof(uint32_t) * (words - crowd_at), st));
            for (size_t m = 0; m < M; ++m)
                if (!dev[m].active) MRGFE_HIP_CHECK(hipMemsetAsync(dc.as<uint32_t>() + ctab.h[m].off, 0, sizeof(uint32_t) * 8, st));
        }
        if (!counts_only) MRGFE_TRY(d_sorted_[level].ensure(sizeof(float4) * std::max<size_t>(tab.total_elems, 1)));
        uint32_t max_cells = 1, max_bricks = 1, max_super = 1;
        for (size_t m = 0; m < M; ++m) {
            NnBuildDev& b = dev[m];
            if (!b.active) continue;
            b.counts = dc.as<uint32_t>() + ctab.h[m].off;
            b.crowd = crowding ? reinterpret_cast<unsigned long long*>(dc.as<uint32_t>() + crowd_at) + kCrowdSlots * m : nullptr;
            b.gapq = gapq_at[m] ? reinterpret_cast<NnGapQueue*>(dc.as<uint32_t>() + gapq_at[m]) : nullptr;
            b.occ[0] = reinterpret_cast<unsigned long long*>(dc.as<uint32_t>() + occ_at[m]);
            b.occ[1] = b.occ[0] + pn[m * 3];
            b.occ[2] = b.occ[1] + pn[m * 3 + 1];
            b.sorted = counts_only ? nullptr : d_sorted_[level].as<float4>() + tab.h[m].off;
            b.lv.cell_start = b.counts;
            b.lv.sorted = b.sorted;
            b.lv.occ = b.occ[0];
            b.lv.occ1 = b.occ[1];
            b.lv.occ2 = b.occ[2];
            max_cells = std::max(max_cells, b.n_cells);
            max_bricks = std::max<uint32_t>(max_bricks, static_cast<uint32_t>(pn[m * 3]));
            max_super = std::max<uint32_t>(max_super, static_cast<uint32_t>(pn[m * 3 + 1]));
        }
        MRGFE_HIP_CHECK(hipMemcpyAsync(ds.as<char>() + at_dev, dev.data(), sizeof(NnBuildDev) * M, hipMemcpyHostToDevice, st));
        if (tab.max_blks) hipLaunchKernelGGL(nn_cellkey_many_kernel, dim3(tab.max_blks * (kTile / 256), count), dim3(256), 0, st, d_dev, dk.as<uint32_t>(), dv.as<uint32_t>(), counts_only ? 1 : 0);
        MRGFE_HIP_CHECK(hipGetLastError());
        std::vector<unsigned long long> slots;
        auto read_crowding = [&]() {
            for (size_t m = 0; m < M; ++m) {
                unsigned long long crowd = 0;
                for (uint32_t k = 0; k < kCrowdSlots; ++k) crowd += slots[m * kCrowdSlots + k];
                (*crowding)[m] = dev[m].active ? 1.0 + 2.0 * double(crowd) / double(bb[m].n_finite) : 0.0;
            }
        };
        if (crowding) slots.resize(M * kCrowdSlots);
        if (counts_only) {
            MRGFE_HIP_CHECK(hipMemcpyAsync(slots.data(), dc.as<uint32_t>() + crowd_at, 8 * slots.size(), hipMemcpyDeviceToHost, st));
            MRGFE_HIP_CHECK(hipStreamSynchronize(st));
            read_crowding();
            return MRGFE_OK;
        }
        int key_bits = 1;
        while (key_bits < 32 && (uint64_t(1) << key_bits) <= max_cells) ++key_bits;
        uint32_t *sk = nullptr, *sv = nullptr;
        if (tab.max_blks) MRGFE_TRY(radix_sort_pairs(ctx, dk.as<uint32_t>(), dv.as<uint32_t>(), dkt.as<uint32_t>(), dvt.as<uint32_t>(), d_slices, tab, key_bits, dh.as<uint32_t>(), &sk, &sv, true));
        // cell tables, brick words, sorted points and crowding figures of all members from their sorted (key, index) pairs in one launch
        (void)max_cells;
        if (tab.max_blks) hipLaunchKernelGGL(nn_fill_many_kernel, dim3(tab.max_blks * (kTile / 256) + 1, count), dim3(256), 0, st, d_dev, sk, sv, level == 0 ? 1 : 0);
        if (tab.max_blks && any_long) hipLaunchKernelGGL(nn_fill_long_many_kernel, dim3(kFillLongBlocks / 4, count), dim3(256), 0, st, d_dev);
        if (level == 0) {  // only the finest level is searched through the pyramid
            hipLaunchKernelGGL(nn_occupancy_up_many_kernel, dim3((max_bricks + 255) / 256, count), dim3(256), 0, st, d_dev, 0);
            hipLaunchKernelGGL(nn_occupancy_up_many_kernel, dim3((max_super + 255) / 256, count), dim3(256), 0, st, d_dev, 1);
        }
        MRGFE_HIP_CHECK(hipGetLastError());
        if (crowding) MRGFE_HIP_CHECK(hipMemcpyAsync(slots.data(), dc.as<uint32_t>() + crowd_at, 8 * slots.size(), hipMemcpyDeviceToHost, st));
        MRGFE_HIP_CHECK(hipStreamSynchronize(st));  // the descriptor tables above were sources of asynchronous copies
        if (crowding) read_crowding();
        for (size_t m = 0; m < M; ++m) {
            if (dev[m].active) h[m].level[level] = dev[m].lv;
            else if (level == 0 && bb[m].n_finite == 0) {  // empty member: one cell, no points, an empty pyramid (as NnGrid::build's empty grid)
                NnGridDev& lv = h[m].level[0];
                lv.cell_start = dc.as<uint32_t>() + ctab.h[m].off;
                lv.occ = reinterpret_cast<const unsigned long long*>(lv.cell_start + 2);
                lv.occ1 = lv.occ + 1;
                lv.occ2 = lv.occ + 2;
                lv.sorted = d_sorted_[0].as<float4>();
            }
        }
        return MRGFE_OK;
    };

    // the finest edge of every member: last time's choice when there is one (verified by the full build's own crowding figure), else the
    // adaptive counting passes of NnGrid::build, all undecided members per launch
    const bool hints = crowding_target > 0 && hint_cell_.size() == M && hint_cell_size_ == cell_size && hint_target_ == crowding_target;
    std::vector<float> cell(M), top(M);
    std::vector<char>  hinted(M, 0), all(M, 1);
    for (size_t m = 0; m < M; ++m) {
        float c = cell_size;
        if (bb[m].n_finite) while (cells_at(m, c) > double(1u << 24)) c *= 2.0f;
        top[m] = cell[m] = c;
        if (hints && bb[m].n_finite && hint_cell_[m] > 0 && hint_cell_[m] <= c && cells_at(m, hint_cell_[m]) <= double(1u << 24)) { cell[m] = hint_cell_[m]; hinted[m] = 1; }
    }
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (crowding_target > 0) {
            std::vector<int>  halvings(M, 0);
            std::vector<char> open(M);
            for (size_t m = 0; m < M; ++m) open[m] = !hinted[m] && bb[m].n_finite > 0 && cells_at(m, cell[m] * 0.5f) <= double(1u << 24);
            for (int pass = 0; pass < 3; ++pass) {
                bool any = false;
                for (char o : open) any = any || o;
                if (!any) break;
                std::vector<double> crowding(M, 0.0);
                MRGFE_TRY(build_level(0, cell, open, true, &crowding));
                for (size_t m = 0; m < M; ++m) {
                    if (!open[m]) continue;
                    if (crowding[m] <= crowding_target) { open[m] = 0; continue; }
                    int step = std::max(1, static_cast<int>(std::ceil(std::log(crowding[m] / crowding_target) / std::log(4.0))));
                    step = std::min(step, 4 - halvings[m]);
                    while (step > 1 && cells_at(m, cell[m] * std::ldexp(1.0f, -step)) > double(1u << 24)) --step;
                    cell[m] *= std::ldexp(1.0f, -step);
                    halvings[m] += step;
                    if (halvings[m] >= 4 || cells_at(m, cell[m] * 0.5f) > double(1u << 24)) open[m] = 0;
                }
            }
        }
        std::vector<double> crowding(M, 0.0);
        MRGFE_TRY(build_level(0, cell, all, false, crowding_target > 0 ? &crowding : nullptr));
        bool redo = false;
        for (size_t m = 0; m < M && crowding_target > 0; ++m) {
            if (!hinted[m]) continue;
            const bool too_crowded = crowding[m] > 1.5 * crowding_target && cells_at(m, cell[m] * 0.5f) <= double(1u << 24) && cell[m] * 16.0f > cell_size * 0.999f;
            if (too_crowded) { hinted[m] = 0; cell[m] = top[m]; redo = true; }
        }
        if (!redo) {
            if (crowding_target > 0) {
                hint_cell_.assign(M, 0.0f);
                for (size_t m = 0; m < M; ++m)
                    if (bb[m].n_finite && !(hinted[m] && crowding[m] * 6.0 < crowding_target && cell[m] < top[m])) hint_cell_[m] = cell[m];  // (a hint that has become too fine is dropped: the next build adapts from the top)
                hint_cell_size_ = cell_size;
                hint_target_ = crowding_target;
            }
            break;
        }
    }
    // coarser levels for the k-NN climb: kLevelRatio x the edge each, while the level below has more than a handful of cells per axis — the coarser
    // levels of ALL members in one pass of launches (an entry per (member, level): its own geometry, cell table and sorted copy; round 3 ran a
    // pass per level)
    float ratio = NnGrid::kLevelRatio;
    if (const char* e = std::getenv("MRGFE_NN_COARSE_RATIO")) ratio = std::max(2.0f, static_cast<float>(std::atof(e)));
    struct Entry { size_t m; int level; float cell; };
    std::vector<Entry> entries;
    for (size_t m = 0; m < M; ++m) {
        if (!bb[m].n_finite) continue;
        float c = h[m].level[0].cell;
        for (int level = 1; level < std::min(max_levels, kNnMaxLevels); ++level) {
            int dmax = 0;
            for (int a = 0; a < 3; ++a) dmax = std::max(dmax, static_cast<int>(std::floor((bb[m].mx[a] - bb[m].mn[a]) / c)) + 1);
            if (dmax <= 4) break;
            c *= ratio;
            entries.push_back({m, level, c});
        }
    }
    if (!entries.empty()) {
        const size_t E = entries.size();
        std::vector<uint32_t> sizes(E);
        for (size_t e = 0; e < E; ++e) sizes[e] = n[entries[e].m];
        SliceTable etab;
        etab.build(sizes.data(), static_cast<int>(E));
        if (etab.total_elems > 0x7fffffffu) { set_error("NnGridSet: %u points in the coarser levels of one set", etab.total_elems); return MRGFE_ERR_INVALID; }
        const size_t ee = std::max<size_t>(etab.total_elems, 4);
        MRGFE_TRY(dk.ensure(ee * 4)); MRGFE_TRY(dv.ensure(ee * 4)); MRGFE_TRY(dkt.ensure(ee * 4)); MRGFE_TRY(dvt.ensure(ee * 4));
        MRGFE_TRY(dh.ensure(sizeof(uint32_t) * 256 * (etab.total_blks + E)));
        std::vector<NnBuildDev> edev(E);
        std::vector<uint64_t>   cell_at(E), gq_at(E, 0);
        uint64_t words = 0;
        uint32_t max_cells = 1;
        bool     any_long = false;
        for (size_t e = 0; e < E; ++e) {
            const size_t m = entries[e].m;
            NnBuildDev&  b = edev[e];
            std::memset(&b, 0, sizeof(b));
            NnGridDev& lv = b.lv;
            float extent = 0.0f;
            for (int a = 0; a < 3; ++a) { lv.origin[a] = bb[m].mn[a]; extent = std::max(extent, bb[m].mx[a] - bb[m].mn[a]); }
            lv.cell = entries[e].cell;
            lv.slack = 1e-6f * (extent + lv.cell);
            lv.n = bb[m].n_finite;
            for (int a = 0; a < 3; ++a) { lv.dim[a] = static_cast<int>(std::floor((bb[m].mx[a] - bb[m].mn[a]) / lv.cell)) + 1; lv.bdim[a] = (lv.dim[a] + 3) / 4; }
            b.n_cells = static_cast<uint32_t>(lv.dim[0]) * lv.dim[1] * lv.dim[2];
            b.pts = d_clouds[m];
            b.n = n[m];
            b.off = etab.h[e].off;
            b.active = 1;
            cell_at[e] = words;
            words += (uint64_t(b.n_cells) + 1 + 3) & ~uint64_t(3);
            if (b.n_cells >= 64u * kLongGap) { gq_at[e] = 1; any_long = true; }
            max_cells = std::max(max_cells, b.n_cells);
        }
        const uint64_t gq_base = words;  // the queues of long gaps behind all tables: the only part that has to be cleared (no pyramid, no counters here)
        for (size_t e = 0; e < E; ++e)
            if (gq_at[e]) { gq_at[e] = words; words += sizeof(NnGapQueue) / 4; }
        if (words > 0xffffffffull) { set_error("NnGridSet: coarse cell tables of %llu words", static_cast<unsigned long long>(words)); return MRGFE_ERR_INVALID; }
        DevBuf &dc = d_cells_[1], &dso = d_sorted_[1];
        MRGFE_TRY(dc.ensure(sizeof(uint32_t) * std::max<uint64_t>(words, 4)));
        MRGFE_TRY(dso.ensure(sizeof(float4) * std::max<size_t>(etab.total_elems, 1)));
        if (words > gq_base) MRGFE_HIP_CHECK(hipMemsetAsync(dc.as<uint32_t>() + gq_base, 0, sizeof(uint32_t) * (words - gq_base), st));
        for (size_t e = 0; e < E; ++e) {
            NnBuildDev& b = edev[e];
            b.counts = dc.as<uint32_t>() + cell_at[e];
            b.sorted = dso.as<float4>() + etab.h[e].off;
            b.gapq = gq_at[e] ? reinterpret_cast<NnGapQueue*>(dc.as<uint32_t>() + gq_at[e]) : nullptr;
            b.lv.cell_start = b.counts;
            b.lv.sorted = b.sorted;
            b.lv.occ = b.lv.occ1 = b.lv.occ2 = nullptr;  // (only the finest level is searched through the pyramid)
        }
        // the entries' slices and descriptors: in the bounding boxes' scratch (free since their read-back), through the pinned staging ring
        const size_t at_edev = (sizeof(Slice) * E + 15) & ~size_t(15);
        MRGFE_TRY(dbb.ensure(at_edev + sizeof(NnBuildDev) * E));
        MRGFE_TRY(ctx->stage_h2d(dbb.p, etab.h.data(), sizeof(Slice) * E, st));
        MRGFE_TRY(ctx->stage_h2d(dbb.as<char>() + at_edev, edev.data(), sizeof(NnBuildDev) * E, st));
        const Slice*      d_eslices = dbb.as<Slice>();
        const NnBuildDev* d_edev = reinterpret_cast<const NnBuildDev*>(dbb.as<char>() + at_edev);
        if (etab.max_blks) {
            hipLaunchKernelGGL(nn_cellkey_many_kernel, dim3(etab.max_blks * (kTile / 256), static_cast<uint32_t>(E)), dim3(256), 0, st, d_edev, dk.as<uint32_t>(), dv.as<uint32_t>(), 0);
            int key_bits = 1;
            while (key_bits < 32 && (uint64_t(1) << key_bits) <= max_cells) ++key_bits;
            uint32_t *sk = nullptr, *sv = nullptr;
            MRGFE_TRY(radix_sort_pairs(ctx, dk.as<uint32_t>(), dv.as<uint32_t>(), dkt.as<uint32_t>(), dvt.as<uint32_t>(), d_eslices, etab, key_bits, dh.as<uint32_t>(), &sk, &sv, true));
            hipLaunchKernelGGL(nn_fill_many_kernel, dim3(etab.max_blks * (kTile / 256) + 1, static_cast<uint32_t>(E)), dim3(256), 0, st, d_edev, sk, sv, 0);
            if (any_long) hipLaunchKernelGGL(nn_fill_long_many_kernel, dim3(kFillLongBlocks / 4, static_cast<uint32_t>(E)), dim3(256), 0, st, d_edev);
            MRGFE_HIP_CHECK(hipGetLastError());
        }
        MRGFE_HIP_CHECK(hipStreamSynchronize(st));  // the set returns with its grids complete
        for (size_t e = 0; e < E; ++e) {
            NnGrid2Dev& g = h[entries[e].m];
            g.level[entries[e].level] = edev[e].lv;
            g.n_levels = std::max(g.n_levels, entries[e].level + 1);
        }
    }
    for (size_t m = 0; m < M; ++m) out[m]->adopt(h[m], n[m]);
    return MRGFE_OK;
}

// ---- a grid the device sizes itself (nn_grid.h: NnDeviceDrivenGrid) ---------------------------------------------------
// n_boxes != nullptr: `bbox` is a list of *n_boxes partial boxes (the producer of the cloud left one per tile), merged here by the 256 threads
__global__ __launch_bounds__(256) void nn_geometry_kernel(const float4* const* __restrict__ cloud_ptr, const Slice* __restrict__ slice, const BBox* __restrict__ bbox,
                                                           const uint32_t* __restrict__ n_boxes, float cell, uint32_t cells_cap, uint32_t* cell_table, float4* sorted,
                                                           NnBuildDev* __restrict__ out, uint32_t* __restrict__ anomaly, BBox* __restrict__ h_box)
{
#pragma clang fp contract(off)
    BBox bb;
    if (n_boxes) bb = block_merge_partials(bbox, *n_boxes);  // (uniform)
    else         bb = *bbox;
    if (threadIdx.x) return;
    if (h_box) *h_box = bb;  // (pinned host memory: read behind the caller's stream wait)
    NnBuildDev b;
    memset(&b, 0, sizeof(b));
    NnGridDev& lv = b.lv;
    // (the arithmetic of NnGrid::build_level, float for float)
    float extent = 0.0f;
    for (int a = 0; a < 3; ++a) { lv.origin[a] = bb.mn[a]; extent = fmaxf(extent, bb.mx[a] - bb.mn[a]); }
    lv.cell = cell;
    lv.slack = 1e-6f * (extent + cell);
    lv.n = bb.n_finite;
    unsigned long long cells = 1;
    for (int a = 0; a < 3; ++a) {
        lv.dim[a] = bb.n_finite ? static_cast<int>(floorf((bb.mx[a] - bb.mn[a]) / cell)) + 1 : 1;
        lv.bdim[a] = (lv.dim[a] + 3) / 4;
        cells *= static_cast<unsigned long long>(lv.dim[a] > 0 ? lv.dim[a] : 1);
    }
    const bool fits = cells <= cells_cap && lv.dim[0] > 0 && lv.dim[1] > 0 && lv.dim[2] > 0;
    if (!fits) { atomicOr(anomaly, kNnAnomalyCells); lv.dim[0] = lv.dim[1] = lv.dim[2] = 1; lv.bdim[0] = lv.bdim[1] = lv.bdim[2] = 1; cells = 1; }
    b.n_cells = static_cast<uint32_t>(cells);
    b.pts = *cloud_ptr;
    b.n = fits ? slice->n : 0u;
    b.off = 0;
    b.active = (fits && bb.n_finite > 0) ? 1u : 0u;
    b.counts = cell_table;
    b.sorted = sorted;
    lv.cell_start = cell_table;
    lv.sorted = sorted;
    if (!b.active) {  // an empty grid: one cell without points (the search kernels test lv.n first)
        lv.n = 0;
        cell_table[0] = 0;
        cell_table[1] = 0;
    }
    *out = b;
}

// Eight lanes per query: a scan's radius filter is 33k queries of a ring walk each — one lane per query left the chip waiting on 25 dependent row
// probes per outlier (65 us per scan).  The lanes of a group take the x-rows of a ring in turn and meet in a sum after every ring; the flag only
// asks whether the count within the radius reaches `need`, which no order of counting changes.
constexpr int kRadiusGroup = 8;
__global__ __launch_bounds__(256) void nn_radius_flags_dd_kernel(const NnBuildDev* __restrict__ d, const float4* __restrict__ q, const Slice* __restrict__ slice, double r2, int need, int rings,
                                                                  uint32_t* __restrict__ flags)
{
    constexpr uint32_t per_blk = 256u / kRadiusGroup;
    const uint32_t i = blockIdx.x * per_blk + threadIdx.x / kRadiusGroup;
    const int      sub = static_cast<int>(threadIdx.x % kRadiusGroup);
    if (blockIdx.x * per_blk >= slice->n) return;  // uniform
    __shared__ NnGridDev s_g;
    static_assert(sizeof(NnGridDev) % 4 == 0 && sizeof(NnGridDev) / 4 <= 256, "one word per thread");
    if (threadIdx.x < sizeof(NnGridDev) / 4) reinterpret_cast<uint32_t*>(&s_g)[threadIdx.x] = reinterpret_cast<const uint32_t*>(&d->lv)[threadIdx.x];
    __syncthreads();
    if (i >= slice->n) return;  // (whole groups)
    const NnGridDev& g = s_g;
    const float4 p = q[i];
    int          c[3];
    int          count = 0;  // the group's count so far (the same in its eight lanes)
    if (g.n > 0 && nn_cell_of(g, p.x, p.y, p.z, c)) {
        int rmax = 0;
#pragma unroll
        for (int a = 0; a < 3; ++a) rmax = max(rmax, max(c[a], g.dim[a] - 1 - c[a]));
        rmax = min(rmax, rings);
        for (int r = 0; r <= rmax && count < need; ++r) {  // (nn_walk_ranges' rings: rows of cells along x are runs of the sorted points)
            int       mine = 0, k = 0;
            auto range = [&](uint32_t b, uint32_t e) {
                for (uint32_t kk = b; kk < e && mine < need; ++kk) {  // (a lane that has counted `need` by itself has settled the flag)
                    const float4 t = load_point(g.sorted + kk);
                    if (static_cast<double>(sqdist3f(t.x, t.y, t.z, p.x, p.y, p.z)) <= r2) ++mine;
                }
            };
            const int z0 = max(c[2] - r, 0), z1 = min(c[2] + r, g.dim[2] - 1);
            const int y0 = max(c[1] - r, 0), y1 = min(c[1] + r, g.dim[1] - 1);
            for (int z = z0; z <= z1; ++z) {
                const bool zface = (z - c[2] == r) || (c[2] - z == r);
                for (int y = y0; y <= y1; ++y, ++k) {
                    if (k % kRadiusGroup != sub) continue;
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
#pragma unroll
            for (int m = 1; m < kRadiusGroup; m <<= 1) mine += __shfl_xor(mine, m);
            count += mine;
        }
    }
    if (sub == 0) flags[i] = count >= need ? 1u : 0u;
}

int nn_build_device_driven(mrgfe_ctx* ctx, const float4* const* d_cloud_ptr, const Slice* d_slice, uint32_t n_cap, const BBox* d_bbox, const uint32_t* d_n_boxes, float cell,
                           uint32_t cells_cap, NnDeviceDrivenGrid& g, uint32_t* d_anomaly, BBox* h_box_out)
{
    hipStream_t st = ctx->stream;
    SliceTable  tab;
    tab.build(&n_cap, 1);
    MRGFE_TRY(g.cells.ensure(sizeof(uint32_t) * (size_t(cells_cap) + 8)));
    MRGFE_TRY(g.sorted.ensure(sizeof(float4) * std::max<size_t>(n_cap, 1)));
    MRGFE_TRY(g.desc.ensure(sizeof(NnBuildDev)));
    DevBuf &dk = ctx->scratch[2], &dv = ctx->scratch[3], &dkt = ctx->scratch[4], &dvt = ctx->scratch[5], &dh = ctx->scratch[6];
    const size_t ne = std::max<size_t>(n_cap, 4);
    MRGFE_TRY(dk.ensure(ne * 4)); MRGFE_TRY(dv.ensure(ne * 4)); MRGFE_TRY(dkt.ensure(ne * 4)); MRGFE_TRY(dvt.ensure(ne * 4));
    MRGFE_TRY(dh.ensure(sizeof(uint32_t) * 256 * (tab.total_blks + 1)));
    NnBuildDev* d_dev = g.desc.as<NnBuildDev>();
    hipLaunchKernelGGL(nn_geometry_kernel, dim3(1), dim3(256), 0, st, d_cloud_ptr, d_slice, d_bbox, d_n_boxes, cell, cells_cap, g.cells.as<uint32_t>(), g.sorted.as<float4>(), d_dev,
                       d_anomaly, h_box_out);
    if (tab.max_blks == 0) { MRGFE_HIP_CHECK(hipGetLastError()); return MRGFE_OK; }
    hipLaunchKernelGGL(nn_cellkey_many_kernel, dim3(tab.max_blks * (kTile / 256), 1), dim3(256), 0, st, d_dev, dk.as<uint32_t>(), dv.as<uint32_t>(), 0);
    int key_bits = 1;
    while (key_bits < 32 && (uint64_t(1) << key_bits) <= cells_cap) ++key_bits;  // the key of a non-finite point is n_cells <= cells_cap
    uint32_t *sk = nullptr, *sv = nullptr;
    MRGFE_TRY(radix_sort_pairs(ctx, dk.as<uint32_t>(), dv.as<uint32_t>(), dkt.as<uint32_t>(), dvt.as<uint32_t>(), d_slice, tab, key_bits, dh.as<uint32_t>(), &sk, &sv, true));
    hipLaunchKernelGGL(nn_fill_many_kernel, dim3(tab.max_blks * (kTile / 256) + 1, 1), dim3(256), 0, st, d_dev, sk, sv, 0);
    MRGFE_HIP_CHECK(hipGetLastError());
    return MRGFE_OK;
}

int nn_radius_flags_device_driven(mrgfe_ctx* ctx, const NnDeviceDrivenGrid& g, const float4* d_q, const Slice* d_slice, uint32_t n_cap, double r2, int need, float cell, uint32_t* d_flags)
{
    if (n_cap == 0) return MRGFE_OK;
    const int rings = static_cast<int>(std::ceil(std::sqrt(r2) / cell)) + 1;  // (NnGrid::radius_count_flags)
    hipLaunchKernelGGL(nn_radius_flags_dd_kernel, dim3((n_cap + 256 / kRadiusGroup - 1) / (256 / kRadiusGroup)), dim3(256), 0, ctx->stream, g.desc.as<NnBuildDev>(), d_q, d_slice, r2, need, rings, d_flags);
    MRGFE_HIP_CHECK(hipGetLastError());
    return MRGFE_OK;
}

// ---- queries ---------------------------------------------------------------------------------------------------
constexpr int kNnGroup = 8;  // lanes that share one 1-NN query (nn_nearest_group)

__global__ __launch_bounds__(256) void nn_nearest_kernel(NnGrid2Dev g, const float4* __restrict__ q, uint32_t n, const float* __restrict__ T12, int32_t* __restrict__ idx,
                                                          float* __restrict__ sqd)
{
    __shared__ float s_T[12];
    const bool       use_T = T12 != nullptr;
    if (use_T && threadIdx.x < 12) s_T[threadIdx.x] = T12[threadIdx.x];
    __syncthreads();
    const uint32_t i = blockIdx.x * (256u / kNnGroup) + threadIdx.x / kNnGroup;
    const int      sub = threadIdx.x % kNnGroup;
    if (i >= n) return;
    const float4 p = q[i];
    float x = p.x, y = p.y, z = p.z;
    if (use_T) transform_point(s_T, p.x, p.y, p.z, x, y, z);
    int32_t bi;
    float   bd;
    nn_nearest_group<kNnGroup>(g, x, y, z, sub, static_cast<double>(INFINITY), bi, bd);
    if (sub == 0) {
        idx[i] = bi;
        sqd[i] = bi >= 0 ? bd : -1.0f;
    }
}

// getFitnessScore for a batch of (grid, source cloud, transform) jobs (blockIdx.y = job) in three passes over one float
// per query.  A wavefront runs as long as its slowest query, and on a loop-closure candidate half of the queries are not settled
// by the 3x3x3 block around them, so in a single pass nearly every wavefront pays for a far search.  Hence:
//   block : own cell + 3x3x3 block of the finest level; writes the squared distance, -1 (nothing within max_range), or
//           queues the query (per-job list, appended through LDS: one global atomic per workgroup flush);
//   far   : the queued queries only, densely packed: full search (brick walk, coarser levels);
//   sum   : per job, fixed-order f64 sum and count of the distances — the far pass fills slots, so the order in which
//           queries were queued does not enter and the result is bitwise reproducible.
// Lanes per query: a batch fills the chip whatever the group size, and the lanes of a group mostly repeat each other's
// bookkeeping (both passes are bound by VALU issue), so ONE lane per query in the block pass and TWO in the far pass
// (config[3], 256 pairs: block 5.3 / 4.0 / 3.2 / 2.7 ms for 8 / 4 / 2 / 1 lanes, far 23.9 / 21.3 / 19.5 / 19.6 ms).
constexpr float    kFitNone = -1.0f;
constexpr uint32_t kFitPendCap = 1024;

__device__ __forceinline__ void nn_load_job(NnFitnessJob& s_job, const NnFitnessJob* job)
{
    const uint32_t* src = reinterpret_cast<const uint32_t*>(job);
    uint32_t*       dst = reinterpret_cast<uint32_t*>(&s_job);
    for (uint32_t w = threadIdx.x; w < sizeof(NnFitnessJob) / 4; w += 256) dst[w] = src[w];
}

// the query of a job: its transform applied in the order of the code the job stands in for
__device__ __forceinline__ void nn_job_transform(const float* __restrict__ T, uint32_t gicp_order, float px, float py, float pz, float& x, float& y, float& z)
{
#pragma clang fp contract(off)
    if (gicp_order) {  // (uniform) trans_f * Vector4f(x, y, z, 1): accumulated column by column, as gicp_corr_query does
        float s;
        s = T[0] * px; s = s + T[1] * py; s = s + T[2] * pz; x = s + T[3];
        s = T[4] * px; s = s + T[5] * py; s = s + T[6] * pz; y = s + T[7];
        s = T[8] * px; s = s + T[9] * py; s = s + T[10] * pz; z = s + T[11];
    } else {
        transform_point(T, px, py, pz, x, y, z);
    }
}
// kIdx passes: is the candidate (d, i) better than (best_d, best_i)?  Ties go to the lower index.
__device__ __forceinline__ unsigned long long nn_key(float d, int32_t i) { return static_cast<unsigned long long>(__float_as_uint(d)) << 32 | static_cast<uint32_t>(i); }

#ifndef MRGFE_BLOCK_GROUP
#define MRGFE_BLOCK_GROUP 1
#endif
constexpr int kBlockGroup = MRGFE_BLOCK_GROUP;  // lanes per query in the block pass
// kIdx (all four passes): the correspondence search of nn_nearest_batch — the index of the nearest point is carried beside its distance
// (ties: the lowest index) in the job's idx_out, a query counts when its squared distance is < max_range (fast_gicp's test) instead of <=.
template <bool kIdx, int G>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5))) void nn_fit_block_kernel(const NnFitnessJob* __restrict__ jobs, const uint32_t* __restrict__ job_off, double max_range, float* __restrict__ sqd,
                                                            uint32_t* __restrict__ pend, uint32_t* __restrict__ pend_cnt)
{
    __shared__ NnFitnessJob s_job;
    __shared__ uint32_t     s_pend[kFitPendCap], s_np, s_base;
    nn_load_job(s_job, jobs + blockIdx.y);
    if (threadIdx.x == 0) s_np = 0;
    __syncthreads();
    const NnGrid2Dev& g = s_job.grid;
    const uint32_t    n = s_job.n, off = job_off[blockIdx.y];
    const int         sub = threadIdx.x % G;
    constexpr uint32_t per_blk = 256u / G;
    auto flush = [&]() {  // called by the whole workgroup
        __syncthreads();
        const uint32_t np = s_np;
        if (np) {
            if (threadIdx.x == 0) s_base = atomicAdd(&pend_cnt[blockIdx.y], np);
            __syncthreads();
            for (uint32_t k = threadIdx.x; k < np; k += 256) pend[off + s_base + k] = s_pend[k];
            __syncthreads();
            if (threadIdx.x == 0) s_np = 0;
        }
        __syncthreads();
    };
    // each workgroup takes a contiguous chunk of the queries and queues in query order, so neighbours in the queue are
    // neighbours in the scan (the far pass hands each group a run of consecutive entries)
    __shared__ uint32_t s_w[4];
    const uint32_t chunk = ((n + gridDim.x - 1) / gridDim.x + per_blk - 1) / per_blk * per_blk;
    const uint32_t i_end = min(n, (blockIdx.x + 1) * chunk);
    for (uint32_t i0 = blockIdx.x * chunk; i0 < i_end; i0 += per_blk) {  // uniform trip count
        if (s_np > kFitPendCap - per_blk) flush();  // s_np is stable here: the appends of the last trip are behind a barrier
        const uint32_t i = i0 + threadIdx.x / G;
        bool           queue = false;
        if (i < i_end) {
            const float4 p = load_point(s_job.src + i);
            float x, y, z;
            nn_job_transform(s_job.T12, s_job.gicp_order, p.x, p.y, p.z, x, y, z);
            int32_t bi = -1;
            float   bd = INFINITY;
            bool    done = true;
            if (g.level[0].n != 0 && finite3(x, y, z)) done = nn_level_search<G>(g.level[0], x, y, z, sub, 1, max_range, bi, bd);
            // queued queries leave what the block gave (INFINITY: nothing) as the far pass's starting bound
            if (sub == 0) {
                if (kIdx) {
                    const bool hit = bi >= 0 && static_cast<double>(bd) < max_range;
                    sqd[off + i] = done ? (hit ? bd : kFitNone) : (bi >= 0 ? bd : INFINITY);
                    s_job.idx_out[i] = done ? (hit ? bi : -1) : bi;
                } else {
                    sqd[off + i] = done ? ((bi >= 0 && static_cast<double>(bd) <= max_range) ? bd : kFitNone) : (bi >= 0 ? bd : INFINITY);
                }
            }
            queue = sub == 0 && !done;
        }
        const uint64_t m = __ballot(queue);
        if (lane_id() == 0) s_w[wave_id()] = static_cast<uint32_t>(__popcll(m));
        __syncthreads();
        if (queue) {
            uint32_t at = s_np + static_cast<uint32_t>(__popcll(m & ((1ull << lane_id()) - 1ull)));
            for (uint32_t w = 0; w < wave_id(); ++w) at += s_w[w];
            s_pend[at] = i;
        }
        __syncthreads();
        if (threadIdx.x == 0) s_np += s_w[0] + s_w[1] + s_w[2] + s_w[3];
        __syncthreads();
    }
    flush();
}

// ---- far pass: seed, then sweep -------------------------------------------------------------------------------------------
// Three quarters of the queued queries of a loop-closure candidate arrive without any bound (their 3 x 3 x 3 block is empty), and
// nothing can be pruned before SOME distance is known: round 2's walk spent its time bounding occupied cells that a known radius
// would have excluded wholesale.  And a walk per lane is a nest of data-dependent loops: a wavefront executes the union of its
// lanes' walks (round 2: ~15 000 VALU instructions per wavefront of 32 queries, the distance evaluations a few per cent of them).
// So a workgroup takes a tile of 256 queued queries, one per lane, and works in phases whose items are equal:
//   seed   (lane = query) the nearest non-empty brick among the 27 around the query's, read off the (at most eight) super-brick
//          words that cover them; failing that the nearest non-empty super-brick among 27 (block words), then the nearest non-empty
//          block among 27; descend to the nearest occupied cell of what was found (nearest = smallest box distance; any occupied
//          cell would do) and measure its points.  `lim` is now a distance that IS attained, usually close to the answer.
//          Everything that could beat it lies in the cube of half edge sqrt(lim) around the query: per axis a range of cells, hence
//          of bricks, super-bricks and blocks.
//   bricks (lane = query) top-down over that cube: a block's word ANDed with the mask of its super-bricks inside the cube (three
//          4-bit ranges spread to 64 bits: ~20 instructions for up to 64 children), a surviving super-brick's word with the mask
//          of its bricks; bricks whose box reaches into the sphere are appended to the brick list in LDS;
//   cells  (lane = brick list entry) the brick's word ANDed with the mask of its cells inside the cube and outside the query's
//          own 3 x 3 x 3 block; cells whose box reaches into the sphere are appended to the cell list;
//   points (lane = cell list entry) two table reads, the cell's points measured, minimum into the query's best distance and lim
//          (LDS atomics on the float bit patterns).
// The lists are bounded: a lane that finds one full keeps its place, the list is worked off, and the phase resumes.  `lim` only ever
// shrinks and is always attained, so the order in which items run changes how much is pruned, never the result: every point within
// sqrt(lim) of the query is in a cell that gets opened (or in the query's own block, which the block pass measured and which is
// where lim starts) — the exact nearest distance, as before.  Queries with nothing within three blocks (8 m and more) keep their bound
// and go to nn_fit_far_kernel's pyramid walk.
// The queue keeps the order of the scan: neighbours in the queue are neighbours in space, and a tile's table and point reads hit the
// caches.  (Measured and dropped: a second queue for the queries whose cube covers more than eight super-bricks — a tenth of them, and a
// tile runs as long as its widest cube — filled by atomic appends: the scrambled order cost more than the even tiles gained, sweep 9.2 ->
// 11.4 ms for config[3].  Also dropped: a batch as two to eight sub-batches on as many contexts and host threads: 28.8 -> 33 / 36 / 44 ms
// per config[3] step.)
__device__ __forceinline__ unsigned long long nn_range_mask(const int L[3], const int H[3], int ox, int oy, int oz)
{
    // children (x + 4 y + 16 z) of the node with first child (ox, oy, oz) whose coordinates lie in [L, H] per axis; 0 if none
    const int xl = max(L[0] - ox, 0), xh = min(H[0] - ox, 3), yl = max(L[1] - oy, 0), yh = min(H[1] - oy, 3), zl = max(L[2] - oz, 0), zh = min(H[2] - oz, 3);
    if (xl > xh || yl > yh || zl > zh) return 0ull;
    const uint32_t mx = (0xfu >> (3 - xh)) & (0xfu << xl) & 0xfu, my = (0xfu >> (3 - yh)) & (0xfu << yl) & 0xfu, mz = (0xfu >> (3 - zh)) & (0xfu << zl) & 0xfu;
    const uint32_t rows = mx * ((my & 1u) | (my & 2u) << 3 | (my & 4u) << 6 | (my & 8u) << 9);  // mx at every y of the range: one z layer
    const uint32_t lo = rows * ((mz & 1u) | (mz & 2u) << 15), hi = rows * (((mz >> 2) & 1u) | ((mz >> 2) & 2u) << 15);
    return static_cast<unsigned long long>(hi) << 32 | lo;
}

constexpr uint32_t kSweepBrickCap = 1024;  // brick list entries
constexpr uint32_t kSweepCellCap = 2048;   // cell list entries
constexpr int      kSweepStats = 16;       // diagnostic counters (MRGFE_FIT_STATS)
constexpr uint32_t kNoSeed = 0x80000000u;  // flag on a queue entry: no seed, the query takes the pyramid walk

// squared distance from a query to the box with centre offset |centre - query| = (ax, ay, az) and half edge + margin h
__device__ __forceinline__ float nn_lower2(float ax, float ay, float az, float h)
{
    const float lx = fmaxf(ax - h, 0.0f), ly = fmaxf(ay - h, 0.0f), lz = fmaxf(az - h, 0.0f);
    return lx * lx + ly * ly + lz * lz;
}
// ... to child (cx, cy, cz) of edge E on a pyramid level, t = query - grid origin
__device__ __forceinline__ float nn_child_lb(const float t[3], int cx, int cy, int cz, float E, float h)
{
    return nn_lower2(fabsf((static_cast<float>(cx) + 0.5f) * E - t[0]), fabsf((static_cast<float>(cy) + 0.5f) * E - t[1]), fabsf((static_cast<float>(cz) + 0.5f) * E - t[2]), h);
}

// seed: one lane per queued query; sqd[query] becomes min(what the block gave, the nearest point of the seed cell) — attained —, or the
// queue entry is flagged kNoSeed and appended to the second queue
template <bool kIdx>
__global__ __launch_bounds__(256) void nn_fit_seed_kernel(const NnFitnessJob* __restrict__ jobs, const uint32_t* __restrict__ job_off, double max_range, uint32_t* __restrict__ pend,
                                                           const uint32_t* __restrict__ pend_cnt, float* __restrict__ sqd, uint32_t* __restrict__ pend2, uint32_t* __restrict__ pend2_cnt,
                                                           unsigned long long* __restrict__ stats)
{
    const uint32_t np = pend_cnt[blockIdx.y];
    if (blockIdx.x * 256u >= np) return;
    const NnFitnessJob& J = jobs[blockIdx.y];  // uniform: scalar loads
    const NnGridDev&    g = J.grid.level[0];
    const uint32_t      off = job_off[blockIdx.y];
    const float         E0 = g.cell, E1 = 4.0f * E0, E2 = 16.0f * E0, E3 = 64.0f * E0, mg = 4.0f * g.slack;
    const float         h0 = 0.5f * E0 + mg, h1 = 0.5f * E1 + mg, h2 = 0.5f * E2 + mg, h3 = 0.5f * E3 + mg;  // half edge plus the margin
    const int           d0[3] = {g.dim[0], g.dim[1], g.dim[2]};
    const int           d1[3] = {g.bdim[0], g.bdim[1], g.bdim[2]};
    const int           d2[3] = {(d1[0] + 3) >> 2, (d1[1] + 3) >> 2, (d1[2] + 3) >> 2};
    const int           d3[3] = {(d2[0] + 3) >> 2, (d2[1] + 3) >> 2, (d2[2] + 3) >> 2};
    const float         max_sq_f = max_range >= 3.0e38 ? INFINITY : static_cast<float>(max_range) * (1.0f + 1e-6f);
    const int           lane = lane_id();
    const uint64_t      below = (1ull << lane) - 1ull;
    uint32_t            n_words = 0, n_points = 0, n_seed2 = 0, n_seed3 = 0, n_noseed = 0;  // diagnostics
    for (uint32_t k = blockIdx.x * 256u + threadIdx.x; k < ((np + 63u) & ~63u); k += gridDim.x * 256u) {  // whole wavefronts: the queue append ballots
        bool     noseed = false;
        uint32_t qi = 0;
        if (k < np) {
            qi = pend[off + k];
            const float4 p = load_point(J.src + qi);
            float x, y, z;
            nn_job_transform(J.T12, J.gicp_order, p.x, p.y, p.z, x, y, z);
            int c[3];
            nn_cell_of(g, x, y, z, c);  // queued queries are finite
            const float t[3] = {x - g.origin[0], y - g.origin[1], z - g.origin[2]};
            // the child of a node (first child at 4 n) with the smallest box distance among those set in `word`
            auto nearest_child = [&](unsigned long long word, int nx, int ny, int nz, float E, float h, int out[3]) {
                float best = INFINITY;
                for (unsigned long long m = word; m != 0ull; m &= m - 1ull) {
                    const int   bit = __ffsll(m) - 1;
                    const int   cx = 4 * nx + (bit & 3), cy = 4 * ny + ((bit >> 2) & 3), cz = 4 * nz + (bit >> 4);
                    const float lb = nn_child_lb(t, cx, cy, cz, E, h);
                    if (lb < best) { best = lb; out[0] = cx; out[1] = cy; out[2] = cz; }
                }
            };
            // candidates of one level: the non-empty children, within one node of the query's, of the parents that cover them; the nearest
            auto seed_level = [&](const unsigned long long* __restrict__ words, const int pdim[3], int shift, const int cdim[3], float E, float h, int out[3]) {
                int Ls[3], Hs[3];
#pragma unroll
                for (int a = 0; a < 3; ++a) { Ls[a] = max((c[a] >> shift) - 1, 0); Hs[a] = min((c[a] >> shift) + 1, cdim[a] - 1); }
                float best = INFINITY;
                for (int pz = Ls[2] >> 2; pz <= Hs[2] >> 2; ++pz)
                    for (int py = Ls[1] >> 2; py <= Hs[1] >> 2; ++py)
                        for (int px = Ls[0] >> 2; px <= Hs[0] >> 2; ++px) {
                            const unsigned long long wd = as_global(words)[(static_cast<uint32_t>(pz) * pdim[1] + py) * pdim[0] + px] & nn_range_mask(Ls, Hs, 4 * px, 4 * py, 4 * pz);
                            ++n_words;
                            for (unsigned long long m = wd; m != 0ull; m &= m - 1ull) {
                                const int   bit = __ffsll(m) - 1;
                                const int   cx = 4 * px + (bit & 3), cy = 4 * py + ((bit >> 2) & 3), cz = 4 * pz + (bit >> 4);
                                const float lb = nn_child_lb(t, cx, cy, cz, E, h);
                                if (lb < best) { best = lb; out[0] = cx; out[1] = cy; out[2] = cz; }
                            }
                        }
                return best < INFINITY;
            };
            int  br[3] = {0, 0, 0};  // the brick the seed cell is taken from
            bool seeded = seed_level(g.occ1, d2, 2, d1, E1, h1, br);
            if (!seeded) {
                int  sb[3] = {0, 0, 0};
                bool found = seed_level(g.occ2, d3, 4, d2, E2, h2, sb);
                ++n_seed2;
                if (!found) {  // blocks: no word above them, so the 27 around the query's one by one
                    ++n_seed3;
                    float best = INFINITY;
                    int   bk[3] = {0, 0, 0};
                    for (int kz = max((c[2] >> 6) - 1, 0); kz <= min((c[2] >> 6) + 1, d3[2] - 1); ++kz)
                        for (int ky = max((c[1] >> 6) - 1, 0); ky <= min((c[1] >> 6) + 1, d3[1] - 1); ++ky)
                            for (int kx = max((c[0] >> 6) - 1, 0); kx <= min((c[0] >> 6) + 1, d3[0] - 1); ++kx) {
                                if (as_global(g.occ2)[(static_cast<uint32_t>(kz) * d3[1] + ky) * d3[0] + kx] == 0ull) continue;
                                const float lb = nn_child_lb(t, kx, ky, kz, E3, h3);
                                if (lb < best) { best = lb; bk[0] = kx; bk[1] = ky; bk[2] = kz; }
                            }
                    if (best < INFINITY) {
                        nearest_child(as_global(g.occ2)[(static_cast<uint32_t>(bk[2]) * d3[1] + bk[1]) * d3[0] + bk[0]], bk[0], bk[1], bk[2], E2, h2, sb);
                        found = true;
                    }
                }
                if (found) {
                    nearest_child(as_global(g.occ1)[(static_cast<uint32_t>(sb[2]) * d2[1] + sb[1]) * d2[0] + sb[0]], sb[0], sb[1], sb[2], E1, h1, br);
                    seeded = true;
                }
            }
            if (seeded) {
                int sc[3] = {0, 0, 0};
                nearest_child(as_global(g.occ)[(static_cast<uint32_t>(br[2]) * d1[1] + br[1]) * d1[0] + br[0]], br[0], br[1], br[2], E0, h0, sc);
                const uint32_t at = (static_cast<uint32_t>(sc[2]) * d0[1] + sc[1]) * d0[0] + sc[0];
                const uint32_t kb = as_global(g.cell_start)[at], ke = as_global(g.cell_start)[at + 1];
                float          best = sqd[off + qi];  // what the block gave (INFINITY: nothing)
                if (kIdx) {
                    int32_t bidx = J.idx_out[qi];
                    for (uint32_t kk = kb; kk < ke; ++kk) {
                        const float4  q = load_point(g.sorted + kk);
                        const float   d = sqdist3f(q.x, q.y, q.z, x, y, z);
                        const int32_t i = __float_as_int(q.w);
                        if (bidx < 0 || d < best || (d == best && i < bidx)) { best = d; bidx = i; }
                    }
                    J.idx_out[qi] = bidx;
                } else if (ke > kb) {
                    const uint32_t last = ke - 1u;
                    for (uint32_t kk = kb; kk < ke; kk += 4) {  // four loads in flight; past the end the last point again
                        const float4 q0 = load_point(g.sorted + kk), q1 = load_point(g.sorted + min(kk + 1, last)), q2 = load_point(g.sorted + min(kk + 2, last)),
                                     q3 = load_point(g.sorted + min(kk + 3, last));
                        best = fminf(fminf(best, sqdist3f(q0.x, q0.y, q0.z, x, y, z)), sqdist3f(q1.x, q1.y, q1.z, x, y, z));
                        best = fminf(fminf(best, sqdist3f(q2.x, q2.y, q2.z, x, y, z)), sqdist3f(q3.x, q3.y, q3.z, x, y, z));
                    }
                }
                n_points += ke - kb;
                sqd[off + qi] = best;
                // the sweep names a brick by its offset from the query's in eight bits per axis: a radius beyond that (60 m at 0.125 m cells: a query
                // far outside the scene) is left to the pyramid walk, with the seed's distance as its bound
                if (sqrtf(fminf(best, max_sq_f)) > 118.0f * E1) seeded = false;
            }
            if (!seeded) {
                noseed = true;
                ++n_noseed;
                pend[off + k] = qi | kNoSeed;
            }
        }
        const uint64_t m = __ballot(noseed);
        if (m != 0) {
            const int leader = __ffsll(static_cast<unsigned long long>(m)) - 1;
            uint32_t  base = 0;
            if (lane == leader) base = atomicAdd(&pend2_cnt[blockIdx.y], static_cast<uint32_t>(__popcll(m)));
            base = wave_read(base, leader);
            if (noseed) pend2[off + base + static_cast<uint32_t>(__popcll(m & below))] = qi;
        }
    }
    if (stats != nullptr) {
        const uint32_t v[5] = {wave_sum(n_words), wave_sum(n_points), wave_sum(n_seed2), wave_sum(n_seed3), wave_sum(n_noseed)};
        if (lane == 0) {
            const int at[5] = {0, 3, 4, 5, 6};
            for (int kk = 0; kk < 5; ++kk)
                if (v[kk]) atomicAdd(&stats[at[kk]], static_cast<unsigned long long>(v[kk]));
        }
    }
}

#ifndef MRGFE_SWEEP_WAVES
#define MRGFE_SWEEP_WAVES 4  // wavefronts per SIMD the register allocation aims for (tuning builds: -DMRGFE_SWEEP_WAVES=...).  The kernel wants ~165
                             // VGPRs: at 4 (128 VGPRs, 84 bytes of scratch per lane) the sweep of a config[3] step takes 9.2 - 9.5 ms, at 3 (no scratch) 9.8 - 9.9 ms
#endif
// a diagnostic counter that costs no register when the diagnostics are compiled out
template <bool kOn>
struct DiagCount {
    uint32_t v = 0;
    __device__ __forceinline__ void operator++() { if (kOn) ++v; }
    __device__ __forceinline__ void operator+=(uint32_t x) { if (kOn) v += x; }
};

template <bool kIdx, int kT, bool kStats>
__global__ __launch_bounds__(kT) __attribute__((amdgpu_waves_per_eu(MRGFE_SWEEP_WAVES))) void nn_fit_sweep_kernel(const NnFitnessJob* __restrict__ jobs, const uint32_t* __restrict__ job_off, double max_range,
                                                            const uint32_t* __restrict__ pend, const uint32_t* __restrict__ pend_cnt, float* __restrict__ sqd,
                                                            unsigned long long* __restrict__ stats, int clocks)
{
    constexpr uint32_t kBrickCap = kSweepBrickCap * kT / 256, kCellCap = kSweepCellCap * kT / 256;  // list entries per tile of kT queries
    const uint32_t np = pend_cnt[blockIdx.y];
    if (blockIdx.x * static_cast<uint32_t>(kT) >= np) return;
    __shared__ float    s_q[3][static_cast<uint32_t>(kT)];                    // the transformed queries
    __shared__ int      s_c[3][static_cast<uint32_t>(kT)];                    // their cells
    __shared__ int      s_L[3][static_cast<uint32_t>(kT)], s_H[3][static_cast<uint32_t>(kT)];   // the cube of half edge sqrt(lim at seed time), in cells
    __shared__ uint32_t s_lim[static_cast<uint32_t>(kT)], s_best[static_cast<uint32_t>(kT)];    // float bit patterns (>= 0: ordered like unsigned integers)
    __shared__ unsigned long long s_key[kIdx ? static_cast<uint32_t>(kT) : 1];  // kIdx: (distance bits, index) of the best point so far, instead of s_best
    __shared__ uint32_t s_bl[kBrickCap];               // query | (brick - query's brick + 128) per axis << 8, 16, 24
    __shared__ uint32_t s_cl[kCellCap];                // query << 24 | cell
    __shared__ float    s_clb[kCellCap];               // the cell's box distance
    __shared__ uint32_t s_nb, s_nc;
    const NnFitnessJob& J = jobs[blockIdx.y];  // uniform: scalar loads
    const NnGridDev&    g = J.grid.level[0];
    const uint32_t      off = job_off[blockIdx.y];
    const float         E0 = g.cell, E1 = 4.0f * E0, E2 = 16.0f * E0, mg = 4.0f * g.slack;
    const float         h0 = 0.5f * E0 + mg, h1 = 0.5f * E1 + mg, h2 = 0.5f * E2 + mg;  // half edge plus the margin
    const float         org[3] = {g.origin[0], g.origin[1], g.origin[2]};
    const int           d0[3] = {g.dim[0], g.dim[1], g.dim[2]};
    const int           d1[3] = {g.bdim[0], g.bdim[1], g.bdim[2]};
    const int           d2[3] = {(d1[0] + 3) >> 2, (d1[1] + 3) >> 2, (d1[2] + 3) >> 2};
    const int           d3[3] = {(d2[0] + 3) >> 2, (d2[1] + 3) >> 2, (d2[2] + 3) >> 2};
    const float         max_sq_f = max_range >= 3.0e38 ? INFINITY : static_cast<float>(max_range) * (1.0f + 1e-6f);
    const int           lane = lane_id();
    const int           tid = static_cast<int>(threadIdx.x);
    // diagnostics (kStats builds only: six counters and the phase clocks would otherwise hold 16 registers of a kernel that spills)
    DiagCount<kStats>   n_words, n_tested, n_cells, n_points, n_bricks, n_listed;
    long long           clk[4] = {0, 0, 0, 0}, tick = 0;  // thread 0: shader clocks spent in the four phases
    auto stamp = [&](int ph) { if (kStats && clocks && threadIdx.x == 0) { const long long now = clock64(); clk[ph] += now - tick; tick = now; } };  // (a clock read waits for the memory operations in flight)
    // every lane of the wavefront asks for `cnt` consecutive slots of a bounded LDS list: returns the lane's first slot; `full` when the list has filled up
    auto reserve = [&](uint32_t cnt, uint32_t* counter, uint32_t cap, bool& full) -> uint32_t {
        const uint32_t incl = wave_inclusive_scan(cnt);
        const uint32_t total = wave_read(incl, kWave - 1);
        uint32_t       base = 0;
        if (total != 0) {  // uniform
            if (lane == kWave - 1) base = atomicAdd(counter, total);
            base = wave_read(base, kWave - 1);
            full = full || base + total >= cap;
        }
        return base + incl - cnt;
    };
    if (kStats && clocks && threadIdx.x == 0) tick = clock64();
    for (uint32_t k0 = blockIdx.x * static_cast<uint32_t>(kT); k0 < np; k0 += gridDim.x * static_cast<uint32_t>(kT)) {
        // ================= the tile's queries (lane = query): lim as the seed left it, the cube it spans =================
        const uint32_t k = k0 + threadIdx.x;
        uint32_t qi = 0;
        bool     mine = false;
        int      kx = 0, ky = 0, kz = 0;  // next block to load (bricks phase)
        if (k < np) {
            qi = as_global(pend)[off + k];
            mine = (qi & kNoSeed) == 0;
        }
        if (mine) {
            // (the query's cell, cube and offset from the grid origin live in LDS from here on: the bricks phase reloads them every time it
            // resumes, so that they do not hold a dozen registers through the cells and points phases of a kernel that spills)
            float t[3];
            int   c[3], L[3], H[3];
            const float4 p = load_point(J.src + qi);
            float x, y, z;
            nn_job_transform(J.T12, J.gicp_order, p.x, p.y, p.z, x, y, z);
            const float best = sqd[off + qi];  // attained (seed)
            nn_cell_of(g, x, y, z, c);
            t[0] = x - org[0];
            t[1] = y - org[1];
            t[2] = z - org[2];
            const float lim = fminf(best, max_sq_f);
            const float r = sqrtf(lim * kNnPrune) * (1.0f + 1e-6f) + mg;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                L[a] = static_cast<int>(fminf(fmaxf(floorf((t[a] - r) / E0), 0.0f), static_cast<float>(d0[a] - 1)));
                H[a] = static_cast<int>(fminf(fmaxf(floorf((t[a] + r) / E0), 0.0f), static_cast<float>(d0[a] - 1)));
                s_c[a][tid] = c[a];
                s_L[a][tid] = L[a];
                s_H[a][tid] = H[a];
            }
            s_lim[tid] = __float_as_uint(lim);
            s_best[tid] = __float_as_uint(best);
            if (kIdx) s_key[tid] = nn_key(best, J.idx_out[qi]);
            s_q[0][tid] = x;
            s_q[1][tid] = y;
            s_q[2][tid] = z;
            kx = L[0] >> 6;
            ky = L[1] >> 6;
            kz = L[2] >> 6;
        }
        if (threadIdx.x == 0) { s_nb = 0; s_nc = 0; }
        __syncthreads();
        stamp(0);
        // ================= bricks (lane = query): blocks -> super-bricks -> bricks of the cube =================
        bool blocks_left = mine;
        bool act = mine;
        unsigned long long m2 = 0ull, m1p = 0ull;  // super-bricks of the current block still to visit / bricks of the current super-brick still to list
        int  s2[3] = {0, 0, 0}, s1[3] = {0, 0, 0};  // first super-brick of the block m2 belongs to / first brick of the super-brick m1p belongs to
        for (;;) {
            bool full = false;
            // (an inactive lane's entries may be stale or unset: it does not use them)
            const int   L[3] = {s_L[0][tid], s_L[1][tid], s_L[2][tid]}, H[3] = {s_H[0][tid], s_H[1][tid], s_H[2][tid]}, c[3] = {s_c[0][tid], s_c[1][tid], s_c[2][tid]};
            const float t[3] = {s_q[0][tid] - org[0], s_q[1][tid] - org[1], s_q[2][tid] - org[2]};
            const int   L1[3] = {L[0] >> 2, L[1] >> 2, L[2] >> 2}, H1[3] = {H[0] >> 2, H[1] >> 2, H[2] >> 2};
            const int   L2[3] = {L[0] >> 4, L[1] >> 4, L[2] >> 4}, H2[3] = {H[0] >> 4, H[1] >> 4, H[2] >> 4};
            while (!full && __ballot(act)) {  // uniform: every lane of the wavefront stays in the loop
                if (act && m1p == 0ull) {  // next super-brick of the cube that reaches into the sphere: its bricks that do
                    const float lim = __uint_as_float(s_lim[tid]) * kNnPrune;
                    while (m2 == 0ull && blocks_left) {
                        s2[0] = 4 * kx;
                        s2[1] = 4 * ky;
                        s2[2] = 4 * kz;
                        m2 = as_global(g.occ2)[(static_cast<uint32_t>(kz) * d3[1] + ky) * d3[0] + kx] & nn_range_mask(L2, H2, s2[0], s2[1], s2[2]);
                        ++n_words;
                        if (++kx > (H[0] >> 6)) {
                            kx = L[0] >> 6;
                            if (++ky > (H[1] >> 6)) {
                                ky = L[1] >> 6;
                                if (++kz > (H[2] >> 6)) blocks_left = false;
                            }
                        }
                    }
                    if (m2 == 0ull) {
                        act = false;
                    } else {
                        const int bit = __ffsll(m2) - 1;
                        m2 &= m2 - 1ull;
                        const int sx = s2[0] + (bit & 3), sy = s2[1] + ((bit >> 2) & 3), sz = s2[2] + (bit >> 4);
                        ++n_tested;
                        if (nn_child_lb(t, sx, sy, sz, E2, h2) <= lim) {
                            s1[0] = 4 * sx;
                            s1[1] = 4 * sy;
                            s1[2] = 4 * sz;
                            ++n_words;
                            for (unsigned long long m1 = as_global(g.occ1)[(static_cast<uint32_t>(sz) * d2[1] + sy) * d2[0] + sx] & nn_range_mask(L1, H1, s1[0], s1[1], s1[2]); m1 != 0ull; m1 &= m1 - 1ull) {
                                const int b1 = __ffsll(m1) - 1;
                                ++n_tested;
                                if (nn_child_lb(t, s1[0] + (b1 & 3), s1[1] + ((b1 >> 2) & 3), s1[2] + (b1 >> 4), E1, h1) <= lim) m1p |= 1ull << b1;
                            }
                        }
                    }
                }
                uint32_t slot = reserve(static_cast<uint32_t>(__popcll(m1p)), &s_nb, kBrickCap, full);
                for (; m1p != 0ull && slot < kBrickCap; ++slot) {  // what does not fit stays in m1p for the next round
                    const int b1 = __ffsll(m1p) - 1;
                    m1p &= m1p - 1ull;
                    const int bx = s1[0] + (b1 & 3), by = s1[1] + ((b1 >> 2) & 3), bz = s1[2] + (b1 >> 4);
                    s_bl[slot] = static_cast<uint32_t>(tid) | static_cast<uint32_t>(bx - (c[0] >> 2) + 128) << 8 | static_cast<uint32_t>(by - (c[1] >> 2) + 128) << 16 |
                                 static_cast<uint32_t>(bz - (c[2] >> 2) + 128) << 24;  // offsets fit: the seed kernel kept wider cubes out
                }
            }
            __syncthreads();
            stamp(1);
            // ================= cells (lane = brick list entry) and points (lane = cell list entry) =================
            const uint32_t nb = min(s_nb, kBrickCap);
            if (threadIdx.x == 0) n_bricks += nb;
            // a lane's entries (its stride of the list: at most four) are independent: their occupancy words are requested together
            constexpr int kEnt = kBrickCap / kT;
            uint32_t           ent[kEnt];
            unsigned long long wrd[kEnt];
            int                n_ent = 0, u_ent = 0;
#pragma unroll
            for (int u = 0; u < kEnt; ++u) {
                const uint32_t e = threadIdx.x + static_cast<uint32_t>(kT) * u;
                ent[u] = 0;
                wrd[u] = 0ull;
                if (e >= nb) continue;
                n_ent = u + 1;
                const uint32_t it = s_bl[e];
                const int      qq = static_cast<int>(it & 0xffu);
                ent[u] = it;
                wrd[u] = as_global(g.occ)[(static_cast<uint32_t>((s_c[2][qq] >> 2) + static_cast<int>(it >> 24) - 128) * d1[1] + ((s_c[1][qq] >> 2) + static_cast<int>((it >> 16) & 0xffu) - 128)) * d1[0] +
                                          ((s_c[0][qq] >> 2) + static_cast<int>((it >> 8) & 0xffu) - 128)];
                ++n_words;
            }
            unsigned long long m0p = 0ull;  // cells of the current entry still to list
            int                q = 0, o0[3] = {0, 0, 0};  // the entry's query and the brick's first cell
            float              tq[3] = {0, 0, 0};
            for (;;) {
                bool cfull = false;
                while (!cfull && __ballot(m0p != 0ull || u_ent < n_ent)) {  // uniform
                    if (m0p == 0ull && u_ent < n_ent) {  // next entry: the brick's cells inside the cube, outside the query's own block, reaching into the sphere
                        uint32_t           it = ent[0];
                        unsigned long long w0 = wrd[0];
#pragma unroll
                        for (int u = 1; u < kEnt; ++u)
                            if (u == u_ent) { it = ent[u]; w0 = wrd[u]; }
                        ++u_ent;
                        q = static_cast<int>(it & 0xffu);
                        const int cq[3] = {s_c[0][q], s_c[1][q], s_c[2][q]};
                        o0[0] = 4 * ((cq[0] >> 2) + static_cast<int>((it >> 8) & 0xffu) - 128);
                        o0[1] = 4 * ((cq[1] >> 2) + static_cast<int>((it >> 16) & 0xffu) - 128);
                        o0[2] = 4 * ((cq[2] >> 2) + static_cast<int>(it >> 24) - 128);
                        const int Lq[3] = {s_L[0][q], s_L[1][q], s_L[2][q]}, Hq[3] = {s_H[0][q], s_H[1][q], s_H[2][q]};
                        const int bl[3] = {cq[0] - 1, cq[1] - 1, cq[2] - 1}, bh[3] = {cq[0] + 1, cq[1] + 1, cq[2] + 1};  // the query's own block: done
                        tq[0] = s_q[0][q] - org[0];
                        tq[1] = s_q[1][q] - org[1];
                        tq[2] = s_q[2][q] - org[2];
                        const float lim = __uint_as_float(s_lim[q]) * kNnPrune;
                        for (unsigned long long m0 = w0 & nn_range_mask(Lq, Hq, o0[0], o0[1], o0[2]) & ~nn_range_mask(bl, bh, o0[0], o0[1], o0[2]); m0 != 0ull; m0 &= m0 - 1ull) {
                            const int b0 = __ffsll(m0) - 1;
                            ++n_tested;
                            if (nn_child_lb(tq, o0[0] + (b0 & 3), o0[1] + ((b0 >> 2) & 3), o0[2] + (b0 >> 4), E0, h0) <= lim) m0p |= 1ull << b0;
                        }
                    }
                    uint32_t slot = reserve(static_cast<uint32_t>(__popcll(m0p)), &s_nc, kCellCap, cfull);
                    for (; m0p != 0ull && slot < kCellCap; ++slot) {  // what does not fit stays in m0p until the points phase has emptied the list
                        const int b0 = __ffsll(m0p) - 1;
                        m0p &= m0p - 1ull;
                        const int cx = o0[0] + (b0 & 3), cy = o0[1] + ((b0 >> 2) & 3), cz = o0[2] + (b0 >> 4);
                        s_cl[slot] = static_cast<uint32_t>(q) << 24 | ((static_cast<uint32_t>(cz) * d0[1] + cy) * d0[0] + cx);
                        s_clb[slot] = nn_child_lb(tq, cx, cy, cz, E0, h0);
                    }
                }
                __syncthreads();
                stamp(2);
                const uint32_t nc = min(s_nc, kCellCap);
                if (threadIdx.x == 0) n_listed += nc;
                {
                    // a lane's entries are independent: the table reads of all of them go out together, then their points (the phase is a chain of
                    // two memory round trips per entry otherwise, eight entries deep)
                    constexpr int U = kCellCap / kT;
                    uint32_t kb[U], ke[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const uint32_t j = threadIdx.x + static_cast<uint32_t>(kT) * u;
                        kb[u] = ke[u] = 0;
                        if (j >= nc) continue;
                        const uint32_t cj = s_cl[j];
                        if (s_clb[j] > __uint_as_float(s_lim[cj >> 24]) * kNnPrune) continue;  // lim has moved since the cell was listed
                        const uint32_t at = cj & 0xffffffu;
                        kb[u] = as_global(g.cell_start)[at];
                        ke[u] = as_global(g.cell_start)[at + 1];
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        if (kb[u] == ke[u]) continue;
                        const int   qj = static_cast<int>(s_cl[threadIdx.x + static_cast<uint32_t>(kT) * u] >> 24);
                        const float x = s_q[0][qj], y = s_q[1][qj], z = s_q[2][qj];
                        float       dm = INFINITY;
                        uint32_t    kk = kb[u];
                        if (kIdx) {
                            unsigned long long km = ~0ull;
                            const uint32_t last = ke[u] - 1u;
                            for (; kk < ke[u]; kk += 4) {  // four loads in flight (past the end: the last point again, see below)
                                const float4 p0 = load_point(g.sorted + kk), p1 = load_point(g.sorted + min(kk + 1, last)), p2 = load_point(g.sorted + min(kk + 2, last)),
                                             p3 = load_point(g.sorted + min(kk + 3, last));
                                const unsigned long long k0 = nn_key(sqdist3f(p0.x, p0.y, p0.z, x, y, z), __float_as_int(p0.w)), k1 = nn_key(sqdist3f(p1.x, p1.y, p1.z, x, y, z), __float_as_int(p1.w));
                                const unsigned long long k2 = nn_key(sqdist3f(p2.x, p2.y, p2.z, x, y, z), __float_as_int(p2.w)), k3 = nn_key(sqdist3f(p3.x, p3.y, p3.z, x, y, z), __float_as_int(p3.w));
                                const unsigned long long ka = k0 < k1 ? k0 : k1, kb2 = k2 < k3 ? k2 : k3, kc = ka < kb2 ? ka : kb2;
                                km = kc < km ? kc : km;
                            }
                            ++n_cells;
                            n_points += ke[u] - kb[u];
                            const uint32_t db = static_cast<uint32_t>(km >> 32);  // a point AT the current radius may still win by its index
                            if (db <= s_lim[qj]) {
                                atomicMin(&s_key[qj], km);
                                atomicMin(&s_lim[qj], db);
                            }
                            continue;
                        }
                        // four loads in flight per round trip, the cell's last one to three points included: past the end the LAST point is
                        // read again (a minimum does not mind) instead of one dependent load per leftover point — cells hold 6.4 points on average
                        const uint32_t last = ke[u] - 1u;
                        for (; kk < ke[u]; kk += 4) {
                            const float4 p0 = load_point(g.sorted + kk), p1 = load_point(g.sorted + min(kk + 1, last)), p2 = load_point(g.sorted + min(kk + 2, last)),
                                         p3 = load_point(g.sorted + min(kk + 3, last));
                            dm = fminf(fminf(dm, sqdist3f(p0.x, p0.y, p0.z, x, y, z)), sqdist3f(p1.x, p1.y, p1.z, x, y, z));
                            dm = fminf(fminf(dm, sqdist3f(p2.x, p2.y, p2.z, x, y, z)), sqdist3f(p3.x, p3.y, p3.z, x, y, z));
                        }
                        ++n_cells;
                        n_points += ke[u] - kb[u];
                        if (dm < __uint_as_float(s_lim[qj])) {
                            atomicMin(&s_best[qj], __float_as_uint(dm));
                            atomicMin(&s_lim[qj], __float_as_uint(dm));
                        }
                    }
                }
                __syncthreads();
                stamp(3);
                if (threadIdx.x == 0) s_nc = 0;
                if (!__syncthreads_or(m0p != 0ull || u_ent < n_ent)) break;
            }
            if (threadIdx.x == 0) s_nb = 0;
            if (!__syncthreads_or(act)) break;
        }
        // ================= results =================
        if (mine) {
            if (kIdx) {
                const unsigned long long key = s_key[tid];
                const float best = __uint_as_float(static_cast<uint32_t>(key >> 32));
                const bool  hit = static_cast<double>(best) < max_range;
                sqd[off + qi] = hit ? best : kFitNone;
                J.idx_out[qi] = hit ? static_cast<int32_t>(static_cast<uint32_t>(key)) : -1;
            } else {
                const float best = __uint_as_float(s_best[tid]);
                sqd[off + qi] = static_cast<double>(best) <= max_range ? best : kFitNone;
            }
        }
        __syncthreads();
    }
    if (kStats && stats != nullptr) {
        const uint32_t v[6] = {wave_sum(n_words.v), wave_sum(n_tested.v), wave_sum(n_cells.v), wave_sum(n_points.v), wave_sum(n_bricks.v), wave_sum(n_listed.v)};
        if (lane == 0) {
            const int at[6] = {0, 1, 2, 3, 7, 8};
            for (int kk = 0; kk < 6; ++kk)
                if (v[kk]) atomicAdd(&stats[at[kk]], static_cast<unsigned long long>(v[kk]));
        }
        if (clocks && threadIdx.x == 0)
            for (int kk = 0; kk < 4; ++kk) atomicAdd(&stats[9 + kk], static_cast<unsigned long long>(clk[kk]));
    }
}

#ifndef MRGFE_FAR_GROUP
#define MRGFE_FAR_GROUP 2
#endif
constexpr int kFarGroup = MRGFE_FAR_GROUP;  // lanes per query in the far pass
// The pyramid walk of nn_device.h, one lane group per queued query: round 2's far pass.  It takes what nn_fit_sweep_kernel could not
// seed (nothing within three blocks), or the whole queue with MRGFE_FIT_SWEEP=0 (the reference the sweep is tested against).
// (Measured and dropped: two queues per job — queries the block gave a first distance from the front, queries with nothing around
// them from the back — so that a wavefront of the far pass holds walks of one kind: far 19.5 -> 19.3 ms, block 2.7 -> 2.9 ms.)
template <bool kIdx>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void nn_fit_far_kernel(const NnFitnessJob* __restrict__ jobs, const uint32_t* __restrict__ job_off, double max_range, const uint32_t* __restrict__ pend,
                                                          const uint32_t* __restrict__ pend_cnt, float* __restrict__ sqd)
{
    const uint32_t np = pend_cnt[blockIdx.y];
    constexpr uint32_t per_blk = 256u / kFarGroup;
    if (blockIdx.x * per_blk >= np) return;
    __shared__ NnFitnessJob s_job;
    nn_load_job(s_job, jobs + blockIdx.y);
    __syncthreads();
    const uint32_t off = job_off[blockIdx.y];
    const int      sub = threadIdx.x % kFarGroup;
    for (uint32_t k = blockIdx.x * per_blk + threadIdx.x / kFarGroup; k < np; k += gridDim.x * per_blk) {
        const uint32_t i = pend[off + k];
        const float4   p = load_point(s_job.src + i);
        float x, y, z;
        nn_job_transform(s_job.T12, s_job.gicp_order, p.x, p.y, p.z, x, y, z);
        const float bound = sqd[off + i];  // what the earlier passes gave
        if (kIdx) {  // the walk continues from the (index, distance) the earlier passes found, if any
            int32_t bi = s_job.idx_out[i];
            float   bd = bi >= 0 ? bound : INFINITY;
            int     c[3];
            nn_cell_of(s_job.grid.level[0], x, y, z, c);
            nn_pyramid_walk<kFarGroup, false>(s_job.grid.level[0], x, y, z, c, sub, max_range, bi, bd);
            if (sub == 0) {
                const bool hit = bi >= 0 && static_cast<double>(bd) < max_range;
                sqd[off + i] = hit ? bd : kFitNone;
                s_job.idx_out[i] = hit ? bi : -1;
            }
            continue;
        }
        int32_t bpos;
        float   bd;
        nn_far_search<kFarGroup>(s_job.grid, x, y, z, sub, max_range, bound, bpos, bd);
        bd = fminf(bd, bound);
        if (sub == 0) sqd[off + i] = static_cast<double>(bd) <= max_range ? bd : kFitNone;
    }
}

constexpr uint32_t kFitSumSlice = 1024;
constexpr size_t   kFitSmallTotal = size_t(1) << 20;  // queries of a launch below which the passes run in their many-small-workgroups form
// block partial = (sum, count) over queries [1024 b, 1024 (b + 1)) of the job: the slices and the order of the additions depend on
// the job alone, not on the other jobs of the batch (a block past the job's end writes zeros, which add exactly), so a pair's
// fitness score is the same whichever batch or rank it is matched in
__global__ __launch_bounds__(256) void nn_fit_sum_kernel(const NnFitnessJob* __restrict__ jobs, const uint32_t* __restrict__ job_off, const float* __restrict__ sqd,
                                                          double* __restrict__ partial)
{
    __shared__ double   s_sum[4];
    __shared__ uint32_t s_cnt[4];
    const uint32_t n = jobs[blockIdx.y].n, off = job_off[blockIdx.y];
    double   sum = 0.0;
    uint32_t cnt = 0;
    for (uint32_t k = 0; k < kFitSumSlice / 256u; ++k) {
        const uint32_t i = blockIdx.x * kFitSumSlice + k * 256u + threadIdx.x;
        if (i < n) {
            const float d = sqd[off + i];
            if (d >= 0.0f) { sum += static_cast<double>(d); ++cnt; }
        }
    }
    sum = wave_sum(sum);
    cnt = wave_sum(cnt);
    if (lane_id() == 0) { s_sum[wave_id()] = sum; s_cnt[wave_id()] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double* o = partial + 2 * (size_t(blockIdx.y) * gridDim.x + blockIdx.x);
        o[0] = ((s_sum[0] + s_sum[1]) + s_sum[2]) + s_sum[3];
        o[1] = static_cast<double>(s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3]);
    }
}

__global__ __launch_bounds__(256) void nn_fitness_final_kernel(const double* __restrict__ partial, uint32_t nblk, double* __restrict__ out)
{
    __shared__ double s_a[4], s_b[4];
    const double* part = partial + 2 * size_t(blockIdx.x) * nblk;
    double a = 0, b = 0;
    for (uint32_t i = threadIdx.x; i < nblk; i += 256) { a += part[2 * i]; b += part[2 * i + 1]; }
    a = wave_sum(a);
    b = wave_sum(b);
    if (lane_id() == 0) { s_a[wave_id()] = a; s_b[wave_id()] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = ((s_a[0] + s_a[1]) + s_a[2]) + s_a[3];
        out[2 * blockIdx.x + 1] = ((s_b[0] + s_b[1]) + s_b[2]) + s_b[3];
    }
}

static std::atomic<int> g_fit_sweep{-1};  // -1: not read yet
static int fit_sweep_mode()
{
    int v = g_fit_sweep.load(std::memory_order_relaxed);
    if (v < 0) { const char* e = std::getenv("MRGFE_FIT_SWEEP"); v = e ? (std::atoi(e) != 0 ? 1 : 0) : 1; g_fit_sweep.store(v, std::memory_order_relaxed); }
    return v;
}
int nn_set_fit_sweep(int mode)
{
    if (mode == 0 || mode == 1) g_fit_sweep.store(mode, std::memory_order_relaxed);
    return fit_sweep_mode();
}
static std::atomic<int> g_fit_stats{-1};  // -1: not read yet; 0 off, 1 counters, 2 counters + phase clocks and a line on stderr
static int fit_stats_mode()
{
    int v = g_fit_stats.load(std::memory_order_relaxed);
    if (v < 0) { const char* e = std::getenv("MRGFE_FIT_STATS"); v = e ? std::max(0, std::min(2, std::atoi(e))) : 0; g_fit_stats.store(v, std::memory_order_relaxed); }
    return v;
}
int nn_set_fit_stats(int mode)
{
    if (mode >= 0 && mode <= 2) g_fit_stats.store(mode, std::memory_order_relaxed);
    return fit_stats_mode();
}

int nn_nearest_batch(mrgfe_ctx* ctx, const NnFitnessJob* jobs, size_t count, double max_sq)
{
    if (count == 0) return MRGFE_OK;
    if (count > 65535) { set_error("nn_nearest_batch: too many jobs"); return MRGFE_ERR_INVALID; }
    hipStream_t st = ctx->stream;
    std::vector<uint32_t> off(count + 1, 0u);
    uint32_t max_n = 0;
    for (size_t j = 0; j < count; ++j) {
        if (uint64_t(off[j]) + jobs[j].n > 0xfffffff0ull) { set_error("nn_nearest_batch: more than 2^32 queries"); return MRGFE_ERR_INVALID; }
        if (jobs[j].n && !jobs[j].idx_out) { set_error("nn_nearest_batch: job without an index array"); return MRGFE_ERR_INVALID; }
        off[j + 1] = off[j] + jobs[j].n;
        max_n = std::max(max_n, jobs[j].n);
    }
    const size_t total = off[count];
    if (max_n == 0) return MRGFE_OK;
    // One or a few clouds do not fill the chip with a lane per query (130k queries: 507 workgroups in the block pass, two wavefronts per SIMD
    // working through dependent loads): eight lanes per query then (161 -> 85 us).  (The sweep stays as it is: 64-query tiles made it slower,
    // 244 -> 312 us per 130k-query cloud — a launch lasts as long as its widest tile either way; round 4, a rank's shard of 32 pairs: 2.28 ms of
    // fitness passes with tiles of 256, 2.46 with 128, 2.56 with 64.)
    const bool     small = total < kFitSmallTotal;
    const uint32_t per_blk = small ? 256u / 8u : 256u / kBlockGroup;
    const uint32_t want = static_cast<uint32_t>(std::max<size_t>(1, (size_t(ctx->cu_count) * 128 + count - 1) / count));
    const uint32_t nblk = std::max<uint32_t>(1, std::min<uint32_t>((max_n + per_blk - 1) / per_blk, want));
    // the workspaces of nn_fitness_batch (scratch 9: jobs, offsets, queue lengths; 12: one float per query; 13: the two queues)
    DevBuf &dw = ctx->scratch[9], &dq = ctx->scratch[12], &dp = ctx->scratch[13];
    const size_t jobs_bytes = (sizeof(NnFitnessJob) * count + 255) & ~size_t(255);
    const size_t off_bytes = (sizeof(uint32_t) * 3 * (count + 1) + 255) & ~size_t(255);
    MRGFE_TRY(dw.ensure(jobs_bytes + off_bytes));
    MRGFE_TRY(dq.ensure(sizeof(float) * total));
    MRGFE_TRY(dp.ensure(sizeof(uint32_t) * 2 * total));
    NnFitnessJob* d_jobs = dw.as<NnFitnessJob>();
    uint32_t*     d_off = reinterpret_cast<uint32_t*>(dw.as<char>() + jobs_bytes);
    uint32_t*     d_cnt = d_off + count + 1;
    uint32_t*     d_pend[2] = {dp.as<uint32_t>(), dp.as<uint32_t>() + total};
    uint32_t*     d_cnts[2] = {d_cnt, d_cnt + (count + 1)};
    // this function does not wait on the host: the job records (the caller refills its array every round) and the local offset table
    // go through the context's pinned staging ring, not straight from pageable memory that may be gone when the copy runs
    MRGFE_TRY(ctx->stage_h2d(d_jobs, jobs, sizeof(NnFitnessJob) * count, st));
    MRGFE_TRY(ctx->stage_h2d(d_off, off.data(), sizeof(uint32_t) * (count + 1), st));
    MRGFE_HIP_CHECK(hipMemsetAsync(d_cnt, 0, sizeof(uint32_t) * 2 * (count + 1), st));
    const dim3 grid(nblk, static_cast<uint32_t>(count));
    if (small) hipLaunchKernelGGL((nn_fit_block_kernel<true, 8>), grid, dim3(256), 0, st, d_jobs, d_off, max_sq, dq.as<float>(), d_pend[0], d_cnts[0]);
    else       hipLaunchKernelGGL((nn_fit_block_kernel<true, kBlockGroup>), grid, dim3(256), 0, st, d_jobs, d_off, max_sq, dq.as<float>(), d_pend[0], d_cnts[0]);
    hipLaunchKernelGGL(nn_fit_seed_kernel<true>, grid, dim3(256), 0, st, d_jobs, d_off, max_sq, d_pend[0], d_cnts[0], dq.as<float>(), d_pend[1], d_cnts[1], static_cast<unsigned long long*>(nullptr));
    // the unseeded queries' pyramid walk beside the sweep, as in nn_fitness_batch
    if (!ctx->side) MRGFE_TRY(ctx->make_stream(&ctx->side));
    for (int e = 0; e < 4; ++e)
        if (!ctx->ev_side[e]) MRGFE_HIP_CHECK((e == 1 || e == 2) ? hipEventCreate(&ctx->ev_side[e]) : hipEventCreateWithFlags(&ctx->ev_side[e], hipEventDisableTiming));
    MRGFE_HIP_CHECK(hipEventRecord(ctx->ev_side[0], st));
    MRGFE_HIP_CHECK(hipStreamWaitEvent(ctx->side, ctx->ev_side[0], 0));
    hipLaunchKernelGGL(nn_fit_far_kernel<true>, grid, dim3(256), 0, ctx->side, d_jobs, d_off, max_sq, d_pend[1], d_cnts[1], dq.as<float>());
    MRGFE_HIP_CHECK(hipEventRecord(ctx->ev_side[3], ctx->side));
    hipLaunchKernelGGL((nn_fit_sweep_kernel<true, 256, false>), grid, dim3(256), 0, st, d_jobs, d_off, max_sq, d_pend[0], d_cnts[0], dq.as<float>(), static_cast<unsigned long long*>(nullptr), 0);
    MRGFE_HIP_CHECK(hipStreamWaitEvent(st, ctx->ev_side[3], 0));
    MRGFE_HIP_CHECK(hipGetLastError());
    return MRGFE_OK;
}

int nn_fitness_batch(mrgfe_ctx* ctx, const NnFitnessJob* jobs, size_t count, double max_range, double* out)
{
    for (size_t j = 0; j < count; ++j) out[j] = DBL_MAX;
    if (count == 0) return MRGFE_OK;
    if (count > 65535) { set_error("nn_fitness_batch: too many jobs"); return MRGFE_ERR_INVALID; }
    hipStream_t st = ctx->stream;
    std::vector<uint32_t> off(count + 1, 0u);
    uint32_t max_n = 0;
    for (size_t j = 0; j < count; ++j) {
        if (uint64_t(off[j]) + jobs[j].n > 0xfffffff0ull) { set_error("nn_fitness_batch: more than 2^32 queries"); return MRGFE_ERR_INVALID; }
        off[j + 1] = off[j] + jobs[j].n;
        max_n = std::max(max_n, jobs[j].n);
    }
    const size_t total = off[count];
    if (max_n == 0) return MRGFE_OK;
    // One or a few clouds do not fill the chip with a lane per query (130k queries: 507 workgroups in the block pass, two wavefronts per SIMD
    // working through dependent loads): eight lanes per query then (161 -> 85 us).  (The sweep stays as it is: 64-query tiles made it slower,
    // 244 -> 312 us per 130k-query cloud — a launch lasts as long as its widest tile either way.)
    const bool     small = total < kFitSmallTotal;
    const uint32_t per_blk = small ? 256u / 8u : 256u / kBlockGroup;
    // enough blocks to fill the chip many times over (the far pass is ragged), few enough that each has a few trips
    const uint32_t want = static_cast<uint32_t>(std::max<size_t>(1, (size_t(ctx->cu_count) * 128 + count - 1) / count));
    const uint32_t nblk = std::max<uint32_t>(1, std::min<uint32_t>((max_n + per_blk - 1) / per_blk, want));
    const uint32_t nblk_sum = (max_n + kFitSumSlice - 1) / kFitSumSlice;
    // scratch 9: jobs, offsets, queue lengths, counters, partial sums; 12: one float per query; 13: the two queues (10 and 11 may
    // hold the caller's clouds, see mrgfe_calc_fitness_score)
    DevBuf &dw = ctx->scratch[9], &dq = ctx->scratch[12], &dp = ctx->scratch[13];
    constexpr size_t kStatBytes = sizeof(unsigned long long) * kSweepStats;
    const size_t jobs_bytes = (sizeof(NnFitnessJob) * count + 255) & ~size_t(255);
    const size_t off_bytes = (sizeof(uint32_t) * 3 * (count + 1) + 8 + kStatBytes + 255) & ~size_t(255);
    MRGFE_TRY(dw.ensure(jobs_bytes + off_bytes + sizeof(double) * 2 * (size_t(nblk_sum) + 1) * count));
    MRGFE_TRY(dq.ensure(sizeof(float) * total));
    MRGFE_TRY(dp.ensure(sizeof(uint32_t) * 2 * total));
    NnFitnessJob* d_jobs = dw.as<NnFitnessJob>();
    uint32_t*     d_off = reinterpret_cast<uint32_t*>(dw.as<char>() + jobs_bytes);
    uint32_t*     d_cnt = d_off + count + 1;  // queue lengths after the block pass and after the sweep: [2][count + 1]
    unsigned long long* d_stats = reinterpret_cast<unsigned long long*>(dw.as<char>() + jobs_bytes + off_bytes - kStatBytes);  // 8-byte aligned
    double*       d_part = reinterpret_cast<double*>(dw.as<char>() + jobs_bytes + off_bytes);
    double*       d_res = d_part + 2 * size_t(nblk_sum) * count;
    uint32_t*     d_pend[2] = {dp.as<uint32_t>(), dp.as<uint32_t>() + total};
    uint32_t*     d_cnts[2] = {d_cnt, d_cnt + (count + 1)};
    const bool    sweep = fit_sweep_mode() != 0, counters = fit_stats_mode() != 0;
    for (auto& e : ctx->ev_fit)
        if (!e) MRGFE_HIP_CHECK(hipEventCreate(&e));
    MRGFE_HIP_CHECK(hipMemcpyAsync(d_jobs, jobs, sizeof(NnFitnessJob) * count, hipMemcpyHostToDevice, st));
    MRGFE_HIP_CHECK(hipMemcpyAsync(d_off, off.data(), sizeof(uint32_t) * (count + 1), hipMemcpyHostToDevice, st));
    MRGFE_HIP_CHECK(hipMemsetAsync(d_cnt, 0, off_bytes - sizeof(uint32_t) * (count + 1), st));  // the queue lengths and the counters behind them
    const dim3 grid(nblk, static_cast<uint32_t>(count));
    bool       side_far = false;
    MRGFE_HIP_CHECK(hipEventRecord(ctx->ev_fit[0], st));
    if (small) hipLaunchKernelGGL((nn_fit_block_kernel<false, 8>), grid, dim3(256), 0, st, d_jobs, d_off, max_range, dq.as<float>(), d_pend[0], d_cnts[0]);
    else       hipLaunchKernelGGL((nn_fit_block_kernel<false, kBlockGroup>), grid, dim3(256), 0, st, d_jobs, d_off, max_range, dq.as<float>(), d_pend[0], d_cnts[0]);
    MRGFE_HIP_CHECK(hipEventRecord(ctx->ev_fit[1], st));
    if (sweep) {
        hipLaunchKernelGGL(nn_fit_seed_kernel<false>, grid, dim3(256), 0, st, d_jobs, d_off, max_range, d_pend[0], d_cnts[0], dq.as<float>(), d_pend[1], d_cnts[1], counters ? d_stats : nullptr);
        // The unseeded queries (a few hundred of millions: nothing within three blocks) walk the pyramid, a handful of long dependent walks
        // that occupy a few wavefronts for ~0.25 ms: on a second stream beside the sweep, which leaves them alone (their queue entries are
        // flagged), instead of behind it.
        if (!ctx->side) MRGFE_TRY(ctx->make_stream(&ctx->side));
        for (int e = 0; e < 4; ++e)
            if (!ctx->ev_side[e]) MRGFE_HIP_CHECK((e == 1 || e == 2) ? hipEventCreate(&ctx->ev_side[e]) : hipEventCreateWithFlags(&ctx->ev_side[e], hipEventDisableTiming));
        MRGFE_HIP_CHECK(hipEventRecord(ctx->ev_side[0], st));
        MRGFE_HIP_CHECK(hipStreamWaitEvent(ctx->side, ctx->ev_side[0], 0));
        MRGFE_HIP_CHECK(hipEventRecord(ctx->ev_side[1], ctx->side));
        hipLaunchKernelGGL(nn_fit_far_kernel<false>, grid, dim3(256), 0, ctx->side, d_jobs, d_off, max_range, d_pend[1], d_cnts[1], dq.as<float>());
        MRGFE_HIP_CHECK(hipEventRecord(ctx->ev_side[2], ctx->side));
        MRGFE_HIP_CHECK(hipEventRecord(ctx->ev_side[3], ctx->side));
        if (counters) hipLaunchKernelGGL((nn_fit_sweep_kernel<false, 256, true>), grid, dim3(256), 0, st, d_jobs, d_off, max_range, d_pend[0], d_cnts[0], dq.as<float>(), d_stats, fit_stats_mode() > 1 ? 1 : 0);
        else          hipLaunchKernelGGL((nn_fit_sweep_kernel<false, 256, false>), grid, dim3(256), 0, st, d_jobs, d_off, max_range, d_pend[0], d_cnts[0], dq.as<float>(), static_cast<unsigned long long*>(nullptr), 0);
        MRGFE_HIP_CHECK(hipEventRecord(ctx->ev_fit[2], st));
        MRGFE_HIP_CHECK(hipStreamWaitEvent(st, ctx->ev_side[3], 0));
        side_far = true;
    } else {
        MRGFE_HIP_CHECK(hipEventRecord(ctx->ev_fit[2], st));
        hipLaunchKernelGGL(nn_fit_far_kernel<false>, grid, dim3(256), 0, st, d_jobs, d_off, max_range, d_pend[0], d_cnts[0], dq.as<float>());
    }
    MRGFE_HIP_CHECK(hipEventRecord(ctx->ev_fit[3], st));
    hipLaunchKernelGGL(nn_fit_sum_kernel, dim3(nblk_sum, static_cast<uint32_t>(count)), dim3(256), 0, st, d_jobs, d_off, dq.as<float>(), d_part);
    hipLaunchKernelGGL(nn_fitness_final_kernel, dim3(static_cast<uint32_t>(count)), dim3(256), 0, st, d_part, nblk_sum, d_res);
    MRGFE_HIP_CHECK(hipGetLastError());
    std::vector<double>   res(2 * count);
    std::vector<uint32_t> cnts(2 * (count + 1));
    unsigned long long    h_stats[kSweepStats] = {0};
    MRGFE_HIP_CHECK(hipMemcpyAsync(res.data(), d_res, sizeof(double) * 2 * count, hipMemcpyDeviceToHost, st));
    MRGFE_HIP_CHECK(hipMemcpyAsync(cnts.data(), d_cnt, sizeof(uint32_t) * 2 * (count + 1), hipMemcpyDeviceToHost, st));
    if (counters) MRGFE_HIP_CHECK(hipMemcpyAsync(h_stats, d_stats, sizeof(h_stats), hipMemcpyDeviceToHost, st));
    MRGFE_HIP_CHECK(hipStreamSynchronize(st));
    for (size_t j = 0; j < count; ++j)
        if (res[2 * j + 1] > 0) out[j] = res[2 * j] / res[2 * j + 1];
    FitStats& fs = ctx->fit_stats;
    float ms[3] = {0, 0, 0};
    for (int k = 0; k < 3; ++k) (void)hipEventElapsedTime(&ms[k], ctx->ev_fit[k], ctx->ev_fit[k + 1]);
    if (side_far) (void)hipEventElapsedTime(&ms[2], ctx->ev_side[1], ctx->ev_side[2]);  // the walk ran beside the sweep: its own duration
    fs.ms_block = ms[0];
    fs.ms_sweep = ms[1];
    fs.ms_far = ms[2];
    fs.queries = total;
    fs.queued = fs.queued_far = 0;
    for (size_t j = 0; j < count; ++j) { fs.queued += cnts[j]; fs.queued_far += sweep ? cnts[count + 1 + j] : cnts[j]; }
    fs.words = h_stats[0];
    fs.tested = h_stats[1];
    fs.cells = h_stats[2];
    fs.points = h_stats[3];
    ++fs.calls;
    if (counters && fit_stats_mode() > 1)
        std::fprintf(stderr, "[mrgfe] fitness sweep: %llu queued, %llu words, %llu boxes tested, %llu brick entries, %llu cells listed, %llu opened, %llu points; seeds from super-bricks %llu, "
                             "from blocks %llu, none %llu; clocks (thread 0 of every workgroup) seed / bricks / cells / points: %llu %llu %llu %llu\n",
                     static_cast<unsigned long long>(fs.queued), h_stats[0], h_stats[1], h_stats[7], h_stats[8], h_stats[2], h_stats[3], h_stats[4], h_stats[5], h_stats[6], h_stats[9], h_stats[10],
                     h_stats[11], h_stats[12]);
    return MRGFE_OK;
}

int NnGrid::fitness(mrgfe_ctx* ctx, const float4* d_src, size_t n_src, const float T[16], double max_range, double* out)
{
    *out = DBL_MAX;
    if (!built_) { set_error("NnGrid::fitness before build"); return MRGFE_ERR_STATE; }
    if (n_src == 0 || h_.level[0].n == 0) return MRGFE_OK;
    NnFitnessJob job = make_fitness_job(d_src, n_src, T);
    return nn_fitness_batch(ctx, &job, 1, max_range, out);
}

NnFitnessJob NnGrid::make_fitness_job(const float4* d_src, size_t n_src, const float T[16]) const
{
    NnFitnessJob job;
    job.grid = h_;
    job.src = d_src;
    job.n = static_cast<uint32_t>(n_src);
    job.gicp_order = 0;
    job.idx_out = nullptr;
    std::memcpy(job.T12, T, 48);
    return job;
}

int NnGrid::nearest_device(mrgfe_ctx* ctx, const float4* d_q, size_t n, const float* d_T12, int32_t* d_idx, float* d_sqd)
{
    if (!built_) { set_error("NnGrid::nearest before build"); return MRGFE_ERR_STATE; }
    if (n == 0) return MRGFE_OK;
    const uint32_t nn = static_cast<uint32_t>(n);
    constexpr uint32_t per_blk = 256u / kNnGroup;
    hipLaunchKernelGGL(nn_nearest_kernel, dim3((nn + per_blk - 1) / per_blk), dim3(256), 0, ctx->stream, h_, d_q, nn, d_T12, d_idx, d_sqd);
    MRGFE_HIP_CHECK(hipGetLastError());
    return MRGFE_OK;
}

int NnGrid::nearest_host(mrgfe_ctx* ctx, const float* q, size_t n, size_t stride, int32_t* idx, float* sqd)
{
    if (n == 0) return MRGFE_OK;
    DevBuf &dq = ctx->scratch[10], &dr = ctx->scratch[11];
    MRGFE_TRY(dq.ensure(n * 16));
    MRGFE_TRY(dr.ensure(n * 8));
    MRGFE_TRY(upload_cloud(ctx, q, n, stride, dq.p));
    int32_t* d_idx = dr.as<int32_t>();
    float*   d_sqd = reinterpret_cast<float*>(d_idx + n);
    MRGFE_TRY(nearest_device(ctx, dq.as<float4>(), n, nullptr, d_idx, d_sqd));
    MRGFE_HIP_CHECK(hipMemcpyAsync(idx, d_idx, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    MRGFE_HIP_CHECK(hipMemcpyAsync(sqd, d_sqd, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    MRGFE_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return MRGFE_OK;
}

// RadiusOutlierRemoval (dense path): inlier iff at least `need` points (the query itself included) lie within r
__global__ __launch_bounds__(256) void nn_radius_flags_kernel(NnGridDev g, const float4* __restrict__ q, uint32_t n, double r2, int need, int rings, uint32_t* __restrict__ flags)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float4 p = q[i];
    int          c[3];
    int          count = 0;
    if (g.n > 0 && nn_cell_of(g, p.x, p.y, p.z, c)) {
        nn_walk(
            g, c, 0.0, rings,
            [&](const float4& t) {
                if (static_cast<double>(sqdist3f(t.x, t.y, t.z, p.x, p.y, p.z)) <= r2) ++count;
            },
            [&](double) { return count >= need; });
    }
    flags[i] = count >= need ? 1u : 0u;
}

int NnGrid::radius_count_flags(mrgfe_ctx* ctx, const float4* d_q, size_t n, double r2, int need, uint32_t* d_flags)
{
    if (!built_) { set_error("NnGrid::radius_count_flags before build"); return MRGFE_ERR_STATE; }
    if (n == 0) return MRGFE_OK;
    const uint32_t nn = static_cast<uint32_t>(n);
    // every point within r of the query lies within ceil(r / cell) + 1 rings of its cell
    const int rings = static_cast<int>(std::ceil(std::sqrt(r2) / h_.level[0].cell)) + 1;
    hipLaunchKernelGGL(nn_radius_flags_kernel, dim3((nn + 255) / 256), dim3(256), 0, ctx->stream, h_.level[0], d_q, nn, r2, need, rings, d_flags);
    MRGFE_HIP_CHECK(hipGetLastError());
    return MRGFE_OK;
}

// k nearest neighbours, one wavefront per query.  The running top-k list lives in registers, entry j in lane j, kept
// sorted by (squared distance, index); the wave reads 64 candidates of a cell run per step (coalesced), ballots the ones
// that beat the current k-th entry and inserts them one at a time with a popcount for the slot and one lane shift.
// (distance, index) is a total order, so the result does not depend on the visiting order.  A query walks kKnnRings rings
// of a level and then starts over on the next coarser one (sparse surroundings: the k-th neighbour is many fine cells
// away); candidates it meets again there are recognised in the list by their (distance, index) and skipped.
constexpr int kKnnRings = 2;
#ifndef MRGFE_KNN_SORT_MIN
#define MRGFE_KNN_SORT_MIN 9
#endif
constexpr int kKnnSortMin = MRGFE_KNN_SORT_MIN;  // candidates of one 64-wide step that beat the k-th entry: from here on sort + merge beats one-by-one insertion

// (distance, index) order of the neighbour lists
// (squared distances are >= +0 or +inf and indices >= 0: their bit patterns order like the values, so the pair compares as ONE unsigned
// 64-bit number — one v_cmp instead of three compares and two logic operations, in a kernel bound by instruction issue)
__device__ __forceinline__ bool knn_less(float ad, int32_t ai, float bd, int32_t bi)
{
    return (static_cast<unsigned long long>(__float_as_uint(ad)) << 32 | static_cast<uint32_t>(ai)) < (static_cast<unsigned long long>(__float_as_uint(bd)) << 32 | static_cast<uint32_t>(bi));
}

// compare-exchange with the lane `stride` away; `up`: this pair ends ascending from the lower lane
__device__ __forceinline__ void knn_cmpx(float& d, int32_t& i, int stride, bool up)
{
    const float   od = __shfl_xor(d, stride);
    const int32_t oi = __shfl_xor(i, stride);
    const bool    lower = (lane_id() & stride) == 0;
    const bool    mine_less = knn_less(d, i, od, oi);
    // the lower lane of an ascending pair keeps the smaller element, its partner the larger; reversed when descending
    const bool keep = (lower == up) ? mine_less : !mine_less;
    if (!keep) { d = od; i = oi; }
}

// The k smallest of (sorted list in td/ti, lanes >= its length hold (inf, max)) and (64 unsorted candidates in d/ci, the
// ones that do not count set to (inf, max)), sorted ascending over the lanes again.  Bitonic sort of the candidates
// (21 compare-exchange stages), elementwise minimum with the reversed list (a bitonic sequence holding the 64 smallest),
// bitonic merge (6 stages).
// `empty`: the list holds nothing yet (uniform) — the sorted candidates ARE the new list, no merge
__device__ __forceinline__ void knn_sort_merge(float& td, int32_t& ti, float d, int32_t ci, bool empty)
{
#pragma unroll
    for (int size = 2; size <= 64; size <<= 1) {
        const bool up = (lane_id() & size) == 0 || size == 64;
#pragma unroll
        for (int stride = size >> 1; stride > 0; stride >>= 1) knn_cmpx(d, ci, stride, up);
    }
    if (empty) {
        td = d;
        ti = ci;
        return;
    }
    const float   rd = __shfl(d, 63 - lane_id());
    const int32_t ri = __shfl(ci, 63 - lane_id());
    if (knn_less(rd, ri, td, ti)) { td = rd; ti = ri; }
#pragma unroll
    for (int stride = 32; stride > 0; stride >>= 1) knn_cmpx(td, ti, stride, true);
}

// one query per wavefront: the k nearest of p, written to row i of idx / sqd; returns the candidates measured
__device__ __forceinline__ uint32_t nn_knn_query(const NnGrid2Dev& g, const float4 p, uint32_t i, int k, int32_t* __restrict__ idx, float* __restrict__ sqd)
{
    const int    lane = lane_id();
    float        td = INFINITY;     // list entry of this lane
    int32_t      ti = 0x7fffffff;
    float        kth_d = INFINITY;  // current k-th entry (uniform); (inf, max) while the list is not full
    int32_t      kth_i = 0x7fffffff;
    int          cnt = 0;
    uint32_t     n_cand = 0;  // candidates measured (uniform; diagnostics)
    if (g.level[0].n > 0 && finite3(p.x, p.y, p.z)) {
        bool done = false;
        for (int l = 0; l < g.n_levels && !done; ++l) {
            const NnGridDev& lv = g.level[l];
            const bool       again = l > 0;  // candidates may already be in the list
            int              c[3];
            nn_cell_of(lv, p.x, p.y, p.z, c);
            int rmax = 0;
#pragma unroll
            for (int a = 0; a < 3; ++a) rmax = max(rmax, max(c[a], lv.dim[a] - 1 - c[a]));
            const int last_ring = l + 1 < g.n_levels ? min(rmax, kKnnRings) : rmax;
            done = true;  // unless the walk below runs out of rings on this level
            // one 64-wide step: lane candidates (valid, position kk in lv.sorted) against the list
            auto step = [&](bool valid, const float4& cand) {
                float   d = INFINITY;
                int32_t ci = 0x7fffffff;
                if (valid) {
                    d = sqdist3f(cand.x, cand.y, cand.z, p.x, p.y, p.z);
                    ci = __float_as_int(cand.w);
                }
                const bool beats = valid && knn_less(d, ci, kth_d, kth_i);
                n_cand += static_cast<uint32_t>(__popcll(__ballot(valid)));
                uint64_t   m = __ballot(beats);
                if (!again && __popcll(m) >= kKnnSortMin) {  // many at once (the first steps of a query): sort + merge
                    knn_sort_merge(td, ti, beats ? d : INFINITY, beats ? ci : 0x7fffffff, cnt == 0);
                    if (lane >= k) { td = INFINITY; ti = 0x7fffffff; }
                    cnt = min(k, cnt + static_cast<int>(__popcll(m)));
                    if (cnt == k) { kth_d = wave_read(td, k - 1); kth_i = wave_read(ti, k - 1); }
                    m = 0;
                }
                while (m) {
                    const int src = __ffsll(static_cast<unsigned long long>(m)) - 1;
                    m &= m - 1;
                    const float   cd = wave_read(d, src);  // (scalar: the candidate is the same for every lane)
                    const int32_t cci = wave_read(ci, src);
                    if (!knn_less(cd, cci, kth_d, kth_i)) continue;  // the k-th entry moved since the ballot
                    if (again && __ballot(td == cd && ti == cci)) continue;       // met on a finer level already
                    const bool    mine_less = knn_less(td, ti, cd, cci);
                    const int     pos = __popcll(__ballot(mine_less));  // sorted list: the lanes below pos hold the smaller entries
                    const float   up_d = __shfl_up(td, 1);
                    const int32_t up_i = __shfl_up(ti, 1);
                    if (lane == pos) { td = cd; ti = cci; }
                    else if (lane > pos) { td = up_d; ti = up_i; }
                    if (lane >= k) { td = INFINITY; ti = 0x7fffffff; }
                    if (cnt < k) ++cnt;
                    if (cnt == k) { kth_d = wave_read(td, k - 1); kth_i = wave_read(ti, k - 1); }
                }
            };
            // the candidates of up to 64 ranges of lv.sorted, lane j holding range [qb, qb + ql): consumed as ONE list, 64 candidates a step whatever the
            // range lengths (a ring's ranges hold a handful of points each where a query has to look that far: a step per range measured ~30 extra
            // candidates per query for a third of the kernel's time).  Candidate v of the list lies in the LAST range j whose exclusive prefix is
            // <= v: six cross-lane probes.
            auto consume_ranges = [&](uint32_t qb, uint32_t ql) {
                const uint32_t incl = wave_inclusive_scan(ql), excl = incl - ql;
                const uint32_t total = wave_read(incl, kWave - 1);
                for (uint32_t base = 0; base < total; base += 64u) {
                    const uint32_t v = base + lane;
                    int            j = 0;
#pragma unroll
                    for (int s = 32; s > 0; s >>= 1)
                        if (__shfl(excl, j + s) <= v) j += s;
                    const uint32_t at = __shfl(qb, j) + (v - __shfl(excl, j));
                    step(v < total, v < total ? lv.sorted[at] : make_float4(0.0f, 0.0f, 0.0f, 0.0f));
                }
            };
            const double margin = nn_face_margin(lv, c, p.x, p.y, p.z);
            int          first_ring = 0;
            if (last_ring >= 1) {
                // rings 0 and 1 together: the nine x-rows of the 3x3x3 block.  Lanes 0..8 fetch the bounds of one row each
                // (one round trip instead of nine dependent ones), the rows are then consumed as ONE list, 64 candidates
                // per step whatever the row lengths, nearest rows first.
                constexpr unsigned long long order = 0x862075314ull;  // row j: dz = j / 3 - 1, dy = j % 3 - 1; 4, 1, 3, 5, 7, 0, 2, 6, 8 (a nibble each)
                uint32_t  rb = 0, rl = 0;
                if (lane < 9) {
                    const int j = static_cast<int>((order >> (4 * lane)) & 15ull), zz = c[2] + j / 3 - 1, yy = c[1] + j % 3 - 1;
                    if (zz >= 0 && zz < lv.dim[2] && yy >= 0 && yy < lv.dim[1]) {
                        const uint32_t row = (static_cast<uint32_t>(zz) * lv.dim[1] + yy) * lv.dim[0];
                        rb = lv.cell_start[row + max(c[0] - 1, 0)];
                        rl = lv.cell_start[row + min(c[0] + 1, lv.dim[0] - 1) + 1] - rb;
                    }
                }
                uint32_t off[10], beg[9];
                off[0] = 0;
#pragma unroll
                for (int j = 0; j < 9; ++j) {
                    beg[j] = wave_read(rb, j);  // (scalars: nineteen registers of the wavefront, not of every lane)
                    off[j + 1] = off[j] + wave_read(rl, j);
                }
                // position of candidate v in lv.sorted: v + (beg[j] - off[j]) for the row j it falls into, i.e. the LAST j with off[j] <= v (off
                // ascends): one compare and one select per row on scalar operands
                uint32_t dlt[9];
#pragma unroll
                for (int j = 0; j < 9; ++j) dlt[j] = beg[j] - off[j];
                for (uint32_t base = 0; base < off[9]; base += 64u) {
                    const uint32_t v = base + lane;
                    uint32_t       dv = dlt[0];
#pragma unroll
                    for (int j = 1; j < 9; ++j)
                        if (v >= off[j]) dv = dlt[j];
                    step(v < off[9], v < off[9] ? lv.sorted[v + dv] : make_float4(0.0f, 0.0f, 0.0f, 0.0f));
                }
                first_ring = 2;
                // ring 2 the same way when it is needed: its 16 face rows and 18 end cells are 34 ranges, their bounds fetched by
                // lanes 0..33 in one round trip instead of 34 dependent probes
                const double b2 = static_cast<double>(lv.cell) + margin;  // everything outside the block is at least this far
                if (last_ring >= 2 && !(cnt == k && static_cast<double>(kth_d) < b2 * b2 * (1.0 - 1e-5))) {
                    uint32_t qb = 0, ql = 0;
                    if (lane < 34) {
                        int dz, dy, x0, x1;
                        if (lane < 16) {  // rows on a y / z face: the full x extent of the ring
                            const int f = lane;  // 0..4: dz = -2; 5..9: dz = +2; 10..12: dy = -2, dz = -1..1; 13..15: dy = +2
                            if (f < 5) { dz = -2; dy = f - 2; }
                            else if (f < 10) { dz = 2; dy = f - 7; }
                            else if (f < 13) { dy = -2; dz = f - 11; }
                            else { dy = 2; dz = f - 14; }
                            x0 = c[0] - 2;
                            x1 = c[0] + 2;
                        } else {  // inner rows: the two end cells
                            const int e = lane - 16;  // 0..17: row e / 2 of the 3x3 inner rows, side e & 1
                            dz = (e >> 1) / 3 - 1;
                            dy = (e >> 1) % 3 - 1;
                            x0 = x1 = (e & 1) ? c[0] + 2 : c[0] - 2;
                        }
                        const int zz = c[2] + dz, yy = c[1] + dy;
                        x0 = max(x0, 0);
                        x1 = min(x1, lv.dim[0] - 1);
                        if (zz >= 0 && zz < lv.dim[2] && yy >= 0 && yy < lv.dim[1] && x0 <= x1) {
                            const uint32_t row = (static_cast<uint32_t>(zz) * lv.dim[1] + yy) * lv.dim[0];
                            qb = lv.cell_start[row + x0];
                            ql = lv.cell_start[row + x1 + 1] - qb;
                        }
                    }
                    consume_ranges(qb, ql);
                    first_ring = 3;
                }
            }
            // The remaining rings.  Ring r is (2r+1)^2 x-rows: the rows on a y / z face contribute their full x extent, the inner ones their
            // two end cells — 2 (2r+1)^2 ranges, most of them empty where a query has to look this far.  Their bounds are fetched 64 at a
            // time, one lane each (probing them one after the other made a handful of isolated points — hundreds of dependent round trips
            // each — the tail that set the duration of the whole launch); the non-empty ones are then consumed 64 candidates a step.
            for (int r = first_ring; r <= last_ring; ++r) {
                if (r >= 1) {
                    const double b = static_cast<double>(r - 1) * static_cast<double>(lv.cell) + margin;
                    if (cnt == k && static_cast<double>(kth_d) < b * b * (1.0 - 1e-5)) break;
                }
                const int      w = 2 * r + 1;
                const uint32_t total = 2u * static_cast<uint32_t>(w) * static_cast<uint32_t>(w);
                for (uint32_t t0 = 0; t0 < total; t0 += 64u) {
                    const uint32_t t = t0 + lane;
                    uint32_t       qb = 0, qe = 0;
                    if (t < total) {
                        const int rowi = static_cast<int>(t >> 1), side = static_cast<int>(t & 1u);
                        const int dz = rowi / w - r, dy = rowi % w - r, zz = c[2] + dz, yy = c[1] + dy;
                        if (zz >= 0 && zz < lv.dim[2] && yy >= 0 && yy < lv.dim[1]) {
                            const uint32_t row = (static_cast<uint32_t>(zz) * lv.dim[1] + yy) * lv.dim[0];
                            const bool     face = dz == r || dz == -r || dy == r || dy == -r;
                            int            x0 = 1, x1 = 0;
                            if (face) {
                                if (side == 0) { x0 = max(c[0] - r, 0); x1 = min(c[0] + r, lv.dim[0] - 1); }
                            } else {
                                x0 = x1 = side ? c[0] + r : c[0] - r;
                                if (x0 < 0 || x0 >= lv.dim[0]) { x0 = 1; x1 = 0; }
                            }
                            if (x0 <= x1) {
                                qb = lv.cell_start[row + x0];
                                qe = lv.cell_start[row + x1 + 1];
                            }
                        }
                    }
                    if (__ballot(qe > qb)) consume_ranges(qb, qe - qb);
                }
            }
            // conclusive iff everything beyond the walked rings is farther than the k-th entry (or the level is exhausted)
            if (last_ring < rmax) {
                const double bnd = static_cast<double>(last_ring) * static_cast<double>(lv.cell) + margin;
                done = cnt == k && static_cast<double>(kth_d) < bnd * bnd * (1.0 - 1e-5);
            }
        }
    }
    if (lane < k) {
        idx[size_t(i) * k + lane] = lane < cnt ? ti : -1;
        sqd[size_t(i) * k + lane] = lane < cnt ? td : -1.0f;
    }
    return n_cand;
}

#ifndef MRGFE_KNN_WAVES
#define MRGFE_KNN_WAVES 8  // the kernel is bound by instruction issue (VALU busy 0.7) with long dependent chains: eight wavefronts per SIMD on 64 VGPRs and
                           // 40 bytes of scratch run 4 % faster than seven on 65 without (0.283 against 0.295 ms per 130k queries, k = 20; six on 76: 0.304)
#endif
__attribute__((amdgpu_waves_per_eu(MRGFE_KNN_WAVES)))
__global__ __launch_bounds__(256) void nn_knn_kernel(NnGrid2Dev g, const float4* __restrict__ q, uint32_t n, int k, int32_t* __restrict__ idx, float* __restrict__ sqd,
                                                      unsigned long long* __restrict__ stats)
{
    const uint32_t i = (blockIdx.x * 256u + threadIdx.x) >> 6;
    if (i >= n) return;  // uniform per wave
    const uint32_t n_cand = nn_knn_query(g, q[i], i, k, idx, sqd);
    if (stats != nullptr && lane_id() == 0 && n_cand) atomicAdd(stats, static_cast<unsigned long long>(n_cand));
}

int NnGrid::knn_device(mrgfe_ctx* ctx, const float4* d_q, size_t n, int k, int32_t* d_idx, float* d_sqd)
{
    if (!built_) { set_error("NnGrid::knn before build"); return MRGFE_ERR_STATE; }
    if (k < 1 || k > 64) { set_error("NnGrid::knn: k must be in [1, 64]"); return MRGFE_ERR_INVALID; }
    if (n == 0) return MRGFE_OK;
    if (n > (0xffffffffu >> 6)) { set_error("NnGrid::knn: too many queries"); return MRGFE_ERR_INVALID; }
    const uint32_t nn = static_cast<uint32_t>(n);
    // measurement hook (mrgfe_ctx_knn_stats): HIP events around the launch; with the diagnostic counters on, the candidates it measures
    KnnStats& ks = ctx->knn_stats;
    for (auto& e : ks.ev)
        if (!e) MRGFE_HIP_CHECK(hipEventCreate(&e));
    unsigned long long* d_cand = nullptr;
    if (fit_stats_mode() != 0) {
        MRGFE_TRY(ks.counter.ensure(8));
        d_cand = ks.counter.as<unsigned long long>();
        MRGFE_HIP_CHECK(hipMemsetAsync(d_cand, 0, 8, ctx->stream));
    }
    MRGFE_HIP_CHECK(hipEventRecord(ks.ev[0], ctx->stream));
    hipLaunchKernelGGL(nn_knn_kernel, dim3((nn + 3) / 4), dim3(256), 0, ctx->stream, h_, d_q, nn, k, d_idx, d_sqd, d_cand);
    MRGFE_HIP_CHECK(hipGetLastError());
    MRGFE_HIP_CHECK(hipEventRecord(ks.ev[1], ctx->stream));
    ks.queries = n;
    ks.k = k;
    ks.counted = d_cand != nullptr;
    ks.launches += 1;
    return MRGFE_OK;
}

}  // namespace mrgfe
