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

namespace mrgfe {

// ---- build ---------------------------------------------------------------------------------------------------
constexpr uint32_t kCrowdSlots = 32;
__global__ __launch_bounds__(256) void nn_cellkey_kernel(const float4* __restrict__ pts, uint32_t n, NnGridDev g, uint32_t n_cells, uint32_t* __restrict__ keys,
                                                          uint32_t* __restrict__ vals, uint32_t* __restrict__ counts, unsigned long long* __restrict__ crowd)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    uint32_t       key = n_cells;  // non-finite points: behind every cell
    if (i < n) {
        const float4 p = pts[i];
        int          c[3];
        if (nn_cell_of(g, p.x, p.y, p.z, c)) key = (static_cast<uint32_t>(c[2]) * g.dim[1] + c[1]) * g.dim[0] + c[0];
        keys[i] = key;
        vals[i] = i;
    }
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
        const uint32_t lkey = __shfl(key, leader);
        const bool     mine = todo && key == lkey;
        const uint64_t grp = __ballot(mine);
        if (mine) {
            uint32_t old = 0;
            if (lane_id() == leader) old = atomicAdd(&counts[lkey], static_cast<uint32_t>(__popcll(grp)));
            old = __shfl(old, leader);
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

// occupancy words of the bricks (after cell_start is final): one thread per cell, an atomic only for occupied cells
__global__ __launch_bounds__(256) void nn_occupancy_kernel(NnGridDev g, uint32_t n_cells, unsigned long long* __restrict__ occ)
{
    const uint32_t at = blockIdx.x * 256u + threadIdx.x;
    if (at >= n_cells || g.cell_start[at + 1] == g.cell_start[at]) return;
    const uint32_t x = at % g.dim[0], y = (at / g.dim[0]) % g.dim[1], z = at / (static_cast<uint32_t>(g.dim[0]) * g.dim[1]);
    const uint32_t brick = ((z >> 2) * g.bdim[1] + (y >> 2)) * g.bdim[0] + (x >> 2);
    atomicOr(&occ[brick], 1ull << ((x & 3u) | ((y & 3u) << 2) | ((z & 3u) << 4)));
}

// next pyramid level: bit of a child node set iff its word is non-zero (dims = child grid, pdim = parent grid)
__global__ __launch_bounds__(256) void nn_occupancy_up_kernel(const unsigned long long* __restrict__ child, int dx, int dy, int dz, int px, int py, unsigned long long* __restrict__ parent)
{
    const uint32_t at = blockIdx.x * 256u + threadIdx.x;
    if (at >= static_cast<uint32_t>(dx) * dy * dz || child[at] == 0ull) return;
    const uint32_t x = at % dx, y = (at / dx) % dy, z = at / (static_cast<uint32_t>(dx) * dy);
    atomicOr(&parent[((z >> 2) * py + (y >> 2)) * px + (x >> 2)], 1ull << ((x & 3u) | ((y & 3u) << 2) | ((z & 3u) << 4)));
}

__global__ __launch_bounds__(256) void nn_gather_kernel(const float4* __restrict__ pts, const uint32_t* __restrict__ sorted_vals, uint32_t n_valid, float4* __restrict__ sorted)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n_valid) return;
    const uint32_t v = sorted_vals[i];
    float4 p = pts[v];
    p.w = __int_as_float(static_cast<int>(v));
    sorted[i] = p;
}

// bins, sorts and gathers one level; `counts_only` stops after the binning kernel (adaptive cell search)
int NnGrid::build_level(mrgfe_ctx* ctx, const float4* d_pts, uint32_t nn, const BBox& bb, float cell, const SliceTable& tab, NnGridDev& lv, DevBuf& d_cells, DevBuf& d_sorted,
                        bool counts_only, double* crowding)
{
    hipStream_t st = ctx->stream;
    DevBuf &ds = ctx->scratch[0], &dk = ctx->scratch[2], &dv = ctx->scratch[3], &dkt = ctx->scratch[4], &dvt = ctx->scratch[5], &dh = ctx->scratch[6], &dblk = ctx->scratch[8];
    float extent = 0.0f;
    for (int a = 0; a < 3; ++a) { lv.origin[a] = bb.mn[a]; extent = std::max(extent, bb.mx[a] - bb.mn[a]); }
    lv.cell = cell;
    lv.slack = 1e-6f * (extent + cell);
    lv.n = bb.n_finite;
    for (int a = 0; a < 3; ++a) lv.dim[a] = static_cast<int>(std::floor((bb.mx[a] - bb.mn[a]) / cell)) + 1;
    const uint32_t n_cells = static_cast<uint32_t>(lv.dim[0]) * lv.dim[1] * lv.dim[2];
    int    pd[3][3];  // node grids of the occupancy pyramid: bricks, super-bricks, blocks
    size_t pn[3] = {1, 1, 1};
    for (int a = 0; a < 3; ++a) {
        lv.bdim[a] = pd[0][a] = (lv.dim[a] + 3) / 4;
        pd[1][a] = (pd[0][a] + 3) / 4;
        pd[2][a] = (pd[1][a] + 3) / 4;
        for (int k = 0; k < 3; ++k) pn[k] *= static_cast<size_t>(pd[k][a]);
    }
    // [counts / cell_start: n_cells + 1][crowd counters][occupancy words of the three pyramid levels], zeroed together
    const size_t head_words = size_t(n_cells) + 4 + 2 * kCrowdSlots, occ_at = (head_words + 1) & ~size_t(1);
    const size_t all_words = occ_at + 2 * (pn[0] + pn[1] + pn[2]);
    MRGFE_TRY(d_cells.ensure(sizeof(uint32_t) * all_words));
    MRGFE_HIP_CHECK(hipMemsetAsync(d_cells.p, 0, sizeof(uint32_t) * all_words, st));
    lv.cell_start = d_cells.as<uint32_t>();
    lv.occ = reinterpret_cast<const unsigned long long*>(d_cells.as<uint32_t>() + occ_at);
    lv.occ1 = lv.occ + pn[0];
    lv.occ2 = lv.occ1 + pn[1];
    // the crowd counters live behind the (n_cells + 1)-entry count table, 8-byte aligned
    unsigned long long* d_crowd = reinterpret_cast<unsigned long long*>(d_cells.as<uint32_t>() + ((size_t(n_cells) + 2) & ~size_t(1)));
    hipLaunchKernelGGL(nn_cellkey_kernel, dim3((nn + 255) / 256), dim3(256), 0, st, d_pts, nn, lv, n_cells, dk.as<uint32_t>(), dv.as<uint32_t>(), d_cells.as<uint32_t>(),
                       crowding ? d_crowd : nullptr);
    MRGFE_HIP_CHECK(hipGetLastError());
    if (crowding) {
        unsigned long long slots[kCrowdSlots], crowd = 0;
        MRGFE_HIP_CHECK(hipMemcpyAsync(slots, d_crowd, sizeof(slots), hipMemcpyDeviceToHost, st));
        MRGFE_HIP_CHECK(hipStreamSynchronize(st));
        for (unsigned long long v : slots) crowd += v;
        // population of the cell an average POINT sits in (queries are distributed like the points, not like the cells)
        *crowding = 1.0 + 2.0 * double(crowd) / double(bb.n_finite);
    }
    if (counts_only) return MRGFE_OK;
    int key_bits = 1;
    while (key_bits < 32 && (uint64_t(1) << key_bits) <= n_cells) ++key_bits;
    uint32_t *sk, *sv;
    MRGFE_TRY(radix_sort_pairs(ctx, dk.as<uint32_t>(), dv.as<uint32_t>(), dkt.as<uint32_t>(), dvt.as<uint32_t>(), ds.as<Slice>(), tab, key_bits, dh.as<uint32_t>(), &sk, &sv));
    // counts -> cell_start (exclusive scan over n_cells + 1 entries, in place)
    uint32_t   nc1 = n_cells + 1;
    SliceTable ctab;
    ctab.build(&nc1, 1);
    MRGFE_HIP_CHECK(hipMemcpyAsync(ds.as<char>() + sizeof(Slice), ctab.h.data(), sizeof(Slice), hipMemcpyHostToDevice, st));
    MRGFE_TRY(dblk.ensure(sizeof(uint32_t) * (ctab.total_blks + 8)));
    MRGFE_TRY(exclusive_scan(ctx, d_cells.as<uint32_t>(), d_cells.as<uint32_t>(), ds.as<Slice>() + 1, ctab, dblk.as<uint32_t>(), dblk.as<uint32_t>() + ctab.total_blks));
    if (&lv == &h_.level[0]) {  // only the finest level is searched through the pyramid (the coarser ones serve the k-NN climb)
        hipLaunchKernelGGL(nn_occupancy_kernel, dim3((n_cells + 255) / 256), dim3(256), 0, st, lv, n_cells, const_cast<unsigned long long*>(lv.occ));
        hipLaunchKernelGGL(nn_occupancy_up_kernel, dim3(static_cast<uint32_t>((pn[0] + 255) / 256)), dim3(256), 0, st, lv.occ, pd[0][0], pd[0][1], pd[0][2], pd[1][0], pd[1][1],
                           const_cast<unsigned long long*>(lv.occ1));
        hipLaunchKernelGGL(nn_occupancy_up_kernel, dim3(static_cast<uint32_t>((pn[1] + 255) / 256)), dim3(256), 0, st, lv.occ1, pd[1][0], pd[1][1], pd[1][2], pd[2][0], pd[2][1],
                           const_cast<unsigned long long*>(lv.occ2));
    }
    MRGFE_TRY(d_sorted.ensure(sizeof(float4) * std::max<size_t>(nn, 1)));
    lv.sorted = d_sorted.as<float4>();
    hipLaunchKernelGGL(nn_gather_kernel, dim3((lv.n + 255) / 256), dim3(256), 0, st, d_pts, sv, lv.n, d_sorted.as<float4>());
    MRGFE_HIP_CHECK(hipGetLastError());
    MRGFE_HIP_CHECK(hipStreamSynchronize(st));  // ctab's host table was the source of an async copy
    return MRGFE_OK;
}

int NnGrid::build(mrgfe_ctx* ctx, const float4* d_pts, size_t n, float cell_size, double crowding_target, int max_levels)
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
    BBox bb;
    MRGFE_HIP_CHECK(hipMemcpyAsync(&bb, d_out, sizeof(BBox), hipMemcpyDeviceToHost, st));
    MRGFE_HIP_CHECK(hipStreamSynchronize(st));
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
    if (crowding_target > 0) {
        // halve the edge while the cell an average point sits in is more crowded than the target (at most four times).
        // LiDAR returns lie on surfaces, so the crowding falls about 4x per halving: jump by the predicted number of
        // halvings, then correct by single steps.
        int halvings = 0;
        for (int pass = 0; pass < 3 && halvings < 4 && cells_at(cell * 0.5f) <= double(1u << 24); ++pass) {
            double crowding = 0;
            MRGFE_TRY(build_level(ctx, d_pts, nn, bb, cell, tab, h_.level[0], d_cell_start_[0], d_sorted_[0], true, &crowding));
            if (crowding <= crowding_target) break;
            int step = std::max(1, static_cast<int>(std::ceil(std::log(crowding / crowding_target) / std::log(4.0))));
            step = std::min(step, 4 - halvings);
            while (step > 1 && cells_at(cell * std::ldexp(1.0f, -step)) > double(1u << 24)) --step;
            cell *= std::ldexp(1.0f, -step);
            halvings += step;
        }
    }
    MRGFE_TRY(build_level(ctx, d_pts, nn, bb, cell, tab, h_.level[0], d_cell_start_[0], d_sorted_[0], false, nullptr));
    // coarser levels for queries whose neighbourhood is empty at the finer scale: kLevelRatio x the edge each, same origin,
    // as long as the level above still has more than a handful of cells per axis
    float ratio = kLevelRatio;
    if (const char* e = std::getenv("MRGFE_NN_COARSE_RATIO")) ratio = std::max(2.0f, static_cast<float>(std::atof(e)));  // tuning hook
    while (h_.n_levels < std::min(max_levels, kNnMaxLevels)) {
        const NnGridDev& below = h_.level[h_.n_levels - 1];
        if (std::max(below.dim[0], std::max(below.dim[1], below.dim[2])) <= 4) break;
        MRGFE_TRY(build_level(ctx, d_pts, nn, bb, below.cell * ratio, tab, h_.level[h_.n_levels], d_cell_start_[h_.n_levels], d_sorted_[h_.n_levels], false, nullptr));
        ++h_.n_levels;
    }
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
    if (!ctx->tmp_grid) return;
    ctx->tmp_grid->release();
    delete ctx->tmp_grid;
    ctx->tmp_grid = nullptr;
}

void NnGrid::release()
{
    for (auto& b : d_cell_start_) b.release();
    for (auto& b : d_sorted_) b.release();
    built_ = false;
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

#ifndef MRGFE_BLOCK_GROUP
#define MRGFE_BLOCK_GROUP 1
#endif
constexpr int kBlockGroup = MRGFE_BLOCK_GROUP;  // lanes per query in the block pass
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
    const int         sub = threadIdx.x % kBlockGroup;
    constexpr uint32_t per_blk = 256u / kBlockGroup;
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
        const uint32_t i = i0 + threadIdx.x / kBlockGroup;
        bool           queue = false;
        if (i < i_end) {
            const float4 p = load_point(s_job.src + i);
            float x, y, z;
            transform_point(s_job.T12, p.x, p.y, p.z, x, y, z);
            int32_t bi = -1;
            float   bd = INFINITY;
            bool    done = true;
            if (g.level[0].n != 0 && finite3(x, y, z)) done = nn_level_search<kBlockGroup>(g.level[0], x, y, z, sub, 1, max_range, bi, bd);
            // queued queries leave what the block gave (INFINITY: nothing) as the far pass's starting bound
            if (sub == 0) sqd[off + i] = done ? ((bi >= 0 && static_cast<double>(bd) <= max_range) ? bd : kFitNone) : (bi >= 0 ? bd : INFINITY);
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

// ---- far pass, first part: the bricks within two shells of the query's, as flat work items ----------------------------
// A queued query's walk over the 5 x 5 x 5 bricks around it is a nest of data-dependent loops (which words are non-empty, which of
// their cells are occupied, how many points those hold): run one query per lane (or lane pair) a wavefront executes the union of
// 32 different walks — ~15 000 VALU instructions per wavefront in round 2's nn_fit_far_kernel, of which the distance evaluations
// themselves are a few per cent.  Here a workgroup takes a tile of 64 queued queries and turns every level of the nest into a list
// of equal work items in LDS:
//   phase 1  item = (query, brick node): lane = query, the node is uniform over the wavefront (its offset is scalar); lower bound of
//            the node's box against the query's `lim`, occupancy word fetched; non-empty ones appended to the brick list;
//   phase 2  item = brick list entry: the lane walks the set bits of the word: an occupied cell pulls `lim` in through its far corner
//            (LDS atomic min, shared by all lanes working for that query) and is appended to the cell list if its near corner is
//            inside `lim`;
//   phase 3  item = cell list entry: two table reads, the cell's points measured, minimum into the query's best distance / `lim`.
// `lim` (squared radius that can still matter) only ever shrinks and is always an upper bound of the answer, so the order in which
// the items run changes how much is pruned, never the result: the exact nearest distance, as before.  Pass 0 covers the query's
// brick and shell 1 (27 nodes); queries whose `lim` still reaches beyond them take shell 2 in four more passes (its 98 nodes nearest
// class first: faces, edges, corners, as nn_small_shell_pos lists them); what is still open then (nothing within ~1 m: a fifth of the
// queued queries of a loop-closure candidate) goes to a second queue for nn_fit_far_kernel, which starts at the super-bricks.
constexpr uint32_t kShellQ = 64;  // queries per tile = lanes per wavefront
constexpr uint32_t kShellNodes = 27;
constexpr uint32_t kShellBrickCap = kShellQ * kShellNodes;
constexpr uint32_t kShellCellCap = 1024;

__global__ __launch_bounds__(256) void nn_fit_shell_kernel(const NnFitnessJob* __restrict__ jobs, const uint32_t* __restrict__ job_off, double max_range,
                                                            const uint32_t* __restrict__ pend, const uint32_t* __restrict__ pend_cnt, float* __restrict__ sqd,
                                                            uint32_t* __restrict__ pend2, uint32_t* __restrict__ pend2_cnt, unsigned long long* __restrict__ stats)
{
    const uint32_t np = pend_cnt[blockIdx.y];
    if (blockIdx.x * kShellQ >= np) return;
    __shared__ float    s_t[3][kShellQ], s_q[3][kShellQ], s_nm[kShellQ];
    __shared__ int      s_c[3][kShellQ], s_smax[kShellQ], s_state[kShellQ];  // state 0: searching, 1: settled, 2: no query in this slot
    __shared__ uint32_t s_lim[kShellQ], s_best[kShellQ], s_i[kShellQ];       // float bit patterns (>= 0: ordered like unsigned integers)
    __shared__ uint16_t s_bmeta[kShellBrickCap];                             // query | node << 8
    __shared__ unsigned long long s_bword[kShellBrickCap];
    __shared__ uint32_t s_cmeta[kShellCellCap];                              // query << 24 | cell
    __shared__ float    s_clb[kShellCellCap];
    __shared__ uint32_t s_nb, s_nc;
    const NnFitnessJob& J = jobs[blockIdx.y];  // uniform: scalar loads
    const NnGridDev&    g = J.grid.level[0];
    const uint32_t      off = job_off[blockIdx.y];
    const float         E0 = g.cell, E1 = 4.0f * g.cell, mg = 4.0f * g.slack;
    const int           d1[3] = {g.bdim[0], g.bdim[1], g.bdim[2]};
    const int           gd0 = g.dim[0], gd1 = g.dim[1];
    const float         max_sq_f = max_range >= 3.0e38 ? INFINITY : static_cast<float>(max_range) * (1.0f + 1e-6f);
    const int           lane = lane_id();
    const int           w = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
    const uint64_t      below = (1ull << lane) - 1ull;
    uint32_t            n_words = 0, n_cells = 0, n_points = 0;  // diagnostics (stats != nullptr)
    NnPyramidQuery      pq;
    pq.m = mg;
    // settle(s): shells 0..s are done; a query is finished when everything beyond them is farther than lim, or there is nothing beyond
    auto settle = [&](int s) {
        if (threadIdx.x < kShellQ && s_state[threadIdx.x] == 0) {
            const float b = static_cast<float>(s) * E1 + s_nm[threadIdx.x];
            if (s_smax[threadIdx.x] <= s || b * b > __uint_as_float(s_lim[threadIdx.x]) * kNnPrune) s_state[threadIdx.x] = 1;
        }
    };
    for (uint32_t k0 = blockIdx.x * kShellQ; k0 < np; k0 += gridDim.x * kShellQ) {
        // ---- phase 0: the tile's queries
        if (threadIdx.x < kShellQ) {
            const uint32_t k = k0 + threadIdx.x;
            int state = 2;
            if (k < np) {
                const uint32_t i = as_global(pend)[off + k];
                const float4   p = load_point(J.src + i);
                float x, y, z;
                transform_point(J.T12, p.x, p.y, p.z, x, y, z);
                const float bound = sqd[off + i];  // what the block gave (INFINITY: nothing)
                int c[3];
                nn_cell_of(g, x, y, z, c);  // queued queries are finite
                const float t[3] = {x - g.origin[0], y - g.origin[1], z - g.origin[2]};
                float nm = INFINITY;
                int   smax = 0;
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    const int   b = c[a] >> 2;
                    const float lo = static_cast<float>(b) * E1;
                    nm = fminf(nm, fminf(t[a] - lo, lo + E1 - t[a]));
                    smax = max(smax, max(b, d1[a] - 1 - b));
                    s_t[a][threadIdx.x] = t[a];
                    s_c[a][threadIdx.x] = c[a];
                }
                s_q[0][threadIdx.x] = x;
                s_q[1][threadIdx.x] = y;
                s_q[2][threadIdx.x] = z;
                s_nm[threadIdx.x] = fmaxf(nm - mg, 0.0f);
                s_smax[threadIdx.x] = smax;
                s_lim[threadIdx.x] = __float_as_uint(fminf(max_sq_f, bound));
                s_best[threadIdx.x] = __float_as_uint(bound);
                s_i[threadIdx.x] = i;
                state = 0;
            }
            s_state[threadIdx.x] = state;
        }
        if (threadIdx.x == 0) { s_nb = 0; s_nc = 0; }
        __syncthreads();
        for (int pass = 0; pass < 5; ++pass) {
            if (pass == 1) settle(1);
            if (pass >= 1 && !__syncthreads_or(threadIdx.x < kShellQ && s_state[threadIdx.x] == 0)) break;
            const int first = pass * static_cast<int>(kShellNodes), last = min(first + static_cast<int>(kShellNodes), 125);
            // ---- phase 1: lane = query, node uniform per wavefront
            {
                const bool act = s_state[lane] == 0;
                const int  b[3] = {s_c[0][lane] >> 2, s_c[1][lane] >> 2, s_c[2][lane] >> 2};
                pq.t[0] = s_t[0][lane];
                pq.t[1] = s_t[1][lane];
                pq.t[2] = s_t[2][lane];
                const float lim = __uint_as_float(s_lim[lane]) * kNnPrune;
                const float e1 = E1 + s_nm[lane], cls = e1 * e1;  // a shell-2 node with k coordinates at +-2 is at least sqrt(k) e1 away
                constexpr int U = (kShellNodes + 3) / 4;
                unsigned long long wd[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    wd[u] = 0ull;
                    const int node = first + w + 4 * u;
                    if (node >= last) continue;  // uniform
                    int d[3];
                    nn_small_shell_pos(node, d);
                    const float kmin = node < 27 ? 0.0f : (node < 27 + 54 ? 1.0f : (node < 27 + 90 ? 2.0f : 3.0f));
                    const int   nx = b[0] + d[0], ny = b[1] + d[1], nz = b[2] + d[2];
                    if (!act || kmin * cls > lim || nx < 0 || nx >= d1[0] || ny < 0 || ny >= d1[1] || nz < 0 || nz >= d1[2]) continue;
                    float lb2, ub2;
                    pq.box(E1, nx, ny, nz, lb2, ub2);
                    if (lb2 <= lim) {
                        wd[u] = as_global(g.occ)[(static_cast<uint32_t>(nz) * d1[1] + ny) * d1[0] + nx];
                        ++n_words;
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const bool     push = wd[u] != 0ull;
                    const uint64_t m = __ballot(push);
                    if (m == 0) continue;  // uniform
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(&s_nb, static_cast<uint32_t>(__popcll(m)));
                    base = __shfl(base, 0);
                    if (push) {
                        const uint32_t slot = base + static_cast<uint32_t>(__popcll(m & below));
                        s_bmeta[slot] = static_cast<uint16_t>(static_cast<uint32_t>(lane) | static_cast<uint32_t>(first + w + 4 * u) << 8);
                        s_bword[slot] = wd[u];
                    }
                }
            }
            __syncthreads();
            // ---- phases 2 and 3, alternating while the cell list fills up
            const uint32_t nb = s_nb;
            uint32_t       e = threadIdx.x;
            bool           have = e < nb;
            uint32_t       meta = 0;
            unsigned long long bits = 0ull;
            if (have) { meta = s_bmeta[e]; bits = s_bword[e]; }
            for (;;) {
                bool full = false;
                while (!full && __ballot(have)) {  // uniform: every lane of the wavefront stays in the loop
                    bool     push = false;
                    uint32_t cm = 0;
                    float    clb = 0.0f;
                    if (have) {
                        const int q = static_cast<int>(meta & 0xffu);
                        int d[3];
                        nn_small_shell_pos(static_cast<int>(meta >> 8), d);
                        const int bit = __ffsll(bits) - 1;
                        const int c0 = s_c[0][q], c1 = s_c[1][q], c2 = s_c[2][q];
                        const int cx = ((c0 >> 2) + d[0]) * 4 + (bit & 3), cy = ((c1 >> 2) + d[1]) * 4 + ((bit >> 2) & 3), cz = ((c2 >> 2) + d[2]) * 4 + (bit >> 4);
                        pq.t[0] = s_t[0][q];
                        pq.t[1] = s_t[1][q];
                        pq.t[2] = s_t[2][q];
                        float lb2, ub2;
                        pq.box(E0, cx, cy, cz, lb2, ub2);
                        float lim = __uint_as_float(s_lim[q]);
                        const float ub = ub2 * kNnPrune;  // an occupied cell: something is no farther than its far corner
                        if (ub < lim) { atomicMin(&s_lim[q], __float_as_uint(ub)); lim = ub; }
                        const bool in_block = cx - c0 >= -1 && cx - c0 <= 1 && cy - c1 >= -1 && cy - c1 <= 1 && cz - c2 >= -1 && cz - c2 <= 1;  // the block pass did these
                        push = !in_block && lb2 <= lim * kNnPrune;
                        cm = static_cast<uint32_t>(q) << 24 | ((static_cast<uint32_t>(cz) * gd1 + cy) * gd0 + cx);
                        clb = lb2;
                    }
                    const uint64_t m = __ballot(push);
                    bool           stalled = false;
                    if (m != 0) {  // uniform
                        uint32_t base = 0;
                        if (lane == 0) base = atomicAdd(&s_nc, static_cast<uint32_t>(__popcll(m)));
                        base = __shfl(base, 0);
                        if (push) {
                            const uint32_t slot = base + static_cast<uint32_t>(__popcll(m & below));
                            if (slot < kShellCellCap) { s_cmeta[slot] = cm; s_clb[slot] = clb; }
                            else stalled = true;  // list full: this cell again after phase 3 has emptied it
                        }
                        full = base + static_cast<uint32_t>(__popcll(m)) >= kShellCellCap;
                    }
                    if (have && !stalled) {
                        bits &= bits - 1ull;
                        if (bits == 0ull) {
                            e += 256u;
                            have = e < nb;
                            if (have) { meta = s_bmeta[e]; bits = s_bword[e]; }
                        }
                    }
                }
                __syncthreads();
                const uint32_t nc = min(s_nc, kShellCellCap);
                for (uint32_t j = threadIdx.x; j < nc; j += 256u) {
                    const uint32_t cmj = s_cmeta[j];
                    const int      q = static_cast<int>(cmj >> 24);
                    const float    lim = __uint_as_float(s_lim[q]);
                    if (s_clb[j] > lim * kNnPrune) continue;  // lim has moved since the cell was listed
                    const uint32_t at = cmj & 0xffffffu;
                    const uint32_t kb = as_global(g.cell_start)[at], ke = as_global(g.cell_start)[at + 1];
                    const float    x = s_q[0][q], y = s_q[1][q], z = s_q[2][q];
                    float          dm = INFINITY;
                    for (uint32_t k = kb; k < ke; ++k) {
                        const float4 p = load_point(g.sorted + k);
                        dm = fminf(dm, sqdist3f(p.x, p.y, p.z, x, y, z));
                    }
                    ++n_cells;
                    n_points += ke - kb;
                    if (dm < lim) {
                        atomicMin(&s_best[q], __float_as_uint(dm));
                        atomicMin(&s_lim[q], __float_as_uint(dm));
                    }
                }
                __syncthreads();
                if (threadIdx.x == 0) s_nc = 0;
                if (!__syncthreads_or(have)) break;
            }
            if (threadIdx.x == 0) s_nb = 0;
            // (the barrier at the head of the next pass, or the one below, orders this reset and the lists' reuse)
        }
        __syncthreads();
        settle(2);
        // ---- results: settled queries get their distance, the others go on to the super-brick walk with what is known so far
        if (threadIdx.x < kShellQ) {  // wavefront 0
            const int   state = s_state[threadIdx.x];
            const float best = __uint_as_float(s_best[threadIdx.x]);
            const bool  on = state == 0;
            if (state == 1) sqd[off + s_i[threadIdx.x]] = static_cast<double>(best) <= max_range ? best : kFitNone;
            if (on) sqd[off + s_i[threadIdx.x]] = best;
            const uint64_t m = __ballot(on);
            if (m != 0) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(&pend2_cnt[blockIdx.y], static_cast<uint32_t>(__popcll(m)));
                base = __shfl(base, 0);
                if (on) pend2[off + base + static_cast<uint32_t>(__popcll(m & below))] = s_i[threadIdx.x];
            }
        }
        __syncthreads();
    }
    if (stats != nullptr) {
        __shared__ uint32_t s_st[3][4];
        const uint32_t a = wave_sum(n_words), b = wave_sum(n_cells), c = wave_sum(n_points);
        if (lane == 0) { s_st[0][w] = a; s_st[1][w] = b; s_st[2][w] = c; }
        __syncthreads();
        if (threadIdx.x < 3) atomicAdd(&stats[threadIdx.x], static_cast<unsigned long long>(s_st[threadIdx.x][0]) + s_st[threadIdx.x][1] + s_st[threadIdx.x][2] + s_st[threadIdx.x][3]);
    }
}

#ifndef MRGFE_FAR_GROUP
#define MRGFE_FAR_GROUP 2
#endif
constexpr int kFarGroup = MRGFE_FAR_GROUP;  // lanes per query in the far pass
// The rest of the far pass: one lane group per queued query, the occupancy pyramid walked as nn_pyramid_walk describes.  kSkipBricks:
// the queue is nn_fit_shell_kernel's — the bricks within two shells are done, the walk starts at the super-bricks (false: round 2's
// single far pass, kept for MRGFE_FIT_SHELL=0 and as the reference the shell pass is tested against).
// (Measured and dropped: two queues per job — queries the block gave a first distance from the front, queries with nothing around
// them from the back — so that a wavefront of the far pass holds walks of one kind: far 19.5 -> 19.3 ms, block 2.7 -> 2.9 ms.)
template <bool kSkipBricks>
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
        transform_point(s_job.T12, p.x, p.y, p.z, x, y, z);
        const float bound = sqd[off + i];  // what the earlier passes gave
        int32_t bpos;
        float   bd;
        nn_far_search<kFarGroup, kSkipBricks>(s_job.grid, x, y, z, sub, max_range, bound, bpos, bd);
        bd = fminf(bd, bound);
        if (sub == 0) sqd[off + i] = static_cast<double>(bd) <= max_range ? bd : kFitNone;
    }
}

constexpr uint32_t kFitSumSlice = 1024;
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

static std::atomic<int> g_fit_shell{-1};  // -1: not read yet
static int fit_shell_mode()
{
    int v = g_fit_shell.load(std::memory_order_relaxed);
    if (v < 0) { const char* e = std::getenv("MRGFE_FIT_SHELL"); v = e ? (std::atoi(e) != 0 ? 1 : 0) : 1; g_fit_shell.store(v, std::memory_order_relaxed); }
    return v;
}
int nn_set_fit_shell(int mode)
{
    if (mode == 0 || mode == 1) g_fit_shell.store(mode, std::memory_order_relaxed);
    return fit_shell_mode();
}
static int fit_stats_mode() { static const int v = [] { const char* e = std::getenv("MRGFE_FIT_STATS"); return e ? std::atoi(e) : 0; }(); return v; }

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
    constexpr uint32_t per_blk = 256u / kBlockGroup;
    // enough blocks to fill the chip many times over (the far pass is ragged), few enough that each has a few trips
    const uint32_t want = static_cast<uint32_t>(std::max<size_t>(1, (size_t(ctx->cu_count) * 128 + count - 1) / count));
    const uint32_t nblk = std::max<uint32_t>(1, std::min<uint32_t>((max_n + per_blk - 1) / per_blk, want));
    const uint32_t nblk_sum = (max_n + kFitSumSlice - 1) / kFitSumSlice;
    // scratch 9: jobs, offsets, queue lengths, counters, partial sums; 12: one float per query; 13: the two queues (10 and 11 may
    // hold the caller's clouds, see mrgfe_calc_fitness_score)
    DevBuf &dw = ctx->scratch[9], &dq = ctx->scratch[12], &dp = ctx->scratch[13];
    const size_t jobs_bytes = (sizeof(NnFitnessJob) * count + 255) & ~size_t(255);
    const size_t off_bytes = (sizeof(uint32_t) * 3 * (count + 1) + 64 + 255) & ~size_t(255);
    MRGFE_TRY(dw.ensure(jobs_bytes + off_bytes + sizeof(double) * 2 * (size_t(nblk_sum) + 1) * count));
    MRGFE_TRY(dq.ensure(sizeof(float) * total));
    MRGFE_TRY(dp.ensure(sizeof(uint32_t) * 2 * total));
    NnFitnessJob* d_jobs = dw.as<NnFitnessJob>();
    uint32_t*     d_off = reinterpret_cast<uint32_t*>(dw.as<char>() + jobs_bytes);
    uint32_t*     d_cnt = d_off + count + 1;   // queue lengths: after the block pass
    uint32_t*     d_cnt2 = d_cnt + count + 1;  // ... after the brick-shell pass
    unsigned long long* d_stats = reinterpret_cast<unsigned long long*>(dw.as<char>() + jobs_bytes + off_bytes - 64);  // 4 counters, 8-byte aligned
    double*       d_part = reinterpret_cast<double*>(dw.as<char>() + jobs_bytes + off_bytes);
    double*       d_res = d_part + 2 * size_t(nblk_sum) * count;
    uint32_t*     d_pend = dp.as<uint32_t>();
    uint32_t*     d_pend2 = d_pend + total;
    const bool    shell = fit_shell_mode() != 0, counters = fit_stats_mode() != 0;
    for (auto& e : ctx->ev_fit)
        if (!e) MRGFE_HIP_CHECK(hipEventCreate(&e));
    MRGFE_HIP_CHECK(hipMemcpyAsync(d_jobs, jobs, sizeof(NnFitnessJob) * count, hipMemcpyHostToDevice, st));
    MRGFE_HIP_CHECK(hipMemcpyAsync(d_off, off.data(), sizeof(uint32_t) * (count + 1), hipMemcpyHostToDevice, st));
    MRGFE_HIP_CHECK(hipMemsetAsync(d_cnt, 0, off_bytes - sizeof(uint32_t) * (count + 1), st));  // both queue lengths and the counters behind them
    const dim3 grid(nblk, static_cast<uint32_t>(count));
    MRGFE_HIP_CHECK(hipEventRecord(ctx->ev_fit[0], st));
    hipLaunchKernelGGL(nn_fit_block_kernel, grid, dim3(256), 0, st, d_jobs, d_off, max_range, dq.as<float>(), d_pend, d_cnt);
    MRGFE_HIP_CHECK(hipEventRecord(ctx->ev_fit[1], st));
    if (shell) {
        // 64 queries per tile: as many workgroups as the block pass has, walking the queue on their stride
        hipLaunchKernelGGL(nn_fit_shell_kernel, grid, dim3(256), 0, st, d_jobs, d_off, max_range, d_pend, d_cnt, dq.as<float>(), d_pend2, d_cnt2, counters ? d_stats : nullptr);
        MRGFE_HIP_CHECK(hipEventRecord(ctx->ev_fit[2], st));
        hipLaunchKernelGGL(nn_fit_far_kernel<true>, grid, dim3(256), 0, st, d_jobs, d_off, max_range, d_pend2, d_cnt2, dq.as<float>());
    } else {
        MRGFE_HIP_CHECK(hipEventRecord(ctx->ev_fit[2], st));
        hipLaunchKernelGGL(nn_fit_far_kernel<false>, grid, dim3(256), 0, st, d_jobs, d_off, max_range, d_pend, d_cnt, dq.as<float>());
    }
    MRGFE_HIP_CHECK(hipEventRecord(ctx->ev_fit[3], st));
    hipLaunchKernelGGL(nn_fit_sum_kernel, dim3(nblk_sum, static_cast<uint32_t>(count)), dim3(256), 0, st, d_jobs, d_off, dq.as<float>(), d_part);
    hipLaunchKernelGGL(nn_fitness_final_kernel, dim3(static_cast<uint32_t>(count)), dim3(256), 0, st, d_part, nblk_sum, d_res);
    MRGFE_HIP_CHECK(hipGetLastError());
    std::vector<double>   res(2 * count);
    std::vector<uint32_t> cnts(2 * (count + 1));
    unsigned long long    h_stats[4] = {0, 0, 0, 0};
    MRGFE_HIP_CHECK(hipMemcpyAsync(res.data(), d_res, sizeof(double) * 2 * count, hipMemcpyDeviceToHost, st));
    MRGFE_HIP_CHECK(hipMemcpyAsync(cnts.data(), d_cnt, sizeof(uint32_t) * 2 * (count + 1), hipMemcpyDeviceToHost, st));
    if (counters) MRGFE_HIP_CHECK(hipMemcpyAsync(h_stats, d_stats, sizeof(h_stats), hipMemcpyDeviceToHost, st));
    MRGFE_HIP_CHECK(hipStreamSynchronize(st));
    for (size_t j = 0; j < count; ++j)
        if (res[2 * j + 1] > 0) out[j] = res[2 * j] / res[2 * j + 1];
    FitStats& fs = ctx->fit_stats;
    float ms[3] = {0, 0, 0};
    for (int k = 0; k < 3; ++k) (void)hipEventElapsedTime(&ms[k], ctx->ev_fit[k], ctx->ev_fit[k + 1]);
    fs.ms_block = ms[0];
    fs.ms_shell = ms[1];
    fs.ms_far = ms[2];
    fs.queries = total;
    fs.queued = fs.queued_far = 0;
    for (size_t j = 0; j < count; ++j) { fs.queued += cnts[j]; fs.queued_far += shell ? cnts[count + 1 + j] : cnts[j]; }
    fs.words = h_stats[0];
    fs.cells = h_stats[1];
    fs.points = h_stats[2];
    ++fs.calls;
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
    job.pad = 0;
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
constexpr int kKnnSortMin = 12;  // candidates of one 64-wide step that beat the k-th entry: from here on sort + merge beats one-by-one insertion

// (distance, index) order of the neighbour lists
__device__ __forceinline__ bool knn_less(float ad, int32_t ai, float bd, int32_t bi) { return ad < bd || (ad == bd && ai < bi); }

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
__device__ __forceinline__ void knn_sort_merge(float& td, int32_t& ti, float d, int32_t ci)
{
#pragma unroll
    for (int size = 2; size <= 64; size <<= 1) {
        const bool up = (lane_id() & size) == 0 || size == 64;
#pragma unroll
        for (int stride = size >> 1; stride > 0; stride >>= 1) knn_cmpx(d, ci, stride, up);
    }
    const float   rd = __shfl(d, 63 - lane_id());
    const int32_t ri = __shfl(ci, 63 - lane_id());
    if (knn_less(rd, ri, td, ti)) { td = rd; ti = ri; }
#pragma unroll
    for (int stride = 32; stride > 0; stride >>= 1) knn_cmpx(td, ti, stride, true);
}

__global__ __launch_bounds__(256) void nn_knn_kernel(NnGrid2Dev g, const float4* __restrict__ q, uint32_t n, int k, int32_t* __restrict__ idx, float* __restrict__ sqd)
{
    const uint32_t i = (blockIdx.x * 256u + threadIdx.x) >> 6;
    if (i >= n) return;  // uniform per wave
    const int    lane = lane_id();
    const float4 p = q[i];
    float        td = INFINITY;     // list entry of this lane
    int32_t      ti = 0x7fffffff;
    float        kth_d = INFINITY;  // current k-th entry (uniform); (inf, max) while the list is not full
    int32_t      kth_i = 0x7fffffff;
    int          cnt = 0;
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
                const bool beats = valid && (d < kth_d || (d == kth_d && ci < kth_i));
                uint64_t   m = __ballot(beats);
                if (!again && __popcll(m) >= kKnnSortMin) {  // many at once (the first steps of a query): sort + merge
                    knn_sort_merge(td, ti, beats ? d : INFINITY, beats ? ci : 0x7fffffff);
                    if (lane >= k) { td = INFINITY; ti = 0x7fffffff; }
                    cnt = min(k, cnt + static_cast<int>(__popcll(m)));
                    if (cnt == k) { kth_d = __shfl(td, k - 1); kth_i = __shfl(ti, k - 1); }
                    m = 0;
                }
                while (m) {
                    const int src = __ffsll(static_cast<unsigned long long>(m)) - 1;
                    m &= m - 1;
                    const float   cd = __shfl(d, src);
                    const int32_t cci = __shfl(ci, src);
                    if (!(cd < kth_d || (cd == kth_d && cci < kth_i))) continue;  // the k-th entry moved since the ballot
                    if (again && __ballot(td == cd && ti == cci)) continue;       // met on a finer level already
                    const bool    mine_less = td < cd || (td == cd && ti < cci);
                    const int     pos = __popcll(__ballot(mine_less));  // sorted list: the lanes below pos hold the smaller entries
                    const float   up_d = __shfl_up(td, 1);
                    const int32_t up_i = __shfl_up(ti, 1);
                    if (lane == pos) { td = cd; ti = cci; }
                    else if (lane > pos) { td = up_d; ti = up_i; }
                    if (lane >= k) { td = INFINITY; ti = 0x7fffffff; }
                    if (cnt < k) ++cnt;
                    if (cnt == k) { kth_d = __shfl(td, k - 1); kth_i = __shfl(ti, k - 1); }
                }
            };
            const double margin = nn_face_margin(lv, c, p.x, p.y, p.z);
            int          first_ring = 0;
            if (last_ring >= 1) {
                // rings 0 and 1 together: the nine x-rows of the 3x3x3 block.  Lanes 0..8 fetch the bounds of one row each
                // (one round trip instead of nine dependent ones), the rows are then consumed as ONE list, 64 candidates
                // per step whatever the row lengths, nearest rows first.
                const int order[9] = {4, 1, 3, 5, 7, 0, 2, 6, 8};  // row j: dz = j / 3 - 1, dy = j % 3 - 1
                uint32_t  rb = 0, rl = 0;
                if (lane < 9) {
                    const int j = order[lane], zz = c[2] + j / 3 - 1, yy = c[1] + j % 3 - 1;
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
                    beg[j] = __shfl(rb, j);
                    off[j + 1] = off[j] + __shfl(rl, j);
                }
                for (uint32_t base = 0; base < off[9]; base += 64u) {
                    const uint32_t v = base + lane;
                    uint32_t       kk = 0;
#pragma unroll
                    for (int j = 0; j < 9; ++j)
                        if (v >= off[j] && v < off[j + 1]) kk = beg[j] + (v - off[j]);
                    step(v < off[9], v < off[9] ? lv.sorted[kk] : make_float4(0.0f, 0.0f, 0.0f, 0.0f));
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
                    for (int r = 0; r < 34; ++r) {
                        const uint32_t rb2 = __shfl(qb, r), rl2 = __shfl(ql, r);
                        for (uint32_t base = 0; base < rl2; base += 64u)
                            step(base + lane < rl2, base + lane < rl2 ? lv.sorted[rb2 + base + lane] : make_float4(0.0f, 0.0f, 0.0f, 0.0f));
                    }
                    first_ring = 3;
                }
            }
            nn_walk_ranges(
                lv, c, margin, last_ring,
                [&](uint32_t b, uint32_t e) {
                    for (uint32_t base = b; base < e; base += 64u) step(base + lane < e, base + lane < e ? lv.sorted[base + lane] : make_float4(0.0f, 0.0f, 0.0f, 0.0f));
                },
                [&](double bound_sq) { return cnt == k && static_cast<double>(kth_d) < bound_sq; }, first_ring);
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
}

int NnGrid::knn_device(mrgfe_ctx* ctx, const float4* d_q, size_t n, int k, int32_t* d_idx, float* d_sqd)
{
    if (!built_) { set_error("NnGrid::knn before build"); return MRGFE_ERR_STATE; }
    if (k < 1 || k > 64) { set_error("NnGrid::knn: k must be in [1, 64]"); return MRGFE_ERR_INVALID; }
    if (n == 0) return MRGFE_OK;
    if (n > (0xffffffffu >> 6)) { set_error("NnGrid::knn: too many queries"); return MRGFE_ERR_INVALID; }
    const uint32_t nn = static_cast<uint32_t>(n);
    hipLaunchKernelGGL(nn_knn_kernel, dim3((nn + 3) / 4), dim3(256), 0, ctx->stream, h_, d_q, nn, k, d_idx, d_sqd);
    MRGFE_HIP_CHECK(hipGetLastError());
    return MRGFE_OK;
}

}  // namespace mrgfe
