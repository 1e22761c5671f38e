// csrc/filters.hip — the prefilter chain of /root/reference/apps/prefiltering_component.cpp:149-151 on MI355X:
//   distance_filter (:206-229)            flag + exclusive scan + order-preserving compaction
//   pcl::VoxelGrid (:158-180)             bounding box -> voxel keys -> stable radix sort -> runs -> f32 centroids
//   pcl::RadiusOutlierRemoval (:195-198)  neighbour counts on the radix-sorted grid (cell = radius) + compaction
//   pcl::StatisticalOutlierRemoval (:189-192) k-NN mean distances on the grid, threshold from f64 statistics
// All passes are streaming / gather kernels bound by HBM and L2; nothing here is GEMM-shaped.
#include "filters.h"

#include <atomic>
#include <cmath>
#include <cstdlib>
#include <vector>

#include "bbox_device.h"
#include "cellsort.h"
#include "dev_float.h"
#include "dev_utils.h"
#include "ndt_build.h"
#include "nn_grid.h"

namespace mrgfe {

// ---- shared: order-preserving compaction -----------------------------------------------------------------------
__global__ __launch_bounds__(256) void compact_kernel(const float4* __restrict__ in, const uint32_t* __restrict__ flags, const uint32_t* __restrict__ pos, uint32_t n,
                                                       float4* __restrict__ out)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n && flags[i]) out[pos[i]] = in[i];
}

// scan flags (one problem) and compact; returns the kept count through *h_total (synchronises)
int compact_by_flags(mrgfe_ctx* ctx, const float4* d_in, uint32_t n, uint32_t* d_flags, float4* d_out, uint32_t* h_total)
{
    *h_total = 0;
    if (n == 0) return MRGFE_OK;
    hipStream_t st = ctx->stream;
    SliceTable  tab;
    tab.build(&n, 1);
    DevBuf &ds = ctx->scratch[0], &dpos = ctx->scratch[4], &dblk = ctx->scratch[8];
    MRGFE_TRY(ds.ensure(sizeof(Slice)));
    MRGFE_TRY(dpos.ensure(size_t(n) * 4));
    MRGFE_TRY(dblk.ensure(sizeof(uint32_t) * (tab.total_blks + 8)));
    MRGFE_HIP_CHECK(hipMemcpyAsync(ds.p, tab.h.data(), sizeof(Slice), hipMemcpyHostToDevice, st));
    uint32_t* d_tot = dblk.as<uint32_t>() + tab.total_blks;
    MRGFE_TRY(exclusive_scan(ctx, d_flags, dpos.as<uint32_t>(), ds.as<Slice>(), tab, dblk.as<uint32_t>(), d_tot));
    hipLaunchKernelGGL(compact_kernel, dim3((n + 255) / 256), dim3(256), 0, st, d_in, d_flags, dpos.as<uint32_t>(), n, d_out);
    MRGFE_HIP_CHECK(hipGetLastError());
    MRGFE_HIP_CHECK(hipMemcpyAsync(h_total, d_tot, 4, hipMemcpyDeviceToHost, st));
    MRGFE_HIP_CHECK(hipStreamSynchronize(st));
    return MRGFE_OK;
}

static int download(mrgfe_ctx* ctx, const void* d_src, size_t n, float* out)
{
    if (n == 0) return MRGFE_OK;
    MRGFE_HIP_CHECK(hipMemcpyAsync(out, d_src, n * 16, hipMemcpyDeviceToHost, ctx->stream));
    MRGFE_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return MRGFE_OK;
}

// ---- distance filter -------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void distance_flags_kernel(const float4* __restrict__ in, uint32_t n, double near_t, double far_t, uint32_t* __restrict__ flags)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float4 p = in[i];
    // p.getVector3fMap().norm(): float sqrt((x*x + y*y) + z*z), compared as double with strict inequalities
    const float  s = dot3f(p.x, p.x, p.y, p.y, p.z, p.z);
    const double d = static_cast<double>(sqrtf(s));
    flags[i] = (d > near_t && d < far_t) ? 1u : 0u;
}

int filter_distance_device(mrgfe_ctx* ctx, const float4* d_in, size_t n, double near_t, double far_t, float4* d_out, size_t* out_n)
{
    *out_n = 0;
    if (n == 0) return MRGFE_OK;
    const uint32_t nn = static_cast<uint32_t>(n);
    DevBuf& dfl = ctx->scratch[7];
    MRGFE_TRY(dfl.ensure(n * 4));
    hipLaunchKernelGGL(distance_flags_kernel, dim3((nn + 255) / 256), dim3(256), 0, ctx->stream, d_in, nn, near_t, far_t, dfl.as<uint32_t>());
    uint32_t kept = 0;
    MRGFE_TRY(compact_by_flags(ctx, d_in, nn, dfl.as<uint32_t>(), d_out, &kept));
    *out_n = kept;
    return MRGFE_OK;
}

// ---- voxel grid ------------------------------------------------------------------------------------------------
// one thread per voxel run: float running sums in ascending point index (the stable order), then divide by the count
__global__ __launch_bounds__(256) void voxel_centroid_kernel(const float4* __restrict__ pts, const uint32_t* __restrict__ sorted_vals, const uint32_t* __restrict__ seg_start,
                                                              uint32_t n_seg, int min_pts, float4* __restrict__ centroids, uint32_t* __restrict__ keep)
{
#pragma clang fp contract(off)
    const uint32_t s = blockIdx.x * 256u + threadIdx.x;
    if (s >= n_seg) return;
    const uint32_t b = seg_start[s], e = seg_start[s + 1];
    float sx = 0, sy = 0, sz = 0, si = 0;
    // four indices, then their four points, in flight at a time (two dependent loads per point otherwise); the additions stay in point order
    for (uint32_t k = b; k < e; k += 4) {
        uint32_t id[4];
        float4   p[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) id[u] = sorted_vals[min(k + u, e - 1)];
#pragma unroll
        for (int u = 0; u < 4; ++u) p[u] = pts[id[u]];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (k + u < e) { sx += p[u].x; sy += p[u].y; sz += p[u].z; si += p[u].w; }
    }
    const float cnt = static_cast<float>(e - b);
    centroids[s] = make_float4(sx / cnt, sy / cnt, sz / cnt, si / cnt);
    keep[s] = (e - b) >= static_cast<uint32_t>(min_pts) ? 1u : 0u;
}

int launch_voxel_centroids(mrgfe_ctx* ctx, const float4* d_pts, const uint32_t* d_sorted_vals, const uint32_t* d_seg_start, uint32_t n_seg, int min_pts, float4* d_centroids,
                           uint32_t* d_keep)
{
    if (n_seg == 0) return MRGFE_OK;
    hipLaunchKernelGGL(voxel_centroid_kernel, dim3((n_seg + 255) / 256), dim3(256), 0, ctx->stream, d_pts, d_sorted_vals, d_seg_start, n_seg, min_pts, d_centroids, d_keep);
    MRGFE_HIP_CHECK(hipGetLastError());
    return MRGFE_OK;
}

int filter_voxelgrid_device(mrgfe_ctx* ctx, const float4* d_in, size_t n, float leaf, int min_pts, float4* d_out, size_t* out_n, int* overflow)
{
    *out_n = 0;
    if (overflow) *overflow = 0;
    if (n == 0) return MRGFE_OK;
    hipStream_t st = ctx->stream;
    uint32_t    nn = static_cast<uint32_t>(n);
    SliceTable  tab;
    tab.build(&nn, 1);
    // descriptor block: slice | cloud ptr | n_valid | voxel params | leaf slice
    struct Desc { Slice sl; const float4* cp; uint32_t nv; uint32_t pad; VoxelParams vp; LeafSlice ls; };
    PinBuf& hp = ctx->pin[1];
    MRGFE_TRY(hp.ensure(sizeof(Desc) + sizeof(BBox) + 16));
    Desc* hd = hp.as<Desc>();
    std::memset(hd, 0, sizeof(Desc));
    hd->sl = tab.h[0];
    hd->cp = d_in;
    DevBuf& dd = ctx->scratch[0];
    MRGFE_TRY(dd.ensure(sizeof(Desc)));
    Desc* d_desc = dd.as<Desc>();
    MRGFE_HIP_CHECK(hipMemcpyAsync(d_desc, hd, sizeof(Desc), hipMemcpyHostToDevice, st));
    DevBuf& dbb = ctx->scratch[1];
    MRGFE_TRY(dbb.ensure(sizeof(BBox) * (tab.total_blks + 1)));
    BBox* d_part = dbb.as<BBox>();
    BBox* d_bbo = d_part + tab.total_blks;
    MRGFE_TRY(bounding_boxes(ctx, &d_desc->cp, &d_desc->sl, tab, d_part, d_bbo));
    BBox* h_bb = reinterpret_cast<BBox*>(hp.as<char>() + sizeof(Desc));
    MRGFE_HIP_CHECK(hipMemcpyAsync(h_bb, d_bbo, sizeof(BBox), hipMemcpyDeviceToHost, st));
    MRGFE_HIP_CHECK(hipStreamSynchronize(st));
    if (h_bb->n_finite == 0) return MRGFE_OK;
    int32_t max_b[3], div_b[3];
    if (voxel_params_from_bbox(*h_bb, leaf, &hd->vp, max_b, div_b) != MRGFE_OK) {
        // PCL: warn and pass the input through
        if (overflow) *overflow = 1;
        MRGFE_HIP_CHECK(hipMemcpyAsync(d_out, d_in, n * 16, hipMemcpyDeviceToDevice, st));
        MRGFE_HIP_CHECK(hipStreamSynchronize(st));
        *out_n = n;
        return MRGFE_OK;
    }
    hd->nv = h_bb->n_finite;
    MRGFE_HIP_CHECK(hipMemcpyAsync(d_desc, hd, sizeof(Desc), hipMemcpyHostToDevice, st));
    int key_bits = 1;
    while (key_bits < 32 && (uint64_t(1) << key_bits) <= hd->vp.n_cells) ++key_bits;
    DevBuf &dk = ctx->scratch[2], &dv = ctx->scratch[3], &dkt = ctx->scratch[4], &dvt = ctx->scratch[5], &dh = ctx->scratch[6], &dblk = ctx->scratch[8];
    MRGFE_TRY(dk.ensure(n * 4)); MRGFE_TRY(dv.ensure(n * 4)); MRGFE_TRY(dkt.ensure(n * 4)); MRGFE_TRY(dvt.ensure(n * 4));
    MRGFE_TRY(dh.ensure(sizeof(uint32_t) * 256 * (tab.total_blks + 1)));
    MRGFE_TRY(dblk.ensure(sizeof(uint32_t) * (tab.total_blks + 8)));
    MRGFE_TRY(ndt_launch_cellkeys(ctx, &d_desc->cp, &d_desc->sl, tab, &d_desc->vp, dk.as<uint32_t>(), dh.as<uint32_t>()));
    uint32_t *sk, *sv;
    MRGFE_TRY(radix_sort_pairs(ctx, dk.as<uint32_t>(), dv.as<uint32_t>(), dkt.as<uint32_t>(), dvt.as<uint32_t>(), &d_desc->sl, tab, key_bits, dh.as<uint32_t>(), &sk, &sv, true, true));
    uint32_t* d_tot = dblk.as<uint32_t>() + tab.total_blks;
    MRGFE_TRY(exclusive_scan_run_heads(ctx, sk, nullptr, &d_desc->sl, tab, &d_desc->nv, dblk.as<uint32_t>(), d_tot));
    uint32_t* h_tot = reinterpret_cast<uint32_t*>(hp.as<char>() + sizeof(Desc) + sizeof(BBox));
    MRGFE_HIP_CHECK(hipMemcpyAsync(h_tot, d_tot, 4, hipMemcpyDeviceToHost, st));
    MRGFE_HIP_CHECK(hipStreamSynchronize(st));
    const uint32_t V = *h_tot;
    if (V == 0) return MRGFE_OK;
    hd->ls.n_leaves = V;
    hd->ls.leaf_off = 0;
    hd->ls.seg_off = 0;
    hd->ls.n_valid = hd->nv;
    MRGFE_HIP_CHECK(hipMemcpyAsync(&d_desc->ls, &hd->ls, sizeof(LeafSlice), hipMemcpyHostToDevice, st));
    DevBuf &dseg = ctx->scratch[9], &dcent = ctx->scratch[10], &dkeep = ctx->scratch[11];
    MRGFE_TRY(dseg.ensure(sizeof(uint32_t) * (size_t(V) + 4) + sizeof(int32_t) * size_t(V)));
    MRGFE_TRY(dcent.ensure(sizeof(float4) * size_t(V)));
    MRGFE_TRY(dkeep.ensure(sizeof(uint32_t) * size_t(V)));
    uint32_t* d_seg = dseg.as<uint32_t>();
    int32_t*  d_segkey = reinterpret_cast<int32_t*>(d_seg + V + 4);
    MRGFE_TRY(ndt_launch_segments(ctx, sk, &d_desc->sl, tab, &d_desc->nv, dblk.as<uint32_t>(), &d_desc->ls, d_seg, d_segkey));
    hipLaunchKernelGGL(voxel_centroid_kernel, dim3((V + 255) / 256), dim3(256), 0, st, d_in, sv, d_seg, V, min_pts, dcent.as<float4>(), dkeep.as<uint32_t>());
    MRGFE_HIP_CHECK(hipGetLastError());
    uint32_t kept = 0;
    MRGFE_TRY(compact_by_flags(ctx, dcent.as<float4>(), V, dkeep.as<uint32_t>(), d_out, &kept));
    *out_n = kept;
    return MRGFE_OK;
}

// ---- approximate voxel grid -----------------------------------------------------------------------------------------------------------
// pcl::ApproximateVoxelGrid (downsample_method APPROX_VOXELGRID: prefiltering_component.cpp:172-175, scan_matching_odometry_component.cpp:180-183)
// is a sequential algorithm on the CPU — every point looks up a 512-entry direct-mapped history keyed by a hash of its cell, flushes the entry
// (emits its centroid) when another cell holds it, and takes it over; the occupied entries are flushed at the end — and SURVEY.md left it there
// ("order dependent, effectively unparallelisable").  It decomposes exactly:
//   * an entry only ever sees the points that hash to it, in arrival order, so the 512 entries are independent sequences — a stable sort of
//     (hash, index) pairs lays them out; inside a sequence a maximal stretch of points of ONE cell is one emitted centroid (float sums in
//     arrival order / float count: one thread per stretch, like the voxel grid's runs);
//   * a stretch is emitted when the first point of the NEXT stretch of its entry arrives, so its place in the output is the number of such
//     "flushing" points with a smaller index — an exclusive scan over the cloud in its original order — and the last stretch of every entry
//     follows behind all of those, in entry order.
// Same output, point for point and bit for bit, as the sequential loop (oracle/filters.cpp approx_voxelgrid): no sequential pass anywhere.
__device__ __forceinline__ int avg_cell(float v)
{
    const float f = floorf(v);
    if (!(f >= -2147483648.0f && f < 2147483648.0f)) return INT_MIN;  // the x86 conversion's answer out of range and for NaN (the GPU's saturates)
    return static_cast<int>(f);
}
__device__ __forceinline__ void avg_cell3(const float4& p, float inv, int c[3])
{
#pragma clang fp contract(off)
    c[0] = avg_cell(p.x * inv); c[1] = avg_cell(p.y * inv); c[2] = avg_cell(p.z * inv);
}
__global__ __launch_bounds__(256) void avg_keys_kernel(const float4* __restrict__ in, uint32_t n, float inv, uint32_t* __restrict__ keys)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    int c[3];
    avg_cell3(in[i], inv, c);
    keys[i] = (static_cast<uint32_t>(c[0]) * 7171u + static_cast<uint32_t>(c[1]) * 3079u + static_cast<uint32_t>(c[2]) * 4231u) & 511u;
}
// sorted position j starts a stretch iff its entry or its cell differs from position j - 1's; the first point of a stretch that is not its entry's first flushes
__global__ __launch_bounds__(256) void avg_heads_kernel(const float4* __restrict__ in, const uint32_t* __restrict__ sk, const uint32_t* __restrict__ sv, uint32_t n, float inv,
                                                         uint32_t* __restrict__ head, uint32_t* __restrict__ flusher)
{
    const uint32_t j = blockIdx.x * 256u + threadIdx.x;
    if (j >= n) return;
    uint32_t h = 1;
    if (j > 0) {
        const bool same_entry = sk[j] == sk[j - 1];
        int a[3], b[3];
        avg_cell3(in[sv[j]], inv, a);
        avg_cell3(in[sv[j - 1]], inv, b);
        const bool same_cell = a[0] == b[0] && a[1] == b[1] && a[2] == b[2];
        h = (same_entry && same_cell) ? 0u : 1u;
        if (h && same_entry) flusher[sv[j]] = 1u;
    }
    head[j] = h;
}
__global__ __launch_bounds__(256) void avg_segments_kernel(const uint32_t* __restrict__ head, const uint32_t* __restrict__ ord, uint32_t n, const uint32_t* __restrict__ n_runs,
                                                            uint32_t* __restrict__ seg_start)
{
    const uint32_t j = blockIdx.x * 256u + threadIdx.x;
    if (j >= n) return;
    if (head[j]) seg_start[ord[j]] = j;
    if (j + 1 == n) seg_start[*n_runs] = n;
}
// rank of every entry among the occupied ones (their last stretches close the output in entry order)
__global__ __launch_bounds__(512) void avg_entry_ranks_kernel(const uint32_t* __restrict__ sk, uint32_t n, uint32_t* __restrict__ rank)
{
    __shared__ uint32_t lds[16];
    const uint32_t b = threadIdx.x;
    auto lower = [&](uint32_t key) {
        uint32_t lo = 0, hi = n;
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (sk[mid] < key) lo = mid + 1; else hi = mid; }
        return lo;
    };
    const uint32_t occupied = lower(b + 1) > lower(b) ? 1u : 0u;
    uint32_t total;
    rank[b] = block_exclusive_scan<512>(occupied, lds, &total);
}
__global__ __launch_bounds__(256) void avg_emit_kernel(const float4* __restrict__ in, const uint32_t* __restrict__ sk, const uint32_t* __restrict__ sv, uint32_t n,
                                                        const uint32_t* __restrict__ seg_start, const uint32_t* __restrict__ totals /* runs, flushing points */,
                                                        const uint32_t* __restrict__ fpos, const uint32_t* __restrict__ entry_rank, float4* __restrict__ out)
{
#pragma clang fp contract(off)
    const uint32_t r = blockIdx.x * 256u + threadIdx.x;
    if (r >= totals[0]) return;
    const uint32_t b = seg_start[r], e = seg_start[r + 1];
    float sx = 0, sy = 0, sz = 0, si = 0;
    for (uint32_t k = b; k < e; ++k) {
        const float4 p = in[sv[k]];
        sx += p.x; sy += p.y; sz += p.z; si += p.w;
    }
    const float cnt = static_cast<float>(e - b);
    const uint32_t entry = sk[b];
    const uint32_t op = (e < n && sk[e] == entry) ? fpos[sv[e]] : totals[1] + entry_rank[entry];
    out[op] = make_float4(sx / cnt, sy / cnt, sz / cnt, si / cnt);
}

int filter_approx_voxelgrid_device(mrgfe_ctx* ctx, const float4* d_in, size_t n, float leaf, float4* d_out, size_t* out_n)
{
    *out_n = 0;
    if (n == 0) return MRGFE_OK;
    if (n > 0x7fffffffu) { set_error("approximate voxel grid: cloud too large"); return MRGFE_ERR_INVALID; }
    hipStream_t st = ctx->stream;
    uint32_t    nn = static_cast<uint32_t>(n);
    SliceTable  tab;
    tab.build(&nn, 1);
    DevBuf &ds = ctx->scratch[0], &drank = ctx->scratch[1], &dk = ctx->scratch[2], &dv = ctx->scratch[3], &dkt = ctx->scratch[4], &dvt = ctx->scratch[5], &dh = ctx->scratch[6],
           &dhead = ctx->scratch[7], &dblk = ctx->scratch[8], &dseg = ctx->scratch[9], &dfpos = ctx->scratch[10], &dflush = ctx->scratch[11], &dord = ctx->scratch[12];
    MRGFE_TRY(ds.ensure(sizeof(Slice)));
    MRGFE_TRY(drank.ensure(sizeof(uint32_t) * 512));
    MRGFE_TRY(dk.ensure(n * 4)); MRGFE_TRY(dv.ensure(n * 4)); MRGFE_TRY(dkt.ensure(n * 4)); MRGFE_TRY(dvt.ensure(n * 4));
    MRGFE_TRY(dh.ensure(sizeof(uint32_t) * 256 * (tab.total_blks + 1)));
    MRGFE_TRY(dhead.ensure(n * 4)); MRGFE_TRY(dord.ensure(n * 4)); MRGFE_TRY(dflush.ensure(n * 4)); MRGFE_TRY(dfpos.ensure(n * 4));
    MRGFE_TRY(dblk.ensure(sizeof(uint32_t) * (tab.total_blks + 8)));
    MRGFE_TRY(dseg.ensure(sizeof(uint32_t) * (n + 4)));
    MRGFE_TRY(ctx->stage_h2d(ds.p, tab.h.data(), sizeof(Slice), st));
    const float inv = 1.0f / leaf;  // inverse_leaf_size_ = Array3f::Ones() / leaf_size_
    const dim3  g256((nn + 255) / 256), b256(256);
    uint32_t*   d_tot = dblk.as<uint32_t>() + tab.total_blks;  // [0] stretches, [1] flushing points
    hipLaunchKernelGGL(avg_keys_kernel, g256, b256, 0, st, d_in, nn, inv, dk.as<uint32_t>());
    uint32_t *sk, *sv;
    MRGFE_TRY(radix_sort_pairs(ctx, dk.as<uint32_t>(), dv.as<uint32_t>(), dkt.as<uint32_t>(), dvt.as<uint32_t>(), ds.as<Slice>(), tab, 9, dh.as<uint32_t>(), &sk, &sv, true));
    MRGFE_HIP_CHECK(hipMemsetAsync(dflush.p, 0, n * 4, st));
    hipLaunchKernelGGL(avg_heads_kernel, g256, b256, 0, st, d_in, sk, sv, nn, inv, dhead.as<uint32_t>(), dflush.as<uint32_t>());
    MRGFE_TRY(exclusive_scan(ctx, dhead.as<uint32_t>(), dord.as<uint32_t>(), ds.as<Slice>(), tab, dblk.as<uint32_t>(), d_tot));
    hipLaunchKernelGGL(avg_segments_kernel, g256, b256, 0, st, dhead.as<uint32_t>(), dord.as<uint32_t>(), nn, d_tot, dseg.as<uint32_t>());
    MRGFE_TRY(exclusive_scan(ctx, dflush.as<uint32_t>(), dfpos.as<uint32_t>(), ds.as<Slice>(), tab, dblk.as<uint32_t>(), d_tot + 1));
    hipLaunchKernelGGL(avg_entry_ranks_kernel, dim3(1), dim3(512), 0, st, sk, nn, drank.as<uint32_t>());
    hipLaunchKernelGGL(avg_emit_kernel, g256, b256, 0, st, d_in, sk, sv, nn, dseg.as<uint32_t>(), d_tot, dfpos.as<uint32_t>(), drank.as<uint32_t>(), d_out);
    MRGFE_HIP_CHECK(hipGetLastError());
    uint32_t runs = 0;
    MRGFE_HIP_CHECK(hipMemcpyAsync(&runs, d_tot, 4, hipMemcpyDeviceToHost, st));
    MRGFE_HIP_CHECK(hipStreamSynchronize(st));
    *out_n = runs;
    return MRGFE_OK;
}

// ---- radius outlier removal ------------------------------------------------------------------------------------
int filter_radius_outlier_device(mrgfe_ctx* ctx, const float4* d_in, size_t n, double radius, int min_neighbors, float4* d_out, size_t* out_n)
{
    *out_n = 0;
    if (n == 0) return MRGFE_OK;
    NnGrid& grid = ctx_tmp_grid(ctx);
    int st = grid.build(ctx, d_in, n, static_cast<float>(radius), NnGrid::kCrowding1nn, 1);
    if (st == MRGFE_OK) {
        DevBuf& dfl = ctx->scratch[7];
        st = dfl.ensure(n * 4);
        // inlier iff #{q: (double)sqdist <= radius*radius} >= min_neighbors + 1 (the point itself counts)
        if (st == MRGFE_OK) st = grid.radius_count_flags(ctx, d_in, n, radius * radius, min_neighbors + 1, dfl.as<uint32_t>());
        uint32_t kept = 0;
        if (st == MRGFE_OK) st = compact_by_flags(ctx, d_in, static_cast<uint32_t>(n), dfl.as<uint32_t>(), d_out, &kept);
        *out_n = kept;
    }
    return st;
}

// ---- statistical outlier removal -------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sor_mean_dist_kernel(const float* __restrict__ sqd, uint32_t n, int k1, float* __restrict__ dist, uint32_t* __restrict__ valid)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float* row = sqd + size_t(i) * k1;
    float  d = 0.0f;
    uint32_t ok = 0;
    if (row[k1 - 1] >= 0.0f) {  // all mean_k + 1 neighbours found
        double sum = 0.0;
        for (int j = 1; j < k1; ++j) sum += static_cast<double>(sqrtf(row[j]));  // k = 0 is the query point
        d = static_cast<float>(sum / static_cast<double>(k1 - 1));
        ok = 1;
    }
    dist[i] = d;
    valid[i] = ok;
}

__global__ __launch_bounds__(256) void sor_flags_kernel(const float* __restrict__ dist, uint32_t n, double thr, uint32_t* __restrict__ flags)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) flags[i] = (static_cast<double>(dist[i]) > thr) ? 0u : 1u;
}

int filter_statistical_outlier_device(mrgfe_ctx* ctx, const float4* d_in, size_t n, int mean_k, double stddev_mul, float4* d_out, size_t* out_n)
{
    *out_n = 0;
    if (n == 0) return MRGFE_OK;
    hipStream_t    st = ctx->stream;
    const uint32_t nn = static_cast<uint32_t>(n);
    const int      k1 = mean_k + 1;
    NnGrid& grid = ctx_tmp_grid(ctx);
    int     rc = grid.build(ctx, d_in, n, 1.0f, NnGrid::kCrowdingKnn);
    if (rc != MRGFE_OK) return rc;
    DevBuf knn_i, knn_d, ddist;
    auto cleanup = [&]() { knn_i.release(); knn_d.release(); ddist.release(); };
    rc = knn_i.ensure(n * k1 * 4);
    if (rc == MRGFE_OK) rc = knn_d.ensure(n * k1 * 4);
    if (rc == MRGFE_OK) rc = ddist.ensure(n * 8);
    if (rc == MRGFE_OK) rc = grid.knn_device(ctx, d_in, n, k1, knn_i.as<int32_t>(), knn_d.as<float>());
    if (rc != MRGFE_OK) { cleanup(); return rc; }
    float*    d_dist = ddist.as<float>();
    uint32_t* d_valid = reinterpret_cast<uint32_t*>(d_dist + n);
    hipLaunchKernelGGL(sor_mean_dist_kernel, dim3((nn + 255) / 256), dim3(256), 0, st, knn_d.as<float>(), nn, k1, d_dist, d_valid);
    // PCL's mean / variance: sequential f64 sums over the float distances (float square) - done on the host in that order
    std::vector<float>    h_dist(n);
    std::vector<uint32_t> h_valid(n);
    if (hipMemcpyAsync(h_dist.data(), d_dist, n * 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipMemcpyAsync(h_valid.data(), d_valid, n * 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) {
        cleanup();
        set_error("statistical outlier: device to host copy failed");
        return MRGFE_ERR_HIP;
    }
    double sum = 0, sq_sum = 0;
    size_t valid = 0;
    for (size_t i = 0; i < n; ++i) {
        const float d2 = h_dist[i] * h_dist[i];
        sum += h_dist[i];
        sq_sum += d2;
        valid += h_valid[i];
    }
    const double mean = sum / static_cast<double>(valid);
    const double variance = (sq_sum - sum * sum / static_cast<double>(valid)) / (static_cast<double>(valid) - 1);
    const double thr = mean + stddev_mul * std::sqrt(variance);
    DevBuf& dfl = ctx->scratch[7];
    rc = dfl.ensure(n * 4);
    if (rc == MRGFE_OK) {
        hipLaunchKernelGGL(sor_flags_kernel, dim3((nn + 255) / 256), dim3(256), 0, st, d_dist, nn, thr, dfl.as<uint32_t>());
        uint32_t kept = 0;
        rc = compact_by_flags(ctx, d_in, nn, dfl.as<uint32_t>(), d_out, &kept);
        *out_n = kept;
    }
    cleanup();
    return rc;
}

// ---- host-pointer wrappers (the C ABI) -------------------------------------------------------------------------
template <class F>
static int host_wrap(mrgfe_ctx* ctx, const float* xyzi, size_t n, size_t stride, float* out, size_t* out_n, F&& f)
{
    *out_n = 0;
    if (n == 0) return MRGFE_OK;
    DevBuf din, dout;  // per-call buffers: the scratch slots are all in use by the algorithms
    int rc = din.ensure(n * 16);
    if (rc == MRGFE_OK) rc = dout.ensure(n * 16);
    if (rc == MRGFE_OK) rc = upload_cloud(ctx, xyzi, n, stride, din.p);
    size_t m = 0;
    if (rc == MRGFE_OK) rc = f(din.as<float4>(), dout.as<float4>(), &m);
    if (rc == MRGFE_OK) rc = download(ctx, dout.p, m, out);
    if (rc == MRGFE_OK) *out_n = m;
    din.release();
    dout.release();
    return rc;
}

// ---- the chain with its counts on the device (round 4) --------------------------------------------------------------------------------
// distance filter -> VoxelGrid -> RadiusOutlierRemoval used to hand every intermediate point COUNT to the host, which sized the next stage's
// tables: 8 host waits and ~18 small copies per scan for 0.45 ms of kernel time (CHANGELOG.md (rounds 1 - 4 notes, §10.6): 1.0 - 1.16 ms per 132k-point scan).  Here
// the counts stay where they are produced: the slices, voxel parameters and leaf slices the batched primitives read are members of ONE state
// record in device memory (PfState), rewritten between the stages by single-thread kernels that repeat the host's arithmetic float for float;
// every launch is sized for the input size and skips what lies beyond the device-side count; the radius filter's grid sizes itself
// (nn_build_device_driven).  The host waits once, reads five counts and an anomaly word from pinned memory, and only then learns how many
// points came out.  Anything unusual (no finite point, PCL's "leaf size too small" pass-through, a grid beyond its table) sets a bit in that
// word and the call is repeated through the host-driven chain below: same kernels for the arithmetic, so the outputs are the same bits
// either way (tests/test_gpu_filters.py holds the two against each other and the oracle).
struct PfState {
    const float4* cp[2];  // cloud read by the voxel grid / by the radius filter
    Slice       sl_in;    // points after the distance filter
    Slice       sl_vox;   // voxel runs (their centroids are compacted by the min-points flags)
    Slice       sl_rad;   // points after the voxel grid
    uint32_t    nv, pad0;
    VoxelParams vp;
    LeafSlice   ls;
    uint32_t    counts[6];  // after the distance filter, finite among them, voxels, after the voxel grid, after the radius filter
    uint32_t    anomaly;
    uint32_t    pad1;
    uint32_t    bb_n[2];    // partial bounding boxes (one per tile of the producer's INPUT) of the cloud the voxel grid / the radius filter reads
};
constexpr uint32_t kPfAnomalyEmpty = 1u, kPfAnomalyOverflow = 2u, kPfAnomalyNoVoxel = 4u;

__device__ __forceinline__ Slice pf_slice(uint32_t n)
{
    Slice s;
    s.n = n; s.off = 0; s.blk_off = 0; s.nblk = (n + kTile - 1) / kTile;
    return s;
}
__device__ __forceinline__ void pf_state_init(PfState* __restrict__ st, const float4* cp0, const float4* cp1, uint32_t n_in)
{
    PfState s;
    memset(&s, 0, sizeof(s));
    s.cp[0] = cp0;
    s.cp[1] = cp1;
    s.sl_in = pf_slice(n_in);
    s.counts[0] = n_in;
    s.bb_n[0] = s.sl_in.nblk;
    *st = s;
}
__global__ void pf_init_kernel(PfState* __restrict__ st, const float4* cp0, const float4* cp1, uint32_t n_in)
{
    if (threadIdx.x || blockIdx.x) return;
    pf_state_init(st, cp0, cp1, n_in);
}
// The distance filter of the chain: flags of a tile of 2048 points and the tile's count of kept points (what scan_tile_sum_kernel would
// count in a launch of its own); workgroup 0 starts the chain's state first.
__global__ __launch_bounds__(256) void pf_distance_tiles_kernel(const float4* __restrict__ in, uint32_t n, double near_t, double far_t, uint32_t* __restrict__ flags,
                                                                 uint32_t* __restrict__ blk, PfState* __restrict__ st, const float4* cp0, const float4* cp1)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) pf_state_init(st, cp0, cp1, n);
    const uint32_t base = blockIdx.x * kTile;
    uint32_t cnt = 0;
#pragma unroll
    for (int k = 0; k < kTile / 256; ++k) {
        const uint32_t i = base + k * 256 + threadIdx.x;
        if (i < n) {
            const float4 p = in[i];
            // (distance_flags_kernel's test)
            const float  s = dot3f(p.x, p.x, p.y, p.y, p.z, p.z);
            const double d = static_cast<double>(sqrtf(s));
            const uint32_t f = (d > near_t && d < far_t) ? 1u : 0u;
            flags[i] = f;
            cnt += f;
        }
    }
    __shared__ uint32_t sw[4];
    cnt = wave_sum(cnt);
    if (lane_id() == 0) sw[wave_id()] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) blk[blockIdx.x] = sw[0] + sw[1] + sw[2] + sw[3];
}
// Order-preserving compaction of the chain in ONE launch behind the tile counts: a workgroup adds up the counts of the tiles before its own
// (and of all tiles) itself, ranks its tile's kept elements by ballots and moves them; workgroup 0 records the kept count in the chain's state, and (kWhich 0 and 1) every workgroup leaves the bounding box of what it kept for the stage that reads the output — the
// exclusive scan (three launches), the compaction, the count and the bounding-box pass were six launches.
//   kWhich 0: the distance filter (n known to the host)      -> sl_in,  counts[0], boxes for the voxel grid
//   kWhich 1: the voxel grid's centroids (n = sl_vox.n)       -> sl_rad, counts[3], boxes for the radius filter's grid
//   kWhich 2: the radius filter (n = sl_rad.n)               -> counts[4] and the host's status words
template <int kWhich>
__global__ __launch_bounds__(256) void pf_compact_kernel(const float4* __restrict__ in, const uint32_t* __restrict__ flags, const uint32_t* __restrict__ blk, uint32_t n_host,
                                                          float4* __restrict__ out, PfState* st, uint32_t* __restrict__ h_status, BBox* __restrict__ partial)
{
    const uint32_t n = kWhich == 0 ? n_host : (kWhich == 1 ? st->sl_vox.n : st->sl_rad.n);
    const uint32_t nblk = (n + kTile - 1) / kTile;
    if (blockIdx.x >= nblk && blockIdx.x != 0) return;  // (workgroup 0 reports the count of an empty input too)
    __shared__ uint32_t s_red[2][4];
    __shared__ uint32_t s_cnt[kTile / 256][4], s_off[kTile / 256][4];
    uint32_t before = 0, total = 0;
    for (uint32_t b = threadIdx.x; b < nblk; b += 256) {
        const uint32_t v = blk[b];
        total += v;
        before += b < blockIdx.x ? v : 0u;
    }
    before = wave_sum(before);
    total = wave_sum(total);
    const int lane = lane_id(), w = wave_id();
    if (lane == 0) { s_red[0][w] = before; s_red[1][w] = total; }
    __syncthreads();
    before = s_red[0][0] + s_red[0][1] + s_red[0][2] + s_red[0][3];
    total = s_red[1][0] + s_red[1][1] + s_red[1][2] + s_red[1][3];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (kWhich == 0) { st->sl_in = pf_slice(total); st->counts[0] = total; st->bb_n[0] = nblk; }
        else if (kWhich == 1) { st->sl_rad = pf_slice(total); st->counts[3] = total; st->bb_n[1] = nblk; }
        else {
            st->counts[4] = total;
            for (int k = 0; k < 5; ++k) h_status[k] = st->counts[k];
            h_status[5] = st->anomaly;
            __threadfence_system();
            h_status[6] = 0x600df00du;  // written last: the record is complete (the host reads it behind its one wait for the stream)
        }
    }
    if (blockIdx.x >= nblk) return;
    // rank of a kept element = kept elements of earlier rounds and wavefronts of the tile + kept lanes below its own
    const uint32_t base = blockIdx.x * kTile;
    uint32_t f[kTile / 256], lr[kTile / 256];
#pragma unroll
    for (int k = 0; k < kTile / 256; ++k) {
        const uint32_t i = base + k * 256 + threadIdx.x;
        f[k] = i < n ? flags[i] : 0u;
        const uint64_t m = __ballot(f[k] != 0u);
        lr[k] = static_cast<uint32_t>(__popcll(m & ((1ull << lane) - 1ull)));
        if (lane == 0) s_cnt[k][w] = static_cast<uint32_t>(__popcll(m));
    }
    __syncthreads();
    if (threadIdx.x < (kTile / 256) * 4) {  // 32 lanes of wavefront 0: exclusive prefix in (round, wavefront) order
        const uint32_t v = s_cnt[threadIdx.x / 4][threadIdx.x % 4];
        uint32_t incl = v;
#pragma unroll
        for (int off = 1; off < (kTile / 256) * 4; off <<= 1) {
            const uint32_t t = __shfl_up(incl, off, kWave);
            if (lane >= off) incl += t;
        }
        s_off[threadIdx.x / 4][threadIdx.x % 4] = incl - v;
    }
    __syncthreads();
    BoxAcc acc;
    acc.init();
#pragma unroll
    for (int k = 0; k < kTile / 256; ++k) {
        if (f[k]) {
            const float4 p = in[base + k * 256 + threadIdx.x];
            out[before + s_off[k][w] + lr[k]] = p;
            if (kWhich != 2) acc.add(p);
        }
    }
    if (kWhich != 2) {
        const BBox b = block_merge_box(acc);
        if (threadIdx.x == 0) partial[blockIdx.x] = b;
    }
}
// voxel_params_from_bbox (ndt_engine.cpp) on the device, float for float
__global__ __launch_bounds__(256) void pf_voxel_params_kernel(PfState* __restrict__ st, const BBox* __restrict__ partial, float leaf)
{
#pragma clang fp contract(off)
    const BBox bb = block_merge_partials(partial, st->bb_n[0]);  // (the boxes the producer of the cloud left, one per tile of its input)
    if (threadIdx.x) return;
    st->counts[1] = bb.n_finite;
    VoxelParams vp;
    memset(&vp, 0, sizeof(vp));
    if (bb.n_finite == 0) { st->anomaly |= kPfAnomalyEmpty; st->sl_in = pf_slice(0); st->vp = vp; st->nv = 0; return; }
    const float   inv_leaf = 1.0f / leaf;
    const int64_t dx = static_cast<int64_t>((bb.mx[0] - bb.mn[0]) * inv_leaf) + 1;
    const int64_t dy = static_cast<int64_t>((bb.mx[1] - bb.mn[1]) * inv_leaf) + 1;
    const int64_t dz = static_cast<int64_t>((bb.mx[2] - bb.mn[2]) * inv_leaf) + 1;
    bool over = dx * dy * dz > static_cast<int64_t>(INT32_MAX);
    int32_t div_b[3] = {1, 1, 1};
    if (!over) {
        for (int a = 0; a < 3; ++a) {
            vp.min_b[a] = static_cast<int32_t>(floorf(bb.mn[a] * inv_leaf));
            div_b[a] = static_cast<int32_t>(floorf(bb.mx[a] * inv_leaf)) - vp.min_b[a] + 1;
        }
        vp.divb_mul[0] = 1;
        vp.divb_mul[1] = div_b[0];
        vp.divb_mul[2] = div_b[0] * div_b[1];
        vp.inv_leaf = inv_leaf;
        const int64_t cells = static_cast<int64_t>(div_b[0]) * div_b[1] * div_b[2];
        over = cells > static_cast<int64_t>(INT32_MAX);
        vp.n_cells = static_cast<uint32_t>(cells);
    }
    if (over) { st->anomaly |= kPfAnomalyOverflow; st->sl_in = pf_slice(0); memset(&vp, 0, sizeof(vp)); st->vp = vp; st->nv = 0; return; }
    st->vp = vp;
    st->nv = bb.n_finite;
}
// blk: the tiles' run-head counts (run_head_tile_counts) -> their exclusive prefix, in place (scan_tiles_kernel's work), and the run count into the state
__global__ __launch_bounds__(256) void pf_leaves_kernel(PfState* __restrict__ st, uint32_t* __restrict__ blk)
{
    __shared__ uint32_t lds[8];
    const uint32_t nblk = st->sl_in.nblk;
    uint32_t carry = 0;
    for (uint32_t b0 = 0; b0 < nblk; b0 += 256) {
        const uint32_t b = b0 + threadIdx.x;
        const uint32_t v = b < nblk ? blk[b] : 0u;
        uint32_t total;
        const uint32_t ex = block_exclusive_scan<256>(v, lds, &total);
        if (b < nblk) blk[b] = carry + ex;
        carry += total;
    }
    if (threadIdx.x) return;
    const uint32_t V = st->anomaly ? 0u : carry;
    LeafSlice ls;
    memset(&ls, 0, sizeof(ls));
    ls.n_leaves = V;
    ls.n_valid = st->nv;
    st->ls = ls;
    st->sl_vox = pf_slice(V);
    st->counts[2] = V;
    if (V == 0) st->anomaly |= kPfAnomalyNoVoxel;
}
__global__ __launch_bounds__(256) void voxel_centroid_dd_kernel(const float4* __restrict__ pts, const uint32_t* __restrict__ sorted_vals, const uint32_t* __restrict__ seg_start,
                                                                 const Slice* __restrict__ runs, int min_pts, float4* __restrict__ centroids, uint32_t* __restrict__ keep)
{
#pragma clang fp contract(off)
    const uint32_t s = blockIdx.x * 256u + threadIdx.x;
    if (s >= runs->n) return;
    const uint32_t b = seg_start[s], e = seg_start[s + 1];
    float sx = 0, sy = 0, sz = 0, si = 0;
    // four indices, then their four points, in flight at a time (two dependent loads per point otherwise); the additions stay in point order
    for (uint32_t k = b; k < e; k += 4) {
        uint32_t id[4];
        float4   p[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) id[u] = sorted_vals[min(k + u, e - 1)];
#pragma unroll
        for (int u = 0; u < 4; ++u) p[u] = pts[id[u]];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (k + u < e) { sx += p[u].x; sy += p[u].y; sz += p[u].z; si += p[u].w; }
    }
    const float cnt = static_cast<float>(e - b);
    centroids[s] = make_float4(sx / cnt, sy / cnt, sz / cnt, si / cnt);
    keep[s] = (e - b) >= static_cast<uint32_t>(min_pts) ? 1u : 0u;
}
// 1 (default): the usual chain keeps its counts on the device; 0: always the host-driven chain of round 3 (MRGFE_PREFILTER_HOST_DRIVEN=1, tests)
static std::atomic<int> g_pf_device_driven{-1};
static int prefilter_device_driven_mode()
{
    int v = g_pf_device_driven.load(std::memory_order_relaxed);
    if (v < 0) { v = std::getenv("MRGFE_PREFILTER_HOST_DRIVEN") ? 0 : 1; g_pf_device_driven.store(v, std::memory_order_relaxed); }
    return v;
}
int prefilter_set_device_driven(int mode)
{
    if (mode == 0 || mode == 1) g_pf_device_driven.store(mode, std::memory_order_relaxed);
    return prefilter_device_driven_mode();
}

static NnDeviceDrivenGrid& pf_grid(mrgfe_ctx* ctx)
{
    if (!ctx->pf_grid) ctx->pf_grid = new NnDeviceDrivenGrid();
    return *static_cast<NnDeviceDrivenGrid*>(ctx->pf_grid);
}

// returns MRGFE_OK with *used = true when the device-driven chain produced the output, *used = false when the caller has to run the host-driven one
static int filter_chain_device_driven(mrgfe_ctx* ctx, const PrefilterChain& ch, const float4* d_in, uint32_t n, float4* d_work, float4* d_final, size_t* out_n, bool* used)
{
    *used = false;
    if (!prefilter_device_driven_mode() || ch.approx_voxelgrid || !ch.voxelgrid || ch.outlier != 1 || n == 0 || !(ch.leaf > 0)) return MRGFE_OK;
    hipStream_t st = ctx->stream;
    SliceTable  tab;
    tab.build(&n, 1);
    MRGFE_TRY(ctx->pf_state.ensure(sizeof(PfState)));
    MRGFE_TRY(ctx->pf_status.ensure(64));
    PfState*  d_st = ctx->pf_state.as<PfState>();
    uint32_t* h_status = ctx->pf_status.as<uint32_t>();
    h_status[6] = 0;
    DevBuf &dbb = ctx->scratch[1], &dk = ctx->scratch[2], &dv = ctx->scratch[3], &dkt = ctx->scratch[4], &dvt = ctx->scratch[5], &dh = ctx->scratch[6], &dfl = ctx->scratch[7],
           &dblk = ctx->scratch[8], &dseg = ctx->scratch[9], &dcent = ctx->scratch[10], &dkeep = ctx->scratch[11];
    MRGFE_TRY(dbb.ensure(sizeof(BBox) * (tab.total_blks + 2)));
    MRGFE_TRY(dk.ensure(size_t(n) * 4)); MRGFE_TRY(dv.ensure(size_t(n) * 4)); MRGFE_TRY(dkt.ensure(size_t(n) * 4)); MRGFE_TRY(dvt.ensure(size_t(n) * 4));
    MRGFE_TRY(dh.ensure(sizeof(uint32_t) * 256 * (tab.total_blks + 1)));
    MRGFE_TRY(dfl.ensure(size_t(n) * 4));
    MRGFE_TRY(dblk.ensure(sizeof(uint32_t) * (tab.total_blks + 8)));
    MRGFE_TRY(dseg.ensure(sizeof(uint32_t) * (size_t(n) + 4) + sizeof(int32_t) * size_t(n)));
    MRGFE_TRY(dcent.ensure(sizeof(float4) * size_t(n)));
    MRGFE_TRY(dkeep.ensure(sizeof(uint32_t) * size_t(n)));
    const dim3 g256((n + 255) / 256), b256(256);
    BBox*      d_part = dbb.as<BBox>();  // one box per tile of the stage that made the cloud
    // ---- distance filter: flags + tile counts (and the state's first values), compaction into the work buffer with the count and the kept points' boxes
    const float4* vox_in = ch.distance ? d_work : d_in;
    const dim3    gtiles(std::max<uint32_t>(1, tab.total_blks));
    uint32_t*     d_blk = dblk.as<uint32_t>();
    if (ch.distance) {
        hipLaunchKernelGGL(pf_distance_tiles_kernel, gtiles, b256, 0, st, d_in, n, ch.near_t, ch.far_t, dfl.as<uint32_t>(), d_blk, d_st, vox_in, d_final);
        hipLaunchKernelGGL(pf_compact_kernel<0>, gtiles, b256, 0, st, d_in, dfl.as<uint32_t>(), d_blk, n, d_work, d_st, h_status, d_part);
    } else {
        hipLaunchKernelGGL(pf_init_kernel, dim3(1), dim3(1), 0, st, d_st, vox_in, d_final, n);
        MRGFE_TRY(bounding_box_partials(ctx, &d_st->cp[0], &d_st->sl_in, tab, d_part));
    }
    // ---- VoxelGrid: voxel parameters from the boxes (device), keys, stable sort (32 key bits: the host does not know the cell count), runs, centroids
    hipLaunchKernelGGL(pf_voxel_params_kernel, dim3(1), b256, 0, st, d_st, d_part, ch.leaf);
    MRGFE_TRY(ndt_launch_cellkeys(ctx, &d_st->cp[0], &d_st->sl_in, tab, &d_st->vp, dk.as<uint32_t>(), dh.as<uint32_t>()));
    uint32_t *sk, *sv;
    MRGFE_TRY(radix_sort_pairs(ctx, dk.as<uint32_t>(), dv.as<uint32_t>(), dkt.as<uint32_t>(), dvt.as<uint32_t>(), &d_st->sl_in, tab, 32, dh.as<uint32_t>(), &sk, &sv, true, true));
    MRGFE_TRY(run_head_tile_counts(ctx, sk, &d_st->sl_in, tab, &d_st->nv, d_blk));
    hipLaunchKernelGGL(pf_leaves_kernel, dim3(1), b256, 0, st, d_st, d_blk);
    uint32_t* d_seg = dseg.as<uint32_t>();
    int32_t*  d_segkey = reinterpret_cast<int32_t*>(d_seg + size_t(n) + 4);
    MRGFE_TRY(ndt_launch_segments(ctx, sk, &d_st->sl_in, tab, &d_st->nv, d_blk, &d_st->ls, d_seg, d_segkey));
    hipLaunchKernelGGL(voxel_centroid_dd_kernel, g256, b256, 0, st, vox_in, sv, d_seg, &d_st->sl_vox, ch.min_pts, dcent.as<float4>(), dkeep.as<uint32_t>());
    MRGFE_TRY(tile_sums(ctx, dkeep.as<uint32_t>(), &d_st->sl_vox, tab, d_blk));
    // (the voxel grid's output goes to d_final for now: the radius filter reads it there and compacts into the work buffer ...)
    hipLaunchKernelGGL(pf_compact_kernel<1>, gtiles, b256, 0, st, dcent.as<float4>(), dkeep.as<uint32_t>(), d_blk, 0u, d_final, d_st, h_status, d_part);
    // ---- RadiusOutlierRemoval: a grid that sizes itself (from the boxes of the centroids kept), neighbour counts, compaction
    NnDeviceDrivenGrid& grid = pf_grid(ctx);
    const float cell = static_cast<float>(ch.radius);
    BBox* h_box = reinterpret_cast<BBox*>(h_status + 8);  // the box of the voxel centroids: it encloses what the outlier filter keeps of them
    MRGFE_TRY(nn_build_device_driven(ctx, &d_st->cp[1], &d_st->sl_rad, n, d_part, &d_st->bb_n[1], cell, 1u << 22, grid, &d_st->anomaly, h_box));
    // inlier iff #{q: (double)sqdist <= radius*radius} >= min_neighbors + 1 (the point itself counts)
    MRGFE_TRY(nn_radius_flags_device_driven(ctx, grid, d_final, &d_st->sl_rad, n, ch.radius * ch.radius, ch.radius_min_neighbors + 1, cell, dfl.as<uint32_t>()));
    MRGFE_TRY(tile_sums(ctx, dfl.as<uint32_t>(), &d_st->sl_rad, tab, d_blk));
    hipLaunchKernelGGL(pf_compact_kernel<2>, gtiles, b256, 0, st, d_final, dfl.as<uint32_t>(), d_blk, 0u, d_work, d_st, h_status, d_part);
    MRGFE_HIP_CHECK(hipGetLastError());
    MRGFE_HIP_CHECK(hipStreamSynchronize(st));  // the chain's one wait
    if (h_status[6] != 0x600df00du) { set_error("prefilter: the device-driven chain did not report"); return MRGFE_ERR_HIP; }
    if (h_status[5] != 0) return MRGFE_OK;  // something unusual: the host-driven chain decides what the reference does with it
    *out_n = h_status[4];
    *used = true;
    if (h_box->n_finite > 0 && *out_n > 0) {  // (remembered for mrgfe_reg_set_source_from_prefilter; filter_chain says whether d_work is the caller's buffer)
        for (int a = 0; a < 3; ++a) { ctx->pf_out_box[a] = h_box->mn[a]; ctx->pf_out_box[3 + a] = h_box->mx[a]; }
        ctx->pf_out_ptr = d_work;
        ctx->pf_out_n = *out_n;
        ctx->pf_out_valid = true;
    }
    return MRGFE_OK;  // the result is in d_work
}

int filter_chain(mrgfe_ctx* ctx, const PrefilterChain& ch, const float* xyzi, size_t n, size_t stride, void* out, size_t* out_n, bool out_on_device)
{
    *out_n = 0;
    ctx->pf_out_valid = false;
    if (n == 0) return MRGFE_OK;
    if (n > 0x7fffffffu) { set_error("prefilter: cloud too large"); return MRGFE_ERR_INVALID; }
    DevBuf &a = ctx->pf_buf[0], &b = ctx->pf_buf[1];  // ping-pong, kept between calls (grow-only: a hipMalloc / hipFree pair per scan is a device-wide wait)
    int rc = a.ensure(n * 16);
    if (rc == MRGFE_OK) rc = b.ensure(n * 16);
    if (rc == MRGFE_OK) rc = upload_cloud(ctx, xyzi, n, stride, a.p);
    if (rc == MRGFE_OK) {
        // the usual chain with its counts on the device: input a, work buffer = the caller's device buffer (capacity n) or b, result in the work buffer
        bool   used = false;
        size_t m = 0;
        float4* work = out_on_device ? static_cast<float4*>(out) : b.as<float4>();
        float4* final_buf = out_on_device ? b.as<float4>() : nullptr;
        DevBuf& c = ctx->scratch[13];
        if (!out_on_device) { rc = c.ensure(n * 16); final_buf = c.as<float4>(); }
        if (rc == MRGFE_OK) rc = filter_chain_device_driven(ctx, ch, a.as<float4>(), static_cast<uint32_t>(n), work, final_buf, &m, &used);
        if (rc != MRGFE_OK) return rc;
        if (used) {
            if (!out_on_device) ctx->pf_out_valid = false;  // (the cloud went to the host: nothing of the caller's stays on the device)
            if (!out_on_device) rc = download(ctx, work, m, static_cast<float*>(out));
            if (rc == MRGFE_OK) *out_n = m;
            return rc;
        }
    }
    float4 *cur = a.as<float4>(), *nxt = b.as<float4>();
    size_t  m = n;
    auto pass = [&](auto&& f) {
        if (rc != MRGFE_OK || m == 0) return;
        size_t k = 0;
        rc = f(cur, m, nxt, &k);
        if (rc == MRGFE_OK) { std::swap(cur, nxt); m = k; }
    };
    if (ch.distance) pass([&](const float4* i, size_t ni, float4* o, size_t* k) { return filter_distance_device(ctx, i, ni, ch.near_t, ch.far_t, o, k); });
    if (ch.approx_voxelgrid) pass([&](const float4* i, size_t ni, float4* o, size_t* k) { return filter_approx_voxelgrid_device(ctx, i, ni, ch.leaf, o, k); });
    else if (ch.voxelgrid) pass([&](const float4* i, size_t ni, float4* o, size_t* k) { int overflow = 0; return filter_voxelgrid_device(ctx, i, ni, ch.leaf, ch.min_pts, o, k, &overflow); });
    if (ch.outlier == 1) pass([&](const float4* i, size_t ni, float4* o, size_t* k) { return filter_radius_outlier_device(ctx, i, ni, ch.radius, ch.radius_min_neighbors, o, k); });
    if (ch.outlier == 2) pass([&](const float4* i, size_t ni, float4* o, size_t* k) { return filter_statistical_outlier_device(ctx, i, ni, ch.mean_k, ch.stddev_mul, o, k); });
    if (rc == MRGFE_OK) {
        if (!out_on_device) {
            rc = download(ctx, cur, m, static_cast<float*>(out));
        } else if (m && (hipMemcpyAsync(out, cur, m * 16, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess)) {
            set_error("prefilter: device copy failed");
            rc = MRGFE_ERR_HIP;
        }
    }
    if (rc == MRGFE_OK) *out_n = m;
    return rc;
}

int filter_distance(mrgfe_ctx* ctx, const float* xyzi, size_t n, size_t stride, double near_t, double far_t, float* out, size_t* out_n)
{
    return host_wrap(ctx, xyzi, n, stride, out, out_n, [&](const float4* i, float4* o, size_t* m) { return filter_distance_device(ctx, i, n, near_t, far_t, o, m); });
}
int filter_voxelgrid(mrgfe_ctx* ctx, const float* xyzi, size_t n, size_t stride, float leaf, int min_pts, float* out, size_t* out_n, int* overflow)
{
    return host_wrap(ctx, xyzi, n, stride, out, out_n, [&](const float4* i, float4* o, size_t* m) { return filter_voxelgrid_device(ctx, i, n, leaf, min_pts, o, m, overflow); });
}
int filter_approx_voxelgrid(mrgfe_ctx* ctx, const float* xyzi, size_t n, size_t stride, float leaf, float* out, size_t* out_n)
{
    return host_wrap(ctx, xyzi, n, stride, out, out_n, [&](const float4* i, float4* o, size_t* m) { return filter_approx_voxelgrid_device(ctx, i, n, leaf, o, m); });
}
int filter_radius_outlier(mrgfe_ctx* ctx, const float* xyzi, size_t n, size_t stride, double radius, int min_neighbors, float* out, size_t* out_n)
{
    return host_wrap(ctx, xyzi, n, stride, out, out_n, [&](const float4* i, float4* o, size_t* m) { return filter_radius_outlier_device(ctx, i, n, radius, min_neighbors, o, m); });
}
int filter_statistical_outlier(mrgfe_ctx* ctx, const float* xyzi, size_t n, size_t stride, int mean_k, double stddev_mul, float* out, size_t* out_n)
{
    return host_wrap(ctx, xyzi, n, stride, out, out_n, [&](const float4* i, float4* o, size_t* m) { return filter_statistical_outlier_device(ctx, i, n, mean_k, stddev_mul, o, m); });
}

}  // namespace mrgfe
