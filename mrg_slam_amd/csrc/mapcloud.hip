// csrc/mapcloud.hip — SURVEY.md §8(f) rows 2 and 4 on MI355X: streaming per-point kernels (16 B in, 16 B out per point,
// HBM bound) in front of the compaction and voxel-grid machinery of filters.hip.
//   MapCloudGenerator::generate         /root/reference/src/mrg_slam/map_cloud_generator.cpp:14-86
//   pcl::ApproximateMeanVoxelGrid       /root/reference/include/pcl/filters/ApproximateMeanVoxelGrid.hpp:63-126
//   other-robot point removal           /root/reference/apps/mrg_slam_component.cpp:396-429
//   PrefilteringComponent::deskewing    /root/reference/apps/prefiltering_component.cpp:231-292
// Float expressions run in the order documented in oracle/mapcloud.cpp (left to right, no FMA).
#include "mapcloud.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "cellsort.h"
#include "dev_float.h"
#include "dev_utils.h"
#include "filters.h"

namespace mrgfe {

// one lane per point of the concatenated keyframe clouds; the keyframe of a point is found by bisection of kf_off
// cat: the keyframe clouds back to back, or nullptr when every keyframe is read through its own pointer srcs[k]
__global__ __launch_bounds__(256) void map_transform_kernel(const float4* __restrict__ cat, const float4* const* __restrict__ srcs, uint32_t n, const uint32_t* __restrict__ kf_off,
                                                             const float* __restrict__ poses, int K, int use_far, float far_sq, float4* __restrict__ out,
                                                             uint32_t* __restrict__ flags)
{
#pragma clang fp contract(off)
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    int lo = 0, hi = K;  // kf_off[lo] <= i < kf_off[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (kf_off[mid] <= i) lo = mid; else hi = mid;
    }
    const float4 p = cat ? cat[i] : srcs[lo][i - kf_off[lo]];
    uint32_t keep = 1u;
    if (use_far) {
        float s = p.x * p.x + p.y * p.y;  // getVector3fMap().squaredNorm()
        s = s + p.z * p.z;
        if (s > far_sq) keep = 0u;        // map_cloud_generator.cpp:39-41
    }
    const float* P = poses + 16 * lo;  // column-major
    float q[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {  // pose * (x, y, z, 1), column by column
        float s = P[0 * 4 + r] * p.x;
        s = s + P[1 * 4 + r] * p.y;
        s = s + P[2 * 4 + r] * p.z;
        q[r] = s + P[3 * 4 + r] * 1.0f;
    }
    out[i] = make_float4(q[0], q[1], q[2], p.w);
    flags[i] = keep;
}

// ---- pcl::ApproximateMeanVoxelGrid on occupied cells ------------------------------------------------------------------------
// The reference keys a hash map on the integer cell (floor(p * inverse_leaf) per axis, ApproximateMeanVoxelGrid.hpp:85-91): no limit
// on the extent of the map.  A dense linear voxel index over the bounding box (what pcl::VoxelGrid uses, and the prefilter with it)
// overflows int32 at 200 m x 200 m x 30 m / 0.1 m — smaller than a KITTI map (round 1 returned MRGFE_ERR_OVERFLOW there).  Here the key
// is the cell itself, bit-packed: (iz - min) | (iy - min) | (ix - min) with just the bits each axis needs (up to 21 per axis), one
// more bit on top marking non-finite points so that they sort behind every cell.  Up to 32 key bits are one stable radix sort of
// (key, index) pairs as before; longer keys are sorted low word first, then — stably — by the gathered high word.  Runs of equal
// keys, float sums in input order, division by float(count) and the count threshold are the voxel-grid pass of filters.hip.
// Output order: ascending (z, y, x) cell — the same order the dense index gave.
struct CellKeyParams { float inv_leaf; int32_t min_c[3]; uint32_t shift_y, shift_z, total_bits; };

__global__ __launch_bounds__(256) void mapvox_keys_kernel(const float4* __restrict__ pts, uint32_t n, CellKeyParams kp, uint32_t* __restrict__ key_lo, uint32_t* __restrict__ key_hi,
                                                           uint32_t* __restrict__ vals)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float4 p = pts[i];
    uint64_t key = uint64_t(1) << kp.total_bits;  // non-finite: behind every cell
    if (finite3(p.x, p.y, p.z)) {
        const uint64_t cx = static_cast<uint64_t>(static_cast<int64_t>(static_cast<int>(floorf(p.x * kp.inv_leaf))) - kp.min_c[0]);
        const uint64_t cy = static_cast<uint64_t>(static_cast<int64_t>(static_cast<int>(floorf(p.y * kp.inv_leaf))) - kp.min_c[1]);
        const uint64_t cz = static_cast<uint64_t>(static_cast<int64_t>(static_cast<int>(floorf(p.z * kp.inv_leaf))) - kp.min_c[2]);
        key = cx | (cy << kp.shift_y) | (cz << kp.shift_z);
    }
    key_lo[i] = static_cast<uint32_t>(key);
    key_hi[i] = static_cast<uint32_t>(key >> 32);
    vals[i] = i;
}
__global__ __launch_bounds__(256) void gather_u32_kernel(const uint32_t* __restrict__ src, const uint32_t* __restrict__ idx, uint32_t n, uint32_t* __restrict__ dst)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}
// run heads of the sorted 64-bit keys among the first n_valid positions (the non-finite points sit behind them)
__global__ __launch_bounds__(256) void heads64_kernel(const uint32_t* __restrict__ key_lo, const uint32_t* __restrict__ key_hi, const uint32_t* __restrict__ sorted_vals, uint32_t n,
                                                       uint32_t n_valid, uint32_t* __restrict__ flags)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    uint32_t f = 0;
    if (i < n_valid) {
        if (i == 0) f = 1;
        else {
            const uint32_t a = sorted_vals[i], b = sorted_vals[i - 1];
            f = (key_lo[a] != key_lo[b] || key_hi[a] != key_hi[b]) ? 1u : 0u;
        }
    }
    flags[i] = f;
}
__global__ __launch_bounds__(256) void seg_from_heads_kernel(const uint32_t* __restrict__ flags, const uint32_t* __restrict__ ordinal, uint32_t n, uint32_t n_valid, uint32_t n_seg,
                                                              uint32_t* __restrict__ seg_start)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    if (flags[i]) seg_start[ordinal[i]] = i;
    if (i + 1 == n_valid) seg_start[n_seg] = n_valid;
}

int mean_voxelgrid_device(mrgfe_ctx* ctx, const float4* d_in, size_t n, float leaf, int min_pts, float4* d_out, size_t* out_n)
{
    *out_n = 0;
    if (n == 0) return MRGFE_OK;
    hipStream_t st = ctx->stream;
    uint32_t    nn = static_cast<uint32_t>(n);
    SliceTable  tab;
    tab.build(&nn, 1);
    struct Desc { Slice sl; const float4* cp; };
    PinBuf& hp = ctx->pin[1];
    MRGFE_TRY(hp.ensure(sizeof(Desc) + sizeof(BBox) + 16));
    Desc* hd = hp.as<Desc>();
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
    const uint32_t n_valid = h_bb->n_finite;
    if (n_valid == 0) return MRGFE_OK;
    // cells of the extreme points (float multiply and floor are monotonic, so these are the extreme cells)
    CellKeyParams kp;
    kp.inv_leaf = 1.0f / leaf;  // ApproximateMeanVoxelGrid::setLeafSize: inverse_leaf_size_ = 1 / leaf_size_ (float)
    uint32_t bits[3];
    for (int a = 0; a < 3; ++a) {
        const double lo = std::floor(static_cast<double>(h_bb->mn[a] * kp.inv_leaf)), hi = std::floor(static_cast<double>(h_bb->mx[a] * kp.inv_leaf));
        if (!(lo > -2147483000.0 && hi < 2147483000.0)) { set_error("map cloud: cell index beyond int32 (the reference's int cast overflows there too)"); return MRGFE_ERR_OVERFLOW; }
        kp.min_c[a] = static_cast<int32_t>(lo);
        const uint64_t span = static_cast<uint64_t>(hi - lo);
        bits[a] = 1;
        while ((uint64_t(1) << bits[a]) <= span) ++bits[a];
    }
    kp.shift_y = bits[0];
    kp.shift_z = bits[0] + bits[1];
    kp.total_bits = bits[0] + bits[1] + bits[2];
    if (kp.total_bits > 62) { set_error("map cloud: %u key bits needed (extent / resolution)", kp.total_bits); return MRGFE_ERR_OVERFLOW; }
    DevBuf &dk = ctx->scratch[2], &dv = ctx->scratch[3], &dkt = ctx->scratch[4], &dvt = ctx->scratch[5], &dh = ctx->scratch[6], &dfl = ctx->scratch[7], &dblk = ctx->scratch[8],
           &dkh = ctx->scratch[12], &dlo = ctx->scratch[13];
    MRGFE_TRY(dk.ensure(n * 4)); MRGFE_TRY(dv.ensure(n * 4)); MRGFE_TRY(dkt.ensure(n * 4)); MRGFE_TRY(dvt.ensure(n * 4)); MRGFE_TRY(dkh.ensure(n * 4)); MRGFE_TRY(dlo.ensure(n * 4));
    MRGFE_TRY(dh.ensure(sizeof(uint32_t) * 256 * (tab.total_blks + 1)));
    MRGFE_TRY(dfl.ensure(n * 4));
    MRGFE_TRY(dblk.ensure(sizeof(uint32_t) * (tab.total_blks + 8)));
    const dim3 grid((nn + 255) / 256);
    // dlo / dkh keep the keys by POINT index; dk is the sort's working copy
    hipLaunchKernelGGL(mapvox_keys_kernel, grid, dim3(256), 0, st, d_in, nn, kp, dlo.as<uint32_t>(), dkh.as<uint32_t>(), dv.as<uint32_t>());
    MRGFE_HIP_CHECK(hipGetLastError());
    MRGFE_HIP_CHECK(hipMemcpyAsync(dk.p, dlo.p, n * 4, hipMemcpyDeviceToDevice, st));
    const int key_bits = static_cast<int>(kp.total_bits) + 1;
    uint32_t *sk, *sv;
    MRGFE_TRY(radix_sort_pairs(ctx, dk.as<uint32_t>(), dv.as<uint32_t>(), dkt.as<uint32_t>(), dvt.as<uint32_t>(), &d_desc->sl, tab, std::min(key_bits, 32), dh.as<uint32_t>(), &sk, &sv));
    if (key_bits > 32) {
        // second, stable pass over the high word, gathered into the order the first pass produced
        uint32_t* k2 = (sk == dk.as<uint32_t>()) ? dk.as<uint32_t>() : dkt.as<uint32_t>();       // the buffer holding the sorted low words: no longer needed
        uint32_t* k2t = (sk == dk.as<uint32_t>()) ? dkt.as<uint32_t>() : dk.as<uint32_t>();
        uint32_t* v2t = (sv == dv.as<uint32_t>()) ? dvt.as<uint32_t>() : dv.as<uint32_t>();
        hipLaunchKernelGGL(gather_u32_kernel, grid, dim3(256), 0, st, dkh.as<uint32_t>(), sv, nn, k2);
        MRGFE_HIP_CHECK(hipGetLastError());
        MRGFE_TRY(radix_sort_pairs(ctx, k2, sv, k2t, v2t, &d_desc->sl, tab, key_bits - 32, dh.as<uint32_t>(), &sk, &sv));
    }
    hipLaunchKernelGGL(heads64_kernel, grid, dim3(256), 0, st, dlo.as<uint32_t>(), dkh.as<uint32_t>(), sv, nn, n_valid, dfl.as<uint32_t>());
    MRGFE_HIP_CHECK(hipGetLastError());
    uint32_t* d_ord = (sk == dk.as<uint32_t>()) ? dkt.as<uint32_t>() : dk.as<uint32_t>();  // the sort buffer that does not hold the result
    uint32_t* d_tot = dblk.as<uint32_t>() + tab.total_blks;
    MRGFE_TRY(exclusive_scan(ctx, dfl.as<uint32_t>(), d_ord, &d_desc->sl, tab, dblk.as<uint32_t>(), d_tot));
    uint32_t* h_tot = reinterpret_cast<uint32_t*>(hp.as<char>() + sizeof(Desc) + sizeof(BBox));
    MRGFE_HIP_CHECK(hipMemcpyAsync(h_tot, d_tot, 4, hipMemcpyDeviceToHost, st));
    MRGFE_HIP_CHECK(hipStreamSynchronize(st));
    const uint32_t V = *h_tot;
    if (V == 0) return MRGFE_OK;
    DevBuf &dseg = ctx->scratch[9], &dcent = ctx->scratch[10], &dkeep = ctx->scratch[11];
    MRGFE_TRY(dseg.ensure(sizeof(uint32_t) * (size_t(V) + 4)));
    MRGFE_TRY(dcent.ensure(sizeof(float4) * size_t(V)));
    MRGFE_TRY(dkeep.ensure(sizeof(uint32_t) * size_t(V)));
    hipLaunchKernelGGL(seg_from_heads_kernel, grid, dim3(256), 0, st, dfl.as<uint32_t>(), d_ord, nn, n_valid, V, dseg.as<uint32_t>());
    MRGFE_HIP_CHECK(hipGetLastError());
    MRGFE_TRY(launch_voxel_centroids(ctx, d_in, sv, dseg.as<uint32_t>(), V, min_pts, dcent.as<float4>(), dkeep.as<uint32_t>()));
    uint32_t kept = 0;
    MRGFE_TRY(compact_by_flags(ctx, dcent.as<float4>(), V, dkeep.as<uint32_t>(), d_out, &kept));
    *out_n = kept;
    return MRGFE_OK;
}

int map_cloud_device(mrgfe_ctx* ctx, const float4* d_cat, const uint32_t* kf_off, const float* poses_f, int K, float resolution, int min_pts, float far_thresh, float4* d_out,
                     size_t* out_n, size_t* n_unfiltered, const float4* const* kf_ptrs)
{
    *out_n = 0;
    *n_unfiltered = 0;
    const uint32_t n = kf_off[K];
    if (n == 0) return MRGFE_OK;
    hipStream_t st = ctx->stream;
    DevBuf dtab, dtrans, dfl;  // per-call buffers: the scratch slots are in use by the voxel-grid pass below
    auto cleanup = [&]() { dtab.release(); dtrans.release(); dfl.release(); };
    const size_t ptr_at = (sizeof(uint32_t) * (K + 1) + sizeof(float) * 16 * K + 15) & ~size_t(15);
    int rc = dtab.ensure(ptr_at + sizeof(void*) * K);
    if (rc == MRGFE_OK) rc = dtrans.ensure(size_t(n) * 16);
    if (rc == MRGFE_OK) rc = dfl.ensure(size_t(n) * 4);
    if (rc != MRGFE_OK) { cleanup(); return rc; }
    uint32_t* d_off = dtab.as<uint32_t>();
    float*    d_pose = reinterpret_cast<float*>(d_off + (K + 1));
    const float4* const* d_ptrs = kf_ptrs ? reinterpret_cast<const float4* const*>(dtab.as<char>() + ptr_at) : nullptr;
    const bool use_far = far_thresh > 0;  // map_cloud_generator.cpp:27-28
    if ((kf_ptrs && hipMemcpyAsync(dtab.as<char>() + ptr_at, kf_ptrs, sizeof(void*) * K, hipMemcpyHostToDevice, st) != hipSuccess) ||
        hipMemcpyAsync(d_off, kf_off, sizeof(uint32_t) * (K + 1), hipMemcpyHostToDevice, st) != hipSuccess ||
        hipMemcpyAsync(d_pose, poses_f, sizeof(float) * 16 * K, hipMemcpyHostToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
        cleanup();
        set_error("map cloud: table upload failed");
        return MRGFE_ERR_HIP;
    }
    hipLaunchKernelGGL(map_transform_kernel, dim3((n + 255) / 256), dim3(256), 0, st, kf_ptrs ? nullptr : d_cat, d_ptrs, n, d_off, d_pose, K, use_far ? 1 : 0, far_thresh * far_thresh, dtrans.as<float4>(),
                       dfl.as<uint32_t>());
    if (hipGetLastError() != hipSuccess) { cleanup(); set_error("map cloud: transform kernel launch failed"); return MRGFE_ERR_HIP; }
    const float4* d_cloud = dtrans.as<float4>();
    size_t        total = n;
    DevBuf        dcomp;
    if (use_far) {
        rc = dcomp.ensure(size_t(n) * 16);
        uint32_t kept = 0;
        if (rc == MRGFE_OK) rc = compact_by_flags(ctx, dtrans.as<float4>(), n, dfl.as<uint32_t>(), dcomp.as<float4>(), &kept);
        if (rc != MRGFE_OK) { cleanup(); dcomp.release(); return rc; }
        d_cloud = dcomp.as<float4>();
        total = kept;
    }
    *n_unfiltered = total;
    if (resolution <= 0.0f) {  // :66-70: the unfiltered cloud
        if (total && hipMemcpyAsync(d_out, d_cloud, total * 16, hipMemcpyDeviceToDevice, st) != hipSuccess) rc = MRGFE_ERR_HIP;
        if (hipStreamSynchronize(st) != hipSuccess) rc = MRGFE_ERR_HIP;
        *out_n = total;
    } else if (total) {
        // ApproximateMeanVoxelGrid: cells floor(p * inverse_leaf), f32 sums in input order (the radix sort is stable), division by
        // float(count), count threshold; only the output order differs (ascending cell instead of the reference's hash-map order)
        rc = mean_voxelgrid_device(ctx, d_cloud, total, resolution, min_pts, d_out, out_n);
    }
    cleanup();
    dcomp.release();
    return rc;
}

// ---- other-robot point removal -----------------------------------------------------------------------------------
constexpr int kMaxCentres = 64;
struct Centres { float xyz[kMaxCentres][3]; };

__global__ __launch_bounds__(256) void near_flags_kernel(const float4* __restrict__ in, uint32_t n, Centres c, int K, float radius_sqr, uint32_t* __restrict__ keep,
                                                          uint32_t* __restrict__ drop)
{
#pragma clang fp contract(off)
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float4 p = in[i];
    uint32_t gone = 0u;
    for (int k = 0; k < K; ++k) {
        const float dx = p.x - c.xyz[k][0], dy = p.y - c.xyz[k][1], dz = p.z - c.xyz[k][2];
        float s = dx * dx + dy * dy;  // (point - other).squaredNorm()
        s = s + dz * dz;
        if (s < radius_sqr) { gone = 1u; break; }  // mrg_slam_component.cpp:413
    }
    keep[i] = gone ^ 1u;
    drop[i] = gone;
}

int remove_points_near_device(mrgfe_ctx* ctx, const float4* d_in, size_t n, const float* centres, int K, float radius_sqr, float4* d_kept, size_t* n_kept, float4* d_removed,
                              size_t* n_removed)
{
    *n_kept = 0;
    if (n_removed) *n_removed = 0;
    if (n == 0) return MRGFE_OK;
    if (K > kMaxCentres) { set_error("remove_points_near: at most %d centres", kMaxCentres); return MRGFE_ERR_INVALID; }
    const uint32_t nn = static_cast<uint32_t>(n);
    Centres c{};
    for (int k = 0; k < K; ++k) for (int a = 0; a < 3; ++a) c.xyz[k][a] = centres[3 * k + a];
    DevBuf dk, dd;
    int rc = dk.ensure(n * 4);
    if (rc == MRGFE_OK) rc = dd.ensure(n * 4);
    if (rc == MRGFE_OK) {
        hipLaunchKernelGGL(near_flags_kernel, dim3((nn + 255) / 256), dim3(256), 0, ctx->stream, d_in, nn, c, K, radius_sqr, dk.as<uint32_t>(), dd.as<uint32_t>());
        if (hipGetLastError() != hipSuccess) { set_error("remove_points_near: kernel launch failed"); rc = MRGFE_ERR_HIP; }
    }
    uint32_t kept = 0, gone = 0;
    if (rc == MRGFE_OK) rc = compact_by_flags(ctx, d_in, nn, dk.as<uint32_t>(), d_kept, &kept);
    if (rc == MRGFE_OK && d_removed) rc = compact_by_flags(ctx, d_in, nn, dd.as<uint32_t>(), d_removed, &gone);
    dk.release();
    dd.release();
    if (rc != MRGFE_OK) return rc;
    *n_kept = kept;
    if (n_removed) *n_removed = d_removed ? gone : nn - kept;
    return MRGFE_OK;
}

// ---- deskewing ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void deskew_kernel(const float4* __restrict__ in, uint32_t n, float avx, float avy, float avz, double scan_period, float4* __restrict__ out)
{
#pragma clang fp contract(off)
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float4 p = in[i];
    const double delta_t = scan_period * static_cast<double>(i) / static_cast<double>(n);  // prefiltering_component.cpp:289
    const float  qw = 1.0f;
    const float  qx = static_cast<float>(delta_t / 2.0 * static_cast<double>(avx)), qy = static_cast<float>(delta_t / 2.0 * static_cast<double>(avy)),
                 qz = static_cast<float>(delta_t / 2.0 * static_cast<double>(avz));
    float n2 = qx * qx + qy * qy;  // delta_q.inverse() = conjugate / squaredNorm
    n2 = n2 + qz * qz;
    n2 = n2 + qw * qw;
    const float ix = -qx / n2, iy = -qy / n2, iz = -qz / n2, iw = qw / n2;
    float uvx = iy * p.z - iz * p.y, uvy = iz * p.x - ix * p.z, uvz = ix * p.y - iy * p.x;  // uv = 2 * vec x v
    uvx = uvx + uvx; uvy = uvy + uvy; uvz = uvz + uvz;
    const float cx = iy * uvz - iz * uvy, cy = iz * uvx - ix * uvz, cz = ix * uvy - iy * uvx;
    out[i] = make_float4((p.x + iw * uvx) + cx, (p.y + iw * uvy) + cy, (p.z + iw * uvz) + cz, p.w);
}

int deskew_device(mrgfe_ctx* ctx, const float4* d_in, size_t n, const float ang_v[3], double scan_period, float4* d_out)
{
    if (n == 0) return MRGFE_OK;
    const uint32_t nn = static_cast<uint32_t>(n);
    // ang_v *= -1 (:275)
    hipLaunchKernelGGL(deskew_kernel, dim3((nn + 255) / 256), dim3(256), 0, ctx->stream, d_in, nn, ang_v[0] * -1.0f, ang_v[1] * -1.0f, ang_v[2] * -1.0f, scan_period, d_out);
    MRGFE_HIP_CHECK(hipGetLastError());
    return MRGFE_OK;
}

// ---- rigid transform of a cloud ------------------------------------------------------------------------------------
struct Transform12 { float m[12]; };
__global__ __launch_bounds__(256) void transform_cloud_kernel(const float4* __restrict__ in, uint32_t n, Transform12 T, float4* __restrict__ out)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    float4 p = in[i];
    if (finite3(p.x, p.y, p.z)) {  // pcl::transformPointCloud leaves the non-finite points of a non-dense cloud as they are
        float x, y, z;
        transform_point(T.m, p.x, p.y, p.z, x, y, z);
        p.x = x; p.y = y; p.z = z;
    }
    out[i] = p;
}

int transform_cloud_device(mrgfe_ctx* ctx, const float4* d_in, size_t n, const float T_rowmajor[16], float4* d_out)
{
    if (n == 0) return MRGFE_OK;
    Transform12 T;
    std::memcpy(T.m, T_rowmajor, sizeof(T.m));
    const uint32_t nn = static_cast<uint32_t>(n);
    hipLaunchKernelGGL(transform_cloud_kernel, dim3((nn + 255) / 256), dim3(256), 0, ctx->stream, d_in, nn, T, d_out);
    MRGFE_HIP_CHECK(hipGetLastError());
    return MRGFE_OK;
}

}  // namespace mrgfe
