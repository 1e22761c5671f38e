// csrc/mapcloud.hip — SURVEY.md §8(f) rows 2 and 4 on MI355X: streaming per-point kernels (16 B in, 16 B out per point,
// HBM bound) in front of the compaction and voxel-grid machinery of filters.hip.
//   MapCloudGenerator::generate         /root/reference/src/mrg_slam/map_cloud_generator.cpp:14-86
//   pcl::ApproximateMeanVoxelGrid       /root/reference/include/pcl/filters/ApproximateMeanVoxelGrid.hpp:63-126
//   other-robot point removal           /root/reference/apps/mrg_slam_component.cpp:396-429
//   PrefilteringComponent::deskewing    /root/reference/apps/prefiltering_component.cpp:231-292
// Float expressions run in the order documented in oracle/mapcloud.cpp (left to right, no FMA).
#include "mapcloud.h"

#include <vector>

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
        // ApproximateMeanVoxelGrid == the voxel-grid pass: same cells (floor(p * inverse_leaf)), f32 sums in input order
        // (the radix sort is stable), division by float(count), count threshold; only the output order differs
        // (ascending voxel index instead of the reference's hash-map order)
        int overflow = 0;
        rc = filter_voxelgrid_device(ctx, d_cloud, total, resolution, min_pts, d_out, out_n, &overflow);
        if (rc == MRGFE_OK && overflow) {
            *out_n = 0;
            set_error("map cloud: extent / resolution needs more than 2^31 voxel indices");
            rc = MRGFE_ERR_OVERFLOW;
        }
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

}  // namespace mrgfe
