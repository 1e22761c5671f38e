// csrc/nn_grid.hip — exact nearest-neighbour search on a radix-sorted uniform grid (see nn_grid.h).
//
// Memory behaviour: the query kernels read one query (16 B, coalesced) and then a handful of contiguous candidate
// ranges (16 B per candidate) out of a cloud that was reordered cell-major at build time, so neighbouring lanes —
// which hold spatially neighbouring queries for a LiDAR scan — hit the same cache lines.  HBM/L2 bound, no MFMA.
#include <cfloat>
#include <cmath>
#include <cstring>

#include "dev_float.h"
#include "dev_utils.h"
#include "nn_device.h"
#include "nn_grid.h"

namespace mrgfe {

// ---- build ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void nn_cellkey_kernel(const float4* __restrict__ pts, uint32_t n, NnGridDev g, uint32_t n_cells, uint32_t* __restrict__ keys,
                                                          uint32_t* __restrict__ vals, uint32_t* __restrict__ counts)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float4 p = pts[i];
    int          c[3];
    uint32_t     key = n_cells;
    if (nn_cell_of(g, p.x, p.y, p.z, c)) {
        key = (static_cast<uint32_t>(c[2]) * g.dim[1] + c[1]) * g.dim[0] + c[0];
        atomicAdd(&counts[key], 1u);
    }
    keys[i] = key;
    vals[i] = i;
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

int NnGrid::build(mrgfe_ctx* ctx, const float4* d_pts, size_t n, float cell_size)
{
    built_ = false;
    n_ = n;
    std::memset(&h_, 0, sizeof(h_));
    h_.cell = cell_size;
    h_.dim[0] = h_.dim[1] = h_.dim[2] = 1;
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
        for (int a = 0; a < 3; ++a) h_.origin[a] = 0;
        MRGFE_TRY(d_cell_start_.ensure(8));
        MRGFE_HIP_CHECK(hipMemsetAsync(d_cell_start_.p, 0, 8, st));
        MRGFE_TRY(d_sorted_.ensure(16));
        h_.n = 0;
        h_.cell_start = d_cell_start_.as<uint32_t>();
        h_.sorted = d_sorted_.as<float4>();
        built_ = true;
        return MRGFE_OK;
    }
    float cell = cell_size;
    for (;;) {
        double prod = 1;
        for (int a = 0; a < 3; ++a) { h_.dim[a] = static_cast<int>(std::floor((bb.mx[a] - bb.mn[a]) / cell)) + 1; prod *= h_.dim[a]; }
        if (prod <= double(1u << 24)) break;
        cell *= 2.0f;
    }
    h_.cell = cell;
    for (int a = 0; a < 3; ++a) h_.origin[a] = bb.mn[a];
    const uint32_t n_cells = static_cast<uint32_t>(h_.dim[0]) * h_.dim[1] * h_.dim[2];
    MRGFE_TRY(d_cell_start_.ensure(sizeof(uint32_t) * (size_t(n_cells) + 4)));
    MRGFE_TRY(d_sorted_.ensure(sizeof(float4) * std::max<size_t>(n, 1)));
    MRGFE_HIP_CHECK(hipMemsetAsync(d_cell_start_.p, 0, sizeof(uint32_t) * (size_t(n_cells) + 1), st));
    h_.cell_start = d_cell_start_.as<uint32_t>();
    h_.sorted = d_sorted_.as<float4>();
    h_.n = bb.n_finite;
    DevBuf &dk = ctx->scratch[2], &dv = ctx->scratch[3], &dkt = ctx->scratch[4], &dvt = ctx->scratch[5], &dh = ctx->scratch[6], &dblk = ctx->scratch[8];
    MRGFE_TRY(dk.ensure(n * 4)); MRGFE_TRY(dv.ensure(n * 4)); MRGFE_TRY(dkt.ensure(n * 4)); MRGFE_TRY(dvt.ensure(n * 4));
    MRGFE_TRY(dh.ensure(sizeof(uint32_t) * 256 * (tab.total_blks + 1)));
    hipLaunchKernelGGL(nn_cellkey_kernel, dim3((nn + 255) / 256), dim3(256), 0, st, d_pts, nn, h_, n_cells, dk.as<uint32_t>(), dv.as<uint32_t>(), d_cell_start_.as<uint32_t>());
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
    MRGFE_TRY(exclusive_scan(ctx, d_cell_start_.as<uint32_t>(), d_cell_start_.as<uint32_t>(), ds.as<Slice>() + 1, ctab, dblk.as<uint32_t>(), dblk.as<uint32_t>() + ctab.total_blks));
    hipLaunchKernelGGL(nn_gather_kernel, dim3((h_.n + 255) / 256), dim3(256), 0, st, d_pts, sv, h_.n, d_sorted_.as<float4>());
    MRGFE_HIP_CHECK(hipGetLastError());
    MRGFE_HIP_CHECK(hipStreamSynchronize(st));  // ctab / tab host tables were sources of async copies
    built_ = true;
    return MRGFE_OK;
}

void NnGrid::release()
{
    d_cell_start_.release();
    d_sorted_.release();
    built_ = false;
}

// ---- queries ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void nn_nearest_kernel(NnGridDev g, const float4* __restrict__ q, uint32_t n, const float* __restrict__ T12, int32_t* __restrict__ idx,
                                                          float* __restrict__ sqd)
{
    __shared__ float s_T[12];
    const bool       use_T = T12 != nullptr;
    if (use_T && threadIdx.x < 12) s_T[threadIdx.x] = T12[threadIdx.x];
    __syncthreads();
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float4 p = q[i];
    float x = p.x, y = p.y, z = p.z;
    if (use_T) transform_point(s_T, p.x, p.y, p.z, x, y, z);
    int32_t bi;
    float   bd;
    nn_nearest(g, x, y, z, bi, bd);
    idx[i] = bi;
    sqd[i] = bi >= 0 ? bd : -1.0f;
}

// getFitnessScore: block partial = (sum of squared distances, count)
__global__ __launch_bounds__(256) void nn_fitness_kernel(NnGridDev g, const float4* __restrict__ src, uint32_t n, const float* __restrict__ T12, double max_range,
                                                          double* __restrict__ partial)
{
    __shared__ float  s_T[12];
    __shared__ double s_sum[4];
    __shared__ uint32_t s_cnt[4];
    if (threadIdx.x < 12) s_T[threadIdx.x] = T12[threadIdx.x];
    __syncthreads();
    double   sum = 0.0;
    uint32_t cnt = 0;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const float4 p = src[i];
        float x, y, z;
        transform_point(s_T, p.x, p.y, p.z, x, y, z);
        int32_t bi;
        float   bd;
        nn_nearest(g, x, y, z, bi, bd);
        if (bi >= 0 && static_cast<double>(bd) <= max_range) { sum += static_cast<double>(bd); ++cnt; }
    }
    sum = wave_sum(sum);
    cnt = wave_sum(cnt);
    if (lane_id() == 0) { s_sum[wave_id()] = sum; s_cnt[wave_id()] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[2 * blockIdx.x] = ((s_sum[0] + s_sum[1]) + s_sum[2]) + s_sum[3];
        partial[2 * blockIdx.x + 1] = static_cast<double>(s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3]);
    }
}

__global__ __launch_bounds__(256) void nn_fitness_final_kernel(const double* __restrict__ partial, uint32_t nblk, double* __restrict__ out)
{
    __shared__ double s_a[4], s_b[4];
    double a = 0, b = 0;
    for (uint32_t i = threadIdx.x; i < nblk; i += 256) { a += partial[2 * i]; b += partial[2 * i + 1]; }
    a = wave_sum(a);
    b = wave_sum(b);
    if (lane_id() == 0) { s_a[wave_id()] = a; s_b[wave_id()] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        out[0] = ((s_a[0] + s_a[1]) + s_a[2]) + s_a[3];
        out[1] = ((s_b[0] + s_b[1]) + s_b[2]) + s_b[3];
    }
}

int NnGrid::fitness(mrgfe_ctx* ctx, const float4* d_src, size_t n_src, const float T[16], double max_range, double* out)
{
    *out = DBL_MAX;
    if (!built_) { set_error("NnGrid::fitness before build"); return MRGFE_ERR_STATE; }
    if (n_src == 0 || h_.n == 0) return MRGFE_OK;
    hipStream_t st = ctx->stream;
    const uint32_t n = static_cast<uint32_t>(n_src);
    const uint32_t nblk = std::min<uint32_t>((n + 255) / 256, 4096);
    DevBuf& dw = ctx->scratch[9];
    MRGFE_TRY(dw.ensure(64 + sizeof(double) * 2 * (nblk + 1)));
    float*  d_T = dw.as<float>();
    double* d_part = reinterpret_cast<double*>(dw.as<char>() + 64);
    double* d_res = d_part + 2 * nblk;
    MRGFE_HIP_CHECK(hipMemcpyAsync(d_T, T, 48, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(nn_fitness_kernel, dim3(nblk), dim3(256), 0, st, h_, d_src, n, d_T, max_range, d_part);
    hipLaunchKernelGGL(nn_fitness_final_kernel, dim3(1), dim3(256), 0, st, d_part, nblk, d_res);
    MRGFE_HIP_CHECK(hipGetLastError());
    double res[2];
    MRGFE_HIP_CHECK(hipMemcpyAsync(res, d_res, sizeof(res), hipMemcpyDeviceToHost, st));
    MRGFE_HIP_CHECK(hipStreamSynchronize(st));
    if (res[1] > 0) *out = res[0] / res[1];
    return MRGFE_OK;
}

int NnGrid::nearest_device(mrgfe_ctx* ctx, const float4* d_q, size_t n, const float* d_T12, int32_t* d_idx, float* d_sqd)
{
    if (!built_) { set_error("NnGrid::nearest before build"); return MRGFE_ERR_STATE; }
    if (n == 0) return MRGFE_OK;
    const uint32_t nn = static_cast<uint32_t>(n);
    hipLaunchKernelGGL(nn_nearest_kernel, dim3((nn + 255) / 256), dim3(256), 0, ctx->stream, h_, d_q, nn, d_T12, d_idx, d_sqd);
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
            g, c, rings,
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
    const int rings = static_cast<int>(std::ceil(std::sqrt(r2) / h_.cell)) + 1;
    hipLaunchKernelGGL(nn_radius_flags_kernel, dim3((nn + 255) / 256), dim3(256), 0, ctx->stream, h_, d_q, nn, r2, need, rings, d_flags);
    MRGFE_HIP_CHECK(hipGetLastError());
    return MRGFE_OK;
}

// k nearest neighbours. Per-thread sorted candidate lists live in LDS, laid out [slot][thread] (conflict-free).
constexpr int kKnnThreads = 128;
__global__ __launch_bounds__(kKnnThreads) void nn_knn_kernel(NnGridDev g, const float4* __restrict__ q, uint32_t n, int k, int32_t* __restrict__ idx, float* __restrict__ sqd)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char knn_lds[];
    float*   ld = reinterpret_cast<float*>(knn_lds);                       // [k][kKnnThreads]
    int32_t* li = reinterpret_cast<int32_t*>(knn_lds) + k * kKnnThreads;   // [k][kKnnThreads]
    const uint32_t i = blockIdx.x * kKnnThreads + threadIdx.x;
    if (i >= n) return;
    const int t = threadIdx.x;
    int       cnt = 0;
    const float4 p = q[i];
    int          c[3];
    if (g.n > 0 && nn_cell_of(g, p.x, p.y, p.z, c)) {
        nn_walk(
            g, c, -1,
            [&](const float4& cand) {
                const float   d = sqdist3f(cand.x, cand.y, cand.z, p.x, p.y, p.z);
                const int32_t ci = __float_as_int(cand.w);
                if (cnt == k) {
                    const float   wd = ld[(k - 1) * kKnnThreads + t];
                    const int32_t wi = li[(k - 1) * kKnnThreads + t];
                    if (!(d < wd || (d == wd && ci < wi))) return;
                }
                int pos = cnt < k ? cnt : k - 1;  // slot that is free / dropped
                while (pos > 0) {
                    const float   pd = ld[(pos - 1) * kKnnThreads + t];
                    const int32_t pi = li[(pos - 1) * kKnnThreads + t];
                    if (d < pd || (d == pd && ci < pi)) {
                        ld[pos * kKnnThreads + t] = pd;
                        li[pos * kKnnThreads + t] = pi;
                        --pos;
                    } else {
                        break;
                    }
                }
                ld[pos * kKnnThreads + t] = d;
                li[pos * kKnnThreads + t] = ci;
                if (cnt < k) ++cnt;
            },
            [&](double bound_sq) { return cnt == k && static_cast<double>(ld[(k - 1) * kKnnThreads + t]) < bound_sq; });
    }
    for (int s = 0; s < k; ++s) {
        idx[size_t(i) * k + s] = s < cnt ? li[s * kKnnThreads + t] : -1;
        sqd[size_t(i) * k + s] = s < cnt ? ld[s * kKnnThreads + t] : -1.0f;
    }
}

int NnGrid::knn_device(mrgfe_ctx* ctx, const float4* d_q, size_t n, int k, int32_t* d_idx, float* d_sqd)
{
    if (!built_) { set_error("NnGrid::knn before build"); return MRGFE_ERR_STATE; }
    if (k < 1 || k > 64) { set_error("NnGrid::knn: k must be in [1, 64]"); return MRGFE_ERR_INVALID; }
    if (n == 0) return MRGFE_OK;
    const uint32_t nn = static_cast<uint32_t>(n);
    const size_t   lds = size_t(k) * kKnnThreads * 8;
    hipLaunchKernelGGL(nn_knn_kernel, dim3((nn + kKnnThreads - 1) / kKnnThreads), dim3(kKnnThreads), lds, ctx->stream, h_, d_q, nn, k, d_idx, d_sqd);
    MRGFE_HIP_CHECK(hipGetLastError());
    return MRGFE_OK;
}

}  // namespace mrgfe
