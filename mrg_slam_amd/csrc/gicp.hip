// csrc/gicp.hip — GICP_HIP: fast_gicp::FastGICP<PointXYZI,PointXYZI> + fast_gicp::LsqRegistration (Levenberg-Marquardt)
// on MI355X, the registration the reference's code default selects (/root/reference/src/mrg_slam/registrations.cpp:55-63;
// parameter "registration_method" default "FAST_GICP", apps/scan_matching_odometry_component.cpp:122). SURVEY.md A.6.
//
// Device side (KNN correspondence path of BASELINE config 3):
//   covariances : k-NN on the radix-sorted grid -> 3x3 sample covariance -> PLANE regularisation (1, 1, 1e-3)   [per cloud, once]
//   linearize   : per source point transform -> exact 1-NN within max_correspondence_distance -> Mahalanobis matrix
//                 (C_B + R C_A R^T)^-1 -> J^T M J (21), J^T M r (6), r^T M r, reduced like the NDT kernel        [per LM outer step]
//   error       : r^T M r with the stored correspondences / Mahalanobis matrices                                  [per LM trial]
// Host side: the LM loop (step_lm, is_converged, se3_exp) - a few 6x6 solves per iteration.
#include <cfloat>
#include <cmath>
#include <cstring>
#include <functional>
#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "dev_float.h"
#include "dev_linalg.h"
#include "dev_utils.h"
#include "bfgs.h"
#include "gicp_engine.h"
#include "nn_device.h"
#include "ndt_derivatives.h"  // launch_transform_cloud

namespace mrgfe {

constexpr int kGicpStride = 32;  // partial record: err, b[6], H upper[21], n_corr, pad

__device__ __forceinline__ int gidx(int i, int j) { return i * 6 - (i * (i - 1)) / 2 + (j - i); }

// ---- covariances -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gicp_cov_kernel(const float4* __restrict__ pts, uint32_t n, const int32_t* __restrict__ knn, int k, double* __restrict__ cov6)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const int32_t* nb = knn + size_t(i) * k;
    double mean[3] = {0, 0, 0};
    for (int j = 0; j < k; ++j) {
        const int32_t id = nb[j];
        if (id < 0) continue;
        const float4 p = pts[id];
        mean[0] += p.x; mean[1] += p.y; mean[2] += p.z;
    }
    mean[0] /= k; mean[1] /= k; mean[2] /= k;
    double c[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int j = 0; j < k; ++j) {
        const int32_t id = nb[j];
        if (id < 0) continue;
        const float4 p = pts[id];
        const double d[3] = {static_cast<double>(p.x) - mean[0], static_cast<double>(p.y) - mean[1], static_cast<double>(p.z) - mean[2]};
        for (int r = 0; r < 3; ++r) for (int cc = 0; cc < 3; ++cc) c[r * 3 + cc] += d[r] * d[cc];
    }
    for (int t = 0; t < 9; ++t) c[t] /= k;
    double w[3], E[9];
    dl_sym_eig3(c, w, E);                       // ascending
    const double vals[3] = {1e-3, 1.0, 1.0};   // RegularizationMethod::PLANE, singular values (1, 1, 1e-3) descending
    double out[9];
    for (int r = 0; r < 3; ++r)
        for (int cc = 0; cc < 3; ++cc) {
            double s = 0;
            for (int m = 0; m < 3; ++m) s += E[r * 3 + m] * vals[m] * E[cc * 3 + m];
            out[r * 3 + cc] = s;
        }
    double* o = cov6 + size_t(i) * 6;
    o[0] = out[0]; o[1] = out[1]; o[2] = out[2]; o[3] = out[4]; o[4] = out[5]; o[5] = out[8];
}

// ---- VGICP: GaussianVoxelMap of the target (fast_gicp::GaussianVoxelMap, ADDITIVE accumulation) -------------------------
constexpr int kVoxRec = 10;  // per voxel: mean[3], covariance xx xy xz yy yz zz, number of points
struct VoxGridDev {
    double        res;
    int32_t       cmin[3], dim[3];
    uint32_t      n_cells;
    const double* vox;  // n_cells records, dense in (z, y, x) of the voxel coordinates; points == 0: empty
};
// voxel_coord: floor(x / resolution - 0.5)
__device__ __forceinline__ int vox_coord(double x, double res) { return static_cast<int>(floor(x / res - 0.5)); }
__device__ __forceinline__ uint32_t vox_cell(const VoxGridDev& g, double x, double y, double z)
{
    if (!(isfinite(x) && isfinite(y) && isfinite(z))) return g.n_cells;
    const int c[3] = {vox_coord(x, g.res) - g.cmin[0], vox_coord(y, g.res) - g.cmin[1], vox_coord(z, g.res) - g.cmin[2]};
    if (c[0] < 0 || c[0] >= g.dim[0] || c[1] < 0 || c[1] >= g.dim[1] || c[2] < 0 || c[2] >= g.dim[2]) return g.n_cells;
    return (static_cast<uint32_t>(c[2]) * g.dim[1] + c[1]) * g.dim[0] + c[0];
}
__global__ __launch_bounds__(256) void vox_key_kernel(const float4* __restrict__ pts, uint32_t n, VoxGridDev g, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float4 p = pts[i];
    keys[i] = vox_cell(g, p.x, p.y, p.z);  // n_cells: non-finite, sorts behind every voxel
    vals[i] = i;
}
// first / one-past-last position of every voxel's run in the sorted keys (first == 0xffffffff: empty voxel)
__global__ __launch_bounds__(256) void vox_runs_kernel(const uint32_t* __restrict__ keys, uint32_t n, uint32_t n_cells, uint32_t* __restrict__ first, uint32_t* __restrict__ last)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t k = keys[i];
    if (k >= n_cells) return;
    if (i == 0 || keys[i - 1] != k) first[k] = i;
    if (i + 1 == n || keys[i + 1] != k) last[k] = i + 1;
}
// one thread per voxel: sums in point order (the radix sort is stable), then the means
__global__ __launch_bounds__(256) void vox_sums_kernel(const float4* __restrict__ pts, const double* __restrict__ cov6, const uint32_t* __restrict__ sorted_vals,
                                                        const uint32_t* __restrict__ first, const uint32_t* __restrict__ last, uint32_t n_cells, double* __restrict__ vox)
{
#pragma clang fp contract(off)
    const uint32_t c = blockIdx.x * 256u + threadIdx.x;
    if (c >= n_cells) return;
    double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t cnt = 0;
    if (first[c] != 0xffffffffu) {
        for (uint32_t k = first[c]; k < last[c]; ++k) {
            const uint32_t i = sorted_vals[k];
            const float4   p = pts[i];
            acc[0] += static_cast<double>(p.x); acc[1] += static_cast<double>(p.y); acc[2] += static_cast<double>(p.z);
            const double* cv = cov6 + size_t(i) * 6;
#pragma unroll
            for (int t = 0; t < 6; ++t) acc[3 + t] += cv[t];
            ++cnt;
        }
        const double dn = static_cast<double>(cnt);
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[t] = acc[t] / dn;
    }
    double* o = vox + size_t(c) * kVoxRec;
#pragma unroll
    for (int t = 0; t < 9; ++t) o[t] = acc[t];
    o[9] = static_cast<double>(cnt);
}
// FastVGICP::update_correspondences, DIRECT1: the voxel trans * mean_A falls in (double), -1 if it is empty
__global__ __launch_bounds__(256) void vox_corr_kernel(const float4* __restrict__ src, uint32_t n, VoxGridDev g, const double* __restrict__ T12, int32_t* __restrict__ corr)
{
#pragma clang fp contract(off)
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float4 a = src[i];
    const double mA[3] = {a.x, a.y, a.z};
    double tA[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) tA[r] = T12[r * 4 + 0] * mA[0] + T12[r * 4 + 1] * mA[1] + T12[r * 4 + 2] * mA[2] + T12[r * 4 + 3];
    const uint32_t c = vox_cell(g, tA[0], tA[1], tA[2]);
    corr[i] = (c < g.n_cells && g.vox[size_t(c) * kVoxRec + 9] > 0.0) ? static_cast<int32_t>(c) : -1;
}

// ---- ICP (pcl::IterativeClosestPoint): correspondences + the moment sums of TransformationEstimationSVD ------------------
// record: [0] correspondences, [1..3] sum src, [4..6] sum dst, [7..15] sum dst * src^T (row-major), [16] sum of squared distances
template <bool kReciprocal>
__global__ __launch_bounds__(256) void icp_corr_sums_kernel(NnGrid2Dev g, NnGrid2Dev g_cur, const float4* __restrict__ cur, const float4* __restrict__ tgt, uint32_t n, double max_sq,
                                                             double* __restrict__ partials);
__global__ __launch_bounds__(256) void icp_transform_kernel(float4* __restrict__ cur, uint32_t n, const float* __restrict__ T12)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    float4 p = cur[i];
    float  x, y, z;
    transform_point(T12, p.x, p.y, p.z, x, y, z);  // pcl::transformPointCloud
    cur[i] = make_float4(x, y, z, p.w);
}

struct GicpPose {
    double T[12];   // row-major 3x4, double
    float  Tf[12];  // trans.cast<float>()
    int    right;   // 1: linearise for T <- T exp(d) (small_gicp), 0: for T <- exp(d) T (fast_gicp)
    int    pad;
};

// shared tail: block reduction of 29 doubles into one partial record
__device__ __forceinline__ void gicp_block_reduce(double (&vals)[29], double* __restrict__ partial_out, int first_used_h)
{
    __shared__ double s_red[4][kGicpStride];
#pragma unroll
    for (int k = 0; k < 29; ++k) {
        if (k >= 1 && k < 28 && k >= first_used_h) continue;  // entries not produced by this kernel variant
        const double r = wave_sum(vals[k]);
        if (lane_id() == 0) s_red[wave_id()][k] = r;
    }
    __syncthreads();
    if (threadIdx.x < kGicpStride) {
        const int  k = threadIdx.x;
        const bool skip = k >= 29 || (k >= 1 && k < 28 && k >= first_used_h);
        double     r = 0.0;
        if (!skip) r = ((s_red[0][k] + s_red[1][k]) + s_red[2][k]) + s_red[3][k];
        partial_out[k] = r;
    }
}

// update_correspondences: exact 1-NN of trans_f * source point in the target, G lanes per query.  A single registration has 130k
// queries and wants eight lanes on each to fill the chip (two: 1.05 -> 1.49 ms over six rounds); a batch of candidates fills it
// anyway, and there the lanes of a group mostly repeat each other's bookkeeping (eight -> two: 22.9 -> 16.7 ms for 64 pairs).
constexpr int kGicpGroup = 8;       // single registration, and the ICP moment kernel
constexpr int kGicpBatchGroup = 2;  // batched candidates
template <int G>
__device__ __forceinline__ void gicp_corr_query(const NnGrid2Dev& g, const float4* __restrict__ src, uint32_t n, const GicpPose& pose, double thr2, int32_t* __restrict__ corr,
                                                uint32_t blk)
{
#pragma clang fp contract(off)
    const uint32_t i = blk * (256u / G) + threadIdx.x / G;
    if (i >= n) return;
    const float4 a = src[i];
    // trans_f * Vector4f(x, y, z, 1): accumulated column by column
    float q[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        float s = pose.Tf[r * 4 + 0] * a.x;
        s = s + pose.Tf[r * 4 + 1] * a.y;
        s = s + pose.Tf[r * 4 + 2] * a.z;
        q[r] = s + pose.Tf[r * 4 + 3];
    }
    int32_t j = -1;
    float   sqd = INFINITY;
    nn_nearest_group<G>(g, q[0], q[1], q[2], static_cast<int>(threadIdx.x % G), thr2, j, sqd);
    if (j >= 0 && !(static_cast<double>(sqd) < thr2)) j = -1;
    if (threadIdx.x % G == 0) corr[i] = j;
}

template <int G>
__global__ __launch_bounds__(256) void gicp_corr_kernel(NnGrid2Dev g, const float4* __restrict__ src, uint32_t n, GicpPose pose, double thr2, int32_t* __restrict__ corr)
{
    gicp_corr_query<G>(g, src, n, pose, thr2, corr, blockIdx.x);
}
// a downsampled odometry scan (33k queries) leaves the chip half empty with eight lanes per query: sixteen there (33k-point frame 0.87 -> 0.82 ms,
// 130k-point frame unchanged); the answer is the exact nearest neighbour with ties to the lower index whatever the group size
constexpr uint32_t kGicpWideGroupBelow = 65536;

template <bool kReciprocal>
__global__ __launch_bounds__(256) void icp_corr_sums_kernel(NnGrid2Dev g, NnGrid2Dev g_cur, const float4* __restrict__ cur, const float4* __restrict__ tgt, uint32_t n, double max_sq,
                                                             double* __restrict__ partials)
{
#pragma clang fp contract(off)
    double vals[29];
#pragma unroll
    for (int k = 0; k < 29; ++k) vals[k] = 0.0;
    // eight lanes search one query; lane 0 of the group carries its contribution into the block sum
    constexpr uint32_t per_blk = 256u / kGicpGroup;
    const uint32_t i = blockIdx.x * per_blk + threadIdx.x / kGicpGroup;
    if (i < n) {
        const float4 p = cur[i];
        int32_t j = -1;
        float   sqd = INFINITY;
        nn_nearest_group<kGicpGroup>(g, p.x, p.y, p.z, static_cast<int>(threadIdx.x % kGicpGroup), max_sq, j, sqd);
        bool keep = j >= 0 && !(static_cast<double>(sqd) > max_sq);  // determineCorrespondences: skipped iff distance > max_dist^2 (uniform within the group)
        if (kReciprocal && keep) {  // determineReciprocalCorrespondences: the target point's nearest source point must be this one, within the limit
            const float4 q = tgt[j];
            int32_t ri = -1;
            float   rd = INFINITY;
            nn_nearest_group<kGicpGroup>(g_cur, q.x, q.y, q.z, static_cast<int>(threadIdx.x % kGicpGroup), max_sq, ri, rd);
            keep = ri == static_cast<int32_t>(i) && !(static_cast<double>(rd) > max_sq);
        }
        if (threadIdx.x % kGicpGroup == 0 && keep) {
            const float4 q = tgt[j];
            vals[0] = 1.0;
            vals[1] = p.x; vals[2] = p.y; vals[3] = p.z;
            vals[4] = q.x; vals[5] = q.y; vals[6] = q.z;
            const double s[3] = {p.x, p.y, p.z}, d[3] = {q.x, q.y, q.z};
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) vals[7 + r * 3 + c] = d[r] * s[c];
            vals[16] = sqd;
        }
    }
    gicp_block_reduce(vals, partials + size_t(blockIdx.x) * kGicpStride, 17);
}

// ---- PCL_GICP_HIP: pcl::GeneralizedIterativeClosestPoint (registrations.cpp:93-103) / pclomp::GICP (:104-114) -----------------------
// computeCovariances: raw second moments of the FLOAT coordinates (float products) summed in double, / k, minus mean mean^T; the SVD of the
// symmetric result = its eigenvectors ordered by |eigenvalue|; singular values replaced by (1, 1, gicp_epsilon)
__global__ __launch_bounds__(256) void pclgicp_cov_kernel(const float4* __restrict__ pts, uint32_t n, const int32_t* __restrict__ knn, int k, double gicp_epsilon, double* __restrict__ cov6)
{
#pragma clang fp contract(off)
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const int32_t* nb = knn + size_t(i) * k;
    double mean[3] = {0, 0, 0}, c[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int j = 0; j < k; ++j) {
        const int32_t id = nb[j];
        if (id < 0) continue;
        const float4 p = pts[id];
        mean[0] += p.x; mean[1] += p.y; mean[2] += p.z;
        const float xx = p.x * p.x, yx = p.y * p.x, yy = p.y * p.y, zx = p.z * p.x, zy = p.z * p.y, zz = p.z * p.z;
        c[0] += xx; c[3] += yx; c[4] += yy; c[6] += zx; c[7] += zy; c[8] += zz;
    }
    for (int a = 0; a < 3; ++a) mean[a] /= static_cast<double>(k);
    for (int r = 0; r < 3; ++r)
        for (int cc = 0; cc <= r; ++cc) {
            c[r * 3 + cc] /= static_cast<double>(k);
            c[r * 3 + cc] -= mean[r] * mean[cc];
            c[cc * 3 + r] = c[r * 3 + cc];
        }
    double w[3], E[9];
    dl_sym_eig3(c, w, E);  // ascending eigenvalues; singular values are their magnitudes
    int order[3] = {0, 1, 2};
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2 - a; ++b)
            if (fabs(w[order[b]]) < fabs(w[order[b + 1]])) { const int t = order[b]; order[b] = order[b + 1]; order[b + 1] = t; }  // descending, stable
    double out[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int kk = 0; kk < 3; ++kk) {
        const double v = kk == 2 ? gicp_epsilon : 1.0;
        const int    col = order[kk];
        for (int r = 0; r < 3; ++r)
            for (int cc = 0; cc < 3; ++cc) out[r * 3 + cc] += v * E[r * 3 + col] * E[cc * 3 + col];
    }
    double* o = cov6 + size_t(i) * 6;
    o[0] = out[0]; o[1] = out[1]; o[2] = out[2]; o[3] = out[4]; o[4] = out[5]; o[5] = out[8];
}

// one outer iteration's search loop (gicp.hpp computeTransformation): query = transformation_ * point in float, nearest target point, kept iff
// squared distance < max_correspondence_distance^2; its Mahalanobis matrix (R C1 R^T + C2)^-1, R = rotation of transformation_ * guess (double)
struct PclGicpIter {
    float  Tf[12];
    double R[9];
    double thr2;
};
__global__ __launch_bounds__(256) void pclgicp_corr_kernel(NnGrid2Dev g, const float4* __restrict__ src, uint32_t n, PclGicpIter it, const double* __restrict__ cov_src,
                                                            const double* __restrict__ cov_tgt, int32_t* __restrict__ corr, double* __restrict__ mahal)
{
#pragma clang fp contract(off)
    const uint32_t i = blockIdx.x * (256u / kGicpGroup) + threadIdx.x / kGicpGroup;
    if (i >= n) return;
    const float4 a = src[i];
    float q[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        float s = it.Tf[r * 4 + 0] * a.x;
        s = s + it.Tf[r * 4 + 1] * a.y;
        s = s + it.Tf[r * 4 + 2] * a.z;
        q[r] = s + it.Tf[r * 4 + 3];
    }
    int32_t j = -1;
    float   sqd = INFINITY;
    nn_nearest_group<kGicpGroup>(g, q[0], q[1], q[2], static_cast<int>(threadIdx.x % kGicpGroup), it.thr2, j, sqd);
    if (threadIdx.x % kGicpGroup != 0) return;
    if (j >= 0 && !(static_cast<double>(sqd) < it.thr2)) j = -1;
    corr[i] = j;
    if (j < 0) return;
    const double* c1 = cov_src + size_t(i) * 6;
    const double* c2 = cov_tgt + size_t(j) * 6;
    const double C1[9] = {c1[0], c1[1], c1[2], c1[1], c1[3], c1[4], c1[2], c1[4], c1[5]};
    const double C2[9] = {c2[0], c2[1], c2[2], c2[1], c2[3], c2[4], c2[2], c2[4], c2[5]};
    double RC[9], Rt[9], tmp[9], M[9];
    dl_mul3(it.R, C1, RC);
    for (int r = 0; r < 3; ++r) for (int cc = 0; cc < 3; ++cc) Rt[r * 3 + cc] = it.R[cc * 3 + r];
    dl_mul3(RC, Rt, tmp);
    for (int t = 0; t < 9; ++t) tmp[t] += C2[t];
    dl_inv3(tmp, M);
    double* o = mahal + size_t(i) * 9;
    for (int t = 0; t < 9; ++t) o[t] = M[t];
}

// OptimizationFunctorWithIndices::fdf over the correspondences: d = T(x) p_src - p_tgt (float), Md = M d, f += d^T Md, g_t += Md,
// dCost_dR_T += p_base_src Md^T (base_transformation_ is the identity in computeTransformation: p_base_src = p_src)
// kTerms: the 13 terms + the correspondence flag of every point go to `terms` ([14][n_pad], a column per sum) instead of into the block tree — the
// serial reference adds them one after the other in point order, and pclgicp_seqsum_kernel does exactly that
template <bool kTerms>
__global__ __launch_bounds__(256) void pclgicp_fdf_kernel(const float4* __restrict__ src, uint32_t n, const float4* __restrict__ tgt, const int32_t* __restrict__ corr,
                                                           const double* __restrict__ mahal, PclGicpIter it, double* __restrict__ partials, double* __restrict__ terms, uint32_t n_pad)
{
#pragma clang fp contract(off)
    double vals[29];
#pragma unroll
    for (int k = 0; k < 29; ++k) vals[k] = 0.0;
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) {
        const int32_t j = corr[i];
        if (j >= 0) {
            const float4 a = src[i], b = tgt[j];
            float q[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                float s = it.Tf[r * 4 + 0] * a.x;
                s = s + it.Tf[r * 4 + 1] * a.y;
                s = s + it.Tf[r * 4 + 2] * a.z;
                q[r] = s + it.Tf[r * 4 + 3];
            }
            const float  df[3] = {q[0] - b.x, q[1] - b.y, q[2] - b.z};
            const double d[3] = {df[0], df[1], df[2]};
            const double* M = mahal + size_t(i) * 9;
            const double Md[3] = {M[0] * d[0] + M[1] * d[1] + M[2] * d[2], M[3] * d[0] + M[4] * d[1] + M[5] * d[2], M[6] * d[0] + M[7] * d[1] + M[8] * d[2]};
            vals[0] = d[0] * Md[0] + d[1] * Md[1] + d[2] * Md[2];
            vals[1] = Md[0]; vals[2] = Md[1]; vals[3] = Md[2];
            const double pb[3] = {a.x, a.y, a.z};
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int cc = 0; cc < 3; ++cc) vals[4 + r * 3 + cc] = pb[r] * Md[cc];
            vals[28] = 1.0;
        }
    }
    if (kTerms) {
        if (i < n) {
#pragma unroll
            for (int k = 0; k < 13; ++k) terms[size_t(k) * n_pad + i] = vals[k];
            terms[size_t(13) * n_pad + i] = vals[28];
        }
        return;
    }
    gicp_block_reduce(vals, partials + size_t(blockIdx.x) * kGicpStride, 13);
}

// The reference's order of additions for serial pcl::GICP (registration_method "GICP"): OptimizationFunctorWithIndices adds the terms of the
// correspondences one after the other, in source-index order, into f, g.head<3>() and the 3x3 dCost_dR_T — thirteen chains of dependent f64
// additions.  A point without a correspondence contributes +0.0, which leaves a sum that started at +0.0 unchanged, so the chains run over ALL
// source points.  Lane c (c < 14; the fourteenth chain counts the correspondences) walks its column: 32 terms in flight, then 32 dependent adds.
// Not parallel and meant not to be: ~4 ns per point and evaluation, which is what bit-identity with a deterministic reference costs
// (the tree of pclgicp_fdf_kernel<false> leaves the bar on one random scene in fourteen).
__global__ __launch_bounds__(256) void pclgicp_seqsum_kernel(const double* __restrict__ terms, uint32_t n, uint32_t n_pad, double* __restrict__ out)
{
    // The chains are bound by the latency of a dependent f64 addition (~8 cycles) only if the terms are THERE: a lane streaming its column from
    // global memory waited a round trip per 32 terms (2.1 ms per 130k-point evaluation).  So the whole workgroup streams the next tile of 256 points
    // x 14 columns into LDS (coalesced, a column at a time) while lanes 0..13 of wavefront 0 add the current one out of LDS.
    constexpr int kT = 256;
    __shared__ double buf[2][14][kT];
    const int      tid = threadIdx.x;
    const uint32_t ntile = (n + kT - 1) / kT;
    double v[14];
    auto fetch = [&](uint32_t t) {
        const uint32_t i = t * kT + tid;
#pragma unroll
        for (int c = 0; c < 14; ++c) v[c] = i < n ? terms[size_t(c) * n_pad + i] : 0.0;  // (+0.0 past the end: exact to add)
    };
    auto stash = [&](int b) {
#pragma unroll
        for (int c = 0; c < 14; ++c) buf[b][c][tid] = v[c];
    };
    if (ntile) { fetch(0); stash(0); }
    __syncthreads();
    double acc = 0.0;
    for (uint32_t t = 0; t < ntile; ++t) {
        const int b = static_cast<int>(t & 1u);
        if (t + 1 < ntile) fetch(t + 1);  // in flight during the additions below
        if (tid < 14) {
            const double* __restrict__ col = buf[b][tid];
#pragma unroll 4
            for (int u = 0; u < kT; u += 8) {
                const double2 a0 = *reinterpret_cast<const double2*>(col + u), a1 = *reinterpret_cast<const double2*>(col + u + 2), a2 = *reinterpret_cast<const double2*>(col + u + 4),
                              a3 = *reinterpret_cast<const double2*>(col + u + 6);
                acc += a0.x; acc += a0.y; acc += a1.x; acc += a1.y; acc += a2.x; acc += a2.y; acc += a3.x; acc += a3.y;
            }
        }
        if (t + 1 < ntile) stash(b ^ 1);
        __syncthreads();
    }
    if (tid < 14) out[tid < 13 ? tid : 28] = acc;
    else if (tid < kGicpStride + 1 && tid - 1 != 28 && tid - 1 >= 13) out[tid - 1] = 0.0;
}

// pclomp::GICP (registration_method "GICP_OMP") accumulates into f_array / g_array / R_array[omp_get_thread_num()] inside `#pragma omp parallel for`
// and adds the per-thread partials afterwards: T chains over the static chunks of the CORRESPONDENCE list (libgomp: thread t takes the
// iterations [q t + min(t, r), ...), q = m / T, r = m mod T), then ((0 + p_0) + p_1) + ... in thread order.  The result depends on T =
// omp_get_max_threads() of the host the reference runs on; PCL_GICP_OMP_HIP reproduces it for a stated T (reg_num_threads, 8 when that is 0).
// pclgicp_chunk_bounds_kernel: the chunk boundaries in SOURCE-POINT space (a point without a correspondence adds +0.0 wherever it falls): bounds[t] =
// index of the point that holds correspondence number start_t, bounds[T] = n.  Once per outer iteration: the correspondences stay while the BFGS runs.
constexpr int kChunkMaxThreads = 16;
__global__ __launch_bounds__(256) void pclgicp_chunk_bounds_kernel(const int32_t* __restrict__ corr, uint32_t n, int T, uint32_t* __restrict__ bounds)
{
    __shared__ uint32_t lds[8];
    __shared__ uint32_t s_start[kChunkMaxThreads + 1];
    const uint32_t seg = (n + 255u) / 256u, lo = min(n, threadIdx.x * seg), hi = min(n, lo + seg);
    uint32_t mine = 0;
    for (uint32_t i = lo; i < hi; ++i) mine += corr[i] >= 0 ? 1u : 0u;
    uint32_t m;
    uint32_t rank = block_exclusive_scan<256>(mine, lds, &m);
    if (threadIdx.x <= static_cast<uint32_t>(T)) {
        const uint32_t t = threadIdx.x, q = m / static_cast<uint32_t>(T), r = m % static_cast<uint32_t>(T);
        s_start[t] = t == static_cast<uint32_t>(T) ? m : q * t + min(t, r);
        bounds[t] = n;  // chunks that start behind the last correspondence are empty
    }
    __syncthreads();
    int next = 0;
    while (next < T && s_start[next] < rank) ++next;
    for (uint32_t i = lo; i < hi; ++i) {
        if (corr[i] < 0) continue;
        while (next < T && s_start[next] == rank) { bounds[next] = i; ++next; }  // (several empty chunks may start at the same rank when m < T)
        ++rank;
    }
}
// The chains.  Thread (column c, chunk t) adds its chunk's terms of column c in point order; the workgroup streams tiles of kChunkTile / T points per
// chunk and all 14 columns into LDS (coalesced along the points) while the chains add the previous tile.  n / T dependent additions per evaluation
// instead of n: 130k points, T = 8: ~55 us (the single chain of serial pcl::GICP: ~430 us; the tree, which matches no reference: ~10 us).
constexpr int kChunkTile = 512;  // points per column and tile, over all chunks
__global__ __launch_bounds__(256) void pclgicp_chunksum_kernel(const double* __restrict__ terms, uint32_t n, uint32_t n_pad, const uint32_t* __restrict__ bounds, int T, double* __restrict__ out)
{
    __shared__ double   buf[2][14][kChunkTile];  // [column][chunk * L + j]
    __shared__ uint32_t s_b[kChunkMaxThreads + 1];
    __shared__ double   s_part[14][kChunkMaxThreads];
    const int tid = threadIdx.x, L = (kChunkTile / T) & ~3;  // points per chunk and tile: a multiple of four (the chains read double2 pairs); T L <= kChunkTile
    if (tid <= T) s_b[tid] = bounds[tid];
    __syncthreads();
    uint32_t longest = 0;
    for (int t = 0; t < T; ++t) longest = max(longest, s_b[t + 1] - s_b[t]);
    const uint32_t ntile = (longest + static_cast<uint32_t>(L) - 1u) / static_cast<uint32_t>(L);
    constexpr int kPer = 14 * kChunkTile / 256;  // elements of a tile per thread: 28
    double v[kPer];
    auto fetch = [&](uint32_t tile) {
#pragma unroll
        for (int e = 0; e < kPer; ++e) {
            const int      flat = e * 256 + tid, c = flat / kChunkTile, w = flat % kChunkTile, t = w / L, j = w % L;
            const uint32_t i = t < T ? s_b[t] + tile * static_cast<uint32_t>(L) + static_cast<uint32_t>(j) : 0xFFFFFFFFu;  // (slots behind T L stay unused)
            v[e] = (t < T && i < s_b[t + 1]) ? terms[size_t(c) * n_pad + i] : 0.0;  // (+0.0 past the chunk's end: exact to add)
        }
    };
    auto stash = [&](int b) {
#pragma unroll
        for (int e = 0; e < kPer; ++e) { const int flat = e * 256 + tid; buf[b][flat / kChunkTile][flat % kChunkTile] = v[e]; }
    };
    if (ntile) { fetch(0); stash(0); }
    __syncthreads();
    const int c = tid / T, t = tid % T;  // chain (c, t) for tid < 14 T
    double    acc = 0.0;
    for (uint32_t tile = 0; tile < ntile; ++tile) {
        const int b = static_cast<int>(tile & 1u);
        if (tile + 1 < ntile) fetch(tile + 1);
        if (tid < 14 * T) {
            const double* __restrict__ col = &buf[b][c][t * L];
            for (int u = 0; u < L; u += 4) {
                const double2 a0 = *reinterpret_cast<const double2*>(col + u), a1 = *reinterpret_cast<const double2*>(col + u + 2);
                acc += a0.x; acc += a0.y; acc += a1.x; acc += a1.y;
            }
        }
        if (tile + 1 < ntile) stash(b ^ 1);
        __syncthreads();
    }
    if (tid < 14 * T) s_part[c][t] = acc;
    __syncthreads();
    if (tid < 14) {
        double s = 0.0;
        for (int k = 0; k < T; ++k) s += s_part[tid][k];  // f = std::accumulate(f_array.begin(), f_array.end(), 0.0): thread order
        out[tid < 13 ? tid : 28] = s;
    } else if (tid < kGicpStride + 1 && tid - 1 != 28 && tid - 1 >= 13) out[tid - 1] = 0.0;
}

// linearize over the correspondences of gicp_corr_kernel
__device__ __forceinline__ void gicp_linearize_block(const float4* __restrict__ src, uint32_t n, const float4* __restrict__ tgt, const double* __restrict__ cov_src,
                                                     const double* __restrict__ cov_tgt, const GicpPose& pose, const int32_t* __restrict__ corr, double* __restrict__ mahal,
                                                     double* __restrict__ partials, uint32_t blk)
{
#pragma clang fp contract(off)
    double vals[29];
#pragma unroll
    for (int k = 0; k < 29; ++k) vals[k] = 0.0;
    const uint32_t i = blk * 256u + threadIdx.x;
    if (i < n) {
        const float4  a = load_point(src + i);
        const int32_t j = as_global(corr)[i];
        if (j >= 0) {
            const bool    voxel = tgt == nullptr;  // VGICP: cov_tgt holds the voxel records (mean 3, covariance 6, points 1) and j names a voxel
            const MRGFE_GLOBAL double* cA = as_global(cov_src) + size_t(i) * 6;
            const MRGFE_GLOBAL double* cB = voxel ? as_global(cov_tgt) + size_t(j) * kVoxRec + 3 : as_global(cov_tgt) + size_t(j) * 6;
            const double A[9] = {cA[0], cA[1], cA[2], cA[1], cA[3], cA[4], cA[2], cA[4], cA[5]};
            const double R[9] = {pose.T[0], pose.T[1], pose.T[2], pose.T[4], pose.T[5], pose.T[6], pose.T[8], pose.T[9], pose.T[10]};
            double RC[9], Rt[9], RCR[9], M[9];
            dl_mul3(R, A, RC);
            for (int r = 0; r < 3; ++r) for (int cc = 0; cc < 3; ++cc) Rt[r * 3 + cc] = R[cc * 3 + r];
            dl_mul3(RC, Rt, RCR);
            RCR[0] += cB[0]; RCR[1] += cB[1]; RCR[2] += cB[2]; RCR[3] += cB[1]; RCR[4] += cB[3]; RCR[5] += cB[4]; RCR[6] += cB[2]; RCR[7] += cB[4]; RCR[8] += cB[5];
            dl_inv3(RCR, M);
            MRGFE_GLOBAL double* mo = (MRGFE_GLOBAL double*)(mahal + size_t(i) * 9);
#pragma unroll
            for (int t = 0; t < 9; ++t) mo[t] = M[t];
            double mB[3], w = 1.0;
            if (voxel) {
                const MRGFE_GLOBAL double* v = as_global(cov_tgt) + size_t(j) * kVoxRec;
                mB[0] = v[0]; mB[1] = v[1]; mB[2] = v[2];
                w = sqrt(v[9]);
            } else {
                const float4 b = load_point(tgt + j);
                mB[0] = b.x; mB[1] = b.y; mB[2] = b.z;
            }
            const double mA[3] = {a.x, a.y, a.z};
            double tA[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) tA[r] = pose.T[r * 4 + 0] * mA[0] + pose.T[r * 4 + 1] * mA[1] + pose.T[r * 4 + 2] * mA[2] + pose.T[r * 4 + 3];
            const double err[3] = {mB[0] - tA[0], mB[1] - tA[1], mB[2] - tA[2]};
            double Me[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) Me[r] = M[r * 3 + 0] * err[0] + M[r * 3 + 1] * err[1] + M[r * 3 + 2] * err[2];
            vals[0] = err[0] * Me[0] + err[1] * Me[1] + err[2] * Me[2];
            if (voxel) {  // every term of the voxelised cost carries w = sqrt(points in the voxel)
                vals[0] = w * vals[0];
#pragma unroll
                for (int r = 0; r < 3; ++r) Me[r] = w * Me[r];
            }
            // J = [ skew(tA) | -I ], or for a right perturbation [ R skew(a) | -R ]
            double J[3][6] = {{0, -tA[2], tA[1], -1, 0, 0}, {tA[2], 0, -tA[0], 0, -1, 0}, {-tA[1], tA[0], 0, 0, 0, -1}};
            if (pose.right) {
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const double r0 = pose.T[r * 4 + 0], r1 = pose.T[r * 4 + 1], r2 = pose.T[r * 4 + 2];
                    J[r][0] = r1 * mA[2] - r2 * mA[1];
                    J[r][1] = r2 * mA[0] - r0 * mA[2];
                    J[r][2] = r0 * mA[1] - r1 * mA[0];
                    J[r][3] = -r0;
                    J[r][4] = -r1;
                    J[r][5] = -r2;
                }
            }
            double MJ[3][6];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int cc = 0; cc < 6; ++cc) MJ[r][cc] = M[r * 3 + 0] * J[0][cc] + M[r * 3 + 1] * J[1][cc] + M[r * 3 + 2] * J[2][cc];
            if (voxel) {
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int cc = 0; cc < 6; ++cc) MJ[r][cc] = w * MJ[r][cc];
            }
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                vals[1 + r] = J[0][r] * Me[0] + J[1][r] * Me[1] + J[2][r] * Me[2];
#pragma unroll
                for (int cc = r; cc < 6; ++cc) vals[7 + gidx(r, cc)] = J[0][r] * MJ[0][cc] + J[1][r] * MJ[1][cc] + J[2][r] * MJ[2][cc];
            }
            vals[28] = 1.0;
        }
    }
    gicp_block_reduce(vals, partials + size_t(blk) * kGicpStride, 28);
}

__global__ __launch_bounds__(256) void gicp_linearize_kernel(const float4* __restrict__ src, uint32_t n, const float4* __restrict__ tgt, const double* __restrict__ cov_src,
                                                              const double* __restrict__ cov_tgt, GicpPose pose, const int32_t* __restrict__ corr, double* __restrict__ mahal,
                                                              double* __restrict__ partials)
{
    gicp_linearize_block(src, n, tgt, cov_src, cov_tgt, pose, corr, mahal, partials, blockIdx.x);
}

// compute_error: stored correspondences and Mahalanobis matrices, new pose
__device__ __forceinline__ void gicp_error_block(const float4* __restrict__ src, uint32_t n, const float4* __restrict__ tgt, const GicpPose& pose, const int32_t* __restrict__ corr,
                                                 const double* __restrict__ mahal, double* __restrict__ partials, uint32_t blk, const double* __restrict__ vox = nullptr)
{
#pragma clang fp contract(off)
    double vals[29];
#pragma unroll
    for (int k = 0; k < 29; ++k) vals[k] = 0.0;
    const uint32_t i = blk * 256u + threadIdx.x;
    if (i < n) {
        const int32_t j = as_global(corr)[i];
        if (j >= 0) {
            const float4  a = load_point(src + i);
            double mB[3], w = 1.0;
            if (vox) {
                const MRGFE_GLOBAL double* v = as_global(vox) + size_t(j) * kVoxRec;
                mB[0] = v[0]; mB[1] = v[1]; mB[2] = v[2];
                w = sqrt(v[9]);
            } else {
                const float4 b = load_point(tgt + j);
                mB[0] = b.x; mB[1] = b.y; mB[2] = b.z;
            }
            const MRGFE_GLOBAL double* M = as_global(mahal) + size_t(i) * 9;
            const double  mA[3] = {a.x, a.y, a.z};
            double tA[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) tA[r] = pose.T[r * 4 + 0] * mA[0] + pose.T[r * 4 + 1] * mA[1] + pose.T[r * 4 + 2] * mA[2] + pose.T[r * 4 + 3];
            const double err[3] = {mB[0] - tA[0], mB[1] - tA[1], mB[2] - tA[2]};
            double Me[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) Me[r] = M[r * 3 + 0] * err[0] + M[r * 3 + 1] * err[1] + M[r * 3 + 2] * err[2];
            vals[0] = err[0] * Me[0] + err[1] * Me[1] + err[2] * Me[2];
            if (vox) vals[0] = w * vals[0];
            vals[28] = 1.0;
        }
    }
    gicp_block_reduce(vals, partials + size_t(blk) * kGicpStride, 1);
}

__global__ __launch_bounds__(256) void gicp_error_kernel(const float4* __restrict__ src, uint32_t n, const float4* __restrict__ tgt, GicpPose pose, const int32_t* __restrict__ corr,
                                                          const double* __restrict__ mahal, double* __restrict__ partials, const double* __restrict__ vox)
{
    gicp_error_block(src, n, tgt, pose, corr, mahal, partials, blockIdx.x, vox);
}

__device__ __forceinline__ void gicp_reduce_record(const double* __restrict__ partials, uint32_t nblk, double* __restrict__ out)
{
    __shared__ double s[8][kGicpStride];
    const int k = threadIdx.x & 31, slice = threadIdx.x >> 5;
    double    acc = 0.0;
    // eight of the slice's records in flight, then added in the same order as one after the other (the loop used to wait for every load in
    // turn: 64 round trips = 17 us per reduction, four of them in a 130k-point frame)
    uint32_t b = slice;
    for (; b + 56 < nblk; b += 64) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = partials[size_t(b + 8 * u) * kGicpStride + k];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; b < nblk; b += 8) acc += partials[size_t(b) * kGicpStride + k];
    s[slice][k] = acc;
    __syncthreads();
    if (threadIdx.x < kGicpStride) {
        double r = s[0][k];
#pragma unroll
        for (int sl = 1; sl < 8; ++sl) r += s[sl][k];
        out[k] = r;
    }
}

__global__ __launch_bounds__(256) void gicp_reduce_kernel(const double* __restrict__ partials, uint32_t nblk, double* __restrict__ out) { gicp_reduce_record(partials, nblk, out); }
// the same record written straight into pinned HOST memory, its last slot = `tag` once the rest is visible: the single registration's LM loop
// polls that slot instead of queueing a device-to-host copy and waiting for the stream (a copy command and a wake-up less per trial)
__global__ __launch_bounds__(256) void gicp_reduce_host_kernel(const double* __restrict__ partials, uint32_t nblk, double* __restrict__ h_out, double tag)
{
    gicp_reduce_record(partials, nblk, h_out);  // writes slots 0 .. 31 (29 .. 31 are zero padding)
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence_system();
        __hip_atomic_store(&h_out[kGicpStride], tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ---- batched variants: blockIdx.y = the y-th busy pair of this kernel in the round --------------------------------------
struct GicpPairDev {  // static per pair
    const float4* src;
    const float4* tgt;
    const double* cov_src;
    const double* cov_tgt;
    int32_t*      corr;
    double*       mahal;
    uint32_t      n;
    uint32_t      part_off;  // first block-partial record of this pair
    uint32_t      target;    // index into the grid array
    uint32_t      pad;
};
struct GicpEvalDev {  // per round
    GicpPose pose;
    double   thr2;
    uint32_t order[2];  // entry k: the k-th pair with a linearize / an error request this round
    int32_t  type;      // 0 linearize, 1 error, -1 idle
    int32_t  pad;
};

__global__ __launch_bounds__(256) void gicp_corr_batch_kernel(const GicpPairDev* __restrict__ pairs, const GicpEvalDev* __restrict__ evals, const NnGrid2Dev* __restrict__ grids)
{
    __shared__ NnGrid2Dev s_grid;
    const uint32_t     pi = evals[blockIdx.y].order[0];
    const GicpPairDev  pr = pairs[pi];
    if (blockIdx.x * (256u / kGicpBatchGroup) >= pr.n) return;  // uniform
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(grids + pr.target);
        uint32_t*       dst = reinterpret_cast<uint32_t*>(&s_grid);
        for (uint32_t w = threadIdx.x; w < sizeof(NnGrid2Dev) / 4; w += 256) dst[w] = src[w];
    }
    __syncthreads();
    const GicpEvalDev& ev = evals[pi];
    gicp_corr_query<kGicpBatchGroup>(s_grid, pr.src, pr.n, ev.pose, ev.thr2, pr.corr, blockIdx.x);
}

__global__ __launch_bounds__(256) void vox_corr_batch_kernel(const GicpPairDev* __restrict__ pairs, const GicpEvalDev* __restrict__ evals, const VoxGridDev* __restrict__ vgrids)
{
#pragma clang fp contract(off)
    const uint32_t    pi = evals[blockIdx.y].order[0];
    const GicpPairDev pr = pairs[pi];
    const uint32_t    i = blockIdx.x * 256u + threadIdx.x;
    if (i >= pr.n) return;
    const VoxGridDev g = vgrids[pr.target];
    const double*    T = evals[pi].pose.T;
    const float4     a = load_point(pr.src + i);
    const double     mA[3] = {a.x, a.y, a.z};
    double tA[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) tA[r] = T[r * 4 + 0] * mA[0] + T[r * 4 + 1] * mA[1] + T[r * 4 + 2] * mA[2] + T[r * 4 + 3];
    const uint32_t c = vox_cell(g, tA[0], tA[1], tA[2]);
    ((MRGFE_GLOBAL int32_t*)pr.corr)[i] = (c < g.n_cells && as_global(g.vox)[size_t(c) * kVoxRec + 9] > 0.0) ? static_cast<int32_t>(c) : -1;
}

__global__ __launch_bounds__(256) void gicp_linearize_batch_kernel(const GicpPairDev* __restrict__ pairs, const GicpEvalDev* __restrict__ evals, double* __restrict__ partials)
{
    const uint32_t    pi = evals[blockIdx.y].order[0];
    const GicpPairDev pr = pairs[pi];
    if (blockIdx.x * 256u >= pr.n) return;
    gicp_linearize_block(pr.src, pr.n, pr.tgt, pr.cov_src, pr.cov_tgt, evals[pi].pose, pr.corr, pr.mahal, partials + size_t(pr.part_off) * kGicpStride, blockIdx.x);
}

__global__ __launch_bounds__(256) void gicp_error_batch_kernel(const GicpPairDev* __restrict__ pairs, const GicpEvalDev* __restrict__ evals, double* __restrict__ partials)
{
    const uint32_t    pi = evals[blockIdx.y].order[1];
    const GicpPairDev pr = pairs[pi];
    if (blockIdx.x * 256u >= pr.n) return;
    gicp_error_block(pr.src, pr.n, pr.tgt, evals[pi].pose, pr.corr, pr.mahal, partials + size_t(pr.part_off) * kGicpStride, blockIdx.x, pr.tgt ? nullptr : pr.cov_tgt);
}

// one workgroup per pair: the same fixed-order sum as gicp_reduce_kernel, written to (pinned host) results[pair][32]
__global__ __launch_bounds__(256) void gicp_reduce_batch_kernel(const GicpPairDev* __restrict__ pairs, const GicpEvalDev* __restrict__ evals, const double* __restrict__ partials,
                                                                 double* __restrict__ results)
{
    if (evals[blockIdx.x].type < 0) return;
    const GicpPairDev pr = pairs[blockIdx.x];
    gicp_reduce_record(partials + size_t(pr.part_off) * kGicpStride, (pr.n + 255u) / 256u, results + size_t(blockIdx.x) * kGicpStride);
}

// ------------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------------
namespace {

void so3_exp_matrix(const double w[3], double R[9])
{
    const double theta_sq = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    double imag, real;
    if (theta_sq < 1e-10) {
        const double quad = theta_sq * theta_sq;
        imag = 0.5 - 1.0 / 48.0 * theta_sq + 1.0 / 3840.0 * quad;
        real = 1.0 - 1.0 / 8.0 * theta_sq + 1.0 / 384.0 * quad;
    } else {
        const double theta = std::sqrt(theta_sq), half = 0.5 * theta;
        imag = std::sin(half) / theta;
        real = std::cos(half);
    }
    const double qw = real, qx = imag * w[0], qy = imag * w[1], qz = imag * w[2];
    const double tx = 2 * qx, ty = 2 * qy, tz = 2 * qz;
    const double twx = tx * qw, twy = ty * qw, twz = tz * qw, txx = tx * qx, txy = ty * qx, txz = tz * qx, tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

void se3_exp(const double a[6], double T[16])
{
    const double w[3] = {a[0], a[1], a[2]};
    const double theta = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    double R[9];
    so3_exp_matrix(w, R);
    const double Om[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
    double Om2[9], V[9];
    dl_mul3(Om, Om, Om2);
    if (theta < 1e-10) {
        std::memcpy(V, R, sizeof(V));
    } else {
        const double tsq = theta * theta;
        const double c1 = (1.0 - std::cos(theta)) / tsq, c2 = (theta - std::sin(theta)) / (tsq * theta);
        for (int t = 0; t < 9; ++t) V[t] = ((t % 4 == 0) ? 1.0 : 0.0) + c1 * Om[t] + c2 * Om2[t];
    }
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) T[r * 4 + c] = R[r * 3 + c];
        T[r * 4 + 3] = V[r * 3 + 0] * a[3] + V[r * 3 + 1] * a[4] + V[r * 3 + 2] * a[5];
    }
    T[12] = T[13] = T[14] = 0;
    T[15] = 1;
}

void mul4(const double A[16], const double B[16], double o[16])
{
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) { double s = 0; for (int k = 0; k < 4; ++k) s += A[r * 4 + k] * B[k * 4 + c]; o[r * 4 + c] = s; }
}

// symmetric positive (semi) definite solve through LDL^T without pivoting, falling back to pivoted elimination
void solve6(const double A_in[36], const double rhs[6], double x[6])
{
    double A[6][7];
    for (int r = 0; r < 6; ++r) { for (int c = 0; c < 6; ++c) A[r][c] = A_in[r * 6 + c]; A[r][6] = rhs[r]; }
    for (int k = 0; k < 6; ++k) {
        int piv = k;
        for (int r = k + 1; r < 6; ++r) if (std::fabs(A[r][k]) > std::fabs(A[piv][k])) piv = r;
        if (piv != k) for (int c = 0; c < 7; ++c) std::swap(A[k][c], A[piv][c]);
        for (int r = k + 1; r < 6; ++r) {
            const double f = A[r][k] / A[k][k];
            for (int c = k; c < 7; ++c) A[r][c] -= f * A[k][c];
        }
    }
    for (int r = 5; r >= 0; --r) {
        double s = A[r][6];
        for (int c = r + 1; c < 6; ++c) s -= A[r][c] * x[c];
        x[r] = s / A[r][r];
    }
}

bool is_converged(const double d[16], double rot_eps, double trans_eps)
{
    double mx = 0;
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) mx = std::max(mx, 1.0 / rot_eps * std::fabs(d[r * 4 + c] - (r == c ? 1.0 : 0.0)));
        mx = std::max(mx, 1.0 / trans_eps * std::fabs(d[r * 4 + 3]));
    }
    return mx < 1;
}

GicpPose make_pose(const double T[16], int variant)
{
    GicpPose p;
    for (int i = 0; i < 12; ++i) { p.T[i] = T[i]; p.Tf[i] = static_cast<float>(T[i]); }
    p.right = variant == 1 ? 1 : 0;
    p.pad = 0;
    return p;
}

}  // namespace

GicpEngine::~GicpEngine()
{
    if (ctx_) (void)hipSetDevice(ctx_->device);
    tgt_grid_.release();
    cov_grid_.release();
    cur_grid_.release();
    d_knn_i_.release(); d_knn_d_.release();
    d_tgt_cov_.release(); d_src_cov_.release(); d_corr_.release(); d_mahal_.release(); d_partial_.release(); d_T_.release();
    d_vox_.release(); d_vox_runs_.release(); d_cur_.release();
    h_rec_.release();
    d_terms_.release();
    d_chunk_bounds_.release();
}

int GicpEngine::set_target(const void* d, size_t n)
{
    d_tgt_ = static_cast<const float4*>(d);
    n_tgt_ = n;
    tgt_grid_valid_ = tgt_cov_valid_ = false;
    return MRGFE_OK;
}
int GicpEngine::set_source(const void* d, size_t n, const float* enclosing_box)
{
    d_src_ = static_cast<const float4*>(d);
    n_src_ = n;
    src_cov_valid_ = false;
    src_box_valid_ = enclosing_box != nullptr;
    if (enclosing_box) std::memcpy(src_box_, enclosing_box, sizeof(src_box_));
    return MRGFE_OK;
}
int GicpEngine::source_becomes_target()
{
    if (!src_cov_valid_ || n_src_ == 0) return set_target(d_src_, n_src_);  // nothing computed yet (an empty source has no grid): an ordinary target, prepared by the next align
    std::swap(tgt_grid_, cov_grid_);  // (plain structs of device pointers: the buffers travel with them, the old target's are reused for the next source)
    std::swap(d_tgt_cov_, d_src_cov_);
    d_tgt_ = d_src_;
    n_tgt_ = n_src_;
    tgt_grid_valid_ = tgt_cov_valid_ = true;
    vox_valid_ = false;
    src_cov_valid_ = false;
    return MRGFE_OK;
}

// k-NN covariances of one cloud on ctx's stream, through the caller's grid and neighbour buffers
// Correspondence search of GICP_HIP / SMALL_GICP_HIP: one lane group per query that walks until its answer is final (gicp_corr_kernel), or the
// passes of getFitnessScore carrying the index (nn_nearest_batch: block / seed / sweep / pyramid walk), which sort the queries by the work they
// still need.  Measured: a batch of 32 x 130k queries 5.6 -> 3.8 ms per call (three rounds), but ONE cloud is slower through the passes
// (130k queries: frame 1.68 -> 1.85 ms, 33k: 1.86 -> 2.13 ms — four launches, two small copies and a stream fork against one launch that
// already fills the chip with eight lanes per query).  Mode 1 (default): the passes for batches of at least kCorrPassMinQueries queries;
// 0: never; 2: always (tests).
static std::atomic<int> g_corr_passes{-1};
constexpr size_t kCorrPassMinQueries = 400000;
static int gicp_corr_mode()
{
    int m = g_corr_passes.load(std::memory_order_relaxed);
    if (m < 0) {
        const char* e = std::getenv("MRGFE_GICP_CORR_PASSES");
        m = e ? std::max(0, std::min(2, std::atoi(e))) : 1;
        g_corr_passes.store(m, std::memory_order_relaxed);
    }
    return m;
}
static bool gicp_corr_passes(size_t queries) { return gicp_corr_mode() == 2 || (gicp_corr_mode() == 1 && queries >= kCorrPassMinQueries); }
int gicp_set_corr_passes(int mode)
{
    g_corr_passes.store(std::max(0, std::min(2, mode)), std::memory_order_relaxed);
    return MRGFE_OK;
}

int gicp_covariances_on_grid(mrgfe_ctx* ctx, int k, const float4* d_pts, size_t n, DevBuf& out, NnGrid& grid, DevBuf& knn_i, DevBuf& knn_d, bool pcl_moments);
int gicp_compute_covariances(mrgfe_ctx* ctx, int k, const float4* d_pts, size_t n, DevBuf& out, NnGrid& grid, DevBuf& knn_i, DevBuf& knn_d, bool pcl_moments, const float* known_box)
{
    MRGFE_TRY(out.ensure(std::max<size_t>(n, 1) * 48));
    if (n == 0) return MRGFE_OK;
    MRGFE_TRY(grid.build(ctx, d_pts, n, 1.0f, NnGrid::kCrowdingKnn, kNnMaxLevels, known_box));
    return gicp_covariances_on_grid(ctx, k, d_pts, n, out, grid, knn_i, knn_d, pcl_moments);
}

// the k-NN search and the covariance kernel on a grid that is already built over d_pts
int gicp_covariances_on_grid(mrgfe_ctx* ctx, int k, const float4* d_pts, size_t n, DevBuf& out, NnGrid& grid, DevBuf& knn_i, DevBuf& knn_d, bool pcl_moments)
{
    MRGFE_TRY(out.ensure(std::max<size_t>(n, 1) * 48));
    if (n == 0) return MRGFE_OK;
    MRGFE_TRY(knn_i.ensure(n * k * 4));
    MRGFE_TRY(knn_d.ensure(n * k * 4));
    MRGFE_TRY(grid.knn_device(ctx, d_pts, n, k, knn_i.as<int32_t>(), knn_d.as<float>()));
    const uint32_t nn = static_cast<uint32_t>(n);
    if (pcl_moments) hipLaunchKernelGGL(pclgicp_cov_kernel, dim3((nn + 255) / 256), dim3(256), 0, ctx->stream, d_pts, nn, knn_i.as<int32_t>(), k, 1e-3 /* gicp_epsilon_ */, out.as<double>());
    else             hipLaunchKernelGGL(gicp_cov_kernel, dim3((nn + 255) / 256), dim3(256), 0, ctx->stream, d_pts, nn, knn_i.as<int32_t>(), k, out.as<double>());
    MRGFE_HIP_CHECK(hipGetLastError());
    return MRGFE_OK;
}

int GicpEngine::compute_covariances(const float4* d_pts, size_t n, DevBuf& out, NnGrid& grid)
{
    // (the source's enclosing box, when its producer handed one over: set_source)
    const float* box = (d_pts == d_src_ && n == n_src_ && src_box_valid_) ? src_box_ : nullptr;
    return gicp_compute_covariances(ctx_, prm_.k_correspondences, d_pts, n, out, grid, d_knn_i_, d_knn_d_, prm_.variant == 4, box);
}

void GicpEngine::voxel_grid(double* res, int32_t cmin[3], int32_t dim[3], uint32_t* n_cells) const
{
    *res = vox_res_;
    for (int a = 0; a < 3; ++a) { cmin[a] = vox_cmin_[a]; dim[a] = vox_dim_[a]; }
    *n_cells = vox_cells_;
}

// fast_gicp::GaussianVoxelMap::create_voxelmap over the target and its covariances
int GicpEngine::build_voxelmap()
{
    vox_valid_ = false;
    vox_res_ = prm_.voxel_resolution;
    vox_cells_ = vox_occupied_ = 0;
    for (int a = 0; a < 3; ++a) { vox_cmin_[a] = 0; vox_dim_[a] = 1; }
    if (!(vox_res_ > 0)) { set_error("VGICP: resolution must be > 0"); return MRGFE_ERR_INVALID; }
    hipStream_t st = ctx_->stream;
    if (n_tgt_ == 0) {
        MRGFE_TRY(d_vox_.ensure(sizeof(double) * kVoxRec));
        MRGFE_HIP_CHECK(hipMemsetAsync(d_vox_.p, 0, sizeof(double) * kVoxRec, st));
        vox_cells_ = 1;
        vox_valid_ = true;
        return MRGFE_OK;
    }
    if (n_tgt_ > 0x7fffffffu) { set_error("VGICP: cloud too large"); return MRGFE_ERR_INVALID; }
    const uint32_t nn = static_cast<uint32_t>(n_tgt_);
    SliceTable tab;
    tab.build(&nn, 1);
    DevBuf &ds = ctx_->scratch[0], &dbb = ctx_->scratch[1], &dk = ctx_->scratch[2], &dv = ctx_->scratch[3], &dkt = ctx_->scratch[4], &dvt = ctx_->scratch[5], &dh = ctx_->scratch[6];
    MRGFE_TRY(ds.ensure(sizeof(Slice) * 2 + sizeof(void*)));
    const void* cp = d_tgt_;
    MRGFE_HIP_CHECK(hipMemcpyAsync(ds.p, tab.h.data(), sizeof(Slice), hipMemcpyHostToDevice, st));
    MRGFE_HIP_CHECK(hipMemcpyAsync(ds.as<char>() + 2 * sizeof(Slice), &cp, sizeof(void*), hipMemcpyHostToDevice, st));
    MRGFE_TRY(dbb.ensure(sizeof(BBox) * (tab.total_blks + 1)));
    BBox* d_part = dbb.as<BBox>();
    BBox* d_out = d_part + tab.total_blks;
    MRGFE_TRY(bounding_boxes(ctx_, reinterpret_cast<const float4* const*>(ds.as<char>() + 2 * sizeof(Slice)), ds.as<Slice>(), tab, d_part, d_out));
    BBox bb;
    MRGFE_HIP_CHECK(hipMemcpyAsync(&bb, d_out, sizeof(BBox), hipMemcpyDeviceToHost, st));
    MRGFE_HIP_CHECK(hipStreamSynchronize(st));
    if (bb.n_finite == 0) {
        MRGFE_TRY(d_vox_.ensure(sizeof(double) * kVoxRec));
        MRGFE_HIP_CHECK(hipMemsetAsync(d_vox_.p, 0, sizeof(double) * kVoxRec, st));
        vox_cells_ = 1;
        vox_valid_ = true;
        return MRGFE_OK;
    }
    double cells = 1;
    for (int a = 0; a < 3; ++a) {  // the voxel coordinate is monotonic in x: the corner voxels come from the bounding box
        const int lo = static_cast<int>(std::floor(static_cast<double>(bb.mn[a]) / vox_res_ - 0.5)), hi = static_cast<int>(std::floor(static_cast<double>(bb.mx[a]) / vox_res_ - 0.5));
        vox_cmin_[a] = lo;
        vox_dim_[a] = hi - lo + 1;
        cells *= static_cast<double>(vox_dim_[a]);
    }
    if (cells > double(1u << 24)) { set_error("VGICP: the target needs %.0f voxels of %.3f m (more than 2^24)", cells, vox_res_); return MRGFE_ERR_OVERFLOW; }
    vox_cells_ = static_cast<uint32_t>(cells);
    MRGFE_TRY(dk.ensure(n_tgt_ * 4)); MRGFE_TRY(dv.ensure(n_tgt_ * 4)); MRGFE_TRY(dkt.ensure(n_tgt_ * 4)); MRGFE_TRY(dvt.ensure(n_tgt_ * 4));
    MRGFE_TRY(dh.ensure(sizeof(uint32_t) * 256 * (tab.total_blks + 1)));
    MRGFE_TRY(d_vox_.ensure(sizeof(double) * kVoxRec * size_t(vox_cells_)));
    MRGFE_TRY(d_vox_runs_.ensure(sizeof(uint32_t) * 2 * size_t(vox_cells_)));
    VoxGridDev g;
    g.res = vox_res_;
    for (int a = 0; a < 3; ++a) { g.cmin[a] = vox_cmin_[a]; g.dim[a] = vox_dim_[a]; }
    g.n_cells = vox_cells_;
    g.vox = d_vox_.as<double>();
    hipLaunchKernelGGL(vox_key_kernel, dim3((nn + 255) / 256), dim3(256), 0, st, d_tgt_, nn, g, dk.as<uint32_t>(), dv.as<uint32_t>());
    int key_bits = 1;
    while (key_bits < 32 && (uint64_t(1) << key_bits) <= vox_cells_) ++key_bits;
    uint32_t *sk, *sv;
    MRGFE_TRY(radix_sort_pairs(ctx_, dk.as<uint32_t>(), dv.as<uint32_t>(), dkt.as<uint32_t>(), dvt.as<uint32_t>(), ds.as<Slice>(), tab, key_bits, dh.as<uint32_t>(), &sk, &sv));
    uint32_t* d_first = d_vox_runs_.as<uint32_t>();
    uint32_t* d_last = d_first + vox_cells_;
    MRGFE_HIP_CHECK(hipMemsetAsync(d_first, 0xff, sizeof(uint32_t) * size_t(vox_cells_), st));
    hipLaunchKernelGGL(vox_runs_kernel, dim3((nn + 255) / 256), dim3(256), 0, st, sk, nn, vox_cells_, d_first, d_last);
    hipLaunchKernelGGL(vox_sums_kernel, dim3((vox_cells_ + 255) / 256), dim3(256), 0, st, d_tgt_, d_tgt_cov_.as<double>(), sv, d_first, d_last, vox_cells_, d_vox_.as<double>());
    MRGFE_HIP_CHECK(hipGetLastError());
    MRGFE_HIP_CHECK(hipStreamSynchronize(st));  // tab's host table was the source of an async copy
    vox_valid_ = true;
    return MRGFE_OK;
}

int GicpEngine::prepare_target()
{
    if (!d_tgt_ && n_tgt_) { set_error("GICP: no target"); return MRGFE_ERR_STATE; }
    MRGFE_TRY(ctx_->bind());
    // the target's k-NN grid also serves the per-iteration correspondence search (one build per setInputTarget)
    if (!tgt_cov_valid_ || !tgt_grid_valid_) {
        MRGFE_TRY(compute_covariances(d_tgt_, n_tgt_, d_tgt_cov_, tgt_grid_));
        if (n_tgt_ == 0) MRGFE_TRY(tgt_grid_.build(ctx_, d_tgt_, 0, 1.0f));
        tgt_cov_valid_ = tgt_grid_valid_ = true;
        vox_valid_ = false;
    }
    if (prm_.variant == 2 && !vox_valid_) MRGFE_TRY(build_voxelmap());
    return MRGFE_OK;
}

int GicpEngine::ensure_ready()
{
    if (!d_tgt_ && n_tgt_) { set_error("GICP: no target"); return MRGFE_ERR_STATE; }
    MRGFE_TRY(ctx_->bind());
    if (!src_cov_valid_) { MRGFE_TRY(compute_covariances(d_src_, n_src_, d_src_cov_, cov_grid_)); src_cov_valid_ = true; }
    MRGFE_TRY(prepare_target());
    const size_t ns = std::max<size_t>(n_src_, 1);
    MRGFE_TRY(d_corr_.ensure(ns * 4));
    MRGFE_TRY(d_mahal_.ensure(ns * 72));
    MRGFE_TRY(d_partial_.ensure(sizeof(double) * kGicpStride * ((ns + 255) / 256 + 1)));
    return MRGFE_OK;
}

int GicpEngine::covariances(int which, double* out9)
{
    MRGFE_TRY(ensure_ready());
    if (!out9) return MRGFE_OK;  // (callers that only want the covariances, grid and buffers in place)
    const size_t n = which == 0 ? n_src_ : n_tgt_;
    std::vector<double> c6(n * 6);
    if (n) MRGFE_HIP_CHECK(hipMemcpy(c6.data(), (which == 0 ? d_src_cov_ : d_tgt_cov_).p, n * 48, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; ++i) {
        const double* c = &c6[i * 6];
        double* o = out9 + i * 9;
        o[0] = c[0]; o[1] = c[1]; o[2] = c[2]; o[3] = c[1]; o[4] = c[3]; o[5] = c[4]; o[6] = c[2]; o[7] = c[4]; o[8] = c[5];
    }
    return MRGFE_OK;
}

// wait for gicp_reduce_host_kernel's record: poll the tag in pinned memory; the stream is asked now and then so that a failed launch cannot hang the caller
static int gicp_wait_record(hipStream_t st, const volatile double* h_rec, double tag)
{
    uint64_t want;
    memcpy(&want, &tag, sizeof(want));
    const volatile uint64_t* p = reinterpret_cast<const volatile uint64_t*>(&h_rec[kGicpStride]);
    return poll_host_record(st, [&] { return __atomic_load_n(p, __ATOMIC_ACQUIRE) == want; }, "GICP reduction");
}

int GicpEngine::run_linearize(const double T[16], bool, double H[36], double b[6], double* err, int* n_corr)
{
    ++n_linearize_;
    *err = 0;
    if (n_corr) *n_corr = 0;
    for (int t = 0; t < 36; ++t) H[t] = 0;
    for (int t = 0; t < 6; ++t) b[t] = 0;
    if (n_src_ == 0) return MRGFE_OK;
    hipStream_t    st = ctx_->stream;
    const uint32_t n = static_cast<uint32_t>(n_src_), nblk = (n + 255) / 256;
    double* d_part = d_partial_.as<double>();
    double* d_res = d_part + size_t(nblk) * kGicpStride;
    MRGFE_HIP_CHECK(hipEventRecord(ctx_->ev0, st));
    const GicpPose pose = make_pose(T, prm_.variant);
    constexpr uint32_t per_blk = 256u / kGicpGroup;
    if (prm_.variant == 2) {
        VoxGridDev g;
        g.res = vox_res_;
        for (int a = 0; a < 3; ++a) { g.cmin[a] = vox_cmin_[a]; g.dim[a] = vox_dim_[a]; }
        g.n_cells = vox_cells_;
        g.vox = d_vox_.as<double>();
        MRGFE_TRY(d_T_.ensure(sizeof(double) * 12));
        MRGFE_HIP_CHECK(hipMemcpyAsync(d_T_.p, pose.T, sizeof(double) * 12, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(vox_corr_kernel, dim3(nblk), dim3(256), 0, st, d_src_, n, g, d_T_.as<double>(), d_corr_.as<int32_t>());
        hipLaunchKernelGGL(gicp_linearize_kernel, dim3(nblk), dim3(256), 0, st, d_src_, n, static_cast<const float4*>(nullptr), d_src_cov_.as<double>(), d_vox_.as<double>(), pose,
                           d_corr_.as<int32_t>(), d_mahal_.as<double>(), d_part);
    } else {
        if (gicp_corr_passes(n)) {  // (a single cloud: only when forced, see gicp_corr_mode)
            NnFitnessJob job = tgt_grid_.make_fitness_job(d_src_, n, pose.Tf);
            std::memcpy(job.T12, pose.Tf, sizeof(job.T12));
            job.gicp_order = 1;
            job.idx_out = d_corr_.as<int32_t>();
            MRGFE_TRY(nn_nearest_batch(ctx_, &job, 1, prm_.max_corr_dist * prm_.max_corr_dist));
        } else {
            if (n < kGicpWideGroupBelow)
                hipLaunchKernelGGL(gicp_corr_kernel<16>, dim3((n + 15) / 16), dim3(256), 0, st, tgt_grid_.dev2(), d_src_, n, pose, prm_.max_corr_dist * prm_.max_corr_dist, d_corr_.as<int32_t>());
            else
                hipLaunchKernelGGL(gicp_corr_kernel<kGicpGroup>, dim3((n + per_blk - 1) / per_blk), dim3(256), 0, st, tgt_grid_.dev2(), d_src_, n, pose, prm_.max_corr_dist * prm_.max_corr_dist, d_corr_.as<int32_t>());
        }
        hipLaunchKernelGGL(gicp_linearize_kernel, dim3(nblk), dim3(256), 0, st, d_src_, n, d_tgt_, d_src_cov_.as<double>(), d_tgt_cov_.as<double>(), pose, d_corr_.as<int32_t>(),
                           d_mahal_.as<double>(), d_part);
    }
    MRGFE_HIP_CHECK(hipEventRecord(ctx_->ev1, st));
    (void)d_res;
    MRGFE_TRY(h_rec_.ensure(sizeof(double) * (kGicpStride + 1)));
    const double tag = static_cast<double>(++rec_tag_);
    h_rec_.as<double>()[kGicpStride] = 0.0;  // (nothing is in flight: the previous record was waited for)
    hipLaunchKernelGGL(gicp_reduce_host_kernel, dim3(1), dim3(256), 0, st, d_part, nblk, h_rec_.as<double>(), tag);
    MRGFE_HIP_CHECK(hipGetLastError());
    MRGFE_TRY(gicp_wait_record(st, h_rec_.as<double>(), tag));
    double r[kGicpStride];
    std::memcpy(r, h_rec_.p, sizeof(r));
    MRGFE_HIP_CHECK(hipEventSynchronize(ctx_->ev1));  // (long passed: the reduction ran behind it)
    float ms = 0;
    MRGFE_HIP_CHECK(hipEventElapsedTime(&ms, ctx_->ev0, ctx_->ev1));
    kernel_ms += ms;
    kernel_launches += 1;
    // SURVEY.md §8(d) GICP byte model with the measured correspondence count: src point + cov + probes + matched target point + cov
    kernel_alg_bytes += double(n) * (16 + 48 + 27 * 8) + r[28] * (16 + 48);
    *err = r[0];
    for (int t = 0; t < 6; ++t) b[t] = r[1 + t];
    int t = 7;
    for (int i = 0; i < 6; ++i)
        for (int j = i; j < 6; ++j) { H[i * 6 + j] = r[t]; H[j * 6 + i] = r[t]; ++t; }
    if (n_corr) *n_corr = static_cast<int>(r[28]);
    return MRGFE_OK;
}

int GicpEngine::run_error(const double T[16], double* err)
{
    ++n_error_;
    *err = 0;
    if (n_src_ == 0) return MRGFE_OK;
    hipStream_t    st = ctx_->stream;
    const uint32_t n = static_cast<uint32_t>(n_src_), nblk = (n + 255) / 256;
    double* d_part = d_partial_.as<double>();
    double* d_res = d_part + size_t(nblk) * kGicpStride;
    hipLaunchKernelGGL(gicp_error_kernel, dim3(nblk), dim3(256), 0, st, d_src_, n, d_tgt_, make_pose(T, prm_.variant), d_corr_.as<int32_t>(), d_mahal_.as<double>(), d_part,
                       prm_.variant == 2 ? d_vox_.as<const double>() : static_cast<const double*>(nullptr));
    (void)d_res;
    MRGFE_TRY(h_rec_.ensure(sizeof(double) * (kGicpStride + 1)));
    const double tag = static_cast<double>(++rec_tag_);
    h_rec_.as<double>()[kGicpStride] = 0.0;  // (nothing is in flight: the previous record was waited for)
    hipLaunchKernelGGL(gicp_reduce_host_kernel, dim3(1), dim3(256), 0, st, d_part, nblk, h_rec_.as<double>(), tag);
    MRGFE_HIP_CHECK(hipGetLastError());
    MRGFE_TRY(gicp_wait_record(st, h_rec_.as<double>(), tag));
    *err = h_rec_.as<double>()[0];
    return MRGFE_OK;
}

int GicpEngine::linearize(const double T[16], double H[36], double b[6], double* err, int* n_corr)
{
    MRGFE_TRY(ensure_ready());
    return run_linearize(T, true, H, b, err, n_corr);
}

namespace {
// Rotation of the Umeyama / Kabsch problem, R = U diag(1, 1, det(U) det(V)) V^T for sigma = U S V^T: one-sided (Hestenes)
// Jacobi SVD in f64; a vanishing singular direction is completed by the cross product of the other two.
void umeyama_rotation(const double sigma[9], double R[9])
{
    double W[3][3], V[3][3];  // W = sigma V converges to U S
    double scale = 0;
    for (int i = 0; i < 9; ++i) scale = std::max(scale, std::fabs(sigma[i]));
    if (!(scale > 0) || !std::isfinite(scale)) { for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0) ? 1.0 : 0.0; return; }
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { W[r][c] = sigma[r * 3 + c] / scale; V[r][c] = r == c ? 1.0 : 0.0; }
    for (int sweep = 0; sweep < 60; ++sweep) {
        bool rotated = false;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double alpha = 0, beta = 0, gamma = 0;
                for (int k = 0; k < 3; ++k) { alpha += W[k][p] * W[k][p]; beta += W[k][q] * W[k][q]; gamma += W[k][p] * W[k][q]; }
                if (gamma == 0.0 || std::fabs(gamma) <= 1e-15 * std::sqrt(alpha * beta)) continue;
                rotated = true;
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / std::sqrt(1.0 + t * t), s = c * t;
                for (int k = 0; k < 3; ++k) {
                    const double wp = W[k][p], wq = W[k][q];
                    W[k][p] = c * wp - s * wq; W[k][q] = s * wp + c * wq;
                    const double vp = V[k][p], vq = V[k][q];
                    V[k][p] = c * vp - s * vq; V[k][q] = s * vp + c * vq;
                }
            }
        if (!rotated) break;
    }
    double sig[3], U[3][3];
    int    order[3] = {0, 1, 2};
    for (int j = 0; j < 3; ++j) sig[j] = std::sqrt(W[0][j] * W[0][j] + W[1][j] * W[1][j] + W[2][j] * W[2][j]);
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2 - a; ++b) if (sig[order[b]] < sig[order[b + 1]]) std::swap(order[b], order[b + 1]);  // descending
    const double tiny = 1e-12 * sig[order[0]];
    for (int jj = 0; jj < 3; ++jj) {
        const int j = order[jj];
        if (sig[j] > tiny) { for (int k = 0; k < 3; ++k) U[k][j] = W[k][j] / sig[j]; }
        else if (jj == 2) {  // rank 2: complete the basis
            const int a = order[0], b = order[1];
            U[0][j] = U[1][a] * U[2][b] - U[2][a] * U[1][b];
            U[1][j] = U[2][a] * U[0][b] - U[0][a] * U[2][b];
            U[2][j] = U[0][a] * U[1][b] - U[1][a] * U[0][b];
        } else { for (int k = 0; k < 3; ++k) U[k][j] = V[k][j]; }  // rank <= 1: no unique answer; stay finite
    }
    auto det3 = [](const double M[3][3]) { return M[0][0] * (M[1][1] * M[2][2] - M[1][2] * M[2][1]) - M[0][1] * (M[1][0] * M[2][2] - M[1][2] * M[2][0]) + M[0][2] * (M[1][0] * M[2][1] - M[1][1] * M[2][0]); };
    double S[3] = {1, 1, 1};
    if (det3(U) * det3(V) < 0) S[order[2]] = -1;  // the smallest singular direction flips
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            double s = 0;
            for (int j = 0; j < 3; ++j) s += U[r][j] * S[j] * V[c][j];
            R[r * 3 + c] = s;
        }
}
}  // namespace

// pcl::IterativeClosestPoint::computeTransformation with TransformationEstimationSVD and DefaultConvergenceCriteria
// (gicp_engine.h, variant 3).  Per iteration: one correspondence + moment kernel, a 17-double record to the host, a 3x3 SVD,
// one in-place transform of the working copy of the source.
int GicpEngine::align_icp(const float guess[16])
{
    if (!d_tgt_ && n_tgt_) { set_error("ICP: no target"); return MRGFE_ERR_STATE; }
    MRGFE_TRY(ctx_->bind());
    hipStream_t st = ctx_->stream;
    kernel_ms = 0; kernel_launches = 0; kernel_alg_bytes = 0;
    n_linearize_ = n_error_ = 0;
    converged_ = false;
    nr_iterations_ = 0;
    for (int t = 0; t < 36; ++t) final_hessian_[t] = 0.0;
    std::memcpy(final_, guess, sizeof(final_));
    if (!tgt_grid_valid_) {
        MRGFE_TRY(tgt_grid_.build(ctx_, d_tgt_, n_tgt_, 1.0f, NnGrid::kCrowding1nn, 1));
        tgt_grid_valid_ = true;
    }
    const uint32_t n = static_cast<uint32_t>(n_src_);
    constexpr uint32_t per_blk = 256u / kGicpGroup;
    const uint32_t nblk_t = (n + 255) / 256, nblk_c = (n + per_blk - 1) / per_blk;
    MRGFE_TRY(d_cur_.ensure(std::max<size_t>(n_src_, 1) * 16));
    MRGFE_TRY(d_partial_.ensure(sizeof(double) * kGicpStride * (size_t(nblk_c) + 2)));
    MRGFE_TRY(d_T_.ensure(64));
    float4* d_cur = d_cur_.as<float4>();
    double* d_part = d_partial_.as<double>();
    double* d_res = d_part + size_t(nblk_c) * kGicpStride;
    if (n) MRGFE_HIP_CHECK(hipMemcpyAsync(d_cur, d_src_, size_t(n) * 16, hipMemcpyDeviceToDevice, st));
    bool identity = true;
    for (int i = 0; i < 16; ++i) identity = identity && guess[i] == ((i % 5 == 0) ? 1.0f : 0.0f);
    if (!identity && n) {
        MRGFE_HIP_CHECK(hipMemcpyAsync(d_T_.p, guess, 48, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(icp_transform_kernel, dim3(nblk_t), dim3(256), 0, st, d_cur, n, d_T_.as<float>());
        MRGFE_HIP_CHECK(hipStreamSynchronize(st));  // d_T_ is rewritten below
    }
    const double max_sq = prm_.max_corr_dist * prm_.max_corr_dist;
    const double rot_thr = 1.0 - prm_.trans_eps, trans_thr = prm_.trans_eps;
    double prev_mse = std::numeric_limits<double>::max();
    for (;;) {
        ++n_linearize_;
        double r[kGicpStride] = {0};
        if (n && n_tgt_) {
            if (prm_.use_reciprocal) {
                MRGFE_TRY(cur_grid_.build(ctx_, d_cur, n, 1.0f, NnGrid::kCrowding1nn, 1));
                hipLaunchKernelGGL(icp_corr_sums_kernel<true>, dim3(nblk_c), dim3(256), 0, st, tgt_grid_.dev2(), cur_grid_.dev2(), d_cur, d_tgt_, n, max_sq, d_part);
            } else {
                hipLaunchKernelGGL(icp_corr_sums_kernel<false>, dim3(nblk_c), dim3(256), 0, st, tgt_grid_.dev2(), tgt_grid_.dev2(), d_cur, d_tgt_, n, max_sq, d_part);
            }
            hipLaunchKernelGGL(gicp_reduce_kernel, dim3(1), dim3(256), 0, st, d_part, nblk_c, d_res);
            MRGFE_HIP_CHECK(hipGetLastError());
            MRGFE_HIP_CHECK(hipMemcpyAsync(r, d_res, sizeof(r), hipMemcpyDeviceToHost, st));
            MRGFE_HIP_CHECK(hipStreamSynchronize(st));
            kernel_launches += 1;
        }
        const double cnt = r[0];
        if (cnt < 3) { converged_ = false; break; }  // "Not enough correspondences found"
        double mu_s[3], mu_d[3], sigma[9], R[9];
        for (int a = 0; a < 3; ++a) { mu_s[a] = r[1 + a] / cnt; mu_d[a] = r[4 + a] / cnt; }
        for (int rr = 0; rr < 3; ++rr) for (int c = 0; c < 3; ++c) sigma[rr * 3 + c] = r[7 + rr * 3 + c] / cnt - mu_d[rr] * mu_s[c];
        umeyama_rotation(sigma, R);
        float Tm[16];
        for (int i = 0; i < 16; ++i) Tm[i] = (i % 5 == 0) ? 1.0f : 0.0f;
        const float ms[3] = {static_cast<float>(mu_s[0]), static_cast<float>(mu_s[1]), static_cast<float>(mu_s[2])};
        for (int rr = 0; rr < 3; ++rr) {
            for (int c = 0; c < 3; ++c) Tm[rr * 4 + c] = static_cast<float>(R[rr * 3 + c]);
            float s = Tm[rr * 4 + 0] * ms[0];  // Rt.col(3).head(3) = dst_mean - R * src_mean, in float
            s = s + Tm[rr * 4 + 1] * ms[1];
            s = s + Tm[rr * 4 + 2] * ms[2];
            Tm[rr * 4 + 3] = static_cast<float>(mu_d[rr]) - s;
        }
        MRGFE_HIP_CHECK(hipMemcpyAsync(d_T_.p, Tm, 48, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(icp_transform_kernel, dim3(nblk_t), dim3(256), 0, st, d_cur, n, d_T_.as<float>());
        MRGFE_HIP_CHECK(hipGetLastError());
        MRGFE_HIP_CHECK(hipStreamSynchronize(st));  // Tm is a local
        float nf[16];
        for (int rr = 0; rr < 4; ++rr)
            for (int c = 0; c < 4; ++c) { float s = 0; for (int k = 0; k < 4; ++k) s += Tm[rr * 4 + k] * final_[k * 4 + c]; nf[rr * 4 + c] = s; }
        std::memcpy(final_, nf, sizeof(nf));
        ++nr_iterations_;
        if (nr_iterations_ >= prm_.max_iterations) { converged_ = true; break; }
        const double cos_angle = 0.5 * (static_cast<double>(Tm[0]) + static_cast<double>(Tm[5]) + static_cast<double>(Tm[10]) - 1.0);
        const double tsq = static_cast<double>(Tm[3]) * Tm[3] + static_cast<double>(Tm[7]) * Tm[7] + static_cast<double>(Tm[11]) * Tm[11];
        if (cos_angle >= rot_thr && tsq <= trans_thr) { converged_ = true; break; }
        const double mse = r[16] / cnt;
        if (std::fabs(mse - prev_mse) < 1e-12) { converged_ = true; break; }
        prev_mse = mse;
    }
    return MRGFE_OK;
}

// pcl::GeneralizedIterativeClosestPoint::computeTransformation + estimateRigidTransformationBFGS (gicp_engine.h, variant 4).  Per outer
// iteration one correspondence + Mahalanobis kernel; per functor evaluation of the inner BFGS one 14-sum kernel and a small record to the host.
namespace {
struct PclGicpFunctor {
    GicpEngine* eng;
    std::function<int(const double x[6], double* f, double g[6])> eval;  // 0 on success
    int  status = MRGFE_OK, evaluations = 0;
    double operator()(const double x[6]) { double f = 0, g[6]; run(x, &f, g); return f; }
    void   df(const double x[6], double g[6]) { double f = 0; run(x, &f, g); }
    void   fdf(const double x[6], double& f, double g[6]) { run(x, &f, g); }
    void   run(const double x[6], double* f, double g[6])
    {
        ++evaluations;
        const int rc = eval(x, f, g);
        if (rc != MRGFE_OK && status == MRGFE_OK) status = rc;
    }
};
// GeneralizedIterativeClosestPoint::applyState: t.topLeftCorner<3,3>() = Rz(x5) Ry(x4) Rx(x3) t.topLeftCorner<3,3>(); t.col(3) += (x0, x1, x2, 0); all float,
// the rotations as Eigen::AngleAxisf::toRotationMatrix() builds them
void angle_axis_unit_f(float angle, int axis, float R[9])
{
    float ax[3] = {0, 0, 0};
    ax[axis] = 1.0f;
    const float sn = std::sin(angle), c = std::cos(angle);
    const float sin_axis[3] = {sn * ax[0], sn * ax[1], sn * ax[2]};
    const float cos1_axis[3] = {(1.0f - c) * ax[0], (1.0f - c) * ax[1], (1.0f - c) * ax[2]};
    float tmp;
    tmp = cos1_axis[0] * ax[1]; R[1] = tmp - sin_axis[2]; R[3] = tmp + sin_axis[2];
    tmp = cos1_axis[0] * ax[2]; R[2] = tmp + sin_axis[1]; R[6] = tmp - sin_axis[1];
    tmp = cos1_axis[1] * ax[2]; R[5] = tmp - sin_axis[0]; R[7] = tmp + sin_axis[0];
    R[0] = cos1_axis[0] * ax[0] + c; R[4] = cos1_axis[1] * ax[1] + c; R[8] = cos1_axis[2] * ax[2] + c;
}
void mul3f(const float a[9], const float b[9], float out[9])
{
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            const float p0 = a[r * 3 + 0] * b[0 * 3 + c], p1 = a[r * 3 + 1] * b[1 * 3 + c], p2 = a[r * 3 + 2] * b[2 * 3 + c];
            const float s = p0 + p1;
            out[r * 3 + c] = s + p2;
        }
}
void pclgicp_apply_state(float t[16], const double x[6])
{
    float Rx[9], Ry[9], Rz[9], Rzy[9], R[9], old[9], nw[9];
    angle_axis_unit_f(static_cast<float>(x[5]), 2, Rz);
    angle_axis_unit_f(static_cast<float>(x[4]), 1, Ry);
    angle_axis_unit_f(static_cast<float>(x[3]), 0, Rx);
    mul3f(Rz, Ry, Rzy);
    mul3f(Rzy, Rx, R);
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) old[r * 3 + c] = t[r * 4 + c];
    mul3f(R, old, nw);
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) t[r * 4 + c] = nw[r * 3 + c];
    for (int r = 0; r < 3; ++r) t[r * 4 + 3] += static_cast<float>(x[r]);
}
// computeRDerivative: g[3..5] = tr(dR/dphi dCost_dR_T), ... for R = Rz(psi) Ry(theta) Rx(phi)
void pclgicp_r_derivative(const double x[6], const double dC[9], double g[6])
{
    const double phi = x[3], theta = x[4], psi = x[5];
    const double cphi = std::cos(phi), sphi = std::sin(phi), ctheta = std::cos(theta), stheta = std::sin(theta), cpsi = std::cos(psi), spsi = std::sin(psi);
    const double dPhi[9] = {0, sphi * spsi + cphi * cpsi * stheta, cphi * spsi - cpsi * sphi * stheta, 0, -cpsi * sphi + cphi * spsi * stheta, -cphi * cpsi - sphi * spsi * stheta,
                            0, cphi * ctheta, -ctheta * sphi};
    const double dTheta[9] = {-cpsi * stheta, cpsi * ctheta * sphi, cphi * cpsi * ctheta, -spsi * stheta, ctheta * sphi * spsi, cphi * ctheta * spsi, -ctheta, -sphi * stheta, -cphi * stheta};
    const double dPsi[9] = {-ctheta * spsi, -cphi * cpsi - sphi * spsi * stheta, cpsi * sphi - cphi * spsi * stheta, cpsi * ctheta, -cphi * spsi + cpsi * sphi * stheta,
                            sphi * spsi + cphi * cpsi * stheta, 0, 0, 0};
    auto inner = [&](const double A[9]) {  // matricesInnerProd(A, dCost_dR_T) = sum_ij A(j, i) dC(i, j)
        double r = 0;
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r += A[j * 3 + i] * dC[i * 3 + j];
        return r;
    };
    g[3] = inner(dPhi);
    g[4] = inner(dTheta);
    g[5] = inner(dPsi);
}
}  // namespace

// serial pcl::GICP adds its cost terms in point order: 1 (default) so does PCL_GICP_HIP (pclgicp_seqsum_kernel), 0 the tree sums of round 3
// (MRGFE_PCLGICP_TREE_SUMS=1; mrgfe_dbg_set_pclgicp_reference_order).  PCL_GICP_OMP_HIP always sums in a tree: pclomp's own per-thread sums have no fixed order.
static std::atomic<int> g_pclgicp_ref_order{-1};
static int pclgicp_reference_order_mode()
{
    int v = g_pclgicp_ref_order.load(std::memory_order_relaxed);
    if (v < 0) { v = std::getenv("MRGFE_PCLGICP_TREE_SUMS") ? 0 : 1; g_pclgicp_ref_order.store(v, std::memory_order_relaxed); }
    return v;
}
int gicp_set_pcl_reference_order(int mode)
{
    if (mode == 0 || mode == 1) g_pclgicp_ref_order.store(mode, std::memory_order_relaxed);
    return pclgicp_reference_order_mode();
}

int GicpEngine::pcl_evaluate(const float T_rowmajor[16], const float guess_rowmajor[16], const float4* d_pts, bool search, const double x[6], double* f, double g[6], int* n_corr)
{
    hipStream_t    st = ctx_->stream;
    const uint32_t n = static_cast<uint32_t>(n_src_), nblk = (n + 255) / 256;
    *f = 0;
    for (int k = 0; k < 6; ++k) g[k] = 0;
    if (n_corr) *n_corr = 0;
    if (n == 0 || n_tgt_ == 0) return MRGFE_OK;
    PclGicpIter it;
    if (search) {  // the outer iteration's correspondences and Mahalanobis matrices at transformation_ = T
        for (int k = 0; k < 12; ++k) it.Tf[k] = T_rowmajor[k];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                double s = 0;
                for (int k = 0; k < 4; ++k) s += static_cast<double>(T_rowmajor[i * 4 + k]) * static_cast<double>(guess_rowmajor[k * 4 + j]);
                it.R[i * 3 + j] = s;
            }
        it.thr2 = prm_.max_corr_dist * prm_.max_corr_dist;
        constexpr uint32_t per_blk = 256u / kGicpGroup;
        hipLaunchKernelGGL(pclgicp_corr_kernel, dim3((n + per_blk - 1) / per_blk), dim3(256), 0, st, tgt_grid_.dev2(), d_pts, n, it, d_src_cov_.as<double>(), d_tgt_cov_.as<double>(),
                           d_corr_.as<int32_t>(), d_mahal_.as<double>());
        MRGFE_HIP_CHECK(hipGetLastError());
        if (prm_.pcl_omp_sum_threads > 1) {  // the chunk boundaries of pclomp's per-thread sums belong to these correspondences
            MRGFE_TRY(d_chunk_bounds_.ensure(sizeof(uint32_t) * (kChunkMaxThreads + 1)));
            hipLaunchKernelGGL(pclgicp_chunk_bounds_kernel, dim3(1), dim3(256), 0, st, d_corr_.as<int32_t>(), n, prm_.pcl_omp_sum_threads, d_chunk_bounds_.as<uint32_t>());
            MRGFE_HIP_CHECK(hipGetLastError());
        }
        if (!x) return MRGFE_OK;
    }
    float Tx[16];
    for (int i = 0; i < 16; ++i) Tx[i] = (i % 5 == 0) ? 1.0f : 0.0f;  // base_transformation_ = Identity
    pclgicp_apply_state(Tx, x);
    for (int k = 0; k < 12; ++k) it.Tf[k] = Tx[k];
    for (int k = 0; k < 9; ++k) it.R[k] = 0;
    it.thr2 = 0;
    double* d_part = d_partial_.as<double>();
    double* d_res = d_part + size_t(nblk) * kGicpStride;
    MRGFE_HIP_CHECK(hipEventRecord(ctx_->ev0, st));
    if (prm_.pcl_reference_order_sums && pclgicp_reference_order_mode()) {
        const uint32_t n_pad = (n + 1u) & ~1u;
        MRGFE_TRY(d_terms_.ensure(sizeof(double) * 14 * size_t(n_pad)));
        hipLaunchKernelGGL(pclgicp_fdf_kernel<true>, dim3(nblk), dim3(256), 0, st, d_pts, n, d_tgt_, d_corr_.as<int32_t>(), d_mahal_.as<double>(), it, d_part, d_terms_.as<double>(), n_pad);
        hipLaunchKernelGGL(pclgicp_seqsum_kernel, dim3(1), dim3(256), 0, st, d_terms_.as<double>(), n, n_pad, d_res);
        MRGFE_HIP_CHECK(hipEventRecord(ctx_->ev1, st));
    } else if (prm_.pcl_omp_sum_threads > 1 && pclgicp_reference_order_mode()) {
        // pclomp::GICP for T = pcl_omp_sum_threads OpenMP threads: T chains over the static chunks of the correspondence list, partials in thread order
        const uint32_t n_pad = (n + 1u) & ~1u;
        const int      T = prm_.pcl_omp_sum_threads;
        MRGFE_TRY(d_terms_.ensure(sizeof(double) * 14 * size_t(n_pad)));
        if (!d_chunk_bounds_.p) { set_error("pclomp sums: no correspondences were searched yet"); return MRGFE_ERR_STATE; }
        const uint32_t* d_bounds = d_chunk_bounds_.as<uint32_t>();
        hipLaunchKernelGGL(pclgicp_fdf_kernel<true>, dim3(nblk), dim3(256), 0, st, d_pts, n, d_tgt_, d_corr_.as<int32_t>(), d_mahal_.as<double>(), it, d_part, d_terms_.as<double>(), n_pad);
        hipLaunchKernelGGL(pclgicp_chunksum_kernel, dim3(1), dim3(256), 0, st, d_terms_.as<double>(), n, n_pad, d_bounds, T, d_res);
        MRGFE_HIP_CHECK(hipEventRecord(ctx_->ev1, st));
    } else {
        hipLaunchKernelGGL(pclgicp_fdf_kernel<false>, dim3(nblk), dim3(256), 0, st, d_pts, n, d_tgt_, d_corr_.as<int32_t>(), d_mahal_.as<double>(), it, d_part, static_cast<double*>(nullptr), 0u);
        MRGFE_HIP_CHECK(hipEventRecord(ctx_->ev1, st));
        hipLaunchKernelGGL(gicp_reduce_kernel, dim3(1), dim3(256), 0, st, d_part, nblk, d_res);
    }
    MRGFE_HIP_CHECK(hipGetLastError());
    double r[kGicpStride];
    MRGFE_HIP_CHECK(hipMemcpyAsync(r, d_res, sizeof(r), hipMemcpyDeviceToHost, st));
    MRGFE_HIP_CHECK(hipStreamSynchronize(st));
    float ms = 0;
    MRGFE_HIP_CHECK(hipEventElapsedTime(&ms, ctx_->ev0, ctx_->ev1));
    kernel_ms += ms;
    kernel_launches += 1;
    kernel_alg_bytes += double(n) * (16 + 4) + r[28] * (16 + 72);
    const double m = r[28];
    if (n_corr) *n_corr = static_cast<int>(m);
    if (m <= 0) return MRGFE_OK;
    *f = r[0] / m;
    for (int k = 0; k < 3; ++k) g[k] = r[1 + k] * (2.0 / m);
    double dC[9];
    for (int k = 0; k < 9; ++k) dC[k] = r[4 + k] * (2.0 / m);
    pclgicp_r_derivative(x, dC, g);
    return MRGFE_OK;
}

int GicpEngine::align_pcl_gicp(const float guess[16])
{
    MRGFE_TRY(ensure_ready());
    hipStream_t st = ctx_->stream;
    kernel_ms = 0; kernel_launches = 0; kernel_alg_bytes = 0;
    n_linearize_ = n_error_ = 0;
    converged_ = false;
    nr_iterations_ = 0;
    for (int t = 0; t < 36; ++t) final_hessian_[t] = 0.0;
    float transformation[16], previous[16];
    for (int i = 0; i < 16; ++i) transformation[i] = previous[i] = final_[i] = (i % 5 == 0) ? 1.0f : 0.0f;
    const uint32_t n = static_cast<uint32_t>(n_src_);
    // pcl::Registration::align copies the source into `output`; computeTransformation moves it by the guess (pcl::transformPointCloud)
    MRGFE_TRY(d_cur_.ensure(std::max<size_t>(n_src_, 1) * 16));
    MRGFE_TRY(d_T_.ensure(64));
    float4* d_out = d_cur_.as<float4>();
    if (n) {
        MRGFE_HIP_CHECK(hipMemcpyAsync(d_out, d_src_, size_t(n) * 16, hipMemcpyDeviceToDevice, st));
        MRGFE_HIP_CHECK(hipMemcpyAsync(d_T_.p, guess, 48, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(icp_transform_kernel, dim3((n + 255) / 256), dim3(256), 0, st, d_out, n, d_T_.as<float>());
        MRGFE_HIP_CHECK(hipGetLastError());
        MRGFE_HIP_CHECK(hipStreamSynchronize(st));
    }
    while (!converged_ && n > 0) {
        double f0 = 0, g0[6];
        int    m = 0;
        MRGFE_TRY(pcl_evaluate(transformation, guess, d_out, true, nullptr, &f0, g0, nullptr));  // the search loop of this iteration
        std::memcpy(previous, transformation, sizeof(previous));
        // ---- estimateRigidTransformationBFGS
        // x[3] = std::atan2(T(2,1), T(2,2)) and x[5] = std::atan2(T(1,0), T(0,0)) on floats, x[4] = asin(-T(2,0)) through the C function (double)
        double x[6] = {transformation[3], transformation[7], transformation[11], static_cast<double>(std::atan2(transformation[9], transformation[10])),
                       std::asin(static_cast<double>(-transformation[8])), static_cast<double>(std::atan2(transformation[4], transformation[0]))};
        PclGicpFunctor fn;
        fn.eng = this;
        fn.eval = [&](const double xx[6], double* f, double g[6]) { return pcl_evaluate(nullptr, nullptr, d_out, false, xx, f, g, &m); };
        // NotEnoughPointsException when fewer than four correspondences: known after the first evaluation (the count comes back with every record)
        BFGS<PclGicpFunctor> bfgs(fn);
        int inner = 0;
        BFGSSpace::Status result = bfgs.minimizeInit(x);
        MRGFE_TRY(fn.status);
        if (m < 4) break;
        result = BFGSSpace::Running;
        do {
            ++inner;
            result = bfgs.minimizeOneStep(x);
            MRGFE_TRY(fn.status);
            if (result) break;
            // testGradient -> OptimizationFunctorWithIndices::checkGradient (PCL >= 1.11: translation and rotation parts apart; pclomp: the whole norm)
            const double* gr = bfgs.gradient;
            const double gt = std::sqrt(gr[0] * gr[0] + gr[1] * gr[1] + gr[2] * gr[2]), grn = std::sqrt(gr[3] * gr[3] + gr[4] * gr[4] + gr[5] * gr[5]);
            const bool ok = prm_.pcl_whole_gradient_norm ? std::sqrt(gt * gt + grn * grn) < 1e-2 : (gt < 1e-2 && grn < 1e-2);
            result = ok ? BFGSSpace::Success : BFGSSpace::Running;
        } while (result == BFGSSpace::Running && inner < prm_.max_inner_iterations);
        n_error_ += fn.evaluations;
        n_linearize_ += 1;
        if (result == BFGSSpace::NoProgress || result == BFGSSpace::Success || inner == prm_.max_inner_iterations) {
            for (int i = 0; i < 16; ++i) transformation[i] = (i % 5 == 0) ? 1.0f : 0.0f;
            pclgicp_apply_state(transformation, x);
        } else {
            break;  // SolverDidntConvergeException
        }
        double delta = 0;
        for (int k = 0; k < 4; ++k)
            for (int l = 0; l < 4; ++l) {
                const double ratio = (k < 3 && l < 3) ? 1.0 / prm_.rot_eps : 1.0 / prm_.trans_eps;
                const double c_delta = ratio * std::fabs(static_cast<double>(previous[k * 4 + l] - transformation[k * 4 + l]));
                if (c_delta > delta) delta = c_delta;
            }
        ++nr_iterations_;
        if (nr_iterations_ >= prm_.max_iterations || delta < 1) {
            converged_ = true;
            std::memcpy(previous, transformation, sizeof(previous));
        }
    }
    for (int r = 0; r < 4; ++r)  // final_transformation_ = previous_transformation_ * guess (float)
        for (int c = 0; c < 4; ++c) {
            float s = 0;
            for (int k = 0; k < 4; ++k) s += previous[r * 4 + k] * guess[k * 4 + c];
            final_[r * 4 + c] = s;
        }
    return MRGFE_OK;
}

int GicpEngine::align(const float guess[16])
{
    if (prm_.variant == 3) return align_icp(guess);
    if (prm_.variant == 4) return align_pcl_gicp(guess);
    MRGFE_TRY(ensure_ready());
    kernel_ms = 0; kernel_launches = 0; kernel_alg_bytes = 0;
    n_linearize_ = n_error_ = 0;
    GicpLmController ctl;
    ctl.start(prm_, guess, static_cast<uint32_t>(n_src_));
    while (!ctl.done()) {
        const GicpRequest& rq = ctl.request();
        double r[kGicpStride] = {0};
        if (rq.type == 0) {
            double H[36], b[6], err;
            int    nc = 0;
            MRGFE_TRY(run_linearize(rq.T, true, H, b, &err, &nc));
            r[0] = err;
            for (int t = 0; t < 6; ++t) r[1 + t] = b[t];
            int t = 7;
            for (int i = 0; i < 6; ++i) for (int j = i; j < 6; ++j) r[t++] = H[i * 6 + j];
            r[28] = nc;
        } else {
            MRGFE_TRY(run_error(rq.T, &r[0]));
        }
        ctl.on_result(r);
    }
    converged_ = ctl.converged();
    nr_iterations_ = ctl.iterations();
    std::memcpy(final_hessian_, ctl.hessian(), sizeof(final_hessian_));
    ctl.final_transformation(final_);
    return MRGFE_OK;
}

// ---- the LM loop of fast_gicp::LsqRegistration, one request at a time ------------------------------------------------
void GicpLmController::start(const GicpParams& prm, const float guess[16], uint32_t)
{
    prm_ = prm;
    for (int i = 0; i < 16; ++i) x0_[i] = static_cast<double>(guess[i]);
    lambda_ = prm_.variant == 1 ? prm_.sg_init_lambda : -1.0;
    converged_ = false;
    nr_iterations_ = 0;
    n_linearize_ = n_error_ = 0;
    outer_ = 0;
    for (int t = 0; t < 36; ++t) final_hessian_[t] = (t % 7 == 0) ? 1.0 : 0.0;
    done_ = prm_.max_iterations <= 0;
    if (!done_) { req_.type = 0; std::memcpy(req_.T, x0_, sizeof(x0_)); }
}

void GicpLmController::propose()
{
    double A[36], nb[6];
    for (int t = 0; t < 36; ++t) A[t] = H_[t] + ((t % 7 == 0) ? lambda_ : 0.0);
    for (int t = 0; t < 6; ++t) nb[t] = -b_[t];
    solve6(A, nb, d_);
    se3_exp(d_, delta_);
    if (prm_.variant == 1) mul4(x0_, delta_, xi_);  // small_gicp perturbs on the right
    else                   mul4(delta_, x0_, xi_);
    req_.type = 1;
    std::memcpy(req_.T, xi_, sizeof(xi_));
}

void GicpLmController::end_outer(bool ok)
{
    if (!ok) { done_ = true; return; }  // "lm not converged!!"
    converged_ = is_converged(delta_, prm_.rot_eps, prm_.trans_eps);
    ++outer_;
    if (converged_ || outer_ >= prm_.max_iterations) { done_ = true; return; }
    req_.type = 0;
    std::memcpy(req_.T, x0_, sizeof(x0_));
}

// small_gicp::LevenbergMarquardtOptimizer::optimize, one request at a time (gicp_engine.h, variant 1)
void GicpLmController::on_result_small(const double r[32])
{
    if (req_.type == 0) {
        ++n_linearize_;
        nr_iterations_ = outer_;
        y0_ = 0.5 * r[0];
        for (int t = 0; t < 6; ++t) b_[t] = r[1 + t];
        int t = 7;
        for (int i = 0; i < 6; ++i)
            for (int j = i; j < 6; ++j) { H_[i * 6 + j] = r[t]; H_[j * 6 + i] = r[t]; ++t; }
        std::memcpy(final_hessian_, H_, sizeof(H_));
        inner_ = 0;
        if (prm_.sg_max_inner_iterations <= 0) { done_ = true; return; }
        propose();
        return;
    }
    ++n_error_;
    const double new_e = 0.5 * r[0];
    if (new_e <= y0_) {
        converged_ = std::sqrt(d_[0] * d_[0] + d_[1] * d_[1] + d_[2] * d_[2]) <= prm_.rot_eps && std::sqrt(d_[3] * d_[3] + d_[4] * d_[4] + d_[5] * d_[5]) <= prm_.trans_eps;
        std::memcpy(x0_, xi_, sizeof(xi_));
        lambda_ /= prm_.sg_lambda_factor;
        ++outer_;
        if (converged_ || outer_ >= prm_.max_iterations) { done_ = true; return; }
        req_.type = 0;
        std::memcpy(req_.T, x0_, sizeof(x0_));
        return;
    }
    lambda_ *= prm_.sg_lambda_factor;
    if (++inner_ >= prm_.sg_max_inner_iterations) { done_ = true; return; }  // no trial was accepted: give up, not converged
    propose();
}

void GicpLmController::on_result(const double r[32])
{
    if (prm_.variant == 1) { on_result_small(r); return; }
    if (req_.type == 0) {
        ++n_linearize_;
        nr_iterations_ = outer_;
        y0_ = r[0];
        for (int t = 0; t < 6; ++t) b_[t] = r[1 + t];
        int t = 7;
        for (int i = 0; i < 6; ++i)
            for (int j = i; j < 6; ++j) { H_[i * 6 + j] = r[t]; H_[j * 6 + i] = r[t]; ++t; }
        if (lambda_ < 0.0) {
            double md = 0;
            for (int d = 0; d < 6; ++d) md = std::max(md, std::fabs(H_[d * 6 + d]));
            lambda_ = prm_.lm_init_lambda_factor * md;
        }
        nu_ = 2.0;
        inner_ = 0;
        if (prm_.lm_max_iterations <= 0) { end_outer(false); return; }
        propose();
        return;
    }
    ++n_error_;
    const double yi = r[0];
    double denom = 0;
    for (int t = 0; t < 6; ++t) denom += d_[t] * (lambda_ * d_[t] - b_[t]);
    const double rho = (y0_ - yi) / denom;
    if (rho < 0) {
        if (is_converged(delta_, prm_.rot_eps, prm_.trans_eps)) { end_outer(true); return; }
        lambda_ = nu_ * lambda_;
        nu_ = 2 * nu_;
        if (++inner_ >= prm_.lm_max_iterations) { end_outer(false); return; }
        propose();
        return;
    }
    std::memcpy(x0_, xi_, sizeof(xi_));
    lambda_ = lambda_ * std::max(1.0 / 3.0, 1 - std::pow(2 * rho - 1, 3));
    std::memcpy(final_hessian_, H_, sizeof(H_));
    end_outer(true);
}

int GicpEngine::aligned_cloud(float* out)
{
    if (n_src_ == 0) return MRGFE_OK;
    MRGFE_TRY(ctx_->bind());
    MRGFE_TRY(d_T_.ensure(64 + n_src_ * 16));
    float*  d_T = d_T_.as<float>();
    float4* d_out = reinterpret_cast<float4*>(d_T_.as<char>() + 64);
    MRGFE_HIP_CHECK(hipMemcpyAsync(d_T, final_, 48, hipMemcpyHostToDevice, ctx_->stream));
    MRGFE_TRY(launch_transform_cloud(ctx_, d_src_, d_out, static_cast<uint32_t>(n_src_), d_T));
    MRGFE_HIP_CHECK(hipMemcpyAsync(out, d_out, n_src_ * 16, hipMemcpyDeviceToHost, ctx_->stream));
    MRGFE_HIP_CHECK(hipStreamSynchronize(ctx_->stream));
    return MRGFE_OK;
}

// ---- batched LM rounds -------------------------------------------------------------------------------------------------
GicpBatch::~GicpBatch()
{
    if (ctx_) (void)hipSetDevice(ctx_->device);
    for (Lane* l : lanes_) {
        l->set.release(); l->views.clear(); l->knn_i.release(); l->knn_d.release();
        mrgfe_ctx_destroy(l->ctx);
        delete l;
    }
    d_pairs_.release(); d_evals_.release(); d_grids_.release(); d_partials_.release();
    h_evals_.release(); h_results_.release();
    if (done_) (void)hipEventDestroy(done_);
}

int GicpBatch::align_all(std::vector<GicpEngine*>& engines, std::vector<GicpBatchPair>& pairs)
{
    MRGFE_TRY(ctx_->bind());
    const int P = static_cast<int>(pairs.size());
    if (P == 0) return MRGFE_OK;
    hipStream_t st = ctx_->stream;
    if (!done_) MRGFE_HIP_CHECK(hipEventCreateWithFlags(&done_, hipEventDisableTiming));
    // targets: covariances + correspondence grid once each; sources: covariances cloud by cloud
    std::vector<NnGrid2Dev> h_grids(engines.size());
    std::vector<VoxGridDev> h_vgrids(engines.size());
    bool voxel = false;  // VGICP: the targets are voxel maps, the correspondence search is a table lookup
    for (size_t t = 0; t < engines.size(); ++t) {
        if (!engines[t]) continue;
        MRGFE_TRY(engines[t]->prepare_target());
        h_grids[t] = engines[t]->target_grid();
        voxel = engines[t]->params().variant == 2;
        if (voxel) {
            VoxGridDev& g = h_vgrids[t];
            engines[t]->voxel_grid(&g.res, g.cmin, g.dim, &g.n_cells);
            g.vox = engines[t]->voxel_records();
        }
    }
    // Source covariances: one cloud is a chain of small launches and a few host round trips (bounding box, cell-size
    // passes, sorts, k-NN, covariance) that leaves most of the chip idle, so `lanes` host threads work through the clouds
    // side by side, each on its own stream and workspace.
    MRGFE_HIP_CHECK(hipStreamSynchronize(st));  // the clouds were uploaded on this stream
    {
        int want = 4;
        if (const char* e = std::getenv("MRGFE_GICP_LANES")) want = std::atoi(e);
        const int n_lanes = std::max(1, std::min(std::min(want, 8), P));
        while (static_cast<int>(lanes_.size()) < n_lanes) {
            Lane* l = new Lane();
            if (ctx_create_like(ctx_, &l->ctx) != MRGFE_OK) { delete l; return MRGFE_ERR_HIP; }
            lanes_.push_back(l);
        }
        std::vector<int> status(n_lanes, MRGFE_OK);
        std::vector<std::string> message(n_lanes);
        // clouds to do: every pair without stored covariances; a stored cloud shared by several pairs once
        std::vector<int> todo;
        for (int i = 0; i < P; ++i) {
            GicpBatchPair& p = pairs[i];
            const int k = engines[p.target]->params().k_correspondences;
            if (p.ext_cov) {
                if (*p.ext_cov_k == k) continue;
                bool dup = false;
                for (int j : todo) dup = dup || pairs[j].ext_cov == p.ext_cov;
                if (dup) continue;
            }
            todo.push_back(i);
        }
        // ... a chunk of clouds at a time: their k-NN grids are built together (NnGridSet: one launch per build step for the chunk, three
        // host waits per chunk instead of per cloud), then each cloud's k-NN search and covariance kernel follow on the lane's stream
        size_t chunk = 8;
        if (const char* e = std::getenv("MRGFE_GICP_CHUNK")) chunk = static_cast<size_t>(std::max(1, std::atoi(e)));
        const size_t n_chunks = (todo.size() + chunk - 1) / chunk;
        std::atomic<size_t> next{0};
        auto work = [&](int li) {
            Lane& l = *lanes_[li];
            if (l.ctx->bind() != MRGFE_OK) { status[li] = MRGFE_ERR_HIP; return; }
            std::vector<const float4*> clouds;
            std::vector<uint32_t>      sizes;
            std::vector<NnGrid*>       views;
            if (l.views.size() < chunk) l.views.resize(chunk);
            for (;;) {
                const size_t c = next.fetch_add(1);
                if (c >= n_chunks) break;
                const size_t w0 = c * chunk, w1 = std::min(todo.size(), w0 + chunk);
                clouds.clear(); sizes.clear(); views.clear();
                for (size_t w = w0; w < w1; ++w) {
                    clouds.push_back(pairs[todo[w]].d_src);
                    sizes.push_back(pairs[todo[w]].n);
                    views.push_back(&l.views[w - w0]);
                }
                int rc = l.set.build(l.ctx, clouds.data(), sizes.data(), static_cast<int>(w1 - w0), 1.0f, NnGrid::kCrowdingKnn, kNnMaxLevels, views.data());
                for (size_t w = w0; w < w1 && rc == MRGFE_OK; ++w) {
                    GicpBatchPair& p = pairs[todo[w]];
                    const int k = engines[p.target]->params().k_correspondences;
                    rc = gicp_covariances_on_grid(l.ctx, k, p.d_src, p.n, p.ext_cov ? *p.ext_cov : p.cov, l.views[w - w0], l.knn_i, l.knn_d, false);
                    if (rc == MRGFE_OK && p.ext_cov) *p.ext_cov_k = k;
                }
                if (rc != MRGFE_OK) { status[li] = rc; message[li] = mrgfe_last_error(); break; }
            }
            if (hipStreamSynchronize(l.ctx->stream) != hipSuccess && status[li] == MRGFE_OK) status[li] = MRGFE_ERR_HIP;
        };
        std::vector<std::thread> th;
        for (int li = 1; li < n_lanes; ++li) th.emplace_back(work, li);
        work(0);
        for (auto& t : th) t.join();
        MRGFE_TRY(ctx_->bind());
        for (int li = 0; li < n_lanes; ++li)
            if (status[li] != MRGFE_OK) { set_error("GICP batch: source covariances failed: %s", message[li].c_str()); return status[li]; }
    }
    std::vector<GicpPairDev> h_pairs(P);
    uint32_t part = 0, max_n = 0;
    for (int i = 0; i < P; ++i) {
        GicpBatchPair& p = pairs[i];
        GicpEngine*    e = engines[p.target];
        MRGFE_TRY(p.corr.ensure(std::max<size_t>(p.n, 1) * 4));
        MRGFE_TRY(p.mahal.ensure(std::max<size_t>(p.n, 1) * 72));
        GicpPairDev d;
        d.src = p.d_src; d.cov_src = (p.ext_cov ? *p.ext_cov : p.cov).as<double>();
        d.tgt = voxel ? nullptr : e->target_points();  // tgt == nullptr tells the linearize / error kernels that cov_tgt holds voxel records
        d.cov_tgt = voxel ? e->voxel_records() : e->target_covariances();
        d.corr = p.corr.as<int32_t>(); d.mahal = p.mahal.as<double>();
        d.n = p.n; d.part_off = part; d.target = static_cast<uint32_t>(p.target); d.pad = 0;
        part += (p.n + 255u) / 256u;
        max_n = std::max(max_n, p.n);
        h_pairs[i] = d;
        p.ctl.start(e->params(), p.guess, p.n);
    }
    MRGFE_TRY(d_pairs_.ensure(sizeof(GicpPairDev) * P));
    MRGFE_TRY(d_grids_.ensure((voxel ? sizeof(VoxGridDev) : sizeof(NnGrid2Dev)) * std::max<size_t>(engines.size(), 1)));
    MRGFE_TRY(d_evals_.ensure(sizeof(GicpEvalDev) * P));
    MRGFE_TRY(d_partials_.ensure(sizeof(double) * kGicpStride * std::max<uint32_t>(part, 1)));
    MRGFE_TRY(h_evals_.ensure(sizeof(GicpEvalDev) * P));
    MRGFE_TRY(h_results_.ensure(sizeof(double) * kGicpStride * P));
    MRGFE_HIP_CHECK(hipMemcpyAsync(d_pairs_.p, h_pairs.data(), sizeof(GicpPairDev) * P, hipMemcpyHostToDevice, st));
    if (voxel) MRGFE_HIP_CHECK(hipMemcpyAsync(d_grids_.p, h_vgrids.data(), sizeof(VoxGridDev) * h_vgrids.size(), hipMemcpyHostToDevice, st));
    else       MRGFE_HIP_CHECK(hipMemcpyAsync(d_grids_.p, h_grids.data(), sizeof(NnGrid2Dev) * h_grids.size(), hipMemcpyHostToDevice, st));
    MRGFE_HIP_CHECK(hipStreamSynchronize(st));  // h_pairs / h_grids are locals
    GicpEvalDev* he = h_evals_.as<GicpEvalDev>();
    double*      hr = h_results_.as<double>();
    const double thr = engines[pairs[0].target]->params().max_corr_dist;
    const GicpParams& prm0 = engines[pairs[0].target]->params();
    const int    round_cap = (prm0.max_iterations + 1) * (std::max(prm0.lm_max_iterations, prm0.sg_max_inner_iterations) + 2) + 4;
    std::vector<NnFitnessJob> corr_jobs;
    for (int round = 0; round < round_cap; ++round) {
        uint32_t n_lin = 0, n_err = 0;
        size_t   lin_queries = 0;
        for (int i = 0; i < P; ++i) {
            GicpLmController& c = pairs[i].ctl;
            if (c.done() || pairs[i].n == 0) { he[i].type = -1; continue; }
            he[i].pose = make_pose(c.request().T, engines[pairs[i].target]->params().variant);
            he[i].thr2 = thr * thr;
            he[i].type = c.request().type;
            if (he[i].type == 0) { he[n_lin++].order[0] = static_cast<uint32_t>(i); lin_queries += pairs[i].n; }
            else                 he[n_err++].order[1] = static_cast<uint32_t>(i);
        }
        // an empty source still walks its LM loop on all-zero records (like the single engine does)
        bool any_empty = false;
        for (int i = 0; i < P; ++i)
            if (!pairs[i].ctl.done() && pairs[i].n == 0) any_empty = true;
        if (n_lin + n_err == 0 && !any_empty) return MRGFE_OK;
        if (n_lin + n_err) {
            MRGFE_HIP_CHECK(hipMemcpyAsync(d_evals_.p, he, sizeof(GicpEvalDev) * P, hipMemcpyHostToDevice, st));
            const GicpPairDev* dp = d_pairs_.as<GicpPairDev>();
            const GicpEvalDev* de = d_evals_.as<GicpEvalDev>();
            if (n_lin) {
                constexpr uint32_t per_blk = 256u / kGicpBatchGroup;
                if (voxel) {
                    hipLaunchKernelGGL(vox_corr_batch_kernel, dim3((max_n + 255) / 256, n_lin), dim3(256), 0, st, dp, de, d_grids_.as<VoxGridDev>());
                } else if (gicp_corr_passes(lin_queries)) {
                    corr_jobs.clear();
                    for (uint32_t w = 0; w < n_lin; ++w) {
                        const uint32_t i = he[w].order[0];
                        NnFitnessJob job;
                        job.grid = h_grids[pairs[i].target];
                        job.src = pairs[i].d_src;
                        job.n = pairs[i].n;
                        job.gicp_order = 1;
                        std::memcpy(job.T12, he[i].pose.Tf, sizeof(job.T12));
                        job.idx_out = pairs[i].corr.as<int32_t>();
                        corr_jobs.push_back(job);
                    }
                    MRGFE_TRY(nn_nearest_batch(ctx_, corr_jobs.data(), corr_jobs.size(), thr * thr));
                } else {
                    hipLaunchKernelGGL(gicp_corr_batch_kernel, dim3((max_n + per_blk - 1) / per_blk, n_lin), dim3(256), 0, st, dp, de, d_grids_.as<NnGrid2Dev>());
                }
                hipLaunchKernelGGL(gicp_linearize_batch_kernel, dim3((max_n + 255) / 256, n_lin), dim3(256), 0, st, dp, de, d_partials_.as<double>());
            }
            if (n_err) hipLaunchKernelGGL(gicp_error_batch_kernel, dim3((max_n + 255) / 256, n_err), dim3(256), 0, st, dp, de, d_partials_.as<double>());
            hipLaunchKernelGGL(gicp_reduce_batch_kernel, dim3(P), dim3(256), 0, st, dp, de, d_partials_.as<double>(), hr);
            MRGFE_HIP_CHECK(hipGetLastError());
            MRGFE_HIP_CHECK(hipEventRecord(done_, st));
            MRGFE_HIP_CHECK(hipEventSynchronize(done_));
        }
        const double zeros[kGicpStride] = {0};
        for (int i = 0; i < P; ++i) {
            GicpLmController& c = pairs[i].ctl;
            if (c.done()) continue;
            c.on_result(pairs[i].n ? hr + size_t(i) * kGicpStride : zeros);
        }
    }
    set_error("GICP batch did not terminate within %d rounds", round_cap);
    return MRGFE_ERR_STATE;
}

}  // namespace mrgfe
