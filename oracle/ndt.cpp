// oracle/ndt.cpp — TEST INFRASTRUCTURE (CPU oracle). Not part of the shipped product path.
//
// Restates (PARITY UNPINNED, see quirks.h / SURVEY.md Appendix A.2-A.5):
//   pclomp::VoxelGridCovariance<PointXYZI>::applyFilter, getNeighborhoodAtPoint{,7,1}, radiusSearch
//   pclomp::NormalDistributionsTransform::{computeTransformation, computeDerivatives, computeAngleDerivatives,
//     computePointDerivatives (float 4x6 / 24x6 form and double 3x6 / 18x6 form), updateDerivatives,
//     computeHessian, updateHessian, computeStepLengthMT, trialValueSelectionMT, updateIntervalMT}
//   pcl::Registration::align / getFitnessScore
// behind the reference seam /root/reference/include/mrg_slam/registrations.hpp:20 (NDT_OMP branch
// src/mrg_slam/registrations.cpp:130-148).
#include "ndt.h"

#include <omp.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>

#include "linalg.h"
#include "nn.h"
#include "quirks.h"

namespace orc {

// ------------------------------------------------------------------------------------------------------
// VoxelGridCovariance
// ------------------------------------------------------------------------------------------------------
int VoxelGridCovariance::build(const float* xyzi, int n, float leaf)
{
    leaves.clear(); index.clear(); n_valid = 0;
    leaf_size = leaf;
    inv_leaf  = 1.0f / leaf;
    // pcl::getMinMax3D over finite points
    float min_p[3] = {std::numeric_limits<float>::max(), std::numeric_limits<float>::max(), std::numeric_limits<float>::max()};
    float max_p[3] = {-min_p[0], -min_p[1], -min_p[2]};
    int   finite   = 0;
    for (int i = 0; i < n; ++i) {
        const float* p = xyzi + 4 * i;
        if (!std::isfinite(p[0]) || !std::isfinite(p[1]) || !std::isfinite(p[2])) continue;
        ++finite;
        for (int a = 0; a < 3; ++a) { min_p[a] = std::min(min_p[a], p[a]); max_p[a] = std::max(max_p[a], p[a]); }
    }
    if (finite == 0) return -2;
    int64_t dx = static_cast<int64_t>((max_p[0] - min_p[0]) * inv_leaf) + 1;
    int64_t dy = static_cast<int64_t>((max_p[1] - min_p[1]) * inv_leaf) + 1;
    int64_t dz = static_cast<int64_t>((max_p[2] - min_p[2]) * inv_leaf) + 1;
    if (dx * dy * dz > static_cast<int64_t>(std::numeric_limits<int32_t>::max())) return -1;
    for (int a = 0; a < 3; ++a) {
        min_b[a] = static_cast<int>(std::floor(min_p[a] * inv_leaf));
        max_b[a] = static_cast<int>(std::floor(max_p[a] * inv_leaf));
        div_b[a] = max_b[a] - min_b[a] + 1;
    }
    divb_mul[0] = 1; divb_mul[1] = div_b[0]; divb_mul[2] = div_b[0] * div_b[1];

    // first pass: accumulate in point order
    struct Acc { int n; double mean[3]; double cov[9]; float centroid[4]; };
    std::vector<Acc>             acc;
    std::vector<int>             keys;
    std::unordered_map<int, int> pos;
    pos.reserve(static_cast<size_t>(n) / 4 + 16);
    for (int i = 0; i < n; ++i) {
        const float* p = xyzi + 4 * i;
        if (!std::isfinite(p[0]) || !std::isfinite(p[1]) || !std::isfinite(p[2])) continue;
        int ijk0 = static_cast<int>(std::floor(p[0] * inv_leaf) - static_cast<float>(min_b[0]));
        int ijk1 = static_cast<int>(std::floor(p[1] * inv_leaf) - static_cast<float>(min_b[1]));
        int ijk2 = static_cast<int>(std::floor(p[2] * inv_leaf) - static_cast<float>(min_b[2]));
        int key  = ijk0 * divb_mul[0] + ijk1 * divb_mul[1] + ijk2 * divb_mul[2];
        auto it  = pos.find(key);
        int  li;
        if (it == pos.end()) {
            li = static_cast<int>(acc.size());
            pos.emplace(key, li);
            keys.push_back(key);
            Acc a; std::memset(&a, 0, sizeof(a));
            acc.push_back(a);
        } else {
            li = it->second;
        }
        Acc& a = acc[li];
        double pt[3] = {p[0], p[1], p[2]};
        for (int r = 0; r < 3; ++r) a.mean[r] += pt[r];
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) a.cov[r * 3 + c] += pt[r] * pt[c];
        for (int r = 0; r < 4; ++r) a.centroid[r] += p[r];
        ++a.n;
    }
    // second pass in ascending key order (std::map iteration)
    std::vector<int> ord(keys.size());
    for (size_t i = 0; i < ord.size(); ++i) ord[i] = static_cast<int>(i);
    std::sort(ord.begin(), ord.end(), [&](int a, int b) { return keys[a] < keys[b]; });
    leaves.resize(keys.size());
    for (size_t o = 0; o < ord.size(); ++o) {
        const Acc& a = acc[ord[o]];
        NdtLeaf&   L = leaves[o];
        std::memset(&L, 0, sizeof(L));
        L.key = keys[ord[o]];
        L.nr_points = a.n;
        index.emplace(L.key, static_cast<int>(o));
        for (int r = 0; r < 4; ++r) L.centroid[r] = a.centroid[r] / static_cast<float>(a.n);
        double pt_sum[3] = {a.mean[0], a.mean[1], a.mean[2]};
        for (int r = 0; r < 3; ++r) L.mean[r] = a.mean[r] / a.n;
        if (a.n < quirks::kNdtMinPointsPerVoxel) continue;
        L.in_search = 1;  // voxel_centroids_.push_back / voxel_centroids_leaf_indices_.push_back come before the checks below
        // single pass covariance, then PCL's (n-1)/n normalisation
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) L.cov[r * 3 + c] = (a.cov[r * 3 + c] - 2 * (pt_sum[r] * L.mean[c])) / a.n + L.mean[r] * L.mean[c];
        double f = (a.n - 1.0) / a.n;
        for (int k = 0; k < 9; ++k) L.cov[k] *= f;
        double ev[3], evec[9];
        sym_eig3(L.cov, ev, evec);
        if (ev[0] < -negative_eigen_tolerance || ev[1] < -negative_eigen_tolerance || ev[2] <= 0) { L.nr_points = -1; continue; }
        double min_ev = quirks::kNdtMinCovarEigvalMult * ev[2];
        if (ev[0] < min_ev) {
            ev[0] = min_ev;
            if (ev[1] < min_ev) ev[1] = min_ev;
            double D[9] = {ev[0], 0, 0, 0, ev[1], 0, 0, 0, ev[2]};
            double ED[9], Einv[9];
            mul3(evec, D, ED);
            inv3(evec, Einv);
            mul3(ED, Einv, L.cov);
        }
        inv3(L.cov, L.icov);
        double mx = L.icov[0], mn = L.icov[0];
        for (int k = 1; k < 9; ++k) { mx = std::max(mx, L.icov[k]); mn = std::min(mn, L.icov[k]); }
        if (mx == std::numeric_limits<double>::infinity() || mn == -std::numeric_limits<double>::infinity() || mx != mx || mn != mn) {
            L.nr_points = -1;
            continue;
        }
        ++n_valid;
    }
    return 0;
}

int VoxelGridCovariance::neighbours(float x, float y, float z, NdtSearch method, int out[27]) const
{
    // getNeighborhoodAtPoint: floor(p / leaf_size) (division, unlike the build's multiply by inverse)
    int ijk[3] = {static_cast<int>(std::floor(x / leaf_size)), static_cast<int>(std::floor(y / leaf_size)), static_cast<int>(std::floor(z / leaf_size))};
    static const int d7[7][3] = {{0, 0, 0}, {1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}};
    int cnt = 0;
    auto probe = [&](int ox, int oy, int oz) -> int {
        int c[3] = {ijk[0] + ox, ijk[1] + oy, ijk[2] + oz};
        for (int a = 0; a < 3; ++a) if (c[a] < min_b[a] || c[a] > max_b[a]) return -1;
        int key = (c[0] - min_b[0]) * divb_mul[0] + (c[1] - min_b[1]) * divb_mul[1] + (c[2] - min_b[2]) * divb_mul[2];
        auto it = index.find(key);
        if (it == index.end()) return -1;
        if (leaves[it->second].nr_points < quirks::kNdtMinPointsPerVoxel) return -1;
        return it->second;
    };
    if (method == NDT_DIRECT7 || method == NDT_DIRECT1) {
        int m = method == NDT_DIRECT7 ? 7 : 1;
        for (int k = 0; k < m; ++k) { int l = probe(d7[k][0], d7[k][1], d7[k][2]); if (l >= 0) out[cnt++] = l; }
        return cnt;
    }
    if (method == NDT_DIRECT26) {
        // pcl::getAllNeighborCellIndices(): 27 offsets, x fastest? -> order only affects f64 summation order
        for (int ox = -1; ox <= 1; ++ox) for (int oy = -1; oy <= 1; ++oy) for (int oz = -1; oz <= 1; ++oz) {
            int l = probe(ox, oy, oz); if (l >= 0) out[cnt++] = l;
        }
        return cnt;
    }
    // KDTREE: radiusSearch(point, resolution) over the centroids the kd-tree holds (sorted by distance)
    return radius_neighbours(x, y, z, out);
}

int VoxelGridCovariance::radius_neighbours(float x, float y, float z, int out[27]) const
{
    // kdtree_.radiusSearch(point, radius) over voxel_centroids_ (float centroids of the leaves with >= 6 points, ascending key), FLANN L2_Simple in
    // float, dist^2 < float(radius * radius), sorted by (distance, index); then leaves_.find(voxel_centroids_leaf_indices_[k]) — no nr_points test.
    // A centroid within one leaf size of the point lies in one of the 27 cells around the point's cell.
    const int ijk[3] = {static_cast<int>(std::floor(x / leaf_size)), static_cast<int>(std::floor(y / leaf_size)), static_cast<int>(std::floor(z / leaf_size))};
    std::pair<float, int> found[27];
    int cnt = 0;
    const float r2 = leaf_size * leaf_size;
    for (int ox = -1; ox <= 1; ++ox) for (int oy = -1; oy <= 1; ++oy) for (int oz = -1; oz <= 1; ++oz) {
        const int c[3] = {ijk[0] + ox, ijk[1] + oy, ijk[2] + oz};
        bool inside = true;
        for (int a = 0; a < 3; ++a) inside = inside && c[a] >= min_b[a] && c[a] <= max_b[a];
        if (!inside) continue;
        const int key = (c[0] - min_b[0]) * divb_mul[0] + (c[1] - min_b[1]) * divb_mul[1] + (c[2] - min_b[2]) * divb_mul[2];
        auto it = index.find(key);
        if (it == index.end() || !leaves[it->second].in_search) continue;
        const float* ce = leaves[it->second].centroid;
        const float d = sqdist_f(ce[0], ce[1], ce[2], x, y, z);
        if (d < r2) found[cnt++] = std::make_pair(d, it->second);
    }
    std::sort(found, found + cnt);
    for (int k = 0; k < cnt; ++k) out[k] = found[k].second;
    return cnt;
}

// ------------------------------------------------------------------------------------------------------
// NDT
// ------------------------------------------------------------------------------------------------------
void Ndt::init_gauss()
{
    double gauss_c1 = 10 * (1 - outlier_ratio);
    double gauss_c2 = outlier_ratio / std::pow(static_cast<double>(resolution), 3);
    gauss_d3 = -std::log(gauss_c2);
    gauss_d1 = -std::log(gauss_c1 + gauss_c2) - gauss_d3;
    gauss_d2 = -2 * std::log((-std::log(gauss_c1 * std::exp(-0.5) + gauss_c2) - gauss_d3) / gauss_d1);
}

int Ndt::set_target(const float* xyzi, int n)
{
    target.assign(xyzi, xyzi + static_cast<size_t>(n) * 4);
    target_status = cells.build(target.data(), n, resolution);
    return target_status;
}

void Ndt::set_source(const float* xyzi, int n) { source.assign(xyzi, xyzi + static_cast<size_t>(n) * 4); }

void Ndt::angle_derivatives(const double p[6], bool compute_hessian)
{
    double cx, cy, cz, sx, sy, sz;
    if (std::fabs(p[3]) < quirks::kNdtSmallAngle) { cx = 1.0; sx = 0.0; } else { cx = std::cos(p[3]); sx = std::sin(p[3]); }
    if (std::fabs(p[4]) < quirks::kNdtSmallAngle) { cy = 1.0; sy = 0.0; } else { cy = std::cos(p[4]); sy = std::sin(p[4]); }
    if (std::fabs(p[5]) < quirks::kNdtSmallAngle) { cz = 1.0; sz = 0.0; } else { cz = std::cos(p[5]); sz = std::sin(p[5]); }
    double j[8][3] = {
        {(-sx * sz + cx * sy * cz), (-sx * cz - cx * sy * sz), (-cx * cy)},  // a
        {(cx * sz + sx * sy * cz), (cx * cz - sx * sy * sz), (-sx * cy)},    // b
        {(-sy * cz), sy * sz, cy},                                           // c
        {sx * cy * cz, (-sx * cy * sz), sx * sy},                            // d
        {(-cx * cy * cz), cx * cy * sz, (-cx * sy)},                         // e
        {(-cy * sz), (-cy * cz), 0},                                         // f
        {(cx * cz - sx * sy * sz), (-cx * sz - sx * sy * cz), 0},            // g
        {(sx * cz + cx * sy * sz), (cx * sy * cz - sx * sz), 0}};            // h
    for (int r = 0; r < 8; ++r) for (int c = 0; c < 3; ++c) { j_ang_d[r][c] = j[r][c]; j_ang_f[r][c] = static_cast<float>(j[r][c]); }
    if (!compute_hessian) return;
    double h[15][3] = {
        {(-cx * sz - sx * sy * cz), (-cx * cz + sx * sy * sz), sx * cy},     // a2
        {(-sx * sz + cx * sy * cz), (-cx * sy * sz - sx * cz), (-cx * cy)},  // a3
        {(cx * cy * cz), (-cx * cy * sz), (cx * sy)},                        // b2
        {(sx * cy * cz), (-sx * cy * sz), (sx * sy)},                        // b3
        {(-sx * cz - cx * sy * sz), (sx * sz - cx * sy * cz), 0},            // c2
        {(cx * cz - sx * sy * sz), (-sx * sy * cz - cx * sz), 0},            // c3
        {(-cy * cz), (cy * sz), (sy)},                                       // d1
        {(-sx * sy * cz), (sx * sy * sz), (sx * cy)},                        // d2
        {(cx * sy * cz), (-cx * sy * sz), (-cx * cy)},                       // d3
        {(sy * sz), (sy * cz), 0},                                           // e1
        {(-sx * cy * sz), (-sx * cy * cz), 0},                               // e2
        {(cx * cy * sz), (cx * cy * cz), 0},                                 // e3
        {(-cy * cz), (cy * sz), 0},                                          // f1
        {(-cx * sz - sx * sy * cz), (-cx * cz + sx * sy * sz), 0},           // f2
        {(-sx * sz + cx * sy * cz), (-cx * sy * sz - sx * cz), 0}};          // f3
    for (int r = 0; r < 15; ++r) for (int c = 0; c < 3; ++c) { h_ang_d[r][c] = h[r][c]; h_ang_f[r][c] = static_cast<float>(h[r][c]); }
}

void Ndt::transform_cloud(const float T[16])
{
    int n = static_cast<int>(source.size() / 4);
    trans_.resize(static_cast<size_t>(n) * 3);
    for (int i = 0; i < n; ++i) {
        const float* p = &source[4 * static_cast<size_t>(i)];
        transform_point_f(T, p[0], p[1], p[2], trans_[3 * i], trans_[3 * i + 1], trans_[3 * i + 2]);
    }
}

// Three-term products accumulated left to right.  FUSED: s = a0*b0; s = fma(a1,b1,s); s = fma(a2,b2,s) - the sequence a build
// of the reference with hardware FMA produces (its aarch64 flags, /root/reference/CMakeLists.txt:19); otherwise the SSE-only
// x86 sequence (CMakeLists.txt:15): every product and sum rounded separately.  The reference's float order is not part of
// its contract (quirks.h); the HIP kernels execute the FUSED sequence bit for bit.
template <bool FUSED>
static inline float dot3f(float a0, float b0, float a1, float b1, float a2, float b2)
{
    if (FUSED) return std::fmaf(a2, b2, std::fmaf(a1, b1, a0 * b0));
    float p0 = a0 * b0, p1 = a1 * b1, p2 = a2 * b2;
    float s = p0 + p1;
    return s + p2;
}
template <bool FUSED>
static inline double dot3d(double a0, double b0, double a1, double b1, double a2, double b2)
{
    if (FUSED) return std::fma(a2, b2, std::fma(a1, b1, a0 * b0));
    return a0 * b0 + a1 * b1 + a2 * b2;
}

// updateDerivatives for one (point, voxel) pair, float form: score increment, gradient terms t_g[6] and (compute_hessian) the 36
// Hessian terms t_h, each the rounded float the reference widens to double and adds.  false: the pair fails the e_x_cov_x range check
// and contributes nothing.
namespace {
struct PointTermsF {
    float xt[3];
    float J3[3], J4[3], J5[3];
    float ha[3], hb[3], hc[3], hd[3], he[3], hf[3];
};
template <bool FUSED>
inline void point_terms_f(const float xt[3], const float x4[3], const float j_ang_f[8][3], const float h_ang_f[15][3], PointTermsF& P)
{
    // computePointDerivatives (float form): x_j_ang = j_ang * x4 ; x_h_ang = h_ang * x4
    float xj[8], xh[15];
    for (int r = 0; r < 8; ++r) xj[r] = dot3f<FUSED>(j_ang_f[r][0], x4[0], j_ang_f[r][1], x4[1], j_ang_f[r][2], x4[2]);
    for (int r = 0; r < 15; ++r) xh[r] = dot3f<FUSED>(h_ang_f[r][0], x4[0], h_ang_f[r][1], x4[1], h_ang_f[r][2], x4[2]);
    for (int a = 0; a < 3; ++a) P.xt[a] = xt[a];
    // point_gradient4 columns 3..5 (rows 0..2); columns 0..2 = identity
    P.J3[0] = 0.0f; P.J3[1] = xj[0]; P.J3[2] = xj[1];
    P.J4[0] = xj[2]; P.J4[1] = xj[3]; P.J4[2] = xj[4];
    P.J5[0] = xj[5]; P.J5[1] = xj[6]; P.J5[2] = xj[7];
    // point_hessian blocks (3-vectors): a,b,c,d,e,f
    P.ha[0] = 0.0f; P.ha[1] = xh[0]; P.ha[2] = xh[1];
    P.hb[0] = 0.0f; P.hb[1] = xh[2]; P.hb[2] = xh[3];
    P.hc[0] = 0.0f; P.hc[1] = xh[4]; P.hc[2] = xh[5];
    P.hd[0] = xh[6]; P.hd[1] = xh[7]; P.hd[2] = xh[8];
    P.he[0] = xh[9]; P.he[1] = xh[10]; P.he[2] = xh[11];
    P.hf[0] = xh[12]; P.hf[1] = xh[13]; P.hf[2] = xh[14];
}
template <bool FUSED>
inline bool pair_terms_f(const PointTermsF& P, const NdtLeaf& cell, float gauss_d2f, double gd1, bool compute_hessian, float* score_inc_out, float t_g[6], float t_h[36])
{
    const float* xt = P.xt;
    const float *J3 = P.J3, *J4 = P.J4, *J5 = P.J5;
    // PH[i][j] for i,j in 3..5 : block (i*4, j)
    const float* PH[3][3] = {{P.ha, P.hb, P.hc}, {P.hb, P.hd, P.he}, {P.hc, P.he, P.hf}};
    // x_trans (double) -= mean ; cast to float
    const float q[3] = {static_cast<float>(static_cast<double>(xt[0]) - cell.mean[0]), static_cast<float>(static_cast<double>(xt[1]) - cell.mean[1]),
                        static_cast<float>(static_cast<double>(xt[2]) - cell.mean[2])};
    float C[9];
    for (int t = 0; t < 9; ++t) C[t] = static_cast<float>(cell.icov[t]);
    // qC = x_trans4 * c_inv4 (row vector times matrix)
    float qC[3];
    for (int c = 0; c < 3; ++c) qC[c] = dot3f<FUSED>(q[0], C[0 * 3 + c], q[1], C[1 * 3 + c], q[2], C[2 * 3 + c]);
    float qCq = dot3f<FUSED>(q[0], qC[0], q[1], qC[1], q[2], qC[2]);
    float arg0 = -gauss_d2f * qCq;
    float arg = arg0 * 0.5f;
    float e_x_cov_x = static_cast<float>(std::exp(static_cast<double>(arg)));
    float score_inc = static_cast<float>(-gd1 * static_cast<double>(e_x_cov_x));
    e_x_cov_x = gauss_d2f * e_x_cov_x;
    if (e_x_cov_x > 1 || e_x_cov_x < 0 || e_x_cov_x != e_x_cov_x) return false;
    e_x_cov_x = static_cast<float>(static_cast<double>(e_x_cov_x) * gd1);
    // CJ = c_inv4 * point_gradient4 : columns 0..2 = C, columns 3..5 = C * J3/J4/J5
    float CJ[3][6];
    for (int r = 0; r < 3; ++r) {
        CJ[r][0] = C[r * 3 + 0]; CJ[r][1] = C[r * 3 + 1]; CJ[r][2] = C[r * 3 + 2];
        CJ[r][3] = dot3f<FUSED>(C[r * 3 + 0], J3[0], C[r * 3 + 1], J3[1], C[r * 3 + 2], J3[2]);
        CJ[r][4] = dot3f<FUSED>(C[r * 3 + 0], J4[0], C[r * 3 + 1], J4[1], C[r * 3 + 2], J4[2]);
        CJ[r][5] = dot3f<FUSED>(C[r * 3 + 0], J5[0], C[r * 3 + 1], J5[1], C[r * 3 + 2], J5[2]);
    }
    float qCJ[6];
    for (int c = 0; c < 6; ++c) qCJ[c] = dot3f<FUSED>(q[0], CJ[0][c], q[1], CJ[1][c], q[2], CJ[2][c]);
    for (int c = 0; c < 6; ++c) t_g[c] = e_x_cov_x * qCJ[c];
    *score_inc_out = score_inc;
    if (!compute_hessian) return true;
    // JtCJ(a,b) = J(:,a) . CJ(:,b)
    const float* Jc[6] = {nullptr, nullptr, nullptr, J3, J4, J5};
    float JtCJ[6][6];
    for (int a = 0; a < 6; ++a)
        for (int b = 0; b < 6; ++b)
            JtCJ[a][b] = (a < 3) ? CJ[a][b] : dot3f<FUSED>(Jc[a][0], CJ[0][b], Jc[a][1], CJ[1][b], Jc[a][2], CJ[2][b]);
    for (int i = 0; i < 6; ++i) {
        float qCH[6] = {0, 0, 0, 0, 0, 0};
        if (i >= 3)
            for (int j = 3; j < 6; ++j) { const float* v = PH[i - 3][j - 3]; qCH[j] = dot3f<FUSED>(qC[0], v[0], qC[1], v[1], qC[2], v[2]); }
        for (int j = 0; j < 6; ++j) {
            float t0 = -gauss_d2f * qCJ[i];
            float t2;
            if (FUSED) { t2 = std::fmaf(t0, qCJ[j], qCH[j]); }
            else       { float t1 = t0 * qCJ[j]; t2 = t1 + qCH[j]; }
            float t3 = t2 + JtCJ[j][i];
            t_h[i * 6 + j] = e_x_cov_x * t3;
        }
    }
    return true;
}
}  // namespace

double Ndt::compute_derivatives(double grad[6], double hess[36], const double p[6], bool compute_hessian)
{
    for (int k = 0; k < 6; ++k) last_p[k] = p[k];
    if (gpu_order_ppt > 0) return compute_derivatives_gpu_order(grad, hess, p, compute_hessian);
    return fused ? compute_derivatives_impl<true>(grad, hess, p, compute_hessian) : compute_derivatives_impl<false>(grad, hess, p, compute_hessian);
}

void Ndt::compute_hessian(double hess[36], const double p[6])
{
    for (int k = 0; k < 6; ++k) last_p[k] = p[k];
    if (gpu_order_ppt > 0) { compute_hessian_gpu_order(hess, p); return; }
    if (fused) compute_hessian_impl<true>(hess, p);
    else       compute_hessian_impl<false>(hess, p);
}

template <bool FUSED>
double Ndt::compute_derivatives_impl(double grad[6], double hess[36], const double p[6], bool compute_hessian)
{
    const int n = static_cast<int>(source.size() / 4);
    angle_derivatives(p, true);
    ++n_evals;
    const float  gauss_d2f = static_cast<float>(gauss_d2);
    const double gd1       = gauss_d1;
    long long    nb_total  = 0;
    if (thread_sums) {
        // ndt_omp's own shape: scores / score_gradients / hessians per thread (`thread_n = omp_get_thread_num()`), summed over the threads afterwards.
        // schedule(static) where upstream has schedule(guided, 8): with a dynamic schedule the sums of two evaluations AT THE SAME POSE differ in the
        // last bits, and More-Thuente with mrg_slam's parameters (every step clamped to [eps / 2, 0.1]) re-evaluates the pose it has just been at —
        // the interval update then divides a rounding difference by zero and the search ends in a NaN step now and then (seen in a quarter of the
        // runs of one bench pair with 4 threads).  Static chunks keep re-evaluations bitwise repeatable; the thread-order sums remain.
        const int nt = std::max(1, num_threads);
        std::vector<double> acc(static_cast<size_t>(nt) * 48, 0.0);
#pragma omp parallel for num_threads(num_threads) schedule(static) reduction(+ : nb_total)
        for (int idx = 0; idx < n; ++idx) {
            const float xt[3] = {trans_[3 * idx], trans_[3 * idx + 1], trans_[3 * idx + 2]};
            int nb[27];
            int cnt = cells.neighbours(xt[0], xt[1], xt[2], search, nb);
            if (cnt == 0) continue;
            nb_total += cnt;
            const float* xp = &source[4 * static_cast<size_t>(idx)];
            const float x4[3] = {xp[0], xp[1], xp[2]};
            PointTermsF P;
            point_terms_f<FUSED>(xt, x4, j_ang_f, h_ang_f, P);
            double pt[43] = {0};
            for (int k = 0; k < cnt; ++k) {
                float score_inc, t_g[6], t_h[36];
                if (!pair_terms_f<FUSED>(P, cells.leaves[nb[k]], gauss_d2f, gd1, compute_hessian, &score_inc, t_g, t_h)) continue;
                pt[0] += static_cast<double>(score_inc);
                for (int c = 0; c < 6; ++c) pt[1 + c] += static_cast<double>(t_g[c]);
                if (!compute_hessian) continue;
                for (int c = 0; c < 36; ++c) pt[7 + c] += static_cast<double>(t_h[c]);
            }
            double* a = &acc[static_cast<size_t>(omp_get_thread_num()) * 48];
            for (int c = 0; c < 43; ++c) a[c] += pt[c];
        }
        neighbours_sum += n > 0 ? static_cast<double>(nb_total) / n : 0.0;
        double score = 0;
        for (int k = 0; k < 6; ++k) grad[k] = 0;
        for (int k = 0; k < 36; ++k) hess[k] = 0;
        for (int t = 0; t < nt; ++t) {
            score += acc[static_cast<size_t>(t) * 48];
            for (int k = 0; k < 6; ++k) grad[k] += acc[static_cast<size_t>(t) * 48 + 1 + k];
            for (int k = 0; k < 36; ++k) hess[k] += acc[static_cast<size_t>(t) * 48 + 7 + k];
        }
        return score;
    }
    scores_.assign(n, 0.0);
    grads_.assign(static_cast<size_t>(n) * 6, 0.0);
    hessians_.assign(static_cast<size_t>(n) * 36, 0.0);

#pragma omp parallel for num_threads(num_threads) schedule(guided, 8) reduction(+ : nb_total)
    for (int idx = 0; idx < n; ++idx) {
        const float xt[3] = {trans_[3 * idx], trans_[3 * idx + 1], trans_[3 * idx + 2]};
        int nb[27];
        int cnt = cells.neighbours(xt[0], xt[1], xt[2], search, nb);
        if (cnt == 0) continue;
        nb_total += cnt;
        const float* xp = &source[4 * static_cast<size_t>(idx)];
        // Vector3d x(x_pt.x, ..) -> Vector4f x4: float -> double -> float is the identity
        const float x4[3] = {xp[0], xp[1], xp[2]};
        PointTermsF P;
        point_terms_f<FUSED>(xt, x4, j_ang_f, h_ang_f, P);

        double  score_pt = 0;
        double* g_pt = &grads_[static_cast<size_t>(idx) * 6];
        double* h_pt = &hessians_[static_cast<size_t>(idx) * 36];
        for (int k = 0; k < cnt; ++k) {
            float score_inc, t_g[6], t_h[36];
            if (!pair_terms_f<FUSED>(P, cells.leaves[nb[k]], gauss_d2f, gd1, compute_hessian, &score_inc, t_g, t_h)) continue;
            for (int c = 0; c < 6; ++c) g_pt[c] += static_cast<double>(t_g[c]);
            score_pt += static_cast<double>(score_inc);
            if (!compute_hessian) continue;
            for (int c = 0; c < 36; ++c) h_pt[c] += static_cast<double>(t_h[c]);
        }
        scores_[idx] = score_pt;
    }
    neighbours_sum += n > 0 ? static_cast<double>(nb_total) / n : 0.0;
    // "Ensure that the result is invariant against the summing up order": sequential sum in index order
    double score = 0;
    for (int k = 0; k < 6; ++k) grad[k] = 0;
    for (int k = 0; k < 36; ++k) hess[k] = 0;
    for (int i = 0; i < n; ++i) {
        score += scores_[i];
        for (int k = 0; k < 6; ++k) grad[k] += grads_[static_cast<size_t>(i) * 6 + k];
        for (int k = 0; k < 36; ++k) hess[k] += hessians_[static_cast<size_t>(i) * 36 + k];
    }
    return score;
}

// computeHessian: serial, double precision (PCL's original 3x6 / 18x6 point derivative forms)
template <bool FUSED>
void Ndt::compute_hessian_impl(double hess[36], const double p[6])
{
    (void)p;  // angular derivative tables are those of the last computeDerivatives call (same pose)
    const int n = static_cast<int>(source.size() / 4);
    for (int k = 0; k < 36; ++k) hess[k] = 0;
    ++n_evals;
    long long nb_total = 0;
    for (int idx = 0; idx < n; ++idx) {
        const float xt[3] = {trans_[3 * idx], trans_[3 * idx + 1], trans_[3 * idx + 2]};
        int nb[27];
        int cnt = cells.neighbours(xt[0], xt[1], xt[2], search, nb);
        if (cnt == 0) continue;
        nb_total += cnt;
        const float* xp = &source[4 * static_cast<size_t>(idx)];
        const double x[3] = {xp[0], xp[1], xp[2]};
        auto dotd = [&](const double* a) { return dot3d<FUSED>(x[0], a[0], x[1], a[1], x[2], a[2]); };
        double J[3][6] = {{1, 0, 0, 0, 0, 0}, {0, 1, 0, 0, 0, 0}, {0, 0, 1, 0, 0, 0}};
        J[1][3] = dotd(j_ang_d[0]); J[2][3] = dotd(j_ang_d[1]);
        J[0][4] = dotd(j_ang_d[2]); J[1][4] = dotd(j_ang_d[3]); J[2][4] = dotd(j_ang_d[4]);
        J[0][5] = dotd(j_ang_d[5]); J[1][5] = dotd(j_ang_d[6]); J[2][5] = dotd(j_ang_d[7]);
        double a[3] = {0, dotd(h_ang_d[0]), dotd(h_ang_d[1])}, b[3] = {0, dotd(h_ang_d[2]), dotd(h_ang_d[3])}, c[3] = {0, dotd(h_ang_d[4]), dotd(h_ang_d[5])};
        double d[3] = {dotd(h_ang_d[6]), dotd(h_ang_d[7]), dotd(h_ang_d[8])}, e[3] = {dotd(h_ang_d[9]), dotd(h_ang_d[10]), dotd(h_ang_d[11])};
        double f[3] = {dotd(h_ang_d[12]), dotd(h_ang_d[13]), dotd(h_ang_d[14])};
        const double  zero3[3] = {0, 0, 0};
        const double* PH[6][6];
        for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) PH[i][j] = zero3;
        PH[3][3] = a; PH[4][3] = b; PH[5][3] = c; PH[3][4] = b; PH[4][4] = d; PH[5][4] = e; PH[3][5] = c; PH[4][5] = e; PH[5][5] = f;
        for (int k = 0; k < cnt; ++k) {
            const NdtLeaf& cell = cells.leaves[nb[k]];
            const double q[3] = {static_cast<double>(xt[0]) - cell.mean[0], static_cast<double>(xt[1]) - cell.mean[1], static_cast<double>(xt[2]) - cell.mean[2]};
            const double* C = cell.icov;
            auto Cv = [&](const double* v, double out[3]) { for (int r = 0; r < 3; ++r) out[r] = dot3d<FUSED>(C[r * 3 + 0], v[0], C[r * 3 + 1], v[1], C[r * 3 + 2], v[2]); };
            auto qdot = [&](const double* v) { return dot3d<FUSED>(q[0], v[0], q[1], v[1], q[2], v[2]); };
            double Cq[3];
            Cv(q, Cq);
            double e_x_cov_x = gauss_d2 * std::exp(-gauss_d2 * qdot(Cq) / 2);
            if (e_x_cov_x > 1 || e_x_cov_x < 0 || e_x_cov_x != e_x_cov_x) continue;
            e_x_cov_x *= gauss_d1;
            for (int i = 0; i < 6; ++i) {
                double Ji[3] = {J[0][i], J[1][i], J[2][i]}, cov_dxd_pi[3];
                Cv(Ji, cov_dxd_pi);
                for (int j = 0; j < 6; ++j) {
                    double Jj[3] = {J[0][j], J[1][j], J[2][j]}, CJj[3], CH[3];
                    Cv(Jj, CJj);
                    Cv(PH[i][j], CH);
                    const double jtcj = dot3d<FUSED>(Jj[0], cov_dxd_pi[0], Jj[1], cov_dxd_pi[1], Jj[2], cov_dxd_pi[2]);
                    const double t0 = -gauss_d2 * qdot(cov_dxd_pi);
                    if (FUSED) hess[i * 6 + j] = std::fma(e_x_cov_x, std::fma(t0, qdot(CJj), qdot(CH)) + jtcj, hess[i * 6 + j]);
                    else       hess[i * 6 + j] += e_x_cov_x * (t0 * qdot(CJj) + qdot(CH) + jtcj);
                }
            }
        }
    }
    neighbours_sum += n > 0 ? static_cast<double>(nb_total) / n : 0.0;
}

// ---- diagnostic: the HIP kernels' operation ORDER on the CPU ------------------------------------------------------------------
// gpu_order_ppt > 0 makes compute_derivatives / compute_hessian add their terms in the order ndt_derivatives_kernel and
// ndt_reduce_kernel do (mrg_slam_amd/csrc/ndt_derivatives.hip): items of gpu_order_ppt tiles of 256 points; per tile the occupied
// (point, voxel) pairs queued point by point in probe order and dealt round-robin to 256 lanes (score + gradient + Hessian) or kept
// on the lane of their point (score + gradient only), every lane adding its pairs' float terms to f64 accumulators; a 64-lane shuffle tree (offsets 32..1), the four waves as ((w0 + w1) + w2) + w3; the item partials of an
// evaluation in four interleaved slices, combined the same way.  The f64 Hessian pass is the kernel's per-point factorisation (one lane
// per point).  With the same per-pair float terms this reproduces the GPU's sums bit for bit (float path; the f64 pass differs where
// the two C libraries' exp differ in the last bit), so a test can tell summation-order noise from an arithmetic difference.
void gpu_tree_reduce(std::vector<double>& acc /* [256][48] */, double out[48])
{
    double waves[4][48];
    for (int w = 0; w < 4; ++w)
        for (int k = 0; k < 48; ++k) {
            double v[64];
            for (int l = 0; l < 64; ++l) v[l] = acc[static_cast<size_t>(w * 64 + l) * 48 + k];
            for (int off = 32; off > 0; off >>= 1)
                for (int l = 0; l < off; ++l) v[l] = v[l] + v[l + off];
            waves[w][k] = v[0];
        }
    for (int k = 0; k < 48; ++k) out[k] = ((waves[0][k] + waves[1][k]) + waves[2][k]) + waves[3][k];
}
void gpu_slice_reduce(const std::vector<double>& partials /* [nblk][48] */, size_t nblk, double out[48])
{
    for (int k = 0; k < 48; ++k) {
        double s[4] = {0, 0, 0, 0};
        for (int sl = 0; sl < 4; ++sl)
            for (size_t b = sl; b < nblk; b += 4) s[sl] += partials[b * 48 + k];
        out[k] = ((s[0] + s[1]) + s[2]) + s[3];
    }
}

int Ndt::neighbours_probe_order(float x, float y, float z, int out[27]) const
{
    if (search != NDT_KDTREE) return cells.neighbours(x, y, z, search, out);
    // the kernel keeps the 27 probes in probe order and drops the centroids outside the radius; the reference sorts by distance
    int found[27];
    const int m = cells.radius_neighbours(x, y, z, found);
    // probe order of the kernel = ascending (ox, oy, oz) = ascending key offset order used by radius_neighbours' loops: re-sort by leaf key offset
    std::pair<int, int> byprobe[27];
    const int ijk[3] = {static_cast<int>(std::floor(x / cells.leaf_size)), static_cast<int>(std::floor(y / cells.leaf_size)), static_cast<int>(std::floor(z / cells.leaf_size))};
    for (int k = 0; k < m; ++k) {
        // recover the cell of the leaf from its key
        int key = cells.leaves[found[k]].key;
        const int cz = key / cells.divb_mul[2]; key -= cz * cells.divb_mul[2];
        const int cy = key / cells.divb_mul[1]; key -= cy * cells.divb_mul[1];
        const int cx = key;
        const int ox = cx + cells.min_b[0] - ijk[0] + 1, oy = cy + cells.min_b[1] - ijk[1] + 1, oz = cz + cells.min_b[2] - ijk[2] + 1;
        byprobe[k] = std::make_pair(ox * 9 + oy * 3 + oz, found[k]);
    }
    std::sort(byprobe, byprobe + m);
    for (int k = 0; k < m; ++k) out[k] = byprobe[k].second;
    return m;
}

double Ndt::compute_derivatives_gpu_order(double grad[6], double hess[36], const double p[6], bool compute_hessian)
{
    const int n = static_cast<int>(source.size() / 4);
    angle_derivatives(p, true);
    ++n_evals;
    const float  gauss_d2f = static_cast<float>(gauss_d2);
    const double gd1 = gauss_d1;
    const int    per_item = 256 * gpu_order_ppt;
    const size_t nblk = static_cast<size_t>((n + per_item - 1) / per_item);
    std::vector<double> partials(std::max<size_t>(nblk, 1) * 48, 0.0);
    long long nb_total = 0;
    // (items are independent and every item owns its partial record: the loop may run on any number of threads without touching the order of a sum)
#pragma omp parallel for num_threads(std::max(1, num_threads)) schedule(dynamic, 1) reduction(+ : nb_total)
    for (size_t item = 0; item < nblk; ++item) {
        std::vector<double> acc(256 * 48, 0.0);
        const int base = static_cast<int>(item) * per_item, last = std::min(n, base + per_item);
        for (int tile0 = base; tile0 < last; tile0 += 256) {
            std::vector<PointTermsF> pts(256);
            std::vector<std::pair<int, int>> queue;  // (slot, leaf)
            for (int t = 0; t < 256 && tile0 + t < last; ++t) {
                const int idx = tile0 + t;
                const float xt[3] = {trans_[3 * idx], trans_[3 * idx + 1], trans_[3 * idx + 2]};
                int nb[27];
                const int cnt = neighbours_probe_order(xt[0], xt[1], xt[2], nb);
                if (!cnt) continue;
                const float* xp = &source[4 * static_cast<size_t>(idx)];
                const float x4[3] = {xp[0], xp[1], xp[2]};
                point_terms_f<true>(xt, x4, j_ang_f, h_ang_f, pts[t]);
                for (int k = 0; k < cnt; ++k) queue.emplace_back(t, nb[k]);
            }
            nb_total += static_cast<long long>(queue.size());
            for (size_t qi = 0; qi < queue.size(); ++qi) {
                // score + gradient + Hessian: queue slots dealt round-robin to the 256 lanes; score + gradient only (line-search trials):
                // the kernel keeps one lane per POINT, which walks its voxels in probe order
                double* a = &acc[static_cast<size_t>(compute_hessian ? qi % 256 : static_cast<size_t>(queue[qi].first)) * 48];
                float score_inc, t_g[6], t_h[36];
                if (!pair_terms_f<true>(pts[queue[qi].first], cells.leaves[queue[qi].second], gauss_d2f, gd1, compute_hessian, &score_inc, t_g, t_h)) continue;
                for (int c = 0; c < 6; ++c) a[1 + c] += static_cast<double>(t_g[c]);
                a[0] += static_cast<double>(score_inc);
                if (compute_hessian)
                    for (int c = 0; c < 36; ++c) a[7 + c] += static_cast<double>(t_h[c]);
            }
        }
        gpu_tree_reduce(acc, &partials[item * 48]);
    }
    double r[48];
    gpu_slice_reduce(partials, nblk, r);
    neighbours_sum += n > 0 ? static_cast<double>(nb_total) / n : 0.0;
    for (int k = 0; k < 6; ++k) grad[k] = r[1 + k];
    for (int k = 0; k < 36; ++k) hess[k] = compute_hessian ? r[7 + k] : 0.0;
    return r[0];
}

void Ndt::compute_hessian_gpu_order(double hess[36], const double p[6])
{
    (void)p;
    const int n = static_cast<int>(source.size() / 4);
    ++n_evals;
    const int    per_item = 256 * gpu_order_ppt;
    const size_t nblk = static_cast<size_t>((n + per_item - 1) / per_item);
    std::vector<double> partials(std::max<size_t>(nblk, 1) * 48, 0.0);
    long long nb_total = 0;
    auto fd3 = [](double a0, double b0, double a1, double b1, double a2, double b2) { return std::fma(a2, b2, std::fma(a1, b1, a0 * b0)); };
    auto fd3z = [](double a1, double b1, double a2, double b2) { return std::fma(a2, b2, a1 * b1); };
    // (items are independent and every item owns its partial record: the loop may run on any number of threads without touching the order of a sum)
#pragma omp parallel for num_threads(std::max(1, num_threads)) schedule(dynamic, 1) reduction(+ : nb_total)
    for (size_t item = 0; item < nblk; ++item) {
        std::vector<double> acc(256 * 48, 0.0);
        const int base = static_cast<int>(item) * per_item, last = std::min(n, base + per_item);
        for (int idx = base; idx < last; ++idx) {
            double* H = &acc[static_cast<size_t>((idx - base) % 256) * 48] + 7;  // this lane's 36 Hessian slots (upper triangle used)
            const float xt[3] = {trans_[3 * idx], trans_[3 * idx + 1], trans_[3 * idx + 2]};
            int nb[27];
            const int cnt = neighbours_probe_order(xt[0], xt[1], xt[2], nb);
            nb_total += cnt;
            if (!cnt) continue;
            const float* xp = &source[4 * static_cast<size_t>(idx)];
            const double x[3] = {xp[0], xp[1], xp[2]};
            double M1[6] = {0, 0, 0, 0, 0, 0}, M2[6] = {0, 0, 0, 0, 0, 0}, w[3] = {0, 0, 0};
            for (int k = 0; k < cnt; ++k) {
                const NdtLeaf& cell = cells.leaves[nb[k]];
                const double*  C = cell.icov;
                const double   q[3] = {static_cast<double>(xt[0]) - cell.mean[0], static_cast<double>(xt[1]) - cell.mean[1], static_cast<double>(xt[2]) - cell.mean[2]};
                double v[3];
                for (int r = 0; r < 3; ++r) v[r] = fd3(C[r * 3 + 0], q[0], C[r * 3 + 1], q[1], C[r * 3 + 2], q[2]);
                double e = gauss_d2 * std::exp(-gauss_d2 * fd3(q[0], v[0], q[1], v[1], q[2], v[2]) / 2);
                if (e > 1 || e < 0 || e != e) continue;
                e *= gauss_d1;
                const double ev3[3] = {e * v[0], e * v[1], e * v[2]};
                M1[0] = std::fma(e, C[0], M1[0]); M1[1] = std::fma(e, C[1], M1[1]); M1[2] = std::fma(e, C[2], M1[2]);
                M1[3] = std::fma(e, C[4], M1[3]); M1[4] = std::fma(e, C[5], M1[4]); M1[5] = std::fma(e, C[8], M1[5]);
                M2[0] = std::fma(ev3[0], v[0], M2[0]); M2[1] = std::fma(ev3[0], v[1], M2[1]); M2[2] = std::fma(ev3[0], v[2], M2[2]);
                M2[3] = std::fma(ev3[1], v[1], M2[3]); M2[4] = std::fma(ev3[1], v[2], M2[4]); M2[5] = std::fma(ev3[2], v[2], M2[5]);
                w[0] += ev3[0]; w[1] += ev3[1]; w[2] += ev3[2];
            }
            double a[6];
            for (int k = 0; k < 6; ++k) a[k] = std::fma(-gauss_d2, M2[k], M1[k]);
            const double A[3][3] = {{a[0], a[1], a[2]}, {a[1], a[3], a[4]}, {a[2], a[4], a[5]}};
            double xj[8], xh[15];
            for (int r = 0; r < 8; ++r) xj[r] = fd3(x[0], j_ang_d[r][0], x[1], j_ang_d[r][1], x[2], j_ang_d[r][2]);
            for (int r = 0; r < 15; ++r) xh[r] = fd3(x[0], h_ang_d[r][0], x[1], h_ang_d[r][1], x[2], h_ang_d[r][2]);
            const double Jr[3][3] = {{0.0, xj[2], xj[5]}, {xj[0], xj[3], xj[6]}, {xj[1], xj[4], xj[7]}};
            double AJ[3][3];
            for (int r = 0; r < 3; ++r) {
                AJ[r][0] = fd3z(A[r][1], Jr[1][0], A[r][2], Jr[2][0]);
                AJ[r][1] = fd3(A[r][0], Jr[0][1], A[r][1], Jr[1][1], A[r][2], Jr[2][1]);
                AJ[r][2] = fd3(A[r][0], Jr[0][2], A[r][1], Jr[1][2], A[r][2], Jr[2][2]);
            }
            H[0 * 6 + 0] += A[0][0]; H[0 * 6 + 1] += A[0][1]; H[0 * 6 + 2] += A[0][2]; H[1 * 6 + 1] += A[1][1]; H[1 * 6 + 2] += A[1][2]; H[2 * 6 + 2] += A[2][2];
            for (int i = 0; i < 3; ++i)
                for (int c = 0; c < 3; ++c) H[i * 6 + 3 + c] += AJ[i][c];
            const double PH[6][3] = {{0, xh[0], xh[1]}, {0, xh[2], xh[3]}, {0, xh[4], xh[5]}, {xh[6], xh[7], xh[8]}, {xh[9], xh[10], xh[11]}, {xh[12], xh[13], xh[14]}};
            for (int i = 0; i < 3; ++i)
                for (int j = i; j < 3; ++j) {
                    const int ph = (i == 0) ? j : (i == 1 ? j + 2 : 5);
                    const double jaj = (i == 0) ? fd3z(Jr[1][0], AJ[1][j], Jr[2][0], AJ[2][j]) : fd3(Jr[0][i], AJ[0][j], Jr[1][i], AJ[1][j], Jr[2][i], AJ[2][j]);
                    const double wph = (ph < 3) ? fd3z(w[1], PH[ph][1], w[2], PH[ph][2]) : fd3(w[0], PH[ph][0], w[1], PH[ph][1], w[2], PH[ph][2]);
                    H[(3 + i) * 6 + 3 + j] += jaj + wph;
                }
        }
        // the kernel mirrors the upper triangle of every lane before the tree
        for (int l = 0; l < 256; ++l) {
            double* H = &acc[static_cast<size_t>(l) * 48] + 7;
            for (int i = 0; i < 6; ++i)
                for (int j = 0; j < i; ++j) H[i * 6 + j] = H[j * 6 + i];
        }
        gpu_tree_reduce(acc, &partials[item * 48]);
    }
    double r[48];
    gpu_slice_reduce(partials, nblk, r);
    neighbours_sum += n > 0 ? static_cast<double>(nb_total) / n : 0.0;
    for (int k = 0; k < 36; ++k) hess[k] = r[7 + k];
}

double Ndt::evaluate(const float T[16], const double p[6], int mode, double grad[6], double hess[36])
{
    init_gauss();
    transform_cloud(T);
    if (mode == 2) {
        angle_derivatives(p, true);
        compute_hessian(hess, p);
        for (int k = 0; k < 6; ++k) grad[k] = 0;
        return 0;
    }
    return compute_derivatives(grad, hess, p, mode == 0);
}

// ---- More-Thuente helpers (ndt_omp_impl.hpp) -----------------------------------------------------------
static inline double psi_mt(double a, double f_a, double f_0, double g_0, double mu) { return f_a - f_0 - mu * g_0 * a; }
static inline double dpsi_mt(double g_a, double g_0, double mu) { return g_a - mu * g_0; }

bool mt_update_interval(double& a_l, double& f_l, double& g_l, double& a_u, double& f_u, double& g_u, double a_t, double f_t, double g_t)
{
    if (f_t > f_l) { a_u = a_t; f_u = f_t; g_u = g_t; return false; }
    else if (g_t * (a_l - a_t) > 0) { a_l = a_t; f_l = f_t; g_l = g_t; return false; }
    else if (g_t * (a_l - a_t) < 0) { a_u = a_l; f_u = f_l; g_u = g_l; a_l = a_t; f_l = f_t; g_l = g_t; return false; }
    return true;
}

double mt_trial_value_selection(double a_l, double f_l, double g_l, double a_u, double f_u, double g_u, double a_t, double f_t, double g_t)
{
    if (f_t > f_l) {  // case 1
        double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l;
        double w = std::sqrt(z * z - g_t * g_l);
        double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
        double a_q = a_l - 0.5 * (a_l - a_t) * g_l / (g_l - (f_l - f_t) / (a_l - a_t));
        if (std::fabs(a_c - a_l) < std::fabs(a_q - a_l)) return a_c;
        return 0.5 * (a_q + a_c);
    } else if (g_t * g_l < 0) {  // case 2
        double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l;
        double w = std::sqrt(z * z - g_t * g_l);
        double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
        double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
        if (std::fabs(a_c - a_t) >= std::fabs(a_s - a_t)) return a_c;
        return a_s;
    } else if (std::fabs(g_t) <= std::fabs(g_l)) {  // case 3
        double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l;
        double w = std::sqrt(z * z - g_t * g_l);
        double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
        double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
        double a_t_next = (std::fabs(a_c - a_t) < std::fabs(a_s - a_t)) ? a_c : a_s;
        if (a_t > a_l) return std::min(a_t + 0.66 * (a_u - a_t), a_t_next);
        return std::max(a_t + 0.66 * (a_u - a_t), a_t_next);
    } else {  // case 4
        double z = 3 * (f_t - f_u) / (a_t - a_u) - g_t - g_u;
        double w = std::sqrt(z * z - g_t * g_u);
        return a_u + (a_t - a_u) * (w - g_u - z) / (g_t - g_u + 2 * w);
    }
}

double Ndt::step_length_mt(const double x[6], double step_dir[6], double step_init, double step_max, double step_min, double& score, double grad[6],
                           double hess[36])
{
    double phi_0 = -score;
    double d_phi_0 = 0;
    for (int k = 0; k < 6; ++k) d_phi_0 += grad[k] * step_dir[k];
    d_phi_0 = -d_phi_0;
    double x_t[6];
    if (d_phi_0 >= 0) {
        if (d_phi_0 == 0) return 0;
        d_phi_0 *= -1;
        for (int k = 0; k < 6; ++k) step_dir[k] *= -1;
    }
    const int    max_step_iterations = quirks::kMtMaxStepIterations;
    int          step_iterations = 0;
    const double mu = quirks::kMtMu, nu = quirks::kMtNu;
    double a_l = 0, a_u = 0;
    double f_l = psi_mt(a_l, phi_0, phi_0, d_phi_0, mu);
    double g_l = dpsi_mt(d_phi_0, d_phi_0, mu);
    double f_u = psi_mt(a_u, phi_0, phi_0, d_phi_0, mu);
    double g_u = dpsi_mt(d_phi_0, d_phi_0, mu);
    bool   interval_converged = (step_max - step_min) < 0, open_interval = true;
    double a_t = step_init;
    a_t = std::min(a_t, step_max);
    a_t = std::max(a_t, step_min);
    for (int k = 0; k < 6; ++k) x_t[k] = x[k] + step_dir[k] * a_t;
    pose_to_matrix_f(x_t, final_);
    transform_cloud(final_);
    score = compute_derivatives(grad, hess, x_t, true);
    double phi_t = -score;
    double d_phi_t = 0;
    for (int k = 0; k < 6; ++k) d_phi_t += grad[k] * step_dir[k];
    d_phi_t = -d_phi_t;
    double psi_t = psi_mt(a_t, phi_t, phi_0, d_phi_0, mu);
    double d_psi_t = dpsi_mt(d_phi_t, d_phi_0, mu);
    while (!interval_converged && step_iterations < max_step_iterations && !(psi_t <= 0 && d_phi_t <= -nu * d_phi_0)) {
        if (open_interval) a_t = mt_trial_value_selection(a_l, f_l, g_l, a_u, f_u, g_u, a_t, psi_t, d_psi_t);
        else               a_t = mt_trial_value_selection(a_l, f_l, g_l, a_u, f_u, g_u, a_t, phi_t, d_phi_t);
        a_t = std::min(a_t, step_max);
        a_t = std::max(a_t, step_min);
        for (int k = 0; k < 6; ++k) x_t[k] = x[k] + step_dir[k] * a_t;
        pose_to_matrix_f(x_t, final_);
        transform_cloud(final_);
        score = compute_derivatives(grad, hess, x_t, false);
        phi_t = -score;
        d_phi_t = 0;
        for (int k = 0; k < 6; ++k) d_phi_t += grad[k] * step_dir[k];
        d_phi_t = -d_phi_t;
        psi_t = psi_mt(a_t, phi_t, phi_0, d_phi_0, mu);
        d_psi_t = dpsi_mt(d_phi_t, d_phi_0, mu);
        if (open_interval && (psi_t <= 0 && d_psi_t >= 0)) {
            open_interval = false;
            f_l = f_l + phi_0 - mu * d_phi_0 * a_l;
            g_l = g_l + mu * d_phi_0;
            f_u = f_u + phi_0 - mu * d_phi_0 * a_u;
            g_u = g_u + mu * d_phi_0;
        }
        if (open_interval) interval_converged = mt_update_interval(a_l, f_l, g_l, a_u, f_u, g_u, a_t, psi_t, d_psi_t);
        else               interval_converged = mt_update_interval(a_l, f_l, g_l, a_u, f_u, g_u, a_t, phi_t, d_phi_t);
        step_iterations++;
    }
    if (step_iterations) compute_hessian(hess, x_t);
    return a_t;
}

void Ndt::align(const float guess[16], float* aligned)
{
    const int n = static_cast<int>(source.size() / 4);
    // pcl::Registration::align
    converged = false;
    nr_iterations = 0;
    n_evals = 0;
    neighbours_sum = 0;
    mat4f_identity(final_);
    float transformation[16], previous[16];
    mat4f_identity(transformation); mat4f_identity(previous);
    for (int k = 0; k < 36; ++k) hessian[k] = 0;
    trans_probability = 0;
    auto write_output = [&]() {
        if (!aligned) return;
        // output == final_transformation_ * input (intensity carried over)
        for (int i = 0; i < n; ++i) {
            const float* p = &source[4 * static_cast<size_t>(i)];
            transform_point_f(final_, p[0], p[1], p[2], aligned[4 * i], aligned[4 * i + 1], aligned[4 * i + 2]);
            aligned[4 * i + 3] = p[3];
        }
    };
    if (n == 0 || target_status != 0) { write_output(); return; }

    // computeTransformation
    init_gauss();
    if (!mat4f_is_identity(guess)) std::memcpy(final_, guess, sizeof(float) * 16);
    transform_cloud(final_);
    float eul[3];
    euler_xyz_f(final_, eul);
    double p[6] = {final_[3], final_[7], final_[11], eul[0], eul[1], eul[2]};
    double delta_p[6], grad[6];
    double score = compute_derivatives(grad, hessian, p, true);
    while (!converged) {
        std::memcpy(previous, transformation, sizeof(previous));
        JacobiSvd6 sv;
        sv.compute(hessian);
        double neg_g[6];
        for (int k = 0; k < 6; ++k) neg_g[k] = -grad[k];
        sv.solve(neg_g, delta_p);
        double delta_p_norm = 0;
        for (int k = 0; k < 6; ++k) delta_p_norm += delta_p[k] * delta_p[k];
        delta_p_norm = std::sqrt(delta_p_norm);
        if (delta_p_norm == 0 || delta_p_norm != delta_p_norm) {
            trans_probability = score / static_cast<double>(n);
            converged = delta_p_norm == delta_p_norm;
            write_output();
            return;
        }
        for (int k = 0; k < 6; ++k) delta_p[k] /= delta_p_norm;
        delta_p_norm = step_length_mt(p, delta_p, delta_p_norm, step_size, trans_eps / 2, score, grad, hessian);
        for (int k = 0; k < 6; ++k) delta_p[k] *= delta_p_norm;
        pose_to_matrix_f(delta_p, transformation);
        for (int k = 0; k < 6; ++k) p[k] = p[k] + delta_p[k];
        if (nr_iterations > max_iterations || (nr_iterations && (std::fabs(delta_p_norm) < trans_eps))) converged = true;
        nr_iterations++;
    }
    trans_probability = score / static_cast<double>(n);
    write_output();
}

double Ndt::fitness(double max_range) const
{
    const int n = static_cast<int>(source.size() / 4);
    const int nt = static_cast<int>(target.size() / 4);
    if (n == 0 || nt == 0) return std::numeric_limits<double>::max();
    NnGrid grid;
    grid.build(target.data(), nt, 1.0f);
    double sum = 0;
    int    nr = 0;
    for (int i = 0; i < n; ++i) {
        const float* p = &source[4 * static_cast<size_t>(i)];
        float x, y, z, d;
        transform_point_f(final_, p[0], p[1], p[2], x, y, z);
        if (grid.nearest(x, y, z, d) < 0) continue;
        if (static_cast<double>(d) <= max_range) { sum += d; nr++; }
    }
    return nr > 0 ? sum / nr : std::numeric_limits<double>::max();
}

}  // namespace orc
