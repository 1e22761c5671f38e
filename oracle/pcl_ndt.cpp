// oracle/pcl_ndt.cpp — TEST INFRASTRUCTURE (CPU oracle). Not part of the shipped product path.
//
// Restates (PARITY UNPINNED, see pcl_ndt.h / quirks.h) pcl::NormalDistributionsTransform (PCL 1.12 registration/impl/ndt.hpp):
//   computeTransformation, computeDerivatives, computeAngleDerivatives, computePointDerivatives, updateDerivatives, computeHessian,
//   updateHessian, computeStepLengthMT (+ trialValueSelectionMT / updateIntervalMT, shared with ndt.cpp), convertTransform,
// and pcl::Registration::align / getFitnessScore around it — what `registration_method: "NDT"` (and every unknown name) runs in the
// reference: /root/reference/src/mrg_slam/registrations.cpp:115-129.
#include "pcl_ndt.h"

#include <omp.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>

#include "linalg.h"
#include "nn.h"
#include "quirks.h"

namespace orc {

void PclNdt::init_gauss()
{
    // "Initializes the gaussian fitting parameters (eq. 6.8) [Magnusson 2009]" — at the top of every computeTransformation
    const double gauss_c1 = 10 * (1 - outlier_ratio);
    const double gauss_c2 = outlier_ratio / std::pow(static_cast<double>(resolution), 3);
    const double gauss_d3 = -std::log(gauss_c2);
    gauss_d1 = -std::log(gauss_c1 + gauss_c2) - gauss_d3;
    gauss_d2 = -2 * std::log((-std::log(gauss_c1 * std::exp(-0.5) + gauss_c2) - gauss_d3) / gauss_d1);
}

int PclNdt::set_target(const float* xyzi, int n)
{
    // setInputTarget -> init(): target_cells_.setLeafSize(resolution_ x3); setInputCloud(target_); filter(true)
    target.assign(xyzi, xyzi + static_cast<size_t>(n) * 4);
    cells.negative_eigen_tolerance = quirks::kPclVgcNegativeEigenTolerance;  // pcl::VoxelGridCovariance of PCL 1.12, not ndt_omp's fork of it
    target_status = cells.build(target.data(), n, resolution);
    return target_status;
}

void PclNdt::set_source(const float* xyzi, int n) { source.assign(xyzi, xyzi + static_cast<size_t>(n) * 4); }

void PclNdt::angle_derivatives(const double p[6])
{
    // computeAngleDerivatives(transform, compute_hessian = true): f64 tables angular_jacobian_ (8 x 4) / angular_hessian_ (15 x 4); the fourth
    // column only ever meets the 0 of Vector4d(x, y, z, 0)
    double cx, cy, cz, sx, sy, sz;
    auto cs = [](double a, double& c, double& s) {
        if (std::fabs(a) < quirks::kNdtSmallAngle) { c = 1.0; s = 0.0; } else { c = std::cos(a); s = std::sin(a); }
    };
    cs(p[3], cx, sx); cs(p[4], cy, sy); cs(p[5], cz, sz);
    const double j[8][3] = {
        {(-sx * sz + cx * sy * cz), (-sx * cz - cx * sy * sz), (-cx * cy)},  // a
        {(cx * sz + sx * sy * cz), (cx * cz - sx * sy * sz), (-sx * cy)},    // b
        {(-sy * cz), sy * sz, cy},                                           // c
        {sx * cy * cz, (-sx * cy * sz), sx * sy},                            // d
        {(-cx * cy * cz), cx * cy * sz, (-cx * sy)},                         // e
        {(-cy * sz), (-cy * cz), 0},                                         // f
        {(cx * cz - sx * sy * sz), (-cx * sz - sx * sy * cz), 0},            // g
        {(sx * cz + cx * sy * sz), (cx * sy * cz - sx * sz), 0}};            // h
    const double h[15][3] = {
        {(-cx * sz - sx * sy * cz), (-cx * cz + sx * sy * sz), sx * cy},     // a2
        {(-sx * sz + cx * sy * cz), (-cx * sy * sz - sx * cz), (-cx * cy)},  // a3
        {(cx * cy * cz), (-cx * cy * sz), (cx * sy)},                        // b2
        {(sx * cy * cz), (-sx * cy * sz), (sx * sy)},                        // b3
        {(-sx * cz - cx * sy * sz), (sx * sz - cx * sy * cz), 0},            // c2
        {(cx * cz - sx * sy * sz), (-sx * sy * cz - cx * sz), 0},            // c3
        {(-cy * cz), (cy * sz), quirks::kNdtHAngD1ZSign * (sy)},             // d1 (upstream's sign, quirks.h)
        {(-sx * sy * cz), (sx * sy * sz), (sx * cy)},                        // d2
        {(cx * sy * cz), (-cx * sy * sz), (-cx * cy)},                       // d3
        {(sy * sz), (sy * cz), 0},                                           // e1
        {(-sx * cy * sz), (-sx * cy * cz), 0},                               // e2
        {(cx * cy * sz), (cx * cy * cz), 0},                                 // e3
        {(-cy * cz), (cy * sz), 0},                                          // f1
        {(-cx * sz - sx * sy * cz), (-cx * cz + sx * sy * sz), 0},           // f2
        {(-sx * sz + cx * sy * cz), (-cx * sy * sz - sx * cz), 0}};          // f3
    std::memcpy(j_ang, j, sizeof(j));
    std::memcpy(h_ang, h, sizeof(h));
}

void PclNdt::transform_cloud(const float T[16])
{
    // transformPointCloud(*input_, trans_cloud, final_transformation_): the float 4x4 on float points
    const int n = static_cast<int>(source.size() / 4);
    trans_.resize(static_cast<size_t>(n) * 3);
    for (int i = 0; i < n; ++i) {
        const float* p = &source[4 * static_cast<size_t>(i)];
        transform_point_f(T, p[0], p[1], p[2], trans_[3 * i], trans_[3 * i + 1], trans_[3 * i + 2]);
    }
}

namespace {
inline double dot3(const double a[3], const double b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
inline void   matvec3(const double C[9], const double v[3], double out[3])
{
    for (int r = 0; r < 3; ++r) out[r] = C[r * 3 + 0] * v[0] + C[r * 3 + 1] * v[1] + C[r * 3 + 2] * v[2];
}
// computePointDerivatives(x): point_jacobian_ (3 x 6) and the nine 3-vectors of point_hessian_ (18 x 6)
struct PointDerivs {
    double J[6][3];       // column i of point_jacobian_
    const double* PH[6][6];
    double a[3], b[3], c[3], d[3], e[3], f[3], zero[3];
};
inline void point_derivatives(const double x[3], const double j_ang[8][3], const double h_ang[15][3], PointDerivs& P)
{
    double xj[8], xh[15];
    for (int r = 0; r < 8; ++r) xj[r] = dot3(j_ang[r], x);
    for (int r = 0; r < 15; ++r) xh[r] = dot3(h_ang[r], x);
    for (int i = 0; i < 6; ++i) for (int r = 0; r < 3; ++r) P.J[i][r] = (i == r) ? 1.0 : 0.0;
    P.J[3][1] = xj[0]; P.J[3][2] = xj[1];
    P.J[4][0] = xj[2]; P.J[4][1] = xj[3]; P.J[4][2] = xj[4];
    P.J[5][0] = xj[5]; P.J[5][1] = xj[6]; P.J[5][2] = xj[7];
    P.a[0] = 0; P.a[1] = xh[0]; P.a[2] = xh[1];
    P.b[0] = 0; P.b[1] = xh[2]; P.b[2] = xh[3];
    P.c[0] = 0; P.c[1] = xh[4]; P.c[2] = xh[5];
    for (int r = 0; r < 3; ++r) { P.d[r] = xh[6 + r]; P.e[r] = xh[9 + r]; P.f[r] = xh[12 + r]; P.zero[r] = 0; }
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) P.PH[i][j] = P.zero;
    // point_hessian_.block<3, 1>(9, 3) = a; (12, 3) = b; (15, 3) = c; (9, 4) = b; (12, 4) = d; (15, 4) = e; (9, 5) = c; (12, 5) = e; (15, 5) = f
    P.PH[3][3] = P.a; P.PH[4][3] = P.b; P.PH[5][3] = P.c;
    P.PH[3][4] = P.b; P.PH[4][4] = P.d; P.PH[5][4] = P.e;
    P.PH[3][5] = P.c; P.PH[4][5] = P.e; P.PH[5][5] = P.f;
}
}  // namespace

double PclNdt::compute_derivatives(double grad[6], double hess[36], const double p[6], bool compute_hessian)
{
    angle_derivatives(p);
    ++n_evals;
    if (gpu_order) return derivatives_point_order(grad, hess, true, compute_hessian);
    const int n = static_cast<int>(source.size() / 4);
    for (int k = 0; k < 6; ++k) grad[k] = 0;
    for (int k = 0; k < 36; ++k) hess[k] = 0;
    double    score = 0;
    long long nb_total = 0;
    for (int idx = 0; idx < n; ++idx) {
        const float xt[3] = {trans_[3 * idx], trans_[3 * idx + 1], trans_[3 * idx + 2]};
        int nb[27];
        const int cnt = cells.radius_neighbours(xt[0], xt[1], xt[2], nb);
        nb_total += cnt;
        if (!cnt) continue;
        const float* xp = &source[4 * static_cast<size_t>(idx)];
        const double x[3] = {xp[0], xp[1], xp[2]};
        PointDerivs P;
        point_derivatives(x, j_ang, h_ang, P);
        for (int k = 0; k < cnt; ++k) {
            const NdtLeaf& cell = cells.leaves[nb[k]];
            const double   q[3] = {static_cast<double>(xt[0]) - cell.mean[0], static_cast<double>(xt[1]) - cell.mean[1], static_cast<double>(xt[2]) - cell.mean[2]};
            const double*  C = cell.icov;
            // updateDerivatives
            double Cq[3];
            matvec3(C, q, Cq);
            double e_x_cov_x = std::exp(-gauss_d2 * dot3(q, Cq) / 2);
            const double score_inc = -gauss_d1 * e_x_cov_x;
            e_x_cov_x = gauss_d2 * e_x_cov_x;
            if (e_x_cov_x > 1 || e_x_cov_x < 0 || e_x_cov_x != e_x_cov_x) continue;  // "return 0": the pair adds nothing, not even its score
            e_x_cov_x *= gauss_d1;
            for (int i = 0; i < 6; ++i) {
                double cov_dxd_pi[3];
                matvec3(C, P.J[i], cov_dxd_pi);
                grad[i] += dot3(q, cov_dxd_pi) * e_x_cov_x;
                if (!compute_hessian) continue;
                for (int j = 0; j < 6; ++j) {
                    double CJj[3], CH[3];
                    matvec3(C, P.J[j], CJj);
                    matvec3(C, P.PH[i][j], CH);
                    hess[i * 6 + j] += e_x_cov_x * (-gauss_d2 * dot3(q, cov_dxd_pi) * dot3(q, CJj) + dot3(q, CH) + dot3(P.J[j], cov_dxd_pi));
                }
            }
            score += score_inc;
        }
    }
    neighbours_sum += n > 0 ? static_cast<double>(nb_total) / n : 0.0;
    return score;
}

void PclNdt::compute_hessian(double hess[36])
{
    // computeHessian(hessian, trans_cloud): the tables of the last computeDerivatives call (same pose) stay in place
    ++n_evals;
    if (gpu_order) { double g[6]; derivatives_point_order(g, hess, false, true); return; }
    const int n = static_cast<int>(source.size() / 4);
    for (int k = 0; k < 36; ++k) hess[k] = 0;
    long long nb_total = 0;
    for (int idx = 0; idx < n; ++idx) {
        const float xt[3] = {trans_[3 * idx], trans_[3 * idx + 1], trans_[3 * idx + 2]};
        int nb[27];
        const int cnt = cells.radius_neighbours(xt[0], xt[1], xt[2], nb);
        nb_total += cnt;
        if (!cnt) continue;
        const float* xp = &source[4 * static_cast<size_t>(idx)];
        const double x[3] = {xp[0], xp[1], xp[2]};
        PointDerivs P;
        point_derivatives(x, j_ang, h_ang, P);
        for (int k = 0; k < cnt; ++k) {
            const NdtLeaf& cell = cells.leaves[nb[k]];
            const double   q[3] = {static_cast<double>(xt[0]) - cell.mean[0], static_cast<double>(xt[1]) - cell.mean[1], static_cast<double>(xt[2]) - cell.mean[2]};
            const double*  C = cell.icov;
            // updateHessian
            double Cq[3];
            matvec3(C, q, Cq);
            double e_x_cov_x = gauss_d2 * std::exp(-gauss_d2 * dot3(q, Cq) / 2);
            if (e_x_cov_x > 1 || e_x_cov_x < 0 || e_x_cov_x != e_x_cov_x) continue;
            e_x_cov_x *= gauss_d1;
            for (int i = 0; i < 6; ++i) {
                double cov_dxd_pi[3];
                matvec3(C, P.J[i], cov_dxd_pi);
                for (int j = 0; j < 6; ++j) {
                    double CJj[3], CH[3];
                    matvec3(C, P.J[j], CJj);
                    matvec3(C, P.PH[i][j], CH);
                    hess[i * 6 + j] += e_x_cov_x * (-gauss_d2 * dot3(q, cov_dxd_pi) * dot3(q, CJj) + dot3(q, CH) + dot3(P.J[j], cov_dxd_pi));
                }
            }
        }
    }
    neighbours_sum += n > 0 ? static_cast<double>(nb_total) / n : 0.0;
}

// ---- diagnostic: the HIP kernel's association and ORDER on the CPU (mrg_slam_amd/csrc/ndt_derivatives.hip, the f64 items) -----------------------
// J and the second-derivative vectors belong to the point, so with v = C q and e the pair's weight the sums over a point's voxels factor into
//     score  += sum -d1 exp(..)          gradient_i += (sum e v) . J_i
//     H_ij   += J_i^T [ sum e C - d2 sum e v v^T ] J_j + (sum e v) . PH_ij
// (exact algebra; in f64 the different association moves results by ~1e-16 relative).  The kernel walks the 27 cells in probe order, one lane per
// point (lane = point index mod 256), items of gpu_order tiles of 256 points, and reduces lanes / waves / items in the tree ndt.cpp restates
// (gpu_tree_reduce / gpu_slice_reduce).  With this mode a GPU evaluation is reproduced to the last bits (up to the two exp implementations).
double PclNdt::derivatives_point_order(double grad[6], double hess[36], bool with_score_grad, bool with_hessian)
{
    const int    n = static_cast<int>(source.size() / 4);
    const int    per_item = 256 * gpu_order;
    const size_t nblk = static_cast<size_t>((n + per_item - 1) / per_item);
    std::vector<double> partials(std::max<size_t>(nblk, 1) * 48, 0.0);
    long long nb_total = 0;
    auto fd3 = [](double a0, double b0, double a1, double b1, double a2, double b2) { return std::fma(a2, b2, std::fma(a1, b1, a0 * b0)); };
    auto fd3z = [](double a1, double b1, double a2, double b2) { return std::fma(a2, b2, a1 * b1); };
#pragma omp parallel for num_threads(std::max(1, num_threads)) schedule(dynamic, 1) reduction(+ : nb_total)
    for (size_t item = 0; item < nblk; ++item) {
        std::vector<double> acc(256 * 48, 0.0);
        const int base = static_cast<int>(item) * per_item, last = std::min(n, base + per_item);
        for (int idx = base; idx < last; ++idx) {
            double* A48 = &acc[static_cast<size_t>((idx - base) % 256) * 48];
            double* H = A48 + 7;
            const float xt[3] = {trans_[3 * idx], trans_[3 * idx + 1], trans_[3 * idx + 2]};
            int found[27];
            const int cnt = cells.radius_neighbours(xt[0], xt[1], xt[2], found);
            nb_total += cnt;
            if (!cnt) continue;
            // probe order (ox, oy, oz ascending) = ascending (cell z, y, x)?  no: the kernel's probe n = (ox + 1) * 9 + (oy + 1) * 3 + (oz + 1)
            std::pair<int, int> byprobe[27];
            const int ijk[3] = {static_cast<int>(std::floor(xt[0] / cells.leaf_size)), static_cast<int>(std::floor(xt[1] / cells.leaf_size)), static_cast<int>(std::floor(xt[2] / cells.leaf_size))};
            for (int k = 0; k < cnt; ++k) {
                int key = cells.leaves[found[k]].key;
                const int cz = key / cells.divb_mul[2]; key -= cz * cells.divb_mul[2];
                const int cy = key / cells.divb_mul[1]; key -= cy * cells.divb_mul[1];
                const int cx = key;
                byprobe[k] = std::make_pair((cx + cells.min_b[0] - ijk[0] + 1) * 9 + (cy + cells.min_b[1] - ijk[1] + 1) * 3 + (cz + cells.min_b[2] - ijk[2] + 1), found[k]);
            }
            std::sort(byprobe, byprobe + cnt);
            const float* xp = &source[4 * static_cast<size_t>(idx)];
            const double x[3] = {xp[0], xp[1], xp[2]};
            double M1[6] = {0, 0, 0, 0, 0, 0}, M2[6] = {0, 0, 0, 0, 0, 0}, w[3] = {0, 0, 0}, sc = 0;
            for (int k = 0; k < cnt; ++k) {
                const NdtLeaf& cell = cells.leaves[byprobe[k].second];
                const double*  C = cell.icov;
                const double   q[3] = {static_cast<double>(xt[0]) - cell.mean[0], static_cast<double>(xt[1]) - cell.mean[1], static_cast<double>(xt[2]) - cell.mean[2]};
                double v[3];
                for (int r = 0; r < 3; ++r) v[r] = fd3(C[r * 3 + 0], q[0], C[r * 3 + 1], q[1], C[r * 3 + 2], q[2]);
                const double e_raw = std::exp(-gauss_d2 * fd3(q[0], v[0], q[1], v[1], q[2], v[2]) / 2);
                double e = gauss_d2 * e_raw;
                if (e > 1 || e < 0 || e != e) continue;
                e *= gauss_d1;
                sc += -gauss_d1 * e_raw;
                const double ev3[3] = {e * v[0], e * v[1], e * v[2]};
                w[0] += ev3[0]; w[1] += ev3[1]; w[2] += ev3[2];
                if (!with_hessian) continue;
                M1[0] = std::fma(e, C[0], M1[0]); M1[1] = std::fma(e, C[1], M1[1]); M1[2] = std::fma(e, C[2], M1[2]);
                M1[3] = std::fma(e, C[4], M1[3]); M1[4] = std::fma(e, C[5], M1[4]); M1[5] = std::fma(e, C[8], M1[5]);
                M2[0] = std::fma(ev3[0], v[0], M2[0]); M2[1] = std::fma(ev3[0], v[1], M2[1]); M2[2] = std::fma(ev3[0], v[2], M2[2]);
                M2[3] = std::fma(ev3[1], v[1], M2[3]); M2[4] = std::fma(ev3[1], v[2], M2[4]); M2[5] = std::fma(ev3[2], v[2], M2[5]);
            }
            double xj[8], xh[15];
            for (int r = 0; r < 8; ++r) xj[r] = fd3(x[0], j_ang[r][0], x[1], j_ang[r][1], x[2], j_ang[r][2]);
            const double Jr[3][3] = {{0.0, xj[2], xj[5]}, {xj[0], xj[3], xj[6]}, {xj[1], xj[4], xj[7]}};
            if (with_score_grad) {
                A48[0] += sc;
                A48[1] += w[0]; A48[2] += w[1]; A48[3] += w[2];
                A48[4] += fd3z(w[1], Jr[1][0], w[2], Jr[2][0]);
                A48[5] += fd3(w[0], Jr[0][1], w[1], Jr[1][1], w[2], Jr[2][1]);
                A48[6] += fd3(w[0], Jr[0][2], w[1], Jr[1][2], w[2], Jr[2][2]);
            }
            if (!with_hessian) continue;
            for (int r = 0; r < 15; ++r) xh[r] = fd3(x[0], h_ang[r][0], x[1], h_ang[r][1], x[2], h_ang[r][2]);
            double a[6];
            for (int k = 0; k < 6; ++k) a[k] = std::fma(-gauss_d2, M2[k], M1[k]);
            const double A[3][3] = {{a[0], a[1], a[2]}, {a[1], a[3], a[4]}, {a[2], a[4], a[5]}};
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
        for (int l = 0; l < 256; ++l) {  // the kernel mirrors the upper triangle of every lane before the tree
            double* H = &acc[static_cast<size_t>(l) * 48] + 7;
            for (int i = 0; i < 6; ++i)
                for (int j = 0; j < i; ++j) H[i * 6 + j] = H[j * 6 + i];
        }
        gpu_tree_reduce(acc, &partials[item * 48]);
    }
    double r[48];
    gpu_slice_reduce(partials, nblk, r);
    neighbours_sum += n > 0 ? static_cast<double>(nb_total) / n : 0.0;
    for (int k = 0; k < 6; ++k) grad[k] = with_score_grad ? r[1 + k] : 0.0;
    for (int k = 0; k < 36; ++k) hess[k] = with_hessian ? r[7 + k] : 0.0;
    return with_score_grad ? r[0] : 0.0;
}

double PclNdt::evaluate(const float T[16], const double p[6], int mode, double grad[6], double hess[36])
{
    init_gauss();
    transform_cloud(T);
    if (mode == 2) {
        angle_derivatives(p);
        compute_hessian(hess);
        for (int k = 0; k < 6; ++k) grad[k] = 0;
        return 0;
    }
    return compute_derivatives(grad, hess, p, mode == 0);
}

double PclNdt::step_length_mt(const double x[6], double step_dir[6], double step_init, double step_max, double step_min, double& score, double grad[6],
                              double hess[36])
{
    // computeStepLengthMT — the text pclomp forked (ndt.cpp Ndt::step_length_mt), here over the f64 derivatives
    const double phi_0 = -score;
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
    auto psi = [](double a, double f_a, double f_0, double g_0, double m) { return f_a - f_0 - m * g_0 * a; };
    auto dpsi = [](double g_a, double g_0, double m) { return g_a - m * g_0; };
    double a_l = 0, a_u = 0;
    double f_l = psi(a_l, phi_0, phi_0, d_phi_0, mu), g_l = dpsi(d_phi_0, d_phi_0, mu);
    double f_u = psi(a_u, phi_0, phi_0, d_phi_0, mu), g_u = dpsi(d_phi_0, d_phi_0, mu);
    bool   interval_converged = (step_max - step_min) < 0, open_interval = true;
    double a_t = step_init;
    a_t = std::min(a_t, step_max);
    a_t = std::max(a_t, step_min);
    for (int k = 0; k < 6; ++k) x_t[k] = x[k] + step_dir[k] * a_t;
    pose_to_matrix_f(x_t, final_);  // convertTransform(x_t, final_transformation_)
    transform_cloud(final_);
    score = compute_derivatives(grad, hess, x_t, true);
    double phi_t = -score, d_phi_t = 0;
    for (int k = 0; k < 6; ++k) d_phi_t += grad[k] * step_dir[k];
    d_phi_t = -d_phi_t;
    double psi_t = psi(a_t, phi_t, phi_0, d_phi_0, mu), d_psi_t = dpsi(d_phi_t, d_phi_0, mu);
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
        psi_t = psi(a_t, phi_t, phi_0, d_phi_0, mu);
        d_psi_t = dpsi(d_phi_t, d_phi_0, mu);
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
    if (step_iterations) compute_hessian(hess);
    return a_t;
}

void PclNdt::align(const float guess[16], float* aligned)
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
    trans_likelihood = 0;
    auto write_output = [&]() {
        if (!aligned) return;
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
    euler_xyz_f(final_, eul);  // eig_transformation.rotation().eulerAngles(0, 1, 2), float
    double p[6] = {final_[3], final_[7], final_[11], eul[0], eul[1], eul[2]};
    double delta[6], grad[6];
    double score = compute_derivatives(grad, hessian, p, true);
    while (!converged) {
        std::memcpy(previous, transformation, sizeof(previous));
        JacobiSvd6 sv;
        sv.compute(hessian);
        double neg_g[6];
        for (int k = 0; k < 6; ++k) neg_g[k] = -grad[k];
        sv.solve(neg_g, delta);
        double delta_norm = 0;
        for (int k = 0; k < 6; ++k) delta_norm += delta[k] * delta[k];
        delta_norm = std::sqrt(delta_norm);
        if (delta_norm == 0 || delta_norm != delta_norm) {
            trans_likelihood = score / static_cast<double>(n);
            converged = quirks::kPclNdtZeroStepConverges ? (delta_norm == 0) : (delta_norm == delta_norm);
            write_output();
            return;
        }
        for (int k = 0; k < 6; ++k) delta[k] /= delta_norm;
        delta_norm = step_length_mt(p, delta, delta_norm, step_size, trans_eps / 2, score, grad, hessian);
        for (int k = 0; k < 6; ++k) delta[k] *= delta_norm;
        pose_to_matrix_f(delta, transformation);  // convertTransform(delta, transformation_)
        for (int k = 0; k < 6; ++k) p[k] += delta[k];
        if (quirks::kPclNdtIterationRule == 1) {
            // PCL >= 1.11.1: the step's float matrix against the two epsilons (the translation SQUARED against the un-squared epsilon)
            const float  trace = transformation[0] + (transformation[5] + transformation[10]);  // Eigen's unrolled reduction of three terms: e0 + (e1 + e2)
            const double cos_angle = 0.5 * (trace - 1);
            const float  tx = transformation[3], ty = transformation[7], tz = transformation[11];
            const float  t2 = tx * tx + (ty * ty + tz * tz);  // Eigen's unrolled reduction of a 3-vector: e0 + (e1 + e2)
            const double translation_sqr = t2;
            nr_iterations++;
            if (nr_iterations >= max_iterations || ((trans_eps > 0 && translation_sqr <= trans_eps) && (rot_eps > 0 && cos_angle >= rot_eps)) ||
                ((trans_eps <= 0) && (rot_eps > 0 && cos_angle >= rot_eps)) || ((trans_eps > 0 && translation_sqr <= trans_eps) && (rot_eps <= 0)))
                converged = true;
        } else {
            // PCL <= 1.11.0 (the text pclomp forked)
            if (nr_iterations > max_iterations || (nr_iterations && (std::fabs(delta_norm) < trans_eps))) converged = true;
            nr_iterations++;
        }
    }
    trans_likelihood = score / static_cast<double>(n);
    write_output();
}

double PclNdt::fitness(double max_range) const
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
