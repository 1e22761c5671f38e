// oracle/gicp.cpp — TEST INFRASTRUCTURE (CPU oracle). Not part of the shipped product path.
//
// Restates (PARITY UNPINNED):
//   fast_gicp::FastGICP::{calculate_covariances (RegularizationMethod::PLANE), update_correspondences,
//     linearize, compute_error}, fast_gicp::LsqRegistration::{computeTransformation, step_lm, is_converged},
//   fast_gicp so3_exp / se3_exp, pcl::Registration::getFitnessScore
// behind /root/reference/src/mrg_slam/registrations.cpp:55-63 (FAST_GICP branch).
#include "gicp.h"

#include <omp.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>

#include "linalg.h"
#include "quirks.h"

namespace orc {

void FastGicp::set_target(const float* xyzi, int n)
{
    target.assign(xyzi, xyzi + static_cast<size_t>(n) * 4);
    target_covs_valid = false;
    target_grid_valid_ = false;
    voxelmap_valid_ = false;
}
void FastGicp::set_source(const float* xyzi, int n)
{
    source.assign(xyzi, xyzi + static_cast<size_t>(n) * 4);
    source_covs_valid = false;
}

void FastGicp::calculate_covariances(const std::vector<float>& cloud, std::vector<double>& covs) const
{
    const int n = static_cast<int>(cloud.size() / 4);
    const int k = k_correspondences;
    covs.assign(static_cast<size_t>(n) * 9, 0.0);
    NnGrid grid;
    grid.build(cloud.data(), n, 0.5f);
#pragma omp parallel for num_threads(num_threads) schedule(guided, 8)
    for (int i = 0; i < n; ++i) {
        std::vector<int>   idx(k);
        std::vector<float> sqd(k);
        const float* p = &cloud[4 * static_cast<size_t>(i)];
        int got = grid.knn(p[0], p[1], p[2], k, idx.data(), sqd.data());
        // neighbors (3 x k, double); colwise -= rowwise mean; cov = N N^T / k
        double mean[3] = {0, 0, 0};
        for (int j = 0; j < got; ++j) for (int a = 0; a < 3; ++a) mean[a] += cloud[4 * static_cast<size_t>(idx[j]) + a];
        for (int a = 0; a < 3; ++a) mean[a] /= k;
        double cov[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int j = 0; j < got; ++j) {
            double d[3];
            for (int a = 0; a < 3; ++a) d[a] = static_cast<double>(cloud[4 * static_cast<size_t>(idx[j]) + a]) - mean[a];
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) cov[r * 3 + c] += d[r] * d[c];
        }
        for (int t = 0; t < 9; ++t) cov[t] /= k;
        // PLANE regularisation: U diag(1,1,1e-3) V^T of the SVD == E diag E^T for a symmetric PSD matrix,
        // singular values descending  <=>  eigenvalues descending
        double ev[3], E[9];
        sym_eig3(cov, ev, E);  // ascending
        const double vals[3] = {quirks::kGicpPlaneEps, 1.0, 1.0};  // matched to ascending order
        double* out = &covs[static_cast<size_t>(i) * 9];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) {
                double s = 0;
                for (int m = 0; m < 3; ++m) s += E[r * 3 + m] * vals[m] * E[c * 3 + m];
                out[r * 3 + c] = s;
            }
    }
}

// fast_gicp::GaussianVoxelMap: voxel_coord, create_voxelmap with VoxelAccumulationMode::ADDITIVE
FastGicp::Coord FastGicp::voxel_coord(const double x[3]) const
{
    Coord c;
    for (int a = 0; a < 3; ++a) c.c[a] = static_cast<int>(std::floor(x[a] / voxel_resolution - 0.5));
    return c;
}
void FastGicp::build_voxelmap()
{
    voxel_index_.clear();
    voxels_.clear();
    const int n = static_cast<int>(target.size() / 4);
    for (int i = 0; i < n; ++i) {  // point order: the sums of a voxel are sequential in the point index
        const double x[3] = {target[4 * static_cast<size_t>(i)], target[4 * static_cast<size_t>(i) + 1], target[4 * static_cast<size_t>(i) + 2]};
        if (!std::isfinite(x[0]) || !std::isfinite(x[1]) || !std::isfinite(x[2])) continue;
        const Coord c = voxel_coord(x);
        auto it = voxel_index_.find(c);
        if (it == voxel_index_.end()) { it = voxel_index_.emplace(c, static_cast<int>(voxels_.size())).first; voxels_.emplace_back(); }
        Voxel& v = voxels_[it->second];
        v.num_points += 1;
        for (int a = 0; a < 3; ++a) v.mean[a] += x[a];
        for (int t = 0; t < 9; ++t) v.cov[t] += target_covs[static_cast<size_t>(i) * 9 + t];
    }
    for (Voxel& v : voxels_) {
        for (int a = 0; a < 3; ++a) v.mean[a] /= v.num_points;
        for (int t = 0; t < 9; ++t) v.cov[t] /= v.num_points;
    }
    voxelmap_valid_ = true;
}

void FastGicp::ensure_covs()
{
    if (!source_covs_valid) { calculate_covariances(source, source_covs); source_covs_valid = true; }
    if (!target_covs_valid) { calculate_covariances(target, target_covs); target_covs_valid = true; }
    if (!target_grid_valid_) { target_grid_.build(target.data(), static_cast<int>(target.size() / 4), 1.0f); target_grid_valid_ = true; }
    if (variant == 2 && !voxelmap_valid_) build_voxelmap();
}

void FastGicp::get_covariances(int which, double* out) const
{
    FastGicp* self = const_cast<FastGicp*>(this);
    self->ensure_covs();
    const std::vector<double>& c = which == 0 ? source_covs : target_covs;
    std::memcpy(out, c.data(), c.size() * sizeof(double));
}

static inline void isometry_apply_d(const double T[16], const double p[3], double out[3])
{
    for (int r = 0; r < 3; ++r) out[r] = T[r * 4 + 0] * p[0] + T[r * 4 + 1] * p[1] + T[r * 4 + 2] * p[2] + T[r * 4 + 3];
}

double FastGicp::linearize(const double T[16], double H[36], double b[6], int* n_corr)
{
    ensure_covs();
    ++n_linearize;
    const int n = static_cast<int>(source.size() / 4);
    correspondences_.assign(n, -1);
    mahalanobis_.assign(static_cast<size_t>(n) * 9, 0.0);
    float Tf[16];
    for (int i = 0; i < 16; ++i) Tf[i] = static_cast<float>(T[i]);
    const double thr2 = max_corr_dist * max_corr_dist;
    // update_correspondences
#pragma omp parallel for num_threads(num_threads) schedule(guided, 8)
    for (int i = 0; i < n; ++i) {
        const float* a = &source[4 * static_cast<size_t>(i)];
        // trans_f * Vector4f(x,y,z,1): Eigen 4x4 * 4x1 float product, accumulated left to right
        float q[3];
        for (int r = 0; r < 3; ++r) { float s = Tf[r * 4 + 0] * a[0]; s = s + Tf[r * 4 + 1] * a[1]; s = s + Tf[r * 4 + 2] * a[2]; q[r] = s + Tf[r * 4 + 3]; }
        int j;
        if (variant == 2) {  // FastVGICP::update_correspondences, DIRECT1: the voxel of trans * mean_A (double)
            const double mA[3] = {a[0], a[1], a[2]};
            double tA[3];
            for (int r = 0; r < 3; ++r) tA[r] = T[r * 4 + 0] * mA[0] + T[r * 4 + 1] * mA[1] + T[r * 4 + 2] * mA[2] + T[r * 4 + 3];
            if (!std::isfinite(tA[0]) || !std::isfinite(tA[1]) || !std::isfinite(tA[2])) continue;
            auto it = voxel_index_.find(voxel_coord(tA));
            if (it == voxel_index_.end()) continue;
            j = it->second;
        } else if (variant == 1 && double_search) {
            double qd[3];
            for (int r = 0; r < 3; ++r) { double s = T[r * 4 + 0] * static_cast<double>(a[0]); s = s + T[r * 4 + 1] * static_cast<double>(a[1]); s = s + T[r * 4 + 2] * static_cast<double>(a[2]); qd[r] = s + T[r * 4 + 3]; }
            int   cid[8];
            float csq[8];
            const int got = target_grid_.knn(static_cast<float>(qd[0]), static_cast<float>(qd[1]), static_cast<float>(qd[2]), 8, cid, csq);
            double best = 0;
            j = -1;
            for (int c = 0; c < got; ++c) {
                const float* t = &target[4 * static_cast<size_t>(cid[c])];
                const double dx = static_cast<double>(t[0]) - qd[0], dy = static_cast<double>(t[1]) - qd[1], dz = static_cast<double>(t[2]) - qd[2];
                const double d = (dx * dx + dy * dy) + dz * dz;
                if (j < 0 || d < best || (d == best && cid[c] < j)) { best = d; j = cid[c]; }
            }
            if (j < 0 || best > thr2) continue;  // DistanceRejector: sq_dist > max_dist_sq
        } else {
            float sqd;
            j = target_grid_.nearest(q[0], q[1], q[2], sqd);
            if (j < 0 || !(static_cast<double>(sqd) < thr2)) continue;
        }
        correspondences_[i] = j;
        const double* cA = &source_covs[static_cast<size_t>(i) * 9];
        const double* cB = variant == 2 ? voxels_[j].cov : &target_covs[static_cast<size_t>(j) * 9];
        double R[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]};
        double RC[9], Rt[9], RCR[9];
        mul3(R, cA, RC);
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Rt[r * 3 + c] = R[c * 3 + r];
        mul3(RC, Rt, RCR);
        for (int t = 0; t < 9; ++t) RCR[t] += cB[t];
        inv3(RCR, &mahalanobis_[static_cast<size_t>(i) * 9]);
    }
    double sum_errors = 0;
    std::vector<double> Hs(static_cast<size_t>(num_threads) * 36, 0.0), bs(static_cast<size_t>(num_threads) * 6, 0.0);
    int corr = 0;
#pragma omp parallel for num_threads(num_threads) reduction(+ : sum_errors, corr) schedule(guided, 8)
    for (int i = 0; i < n; ++i) {
        int j = correspondences_[i];
        if (j < 0) continue;
        ++corr;
        const double mean_A[3] = {source[4 * static_cast<size_t>(i)], source[4 * static_cast<size_t>(i) + 1], source[4 * static_cast<size_t>(i) + 2]};
        double mean_B[3] = {0, 0, 0}, w = 1.0;
        if (variant == 2) {
            for (int a = 0; a < 3; ++a) mean_B[a] = voxels_[j].mean[a];
            w = std::sqrt(static_cast<double>(voxels_[j].num_points));
        } else {
            for (int a = 0; a < 3; ++a) mean_B[a] = target[4 * static_cast<size_t>(j) + a];
        }
        double tA[3];
        isometry_apply_d(T, mean_A, tA);
        const double  err[3] = {mean_B[0] - tA[0], mean_B[1] - tA[1], mean_B[2] - tA[2]};
        const double* M = &mahalanobis_[static_cast<size_t>(i) * 9];
        double Me[3];
        for (int r = 0; r < 3; ++r) Me[r] = M[r * 3 + 0] * err[0] + M[r * 3 + 1] * err[1] + M[r * 3 + 2] * err[2];
        if (variant == 2) {  // every term of the voxelised cost carries w = sqrt(points in the voxel)
            sum_errors += w * (err[0] * Me[0] + err[1] * Me[1] + err[2] * Me[2]);
            for (int r = 0; r < 3; ++r) Me[r] = w * Me[r];
        } else {
            sum_errors += err[0] * Me[0] + err[1] * Me[1] + err[2] * Me[2];
        }
        if (!H || !b) continue;
        // J = [ skew(tA) | -I ]  (3 x 6)
        double J[3][6] = {{0, -tA[2], tA[1], -1, 0, 0}, {tA[2], 0, -tA[0], 0, -1, 0}, {-tA[1], tA[0], 0, 0, 0, -1}};
        if (variant == 1) {  // small_gicp: J = [ R skew(a) | -R ]
            for (int r = 0; r < 3; ++r) {
                const double r0 = T[r * 4 + 0], r1 = T[r * 4 + 1], r2 = T[r * 4 + 2];
                J[r][0] = r1 * mean_A[2] - r2 * mean_A[1];
                J[r][1] = r2 * mean_A[0] - r0 * mean_A[2];
                J[r][2] = r0 * mean_A[1] - r1 * mean_A[0];
                J[r][3] = -r0;
                J[r][4] = -r1;
                J[r][5] = -r2;
            }
        }
        double MJ[3][6];
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 6; ++c) MJ[r][c] = M[r * 3 + 0] * J[0][c] + M[r * 3 + 1] * J[1][c] + M[r * 3 + 2] * J[2][c];
        if (variant == 2)
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 6; ++c) MJ[r][c] = w * MJ[r][c];
        double* Ht = &Hs[static_cast<size_t>(omp_get_thread_num()) * 36];
        double* bt = &bs[static_cast<size_t>(omp_get_thread_num()) * 6];
        for (int r = 0; r < 6; ++r) {
            for (int c = 0; c < 6; ++c) Ht[r * 6 + c] += J[0][r] * MJ[0][c] + J[1][r] * MJ[1][c] + J[2][r] * MJ[2][c];
            bt[r] += J[0][r] * Me[0] + J[1][r] * Me[1] + J[2][r] * Me[2];
        }
    }
    if (H && b) {
        for (int t = 0; t < 36; ++t) H[t] = 0;
        for (int t = 0; t < 6; ++t) b[t] = 0;
        for (int th = 0; th < num_threads; ++th) {
            for (int t = 0; t < 36; ++t) H[t] += Hs[static_cast<size_t>(th) * 36 + t];
            for (int t = 0; t < 6; ++t) b[t] += bs[static_cast<size_t>(th) * 6 + t];
        }
    }
    if (n_corr) *n_corr = corr;
    return sum_errors;
}

double FastGicp::compute_error(const double T[16]) const
{
    const int n = static_cast<int>(source.size() / 4);
    double sum_errors = 0;
#pragma omp parallel for num_threads(num_threads) reduction(+ : sum_errors) schedule(guided, 8)
    for (int i = 0; i < n; ++i) {
        int j = correspondences_[i];
        if (j < 0) continue;
        const double mean_A[3] = {source[4 * static_cast<size_t>(i)], source[4 * static_cast<size_t>(i) + 1], source[4 * static_cast<size_t>(i) + 2]};
        double mean_B[3] = {0, 0, 0}, w = 1.0;
        if (variant == 2) {
            for (int a = 0; a < 3; ++a) mean_B[a] = voxels_[j].mean[a];
            w = std::sqrt(static_cast<double>(voxels_[j].num_points));
        } else {
            for (int a = 0; a < 3; ++a) mean_B[a] = target[4 * static_cast<size_t>(j) + a];
        }
        double tA[3];
        isometry_apply_d(T, mean_A, tA);
        const double  err[3] = {mean_B[0] - tA[0], mean_B[1] - tA[1], mean_B[2] - tA[2]};
        const double* M = &mahalanobis_[static_cast<size_t>(i) * 9];
        double Me[3];
        for (int r = 0; r < 3; ++r) Me[r] = M[r * 3 + 0] * err[0] + M[r * 3 + 1] * err[1] + M[r * 3 + 2] * err[2];
        const double e = err[0] * Me[0] + err[1] * Me[1] + err[2] * Me[2];
        sum_errors += variant == 2 ? w * e : e;
    }
    return sum_errors;
}

// fast_gicp/so3/so3.hpp
static void so3_exp_matrix(const double w[3], double R[9])
{
    double theta_sq = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    double imag_factor, real_factor;
    if (theta_sq < 1e-10) {
        double theta_quad = theta_sq * theta_sq;
        imag_factor = 0.5 - 1.0 / 48.0 * theta_sq + 1.0 / 3840.0 * theta_quad;
        real_factor = 1.0 - 1.0 / 8.0 * theta_sq + 1.0 / 384.0 * theta_quad;
    } else {
        double theta = std::sqrt(theta_sq);
        double half_theta = 0.5 * theta;
        imag_factor = std::sin(half_theta) / theta;
        real_factor = std::cos(half_theta);
    }
    // Eigen::Quaterniond(w,x,y,z).toRotationMatrix() (no normalisation)
    double qw = real_factor, qx = imag_factor * w[0], qy = imag_factor * w[1], qz = imag_factor * w[2];
    double tx = 2 * qx, ty = 2 * qy, tz = 2 * qz;
    double twx = tx * qw, twy = ty * qw, twz = tz * qw;
    double txx = tx * qx, txy = ty * qx, txz = tz * qx;
    double tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}
static void se3_exp(const double a[6], double T[16])
{
    const double w[3] = {a[0], a[1], a[2]};
    double theta = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    double R[9];
    so3_exp_matrix(w, R);
    double Om[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
    double Om2[9];
    mul3(Om, Om, Om2);
    double V[9];
    if (theta < 1e-10) {
        std::memcpy(V, R, sizeof(V));
    } else {
        double theta_sq = theta * theta;
        double c1 = (1.0 - std::cos(theta)) / theta_sq, c2 = (theta - std::sin(theta)) / (theta_sq * theta);
        for (int t = 0; t < 9; ++t) V[t] = ((t % 4 == 0) ? 1.0 : 0.0) + c1 * Om[t] + c2 * Om2[t];
    }
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) T[r * 4 + c] = R[r * 3 + c];
        T[r * 4 + 3] = V[r * 3 + 0] * a[3] + V[r * 3 + 1] * a[4] + V[r * 3 + 2] * a[5];
    }
    T[12] = T[13] = T[14] = 0; T[15] = 1;
}
static void mul4d(const double A[16], const double B[16], double out[16])
{
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) { double s = 0; for (int k = 0; k < 4; ++k) s += A[r * 4 + k] * B[k * 4 + c]; out[r * 4 + c] = s; }
}
// solve A x = rhs, A symmetric 6x6 (H + lambda I): Gaussian elimination with partial pivoting
// (stands in for Eigen::LDLT<Matrix6d>::solve; same solution up to rounding)
static void solve6(const double A_in[36], const double rhs[6], double x[6])
{
    double A[6][7];
    for (int r = 0; r < 6; ++r) { for (int c = 0; c < 6; ++c) A[r][c] = A_in[r * 6 + c]; A[r][6] = rhs[r]; }
    for (int k = 0; k < 6; ++k) {
        int piv = k;
        for (int r = k + 1; r < 6; ++r) if (std::fabs(A[r][k]) > std::fabs(A[piv][k])) piv = r;
        if (piv != k) for (int c = 0; c < 7; ++c) std::swap(A[k][c], A[piv][c]);
        for (int r = k + 1; r < 6; ++r) {
            double f = A[r][k] / A[k][k];
            for (int c = k; c < 7; ++c) A[r][c] -= f * A[k][c];
        }
    }
    for (int r = 5; r >= 0; --r) {
        double s = A[r][6];
        for (int c = r + 1; c < 6; ++c) s -= A[r][c] * x[c];
        x[r] = s / A[r][r];
    }
}

static bool is_converged(const double delta[16], double rot_eps, double trans_eps)
{
    double mx = 0;
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) mx = std::max(mx, 1.0 / rot_eps * std::fabs(delta[r * 4 + c] - (r == c ? 1.0 : 0.0)));
        mx = std::max(mx, 1.0 / trans_eps * std::fabs(delta[r * 4 + 3]));
    }
    return mx < 1;
}

// small_gicp::LevenbergMarquardtOptimizer::optimize over the GICP factors (header of gicp.h)
void FastGicp::align_small_gicp(const float guess[16])
{
    double T[16];
    for (int i = 0; i < 16; ++i) T[i] = static_cast<double>(guess[i]);
    double lambda = sg_init_lambda;
    converged = false;
    nr_iterations = 0;
    n_linearize = n_error_evals = 0;
    for (int t = 0; t < 36; ++t) final_hessian[t] = (t % 7 == 0) ? 1.0 : 0.0;
    for (int i = 0; i < max_iterations && !converged; ++i) {
        double H[36], b[6];
        double e = 0.5 * linearize(T, H, b, nullptr);
        bool   success = false;
        for (int j = 0; j < sg_max_inner_iterations; ++j) {
            double A[36], nb[6], d[6], delta[16], new_T[16];
            for (int t = 0; t < 36; ++t) A[t] = H[t] + ((t % 7 == 0) ? lambda : 0.0);
            for (int t = 0; t < 6; ++t) nb[t] = -b[t];
            solve6(A, nb, d);
            se3_exp(d, delta);
            mul4d(T, delta, new_T);
            const double new_e = 0.5 * compute_error(new_T);
            ++n_error_evals;
            if (new_e <= e) {
                converged = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]) <= rot_eps && std::sqrt(d[3] * d[3] + d[4] * d[4] + d[5] * d[5]) <= trans_eps;
                std::memcpy(T, new_T, sizeof(new_T));
                lambda /= sg_lambda_factor;
                success = true;
                e = new_e;
                break;
            }
            lambda *= sg_lambda_factor;
        }
        nr_iterations = i;
        std::memcpy(final_hessian, H, sizeof(H));
        if (!success) break;
    }
    for (int i = 0; i < 16; ++i) final_[i] = static_cast<float>(T[i]);
}

// pcl::IterativeClosestPoint::computeTransformation + DefaultConvergenceCriteria::hasConverged +
// TransformationEstimationSVD (header of gicp.h)
void FastGicp::align_icp(const float guess[16])
{
    const int n = static_cast<int>(source.size() / 4), nt = static_cast<int>(target.size() / 4);
    if (!target_grid_valid_) { target_grid_.build(target.data(), nt, 1.0f); target_grid_valid_ = true; }
    std::vector<float> cur(source);
    bool guess_is_identity = true;
    for (int i = 0; i < 16; ++i) guess_is_identity = guess_is_identity && guess[i] == ((i % 5 == 0) ? 1.0f : 0.0f);
    if (!guess_is_identity)
        for (int i = 0; i < n; ++i) transform_point_f(guess, source[4 * static_cast<size_t>(i)], source[4 * static_cast<size_t>(i) + 1], source[4 * static_cast<size_t>(i) + 2],
                                                      cur[4 * static_cast<size_t>(i)], cur[4 * static_cast<size_t>(i) + 1], cur[4 * static_cast<size_t>(i) + 2]);
    float fin[16];
    std::memcpy(fin, guess, sizeof(fin));
    converged = false;
    nr_iterations = 0;
    n_linearize = n_error_evals = 0;
    for (int t = 0; t < 36; ++t) final_hessian[t] = 0.0;
    const double max_sq = max_corr_dist * max_corr_dist;
    const double rot_thr = 1.0 - trans_eps, trans_thr = trans_eps;  // setRotationThreshold(1 - eps), setTranslationThreshold(eps)
    double prev_mse = std::numeric_limits<double>::max();
    for (;;) {
        ++n_linearize;
        double cnt = 0, Ss[3] = {0, 0, 0}, Sd[3] = {0, 0, 0}, Sds[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, Smse = 0;
        NnGrid cur_grid;  // determineReciprocalCorrespondences: a search structure over the source as transformed so far, rebuilt every iteration
        if (use_reciprocal && n) cur_grid.build(cur.data(), n, 1.0f);
        for (int i = 0; i < n; ++i) {
            const float* p = &cur[4 * static_cast<size_t>(i)];
            float sqd;
            const int j = nt ? target_grid_.nearest(p[0], p[1], p[2], sqd) : -1;
            if (j < 0 || static_cast<double>(sqd) > max_sq) continue;
            const float* q = &target[4 * static_cast<size_t>(j)];
            if (use_reciprocal) {
                float     rsqd;
                const int ri = cur_grid.nearest(q[0], q[1], q[2], rsqd);
                if (ri < 0 || static_cast<double>(rsqd) > max_sq || ri != i) continue;
            }
            cnt += 1;
            for (int a = 0; a < 3; ++a) { Ss[a] += p[a]; Sd[a] += q[a]; }
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Sds[r * 3 + c] += static_cast<double>(q[r]) * static_cast<double>(p[c]);
            Smse += sqd;
        }
        if (cnt < 3) { converged = false; break; }  // "Not enough correspondences found"
        double mu_s[3], mu_d[3], sigma[9], R[9];
        for (int a = 0; a < 3; ++a) { mu_s[a] = Ss[a] / cnt; mu_d[a] = Sd[a] / cnt; }
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) sigma[r * 3 + c] = Sds[r * 3 + c] / cnt - mu_d[r] * mu_s[c];
        umeyama_rotation(sigma, R);
        float Tm[16];
        mat4f_identity(Tm);
        const float ms[3] = {static_cast<float>(mu_s[0]), static_cast<float>(mu_s[1]), static_cast<float>(mu_s[2])};
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) Tm[r * 4 + c] = static_cast<float>(R[r * 3 + c]);
            float s = Tm[r * 4 + 0] * ms[0];  // Rt.col(3).head(3) = dst_mean - R * src_mean, in float
            s = s + Tm[r * 4 + 1] * ms[1];
            s = s + Tm[r * 4 + 2] * ms[2];
            Tm[r * 4 + 3] = static_cast<float>(mu_d[r]) - s;
        }
        for (int i = 0; i < n; ++i) {
            float* p = &cur[4 * static_cast<size_t>(i)];
            float x, y, z;
            transform_point_f(Tm, p[0], p[1], p[2], x, y, z);
            p[0] = x; p[1] = y; p[2] = z;
        }
        float nf[16];
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 4; ++c) { float s = 0; for (int k = 0; k < 4; ++k) s += Tm[r * 4 + k] * fin[k * 4 + c]; nf[r * 4 + c] = s; }
        std::memcpy(fin, nf, sizeof(fin));
        ++nr_iterations;
        // DefaultConvergenceCriteria::hasConverged
        if (nr_iterations >= max_iterations) { converged = true; break; }
        const double cos_angle = 0.5 * (static_cast<double>(Tm[0]) + static_cast<double>(Tm[5]) + static_cast<double>(Tm[10]) - 1.0);
        const double tsq = static_cast<double>(Tm[3]) * Tm[3] + static_cast<double>(Tm[7]) * Tm[7] + static_cast<double>(Tm[11]) * Tm[11];
        if (cos_angle >= rot_thr && tsq <= trans_thr) { converged = true; break; }
        const double mse = Smse / cnt;
        if (std::fabs(mse - prev_mse) < 1e-12) { converged = true; break; }
        prev_mse = mse;
    }
    std::memcpy(final_, fin, sizeof(fin));
}

void FastGicp::align(const float guess[16], float* aligned)
{
    const int n = static_cast<int>(source.size() / 4);
    if (variant == 3) {
        align_icp(guess);
        if (aligned)
            for (int i = 0; i < n; ++i) {
                const float* p = &source[4 * static_cast<size_t>(i)];
                transform_point_f(final_, p[0], p[1], p[2], aligned[4 * i], aligned[4 * i + 1], aligned[4 * i + 2]);
                aligned[4 * i + 3] = p[3];
            }
        return;
    }
    ensure_covs();
    if (variant == 1) {
        align_small_gicp(guess);
        if (aligned)
            for (int i = 0; i < n; ++i) {
                const float* p = &source[4 * static_cast<size_t>(i)];
                transform_point_f(final_, p[0], p[1], p[2], aligned[4 * i], aligned[4 * i + 1], aligned[4 * i + 2]);
                aligned[4 * i + 3] = p[3];
            }
        return;
    }
    double x0[16];
    for (int i = 0; i < 16; ++i) x0[i] = static_cast<double>(guess[i]);
    double lm_lambda = -1.0;
    converged = false;
    nr_iterations = 0;
    n_linearize = n_error_evals = 0;
    for (int t = 0; t < 36; ++t) final_hessian[t] = (t % 7 == 0) ? 1.0 : 0.0;
    for (int i = 0; i < max_iterations && !converged; ++i) {
        nr_iterations = i;
        // step_lm
        double H[36], b[6], delta[16];
        double y0 = linearize(x0, H, b, nullptr);
        if (lm_lambda < 0.0) {
            double md = 0;
            for (int d = 0; d < 6; ++d) md = std::max(md, std::fabs(H[d * 6 + d]));
            lm_lambda = lm_init_lambda_factor * md;
        }
        double nu = 2.0;
        bool   ok = false;
        for (int it = 0; it < lm_max_iterations; ++it) {
            double A[36], nb[6], d[6];
            for (int t = 0; t < 36; ++t) A[t] = H[t] + ((t % 7 == 0) ? lm_lambda : 0.0);
            for (int t = 0; t < 6; ++t) nb[t] = -b[t];
            solve6(A, nb, d);
            se3_exp(d, delta);
            double xi[16];
            mul4d(delta, x0, xi);
            double yi = compute_error(xi);
            ++n_error_evals;
            double denom = 0;
            for (int t = 0; t < 6; ++t) denom += d[t] * (lm_lambda * d[t] - b[t]);
            double rho = (y0 - yi) / denom;
            if (rho < 0) {
                if (is_converged(delta, rot_eps, trans_eps)) { ok = true; break; }
                lm_lambda = nu * lm_lambda;
                nu = 2 * nu;
                continue;
            }
            std::memcpy(x0, xi, sizeof(xi));
            lm_lambda = lm_lambda * std::max(1.0 / 3.0, 1 - std::pow(2 * rho - 1, 3));
            std::memcpy(final_hessian, H, sizeof(H));
            ok = true;
            break;
        }
        if (!ok) break;  // "lm not converged!!"
        converged = is_converged(delta, rot_eps, trans_eps);
    }
    for (int i = 0; i < 16; ++i) final_[i] = static_cast<float>(x0[i]);
    if (aligned)
        for (int i = 0; i < n; ++i) {
            const float* p = &source[4 * static_cast<size_t>(i)];
            transform_point_f(final_, p[0], p[1], p[2], aligned[4 * i], aligned[4 * i + 1], aligned[4 * i + 2]);
            aligned[4 * i + 3] = p[3];
        }
}

double FastGicp::fitness(double max_range) const
{
    const int n = static_cast<int>(source.size() / 4), nt = static_cast<int>(target.size() / 4);
    if (n == 0 || nt == 0) return std::numeric_limits<double>::max();
    NnGrid grid;
    grid.build(target.data(), nt, 1.0f);
    double sum = 0; int nr = 0;
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
