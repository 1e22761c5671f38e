// oracle/pcl_gicp.cpp — TEST INFRASTRUCTURE (CPU oracle). Not part of the shipped product path.
// pcl::GeneralizedIterativeClosestPoint restated (pcl_gicp.h); the minimiser is bfgs.h.
#include "pcl_gicp.h"

#include <omp.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>

#include "bfgs.h"
#include "linalg.h"

namespace orc {

namespace {
// Eigen's Matrix4f * Vector4f for a point with w = 1 (PCL keeps 1.0f in the padding word): ((m0 x + m1 y) + m2 z) + m3, row by row
inline void mat4f_point(const float M[16], const float* p, float out[3])
{
    for (int r = 0; r < 3; ++r) {
        float s = M[r * 4 + 0] * p[0];
        s = s + M[r * 4 + 1] * p[1];
        s = s + M[r * 4 + 2] * p[2];
        out[r] = s + M[r * 4 + 3];
    }
}
inline void mat4f_mul(const float A[16], const float B[16], float out[16])
{
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {
            float s = 0;
            for (int k = 0; k < 4; ++k) s += A[r * 4 + k] * B[k * 4 + c];
            out[r * 4 + c] = s;
        }
}
// GeneralizedIterativeClosestPoint::applyState: t.topLeftCorner<3,3>() = Rz(x5) Ry(x4) Rx(x3) * t.topLeftCorner<3,3>(); t.col(3) += (x0, x1, x2, 0); float
inline void apply_state(float t[16], const double x[6])
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

struct Functor {  // OptimizationFunctorWithIndices
    const std::vector<float>*  src;   // `output`: the source moved by the guess
    const std::vector<float>*  tgt;
    const std::vector<int>*    idx_src;
    const std::vector<int>*    idx_tgt;
    const std::vector<double>* mahal;  // 9 per source point
    float base[16];                    // base_transformation_ (identity in computeTransformation)
    int   gpu_order = 0;               // diagnostic: add the per-point terms in the order the HIP kernels do (see terms_gpu_order)
    int   sum_threads = 1;             // pclomp: per-thread partial sums over static chunks, added in thread order (see terms)

    // The same per-point terms as terms(), added in the order of mrg_slam_amd/csrc/gicp.hip (pclgicp_fdf_kernel + gicp_reduce_record): one
    // lane per SOURCE point (points without a correspondence contribute exact zeros), workgroups of 256 points, per wavefront of 64 a
    // shuffle-down tree (offsets 32 ... 1), the four wavefronts as ((w0 + w1) + w2) + w3, the workgroup partials in eight interleaved slices
    // (slice s takes workgroups s, s + 8, ...), the slices in order.  A diagnostic mode: it shows that what separates a HIP result from the
    // reference-order result is the order of these f64 additions and nothing else.
    static double sum_gpu_order(const std::vector<double>& v)
    {
        const size_t n = v.size(), nblk = (n + 255) / 256;
        double slice[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (size_t b = 0; b < nblk; ++b) {
            double w[4];
            for (int wv = 0; wv < 4; ++wv) {
                double a[64];
                for (int l = 0; l < 64; ++l) {
                    const size_t i = b * 256 + static_cast<size_t>(wv) * 64 + l;
                    a[l] = i < n ? v[i] : 0.0;
                }
                for (int off = 32; off > 0; off >>= 1)
                    for (int l = 0; l < off; ++l) a[l] = a[l] + a[l + off];
                w[wv] = a[0];
            }
            slice[b & 7] += ((w[0] + w[1]) + w[2]) + w[3];
        }
        double r = slice[0];
        for (int sl = 1; sl < 8; ++sl) r += slice[sl];
        return r;
    }
    void terms_gpu_order(const double x[6], double* fsum, double g_t[3], double dC[9]) const
    {
        float T[16];
        std::memcpy(T, base, sizeof(T));
        apply_state(T, x);
        const size_t n = src->size() / 4;
        std::vector<std::vector<double>> col(13, std::vector<double>(n, 0.0));
        const int m = static_cast<int>(idx_src->size());
        for (int i = 0; i < m; ++i) {
            const int    is = (*idx_src)[i], it = (*idx_tgt)[i];
            const float* ps = &(*src)[4 * static_cast<size_t>(is)];
            const float* pt = &(*tgt)[4 * static_cast<size_t>(it)];
            float q[3];
            mat4f_point(T, ps, q);
            const double d[3] = {static_cast<double>(q[0] - pt[0]), static_cast<double>(q[1] - pt[1]), static_cast<double>(q[2] - pt[2])};
            const double* M = &(*mahal)[9 * static_cast<size_t>(is)];
            const double Md[3] = {M[0] * d[0] + M[1] * d[1] + M[2] * d[2], M[3] * d[0] + M[4] * d[1] + M[5] * d[2], M[6] * d[0] + M[7] * d[1] + M[8] * d[2]};
            col[0][is] = d[0] * Md[0] + d[1] * Md[1] + d[2] * Md[2];
            float pb[3];
            mat4f_point(base, ps, pb);
            for (int k = 0; k < 3; ++k) col[1 + k][is] = Md[k];
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) col[4 + r * 3 + c][is] = static_cast<double>(pb[r]) * Md[c];
        }
        *fsum = sum_gpu_order(col[0]);
        if (g_t) {
            for (int k = 0; k < 3; ++k) g_t[k] = sum_gpu_order(col[1 + k]);
            for (int k = 0; k < 9; ++k) dC[k] = sum_gpu_order(col[4 + k]);
        }
    }

    void terms(const double x[6], double* fsum, double g_t[3], double dC[9]) const
    {
        if (gpu_order) { terms_gpu_order(x, fsum, g_t, dC); return; }
        float T[16];
        std::memcpy(T, base, sizeof(T));
        apply_state(T, x);
        const int m = static_cast<int>(idx_src->size());
        // pclomp (gicp_omp_impl.hpp OptimizationFunctorWithIndices): `#pragma omp parallel for` over the correspondences with f_array / g_array / R_array
        // indexed by omp_get_thread_num(), summed over the threads afterwards.  schedule(static) without a chunk size: thread t of T takes iterations
        // [q t + min(t, r), ...) with q = m / T, r = m mod T, the first r threads one more (libgomp).  T = sum_threads; 1 = serial pcl::GICP: one chunk.
        const int NT = std::max(1, sum_threads);
        double f = 0;
        if (g_t) { g_t[0] = g_t[1] = g_t[2] = 0; for (int k = 0; k < 9; ++k) dC[k] = 0; }
        const int q_len = m / NT, r_len = m % NT;
        for (int th = 0; th < NT; ++th) {
          const int i0 = q_len * th + std::min(th, r_len);
          const int i1 = i0 + q_len + (th < r_len ? 1 : 0);
          double f_th = 0, g_th[3] = {0, 0, 0}, dC_th[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
          for (int i = i0; i < i1; ++i) {
            const int    is = (*idx_src)[i], it = (*idx_tgt)[i];
            const float* ps = &(*src)[4 * static_cast<size_t>(is)];
            const float* pt = &(*tgt)[4 * static_cast<size_t>(it)];
            float q[3];
            mat4f_point(T, ps, q);
            const double d[3] = {static_cast<double>(q[0] - pt[0]), static_cast<double>(q[1] - pt[1]), static_cast<double>(q[2] - pt[2])};
            const double* M = &(*mahal)[9 * static_cast<size_t>(is)];
            const double Md[3] = {M[0] * d[0] + M[1] * d[1] + M[2] * d[2], M[3] * d[0] + M[4] * d[1] + M[5] * d[2], M[6] * d[0] + M[7] * d[1] + M[8] * d[2]};
            f_th += d[0] * Md[0] + d[1] * Md[1] + d[2] * Md[2];
            if (g_t) {
                for (int k = 0; k < 3; ++k) g_th[k] += Md[k];
                float pb[3];
                mat4f_point(base, ps, pb);  // p_base_src = base_transformation_ * p_src
                for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) dC_th[r * 3 + c] += static_cast<double>(pb[r]) * Md[c];  // p_base_src * Md^T
            }
          }
          // the partials in thread order (a sum that starts at +0.0 takes its first term exactly: T = 1 is the serial chain)
          f += f_th;
          if (g_t) { for (int k = 0; k < 3; ++k) g_t[k] += g_th[k]; for (int k = 0; k < 9; ++k) dC[k] += dC_th[k]; }
        }
        *fsum = f;
    }
    // computeRDerivative: g[3..5] = tr(dR/dphi dCost_dR_T) etc. for R = Rz(psi) Ry(theta) Rx(phi)
    static void r_derivative(const double x[6], const double dC[9], double g[6])
    {
        const double phi = x[3], theta = x[4], psi = x[5];
        const double cphi = std::cos(phi), sphi = std::sin(phi), ctheta = std::cos(theta), stheta = std::sin(theta), cpsi = std::cos(psi), spsi = std::sin(psi);
        const double dPhi[9] = {0, sphi * spsi + cphi * cpsi * stheta, cphi * spsi - cpsi * sphi * stheta,
                                0, -cpsi * sphi + cphi * spsi * stheta, -cphi * cpsi - sphi * spsi * stheta,
                                0, cphi * ctheta, -ctheta * sphi};
        const double dTheta[9] = {-cpsi * stheta, cpsi * ctheta * sphi, cphi * cpsi * ctheta,
                                  -spsi * stheta, ctheta * sphi * spsi, cphi * ctheta * spsi,
                                  -ctheta, -sphi * stheta, -cphi * stheta};
        const double dPsi[9] = {-ctheta * spsi, -cphi * cpsi - sphi * spsi * stheta, cpsi * sphi - cphi * spsi * stheta,
                                cpsi * ctheta, -cphi * spsi + cpsi * sphi * stheta, sphi * spsi + cphi * cpsi * stheta,
                                0, 0, 0};
        auto inner = [&](const double A[9]) {  // matricesInnerProd(A, dCost_dR_T) = sum_ij A(j, i) dC(i, j)
            double r = 0;
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r += A[j * 3 + i] * dC[i * 3 + j];
            return r;
        };
        g[3] = inner(dPhi);
        g[4] = inner(dTheta);
        g[5] = inner(dPsi);
    }
    double f(const double x[6]) const
    {
        double fs;
        terms(x, &fs, nullptr, nullptr);
        return fs / static_cast<double>(idx_src->size());
    }
    void fdf(const double x[6], double& fv, double g[6]) const
    {
        double fs, gt[3], dC[9];
        terms(x, &fs, gt, dC);
        const double m = static_cast<double>(idx_src->size());
        fv = fs / m;
        for (int k = 0; k < 3; ++k) g[k] = gt[k] * (2.0 / m);
        for (int k = 0; k < 9; ++k) dC[k] *= 2.0 / m;
        r_derivative(x, dC, g);
    }
    void df(const double x[6], double g[6]) const
    {
        double fv;
        fdf(x, fv, g);
    }
};
}  // namespace

void PclGicp::set_target(const float* xyzi, int n)
{
    target.assign(xyzi, xyzi + static_cast<size_t>(n) * 4);
    target_covs_valid = false;
    target_grid_valid_ = false;
}
void PclGicp::set_source(const float* xyzi, int n)
{
    source.assign(xyzi, xyzi + static_cast<size_t>(n) * 4);
    source_covs_valid = false;
}

// GeneralizedIterativeClosestPoint::computeCovariances
void PclGicp::compute_covariances(const std::vector<float>& cloud, std::vector<double>& covs) const
{
    const int n = static_cast<int>(cloud.size() / 4), k = k_correspondences;
    covs.assign(static_cast<size_t>(n) * 9, 0.0);
    NnGrid grid;
    grid.build(cloud.data(), n, 0.5f);
#pragma omp parallel for num_threads(num_threads) schedule(guided, 8)
    for (int i = 0; i < n; ++i) {
        std::vector<int>   idx(k);
        std::vector<float> sqd(k);
        const float* p = &cloud[4 * static_cast<size_t>(i)];
        const int got = grid.knn(p[0], p[1], p[2], k, idx.data(), sqd.data());
        double mean[3] = {0, 0, 0}, cov[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int j = 0; j < got; ++j) {
            const float* q = &cloud[4 * static_cast<size_t>(idx[j])];
            mean[0] += q[0]; mean[1] += q[1]; mean[2] += q[2];
            // float products accumulated in double: cov(0,0) += pt.x * pt.x; cov(1,0) += pt.y * pt.x; ...
            cov[0] += q[0] * q[0];
            cov[3] += q[1] * q[0];
            cov[4] += q[1] * q[1];
            cov[6] += q[2] * q[0];
            cov[7] += q[2] * q[1];
            cov[8] += q[2] * q[2];
        }
        for (int a = 0; a < 3; ++a) mean[a] /= static_cast<double>(k);
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c <= r; ++c) {
                cov[r * 3 + c] /= static_cast<double>(k);
                cov[r * 3 + c] -= mean[r] * mean[c];
                cov[c * 3 + r] = cov[r * 3 + c];
            }
        // JacobiSVD of a symmetric matrix: singular values |lambda| descending, U = the eigenvectors; the smallest one becomes gicp_epsilon
        double ev[3], E[9];
        sym_eig3(cov, ev, E);
        int order[3] = {0, 1, 2};
        std::sort(order, order + 3, [&](int a, int b) { return std::fabs(ev[a]) > std::fabs(ev[b]); });
        double* out = &covs[static_cast<size_t>(i) * 9];
        for (int kk = 0; kk < 3; ++kk) {
            const double v = kk == 2 ? gicp_epsilon : 1.0;
            const int    col = order[kk];
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) out[r * 3 + c] += v * E[r * 3 + col] * E[c * 3 + col];
        }
    }
}

void PclGicp::ensure()
{
    if (!target_covs_valid) { compute_covariances(target, target_covs); target_covs_valid = true; }
    if (!source_covs_valid) { compute_covariances(source, source_covs); source_covs_valid = true; }
    if (!target_grid_valid_) { target_grid_.build(target.data(), static_cast<int>(target.size() / 4), 1.0f); target_grid_valid_ = true; }
}

void PclGicp::get_covariances(int which, double* out9)
{
    ensure();
    const std::vector<double>& c = which == 0 ? source_covs : target_covs;
    std::memcpy(out9, c.data(), sizeof(double) * c.size());
}

namespace {
// correspondences and Mahalanobis matrices of one outer iteration (gicp.hpp computeTransformation, the loop over the source)
void correspondences(const PclGicp& g, const NnGrid& grid, const std::vector<float>& output, const float transformation[16], const float guess[16],
                     std::vector<int>& idx_src, std::vector<int>& idx_tgt, std::vector<double>& mahal)
{
    const int n = static_cast<int>(output.size() / 4), nt = static_cast<int>(g.target.size() / 4);
    const double thr = g.max_corr_dist * g.max_corr_dist;
    double R[9];  // rotation of transformation_ * guess, in double
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 4; ++k) s += static_cast<double>(transformation[i * 4 + k]) * static_cast<double>(guess[k * 4 + j]);
            R[i * 3 + j] = s;
        }
    idx_src.clear();
    idx_tgt.clear();
    mahal.assign(static_cast<size_t>(n) * 9, 0.0);
    for (int i = 0; i < n; ++i) { mahal[9 * static_cast<size_t>(i)] = mahal[9 * static_cast<size_t>(i) + 4] = mahal[9 * static_cast<size_t>(i) + 8] = 1.0; }  // Identity
    for (int i = 0; i < n; ++i) {
        float q[3];
        mat4f_point(transformation, &output[4 * static_cast<size_t>(i)], q);
        float sqd;
        const int j = nt ? grid.nearest(q[0], q[1], q[2], sqd) : -1;
        if (j < 0) continue;
        if (static_cast<double>(sqd) < thr) {
            const double* C1 = &g.source_covs[9 * static_cast<size_t>(i)];
            const double* C2 = &g.target_covs[9 * static_cast<size_t>(j)];
            double RC[9], Rt[9], tmp[9];
            mul3(R, C1, RC);
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Rt[r * 3 + c] = R[c * 3 + r];
            mul3(RC, Rt, tmp);
            for (int k = 0; k < 9; ++k) tmp[k] += C2[k];
            inv3(tmp, &mahal[9 * static_cast<size_t>(i)]);
            idx_src.push_back(i);
            idx_tgt.push_back(j);
        }
    }
}
}  // namespace

double PclGicp::evaluate(const float T[16], const double x[6], double g[6], int* n_corr)
{
    ensure();
    float eye[16];
    mat4f_identity(eye);
    std::vector<int>    is, it;
    std::vector<double> mahal;
    correspondences(*this, target_grid_, source, T, eye, is, it, mahal);
    if (n_corr) *n_corr = static_cast<int>(is.size());
    Functor fn{&source, &target, &is, &it, &mahal, {}, gpu_order, sum_threads};
    mat4f_identity(fn.base);
    double f = 0;
    if (is.empty()) { for (int k = 0; k < 6; ++k) g[k] = 0; return 0.0; }
    fn.fdf(x, f, g);
    return f;
}

void PclGicp::align(const float guess[16], float* aligned)
{
    const int n = static_cast<int>(source.size() / 4);
    ensure();
    converged = false;
    nr_iterations = 0;
    n_evaluations = n_inner_steps = 0;
    float transformation[16], previous[16];
    mat4f_identity(transformation);
    mat4f_identity(previous);
    mat4f_identity(final_);
    // pcl::Registration::align copies the source into `output`; computeTransformation moves it by the guess
    std::vector<float> output(source);
    for (int i = 0; i < n; ++i) {
        float* p = &output[4 * static_cast<size_t>(i)];
        float  x, y, z;
        transform_point_f(guess, p[0], p[1], p[2], x, y, z);  // pcl::transformPointCloud
        p[0] = x; p[1] = y; p[2] = z;
    }
    std::vector<int>    idx_src, idx_tgt;
    std::vector<double> mahal;
    bool stopped_by_exception = false;
    while (!converged && n > 0) {
        correspondences(*this, target_grid_, output, transformation, guess, idx_src, idx_tgt, mahal);
        std::memcpy(previous, transformation, sizeof(previous));
        // ---- estimateRigidTransformationBFGS
        if (idx_src.size() < 4) { stopped_by_exception = true; break; }  // NotEnoughPointsException -> break
        // x[3] = std::atan2(T(2,1), T(2,2)) and x[5] = std::atan2(T(1,0), T(0,0)) on floats (the float overload), x[4] = asin(-T(2,0)) through
        // the unqualified C function, i.e. in double
        double x[6] = {transformation[3], transformation[7], transformation[11], static_cast<double>(std::atan2(transformation[9], transformation[10])),
                       std::asin(static_cast<double>(-transformation[8])), static_cast<double>(std::atan2(transformation[4], transformation[0]))};
        Functor fn{&output, &target, &idx_src, &idx_tgt, &mahal, {}, gpu_order, sum_threads};
        mat4f_identity(fn.base);
        Bfgs<Functor> bfgs(fn);
        int inner = 0;
        int result = bfgs.minimize_init(x);
        result = BFGS_RUNNING;
        do {
            ++inner;
            result = bfgs.minimize_one_step(x);
            if (result) break;
            // testGradient -> OptimizationFunctorWithIndices::checkGradient (PCL >= 1.11: translation and rotation parts apart)
            const double* gr = bfgs.gradient;
            const double gt = std::sqrt(gr[0] * gr[0] + gr[1] * gr[1] + gr[2] * gr[2]), grn = std::sqrt(gr[3] * gr[3] + gr[4] * gr[4] + gr[5] * gr[5]);
            const bool ok = whole_gradient_norm ? std::sqrt(gt * gt + grn * grn) < translation_gradient_tolerance : (gt < translation_gradient_tolerance && grn < rotation_gradient_tolerance);
            result = ok ? BFGS_SUCCESS : BFGS_RUNNING;
        } while (result == BFGS_RUNNING && inner < max_inner_iterations);
        n_evaluations += bfgs.evaluations;
        n_inner_steps += inner;
        if (result == BFGS_NO_PROGRESS || result == BFGS_SUCCESS || inner == max_inner_iterations) {
            mat4f_identity(transformation);
            apply_state(transformation, x);
        } else {
            stopped_by_exception = true;  // SolverDidntConvergeException -> break
            break;
        }
        double delta = 0;
        for (int k = 0; k < 4; ++k)
            for (int l = 0; l < 4; ++l) {
                const double ratio = (k < 3 && l < 3) ? 1.0 / rot_eps : 1.0 / trans_eps;
                const double c_delta = ratio * std::fabs(static_cast<double>(previous[k * 4 + l] - transformation[k * 4 + l]));
                if (c_delta > delta) delta = c_delta;
            }
        ++nr_iterations;
        if (nr_iterations >= max_iterations || delta < 1) {
            converged = true;
            std::memcpy(previous, transformation, sizeof(previous));
        }
    }
    (void)stopped_by_exception;
    mat4f_mul(previous, guess, final_);  // final_transformation_ = previous_transformation_ * guess
    if (aligned)
        for (int i = 0; i < n; ++i) {
            const float* p = &source[4 * static_cast<size_t>(i)];
            transform_point_f(final_, p[0], p[1], p[2], aligned[4 * i], aligned[4 * i + 1], aligned[4 * i + 2]);
            aligned[4 * i + 3] = p[3];
        }
}

double PclGicp::fitness(double max_range) const
{
    const int n = static_cast<int>(source.size() / 4), nt = static_cast<int>(target.size() / 4);
    if (!n || !nt) return std::numeric_limits<double>::max();
    NnGrid grid;
    grid.build(target.data(), nt, 1.0f);
    double sum = 0;
    int    nr = 0;
    for (int i = 0; i < n; ++i) {
        const float* p = &source[4 * static_cast<size_t>(i)];
        float x, y, z, sqd;
        transform_point_f(final_, p[0], p[1], p[2], x, y, z);
        if (grid.nearest(x, y, z, sqd) < 0) continue;
        if (static_cast<double>(sqd) <= max_range) { sum += sqd; ++nr; }
    }
    return nr ? sum / nr : std::numeric_limits<double>::max();
}

}  // namespace orc
