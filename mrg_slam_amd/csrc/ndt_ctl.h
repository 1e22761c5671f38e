// csrc/ndt_ctl.h — the optimiser of NDT_HIP as ONE piece of host/device code: pclomp::NormalDistributionsTransform::
// computeTransformation + computeStepLengthMT + trialValueSelectionMT + updateIntervalMT + computeAngleDerivatives
// (reached through registration_->align(): /root/reference/apps/scan_matching_odometry_component.cpp:
// 265-266, src/mrg_slam/loop_detector.cpp:134) unrolled into a resumable state machine over a plain struct.
//
// Every computeDerivatives / computeHessian call of the reference is one REQUEST (an NdtEvalDev record read by the derivative
// kernels); the machine resumes when the 44 reduced sums of that evaluation are known (ctl::on_result).  The same source runs
//   * on the DEVICE inside ndt_reduce_control_kernel (ndt_derivatives.hip): a batch advances round after round without the
//     host — reduce -> controller step -> next request, all in HBM;
//   * on the HOST behind class NdtController (ndt_controller.cpp), the host-stepped path kept for single registrations and as
//     the A/B check of the device path (MRGFE_HOST_CONTROL).
// All arithmetic is f64 (or the reference's f32 where it uses f32) in the written order: compiled with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ndt_types.h"

#define MRGFE_HD __host__ __device__ inline

namespace mrgfe {

enum NdtPhase : int32_t { NDT_IDLE = 0, NDT_INIT = 1, NDT_LS_FIRST = 2, NDT_LS_ITER = 3, NDT_LS_HESS = 4, NDT_DONE = 5, NDT_LS_FIRST_HESS = 6 };

// Complete state of one alignment (device-resident during a batch; ~1.4 KB)
struct NdtCtlState {
    // parameters
    double   step_size, trans_eps, outlier_ratio;
    float    resolution;
    int32_t  max_iterations, search, reuse;
    // status
    int32_t  phase;
    uint32_t n_src;
    int32_t  converged, nr_iterations, n_evals, n_reused, cache_valid;
    int32_t  split_first;  // 1: the first trial of a line search is evaluated without its Hessian, which is fetched afterwards if used (below)
    int32_t  svd_only;     // 2: every Newton solve through Eigen's two-sided JacobiSVD restated (reference-order mode, host-stepped); 1: every Newton solve through the Jacobi SVD (MRGFE_NEWTON_SVD=1: round 4's solve, kept to hold the LU fast path against)
    int32_t  formulation;  // 0: pclomp::NormalDistributionsTransform (NDT_OMP); 1: pcl::NormalDistributionsTransform of PCL 1.12 ("NDT",
                           // registrations.cpp:115-129): f64 pair terms (the kernels), PCL's iteration test and zero-step rule (below)
    float    final_[16], transformation_[16], previous_[16];  // row-major
    double   cache_p[6], cache_nb;
    double   trans_probability, nb_sum;
    double   gauss_d1, gauss_d2, gauss_d3;
    double   p[6], score, g[6], H[36];
    // More-Thuente
    double   x[6], x_t[6], dir[6];
    double   phi_0, d_phi_0, a_l, f_l, g_l, a_u, f_u, g_u, a_t, phi_t, d_phi_t, psi_t, d_psi_t;
    int32_t  interval_converged, open_interval, step_iterations;
    // pending request
    int32_t  req_mode;
    double   req_p[6];
    // accounting of the evaluations that were launched for this pair, per kernel variant (SURVEY.md §8d byte model)
    double   acct_points[3], acct_nb[3];
};

namespace ctl {

constexpr double kMu = 1.e-4, kNu = 0.9;
constexpr int    kMaxStepIterations = 10;

// std::min / std::max exactly — (b < a) ? b : a and (a < b) ? b : a — because a NaN trial step (the cubic of
// trialValueSelectionMT takes the square root of a negative number now and then) must propagate the way it does in the reference
MRGFE_HD double dmin(double a, double b) { return (b < a) ? b : a; }
MRGFE_HD double dmax(double a, double b) { return (a < b) ? b : a; }
MRGFE_HD bool   finite_d(double v) { return (v - v) == 0.0; }

// ---- float sine / cosine of Eigen::AngleAxisf ------------------------------------------------------------------------
// The reference calls std::sin / std::cos on floats, i.e. the C library's sinf / cosf, and those are NOT correctly rounded
// (glibc >= 2.28 documents 0.56 ULP): rounding a double-precision sine to float differs from glibc in ~1 % of the arguments,
// which moved 6 % of the pose matrices by an ulp when the device did that.  So both builds of this file evaluate glibc's own
// algorithm (the ARM optimized-routines sincosf: double-precision minimax polynomials on [-pi/4, pi/4] after a reduction by
// multiples of pi/2, written here from its published description): tests/test_controller_cpu.py compares it with the C
// library's sinf / cosf on every float with 2^-31 <= |x| < 120 of a strided sweep (identical on glibc 2.35 / x86-64 with FMA:
// the reduction x - n * (pi/2) is fused there).  |x| >= 120, NaN and infinities take the plain double routine.
struct SinCosPoly { double c0, c1, c2, c3, c4, s1, s2, s3; };
MRGFE_HD uint32_t f_abstop12(float x) { return (__builtin_bit_cast(uint32_t, x) >> 20) & 0x7ffu; }
MRGFE_HD float sincosf_poly(double x, double x2, bool negate_cos, int n)
{
    const double S1 = -0x1.555545995a603p-3, S2 = 0x1.1107605230bc4p-7, S3 = -0x1.994eb3774cf24p-13;
    const double sg = negate_cos ? -1.0 : 1.0;
    const double C0 = sg * 0x1p0, C1 = sg * -0x1.ffffffd0c621cp-2, C2 = sg * 0x1.55553e1068f19p-5, C3 = sg * -0x1.6c087e89a359dp-10, C4 = sg * 0x1.99343027bf8c3p-16;
    if ((n & 1) == 0) {
        const double x3 = x * x2;
        const double s1 = S2 + x2 * S3;
        const double x7 = x3 * x2;
        const double s = x + x3 * S1;
        return static_cast<float>(s + x7 * s1);
    }
    const double x4 = x2 * x2;
    const double c2 = C3 + x2 * C4;
    const double c1 = C0 + x2 * C1;
    const double x6 = x4 * x2;
    const double c = c1 + x4 * C2;
    return static_cast<float>(c + x6 * c2);
}
MRGFE_HD double sincosf_reduce(double x, int* np)
{
    const double hpi_inv = 0x1.45F306DC9C883p+23, hpi = 0x1.921FB54442D18p0;
    const double r = x * hpi_inv;
    const int    n = (static_cast<int32_t>(r) + 0x800000) >> 24;
    *np = n;
    return __builtin_fma(-static_cast<double>(n), hpi, x);
}
MRGFE_HD float sin_f(float y)
{
    double x = y;
    if (f_abstop12(y) < f_abstop12(0x1.921FB6p-1f)) {
        if (f_abstop12(y) < f_abstop12(0x1p-12f)) return y;
        return sincosf_poly(x, x * x, false, 0);
    }
    if (f_abstop12(y) < f_abstop12(120.0f)) {
        int n;
        x = sincosf_reduce(x, &n);
        const double sign = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;  // {1, -1, -1, 1}[n & 3]
        return sincosf_poly(x * sign, x * x, (n & 2) != 0, n);
    }
    return static_cast<float>(sin(x));
}
MRGFE_HD float cos_f(float y)
{
    double x = y;
    if (f_abstop12(y) < f_abstop12(0x1.921FB6p-1f)) {
        if (f_abstop12(y) < f_abstop12(0x1p-12f)) return 1.0f;
        return sincosf_poly(x, x * x, false, 1);
    }
    if (f_abstop12(y) < f_abstop12(120.0f)) {
        int n;
        x = sincosf_reduce(x, &n);
        const int m = n + 1;
        const double sign = ((m & 3) == 1 || (m & 3) == 2) ? -1.0 : 1.0;
        return sincosf_poly(x * sign, x * x, (m & 2) != 0, n ^ 1);
    }
    return static_cast<float>(cos(x));
}

// ---- 6x6 solve through a one-sided (Hestenes) Jacobi SVD: x = V diag(1/s) U^T b over singular values above
// 6*eps*s_max, i.e. the minimum-norm solution Eigen::JacobiSVD<Matrix6d>::solve returns (no PD fix-up). -----------
// Column pairs are visited in round-robin order: five steps of three DISJOINT pairs per sweep.  The rotations of a step touch
// different columns, so executing them one after the other (this function: host, single lane) or side by side (svd_solve6_wave:
// three lanes compute the rotations, 36 lanes apply them) gives the same doubles; both are held against each other in
// tests/test_gpu_control.py.
struct SvdPair { int p, q; };
MRGFE_HD SvdPair svd_schedule(int step, int m)
{
    // (0,5)(1,4)(2,3) | (0,4)(3,5)(1,2) | (0,3)(2,4)(1,5) | (0,2)(1,3)(4,5) | (0,1)(2,5)(3,4): every pair once per sweep
    const int P[5][3] = {{0, 1, 2}, {0, 3, 1}, {0, 2, 1}, {0, 1, 4}, {0, 2, 3}};
    const int Q[5][3] = {{5, 4, 3}, {4, 5, 2}, {3, 4, 5}, {2, 3, 5}, {1, 5, 4}};
    return SvdPair{P[step][m], Q[step][m]};
}
// rotation that orthogonalises two columns with squared norms alpha, beta and inner product gamma; false: leave them
MRGFE_HD bool svd_rotation(double alpha, double beta, double gamma, double* c, double* sn)
{
    if (gamma == 0.0 || fabs(gamma) <= 1e-15 * sqrt(alpha * beta)) return false;
    const double zeta = (beta - alpha) / (2.0 * gamma);
    const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
    *c = 1.0 / sqrt(1.0 + t * t);
    *sn = *c * t;
    return true;
}
MRGFE_HD void svd_solve6(const double A[36], const double b[6], double x[6])
{
    const double kNaN = __builtin_nan("");
    for (int i = 0; i < 36; ++i)
        if (!finite_d(A[i])) { for (int k = 0; k < 6; ++k) x[k] = kNaN; return; }
    double U[6][6], V[6][6];
    double scale = 0;
    for (int i = 0; i < 36; ++i) scale = dmax(scale, fabs(A[i]));
    if (scale == 0) { for (int k = 0; k < 6; ++k) x[k] = 0; return; }
    for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) { U[r][c] = A[r * 6 + c] / scale; V[r][c] = r == c ? 1.0 : 0.0; }
    for (int sweep = 0; sweep < 60; ++sweep) {
        bool rotated = false;
        for (int step = 0; step < 5; ++step)
            for (int m = 0; m < 3; ++m) {
                const SvdPair pq = svd_schedule(step, m);
                const int p = pq.p, q = pq.q;
                double alpha = 0, beta = 0, gamma = 0;
                for (int k = 0; k < 6; ++k) { alpha += U[k][p] * U[k][p]; beta += U[k][q] * U[k][q]; gamma += U[k][p] * U[k][q]; }
                double c, sn;
                if (!svd_rotation(alpha, beta, gamma, &c, &sn)) continue;
                rotated = true;
                for (int k = 0; k < 6; ++k) {
                    const double up = U[k][p], uq = U[k][q];
                    U[k][p] = c * up - sn * uq; U[k][q] = sn * up + c * uq;
                    const double vp = V[k][p], vq = V[k][q];
                    V[k][p] = c * vp - sn * vq; V[k][q] = sn * vp + c * vq;
                }
            }
        if (!rotated) break;
    }
    double sig[6], smax = 0;
    for (int j = 0; j < 6; ++j) {
        double n2 = 0;
        for (int k = 0; k < 6; ++k) n2 += U[k][j] * U[k][j];
        sig[j] = sqrt(n2);
        smax = dmax(smax, sig[j]);
    }
    const double thr = dmax(smax * 6.0 * 2.2204460492503131e-16, 2.2250738585072014e-308 / scale);
    for (int k = 0; k < 6; ++k) x[k] = 0;
    for (int j = 0; j < 6; ++j) {
        if (!(sig[j] > thr)) continue;
        double ub = 0;
        for (int k = 0; k < 6; ++k) ub += U[k][j] * b[k];
        const double coef = ub / (sig[j] * sig[j] * scale);  // U[:,j] is sig_j * u_j
        for (int k = 0; k < 6; ++k) x[k] += V[k][j] * coef;
    }
}

// ---- the REFERENCE's solve, operation for operation (reference-order mode only) --------------------------------------------------------------
// Eigen::JacobiSVD<Matrix6d>(H, ComputeFullU | ComputeFullV).solve(-g) as Eigen 3.4 runs it for a square real matrix (no QR preconditioner): scale by
// the largest |entry|; sweeps over the pairs p = 1..5, q = 0..p-1; for a pair whose off-diagonal entries exceed max(DBL_MIN, 2 eps maxDiagEntry) the
// 2 x 2 block is diagonalised by real_2x2_jacobi_svd (a rotation that symmetrises it, then makeJacobi) and the rotations are applied to the working
// matrix, U and V; singular values |diag|, signs moved into U, sorted descending; solve = V diag(1 / s_i, i < rank) U^T b with rank from
// s_i > max(6 eps s_0, DBL_MIN).  The one-sided Hestenes SVD above gives the same solution to ~1e-16 — but the optimiser's pose vector then differs
// from the reference's in its last bits, and an optimisation that runs to the iteration limit without settling amplifies that like it amplifies
// summation order (2 of 160 random scenes left the bar with every SUM already in the reference's order).  With this solve the whole double trajectory
// is the reference-order oracle's.  ~4 us on the host; never used on the device.
MRGFE_HD void jacobi2_rot_rows(double M[36], int p, int q, double c, double s)  // applyOnTheLeft(p, q, j): row_p' = c row_p + s row_q, row_q' = -s row_p + c row_q
{
    for (int k = 0; k < 6; ++k) {
        const double xp = M[p * 6 + k], xq = M[q * 6 + k];
        M[p * 6 + k] = c * xp + s * xq;
        M[q * 6 + k] = -s * xp + c * xq;
    }
}
MRGFE_HD void jacobi2_rot_cols(double M[36], int p, int q, double c, double s)  // applyOnTheRight(p, q, j): col_p' = c col_p - s col_q, col_q' = s col_p + c col_q
{
    for (int k = 0; k < 6; ++k) {
        const double xp = M[k * 6 + p], xq = M[k * 6 + q];
        M[k * 6 + p] = c * xp - s * xq;
        M[k * 6 + q] = s * xp + c * xq;
    }
}
MRGFE_HD void jacobi2_solve6(const double A[36], const double b[6], double x[6])
{
    const double kNaN = __builtin_nan(""), kMin = 2.2250738585072014e-308, kEps = 2.2204460492503131e-16;
    double scale = 0;
    bool   fin = true;
    for (int i = 0; i < 36; ++i) { scale = dmax(scale, fabs(A[i])); fin = fin && finite_d(A[i]); }
    if (!fin) { for (int k = 0; k < 6; ++k) x[k] = kNaN; return; }  // Eigen: m_info = InvalidInput
    if (scale == 0) scale = 1;
    double W[36], U[36], V[36], S[6];
    for (int i = 0; i < 36; ++i) { W[i] = A[i] / scale; U[i] = V[i] = (i % 7 == 0) ? 1.0 : 0.0; }
    const double precision = 2.0 * kEps;
    double max_diag = 0;
    for (int i = 0; i < 6; ++i) max_diag = dmax(max_diag, fabs(W[i * 7]));
    bool finished = false;
    for (int guard = 0; !finished && guard < 200; ++guard) {
        finished = true;
        for (int p = 1; p < 6; ++p)
            for (int q = 0; q < p; ++q) {
                const double thr = dmax(kMin, precision * max_diag);
                if (!(fabs(W[p * 6 + q]) > thr || fabs(W[q * 6 + p]) > thr)) continue;
                finished = false;
                // real_2x2_jacobi_svd of [[W(p,p) W(p,q)], [W(q,p) W(q,q)]]
                const double m00 = W[p * 6 + p], m01 = W[p * 6 + q], m10 = W[q * 6 + p], m11 = W[q * 6 + q];
                const double t = m00 + m11, d = m10 - m01;
                double c1 = 1, s1 = 0;
                if (!(fabs(d) < kMin)) { const double u = t / d, h = sqrt(1.0 + u * u); s1 = 1.0 / h; c1 = u / h; }
                const double n00 = c1 * m00 + s1 * m10, n01 = c1 * m01 + s1 * m11, n11 = -s1 * m01 + c1 * m11;  // the block after the symmetrising rotation
                double cr = 1, sr = 0;  // makeJacobi(n00, n01, n11)
                const double deno = 2.0 * fabs(n01);
                if (!(deno < kMin)) {
                    const double tau = (n00 - n11) / deno, w = sqrt(tau * tau + 1.0);
                    const double tt = (tau > 0) ? 1.0 / (tau + w) : 1.0 / (tau - w);
                    const double sign_t = tt > 0 ? 1.0 : -1.0, nn = 1.0 / sqrt(tt * tt + 1.0);
                    sr = -sign_t * (n01 / fabs(n01)) * fabs(tt) * nn;
                    cr = nn;
                }
                const double cl = c1 * cr + s1 * sr, sl = s1 * cr - c1 * sr;  // j_left = rot1 * j_right^T
                jacobi2_rot_rows(W, p, q, cl, sl);
                jacobi2_rot_cols(U, p, q, cl, -sl);
                jacobi2_rot_cols(W, p, q, cr, sr);
                jacobi2_rot_cols(V, p, q, cr, sr);
                max_diag = dmax(max_diag, dmax(fabs(W[p * 7]), fabs(W[q * 7])));
            }
    }
    for (int i = 0; i < 6; ++i) {
        const double a = W[i * 7];
        S[i] = fabs(a);
        if (a < 0) for (int k = 0; k < 6; ++k) U[k * 6 + i] = -U[k * 6 + i];
    }
    for (int i = 0; i < 6; ++i) {  // descending; columns of U and V follow
        int best = i;
        for (int k = i + 1; k < 6; ++k) if (S[k] > S[best]) best = k;
        if (S[best] == 0) break;
        if (best != i) {
            const double ts = S[i]; S[i] = S[best]; S[best] = ts;
            for (int k = 0; k < 6; ++k) {
                const double tu = U[k * 6 + i]; U[k * 6 + i] = U[k * 6 + best]; U[k * 6 + best] = tu;
                const double tv = V[k * 6 + i]; V[k * 6 + i] = V[k * 6 + best]; V[k * 6 + best] = tv;
            }
        }
    }
    for (int i = 0; i < 6; ++i) S[i] *= scale;
    const double rthr = dmax(S[0] * 6.0 * kEps, kMin);
    int rank = 0;
    for (int i = 0; i < 6; ++i) if (S[i] > rthr) ++rank;
    double tmp[6];
    for (int i = 0; i < 6; ++i) {
        double acc = 0;
        for (int k = 0; k < 6; ++k) acc += U[k * 6 + i] * b[k];
        tmp[i] = (i < rank) ? acc / S[i] : 0.0;
    }
    for (int r = 0; r < 6; ++r) {
        double acc = 0;
        for (int i = 0; i < rank; ++i) acc += V[r * 6 + i] * tmp[i];
        x[r] = acc;
    }
}

// ---- the fast path of the Newton solve ------------------------------------------------------------------------------------------
// JacobiSVD(H).solve(-g) is the minimum-norm solution over the singular values above 6 eps s_max.  For a matrix that is far from
// rank-deficient that IS H^-1 (-g), and an LU factorisation with partial pivoting delivers it with the same forward error (cond * eps)
// in ~150 dependent f64 operations instead of the ~8 sweeps x 15 rotations — each a chain of three divisions and three square roots —
// of the Jacobi SVD, which was two thirds of a controller step on the device (profiles/r05: 29 us per step, the step sits on every round's
// critical path).  The factorisation tells how far from singular the matrix is: the solve is accepted when the smallest pivot is more
// than kLuMinPivotRatio of the largest (cond(U) below ~1e6 for the 6 x 6 Hessians seen here, typically 1e3-1e4); everything else —
// a vanishing score (H = 0), a degenerate scene, non-finite sums — takes the SVD as before.  One source for host and device.
constexpr double kLuMinPivotRatio = 1e-6;
MRGFE_HD bool lu_solve6(const double A[36], const double b[6], double x[6])
{
    // every index below is a compile-time constant once the loops are unrolled, and the row exchange is a select per element: the 6 x 7 tableau
    // lives in registers on the device (with a run-time row index it sat in scratch memory: 9 us per solve, a third of that now)
    double M[6][7];
    bool   fin = true;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
#pragma unroll
        for (int c = 0; c < 6; ++c) { M[r][c] = A[r * 6 + c]; fin = fin && finite_d(M[r][c]); }
        M[r][6] = b[r];
        fin = fin && finite_d(b[r]);
    }
    if (!fin) return false;
    double pmax = 0, pmin = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        int    piv = k;
        double best = fabs(M[k][k]);
#pragma unroll
        for (int r = k + 1; r < 6; ++r) { const double v = fabs(M[r][k]); const bool gt = v > best; best = gt ? v : best; piv = gt ? r : piv; }
        if (!(best > 0)) return false;
#pragma unroll
        for (int r = k + 1; r < 6; ++r) {
            const bool sw = piv == r;
#pragma unroll
            for (int c = k; c < 7; ++c) { const double u = M[k][c], v = M[r][c]; M[k][c] = sw ? v : u; M[r][c] = sw ? u : v; }
        }
        pmax = k == 0 ? best : dmax(pmax, best);
        pmin = k == 0 ? best : dmin(pmin, best);
        const double inv = 1.0 / M[k][k];
#pragma unroll
        for (int r = k + 1; r < 6; ++r) {
            const double f = M[r][k] * inv;
#pragma unroll
            for (int c = k + 1; c < 7; ++c) M[r][c] = M[r][c] - f * M[k][c];
        }
    }
    if (!(pmin > kLuMinPivotRatio * pmax)) return false;
    double y[6];
#pragma unroll
    for (int r = 5; r >= 0; --r) {
        double acc = M[r][6];
#pragma unroll
        for (int c = r + 1; c < 6; ++c) acc = acc - M[r][c] * y[c];
        y[r] = acc / M[r][r];
    }
    fin = true;
#pragma unroll
    for (int r = 0; r < 6; ++r) fin = fin && finite_d(y[r]);
    if (!fin) return false;
#pragma unroll
    for (int r = 0; r < 6; ++r) x[r] = y[r];
    return true;
}
// MRGFE_NEWTON_SVD=1 (host: environment; device: the flag rides in the state): always the SVD, the solve of round 4
MRGFE_HD void newton_solve6(const double A[36], const double b[6], double x[6], int svd_only)
{
    if (svd_only == 2) { jacobi2_solve6(A, b, x); return; }  // reference-order mode: Eigen's two-sided JacobiSVD, operation for operation
    if (!svd_only && lu_solve6(A, b, x)) return;
    svd_solve6(A, b, x);
}

#if defined(__HIPCC__)
// The same solve by the 64 lanes of one wavefront (all of them must call it; A, b, x may be LDS or global; `w` is LDS scratch).
struct SvdWaveScratch { double U[6][6], V[6][6], c[3], s[3], sig[6], coef[6]; int rot[3], bad; };
__device__ inline void svd_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ inline void svd_solve6_wave(const double* A, const double* b, double* x, SvdWaveScratch& w)
{
    const int lane = threadIdx.x & 63;
    const double kNaN = __builtin_nan("");
    double a = 0;
    bool   fin = true;
    if (lane < 36) { a = A[lane]; fin = finite_d(a); }
    const bool all_finite = __ballot(!fin) == 0;
    double scale = fabs(a);  // max over the 36 entries (lanes >= 36 hold 0): exact, whatever the order
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) scale = dmax(scale, __shfl_xor(scale, off, 64));
    if (!all_finite) { if (lane < 6) x[lane] = kNaN; svd_wave_sync(); return; }
    if (scale == 0) { if (lane < 6) x[lane] = 0; svd_wave_sync(); return; }
    if (lane < 36) { w.U[lane / 6][lane % 6] = a / scale; w.V[lane / 6][lane % 6] = (lane / 6 == lane % 6) ? 1.0 : 0.0; }
    svd_wave_sync();
    for (int sweep = 0; sweep < 60; ++sweep) {
        bool rotated = false;
        for (int step = 0; step < 5; ++step) {
            if (lane < 3) {
                const SvdPair pq = svd_schedule(step, lane);
                double alpha = 0, beta = 0, gamma = 0;
                for (int k = 0; k < 6; ++k) { const double up = w.U[k][pq.p], uq = w.U[k][pq.q]; alpha += up * up; beta += uq * uq; gamma += up * uq; }
                double c = 1, sn = 0;
                w.rot[lane] = svd_rotation(alpha, beta, gamma, &c, &sn) ? 1 : 0;
                w.c[lane] = c; w.s[lane] = sn;
            }
            svd_wave_sync();
            if (lane < 36) {
                const int m = lane / 12, rem = lane % 12, k = rem % 6;
                if (w.rot[m]) {
                    const SvdPair pq = svd_schedule(step, m);
                    double (*M)[6] = rem < 6 ? w.U : w.V;
                    const double c = w.c[m], sn = w.s[m], up = M[k][pq.p], uq = M[k][pq.q];
                    M[k][pq.p] = c * up - sn * uq; M[k][pq.q] = sn * up + c * uq;
                }
            }
            rotated = rotated || (w.rot[0] | w.rot[1] | w.rot[2]);
            svd_wave_sync();
        }
        if (!rotated) break;
    }
    double my_sig = 0;
    if (lane < 6) {
        double n2 = 0;
        for (int k = 0; k < 6; ++k) n2 += w.U[k][lane] * w.U[k][lane];
        my_sig = sqrt(n2);
        w.sig[lane] = my_sig;
    }
    double smax = my_sig;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) smax = dmax(smax, __shfl_xor(smax, off, 64));
    const double thr = dmax(smax * 6.0 * 2.2204460492503131e-16, 2.2250738585072014e-308 / scale);
    if (lane < 6) {
        double ub = 0;
        for (int k = 0; k < 6; ++k) ub += w.U[k][lane] * b[k];
        w.coef[lane] = ub / (my_sig * my_sig * scale);
    }
    svd_wave_sync();
    if (lane < 6) {
        double acc = 0;
        for (int j = 0; j < 6; ++j)
            if (w.sig[j] > thr) acc += w.V[lane][j] * w.coef[j];
        x[lane] = acc;
    }
    svd_wave_sync();
}
#endif

MRGFE_HD double psi_mt(double a, double f_a, double f_0, double g_0, double mu) { return f_a - f_0 - mu * g_0 * a; }
MRGFE_HD double dpsi_mt(double g_a, double g_0, double mu) { return g_a - mu * g_0; }

MRGFE_HD bool update_interval(double& a_l, double& f_l, double& g_l, double& a_u, double& f_u, double& g_u, double a_t, double f_t, double g_t)
{
    if (f_t > f_l) { a_u = a_t; f_u = f_t; g_u = g_t; return false; }
    if (g_t * (a_l - a_t) > 0) { a_l = a_t; f_l = f_t; g_l = g_t; return false; }
    if (g_t * (a_l - a_t) < 0) { a_u = a_l; f_u = f_l; g_u = g_l; a_l = a_t; f_l = f_t; g_l = g_t; return false; }
    return true;
}

// minimiser of the cubic through (a0,f0,g0) and (a1,f1,g1), Sun & Yuan eq. 2.4.52/2.4.56
MRGFE_HD double cubic_min(double a0, double f0, double g0, double a1, double f1, double g1)
{
    const double z = 3 * (f1 - f0) / (a1 - a0) - g1 - g0;
    const double w = sqrt(z * z - g1 * g0);
    return a0 + (a1 - a0) * (w - g0 - z) / (g1 - g0 + 2 * w);
}

MRGFE_HD double trial_value(double a_l, double f_l, double g_l, double a_u, double f_u, double g_u, double a_t, double f_t, double g_t)
{
    if (f_t > f_l) {
        const double a_c = cubic_min(a_l, f_l, g_l, a_t, f_t, g_t);
        const double a_q = a_l - 0.5 * (a_l - a_t) * g_l / (g_l - (f_l - f_t) / (a_l - a_t));
        return (fabs(a_c - a_l) < fabs(a_q - a_l)) ? a_c : 0.5 * (a_q + a_c);
    }
    if (g_t * g_l < 0) {
        const double a_c = cubic_min(a_l, f_l, g_l, a_t, f_t, g_t);
        const double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
        return (fabs(a_c - a_t) >= fabs(a_s - a_t)) ? a_c : a_s;
    }
    if (fabs(g_t) <= fabs(g_l)) {
        const double a_c = cubic_min(a_l, f_l, g_l, a_t, f_t, g_t);
        const double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
        const double a_next = (fabs(a_c - a_t) < fabs(a_s - a_t)) ? a_c : a_s;
        return (a_t > a_l) ? dmin(a_t + 0.66 * (a_u - a_t), a_next) : dmax(a_t + 0.66 * (a_u - a_t), a_next);
    }
    return cubic_min(a_u, f_u, g_u, a_t, f_t, g_t);
}

MRGFE_HD void identity16(float M[16]) { for (int i = 0; i < 16; ++i) M[i] = (i % 5 == 0) ? 1.0f : 0.0f; }
MRGFE_HD bool is_identity16(const float M[16])
{
    for (int i = 0; i < 16; ++i) if (M[i] != ((i % 5 == 0) ? 1.0f : 0.0f)) return false;
    return true;
}

// Eigen::AngleAxisf(angle, e_axis).toRotationMatrix()
MRGFE_HD void axis_rotation(float angle, int axis, float R[9])
{
    float ax[3] = {0, 0, 0};
    ax[axis] = 1.0f;
    const float sn = sin_f(angle), c = cos_f(angle);
    const float sa[3] = {sn * ax[0], sn * ax[1], sn * ax[2]};
    const float ca[3] = {(1.0f - c) * ax[0], (1.0f - c) * ax[1], (1.0f - c) * ax[2]};
    float tmp;
    tmp = ca[0] * ax[1]; R[1] = tmp - sa[2]; R[3] = tmp + sa[2];
    tmp = ca[0] * ax[2]; R[2] = tmp + sa[1]; R[6] = tmp - sa[1];
    tmp = ca[1] * ax[2]; R[5] = tmp - sa[0]; R[7] = tmp + sa[0];
    R[0] = ca[0] * ax[0] + c; R[4] = ca[1] * ax[1] + c; R[8] = ca[2] * ax[2] + c;
}
MRGFE_HD void matmul3f(const float a[9], const float b[9], float o[9])
{
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            const float p0 = a[r * 3] * b[c], p1 = a[r * 3 + 1] * b[3 + c], p2 = a[r * 3 + 2] * b[6 + c];
            const float s = p0 + p1;
            o[r * 3 + c] = s + p2;
        }
}

// Translation(p0..2) * AngleAxis(p3, X) * AngleAxis(p4, Y) * AngleAxis(p5, Z) as a float matrix (row-major)
MRGFE_HD void pose_to_matrix(const double p[6], float M[16])
{
    float Rx[9], Ry[9], Rz[9], Rxy[9], R[9];
    axis_rotation(static_cast<float>(p[3]), 0, Rx);
    axis_rotation(static_cast<float>(p[4]), 1, Ry);
    axis_rotation(static_cast<float>(p[5]), 2, Rz);
    matmul3f(Rx, Ry, Rxy);
    matmul3f(Rxy, Rz, R);
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) M[r * 4 + c] = R[r * 3 + c];
        M[r * 4 + 3] = static_cast<float>(p[r]);
    }
    M[12] = M[13] = M[14] = 0.0f;
    M[15] = 1.0f;
}

// computeAngleDerivatives: rows a..h of j_ang and a2..f3 of h_ang
// cs = {cx, sx, cy, sy, cz, sz}: cosine / sine of the three angles with upstream's small-angle rule (|a| < 10e-5: cos 1, sin 0)
MRGFE_HD double angle_trig(const double p[6], int which)
{
    const double a = p[3 + which / 2];
    if (fabs(a) < 10e-5) return (which & 1) ? 0.0 : 1.0;
    return (which & 1) ? sin(a) : cos(a);
}
MRGFE_HD void angle_tables_from(const double cs[6], double j[8][3], double h[15][3]);
MRGFE_HD void angle_tables(const double p[6], double j[8][3], double h[15][3])
{
    double cs[6];
    for (int k = 0; k < 6; ++k) cs[k] = angle_trig(p, k);
    angle_tables_from(cs, j, h);
}
MRGFE_HD void angle_tables_from(const double cs[6], double j[8][3], double h[15][3])
{
    const double cx = cs[0], sx = cs[1], cy = cs[2], sy = cs[3], cz = cs[4], sz = cs[5];
    j[0][0] = (-sx * sz + cx * sy * cz); j[0][1] = (-sx * cz - cx * sy * sz); j[0][2] = (-cx * cy);
    j[1][0] = (cx * sz + sx * sy * cz);  j[1][1] = (cx * cz - sx * sy * sz);  j[1][2] = (-sx * cy);
    j[2][0] = (-sy * cz);                j[2][1] = sy * sz;                   j[2][2] = cy;
    j[3][0] = sx * cy * cz;              j[3][1] = (-sx * cy * sz);           j[3][2] = sx * sy;
    j[4][0] = (-cx * cy * cz);           j[4][1] = cx * cy * sz;              j[4][2] = (-cx * sy);
    j[5][0] = (-cy * sz);                j[5][1] = (-cy * cz);                j[5][2] = 0;
    j[6][0] = (cx * cz - sx * sy * sz);  j[6][1] = (-cx * sz - sx * sy * cz); j[6][2] = 0;
    j[7][0] = (sx * cz + cx * sy * sz);  j[7][1] = (cx * sy * cz - sx * sz);  j[7][2] = 0;
    h[0][0] = (-cx * sz - sx * sy * cz);  h[0][1] = (-cx * cz + sx * sy * sz);  h[0][2] = sx * cy;
    h[1][0] = (-sx * sz + cx * sy * cz);  h[1][1] = (-cx * sy * sz - sx * cz);  h[1][2] = (-cx * cy);
    h[2][0] = (cx * cy * cz);             h[2][1] = (-cx * cy * sz);            h[2][2] = (cx * sy);
    h[3][0] = (sx * cy * cz);             h[3][1] = (-sx * cy * sz);            h[3][2] = (sx * sy);
    h[4][0] = (-sx * cz - cx * sy * sz);  h[4][1] = (sx * sz - cx * sy * cz);   h[4][2] = 0;
    h[5][0] = (cx * cz - sx * sy * sz);   h[5][1] = (-sx * sy * cz - cx * sz);  h[5][2] = 0;
    h[6][0] = (-cy * cz);                 h[6][1] = (cy * sz);                  h[6][2] = (sy);
    h[7][0] = (-sx * sy * cz);            h[7][1] = (sx * sy * sz);             h[7][2] = (sx * cy);
    h[8][0] = (cx * sy * cz);             h[8][1] = (-cx * sy * sz);            h[8][2] = (-cx * cy);
    h[9][0] = (sy * sz);                  h[9][1] = (sy * cz);                  h[9][2] = 0;
    h[10][0] = (-sx * cy * sz);           h[10][1] = (-sx * cy * cz);           h[10][2] = 0;
    h[11][0] = (cx * cy * sz);            h[11][1] = (cx * cy * cz);            h[11][2] = 0;
    h[12][0] = (-cy * cz);                h[12][1] = (cy * sz);                 h[12][2] = 0;
    h[13][0] = (-cx * sz - sx * sy * cz); h[13][1] = (-cx * cz + sx * sy * sz); h[13][2] = 0;
    h[14][0] = (-sx * sz + cx * sy * cz); h[14][1] = (-cx * sy * sz - sx * cz); h[14][2] = 0;
}

MRGFE_HD bool done(const NdtCtlState& s) { return s.phase == NDT_DONE || s.phase == NDT_IDLE; }

MRGFE_HD void make_request(NdtCtlState& s, int mode, const double p[6])
{
    s.req_mode = mode;
    for (int k = 0; k < 6; ++k) s.req_p[k] = p[k];
    ++s.n_evals;
}

// the record the derivative kernels read for the pending request of `s` (its transform is final_ as of the request)
MRGFE_HD void fill_eval_from(const NdtCtlState& s, const double cs[6], NdtEvalDev& e)
{
    if (done(s)) { e.active = 0; return; }
    for (int k = 0; k < 12; ++k) e.T[k] = s.final_[k];
    double j[8][3], h[15][3];
    angle_tables_from(cs, j, h);
    for (int a = 0; a < 8; ++a) for (int b = 0; b < 3; ++b) { e.j_ang_d[a][b] = j[a][b]; e.j_ang[a][b] = static_cast<float>(j[a][b]); }
    for (int a = 0; a < 15; ++a) for (int b = 0; b < 3; ++b) { e.h_ang_d[a][b] = h[a][b]; e.h_ang[a][b] = static_cast<float>(h[a][b]); }
    e.gauss_d1 = s.gauss_d1;
    e.gauss_d2 = s.gauss_d2;
    e.mode = s.req_mode;
    e.active = 1;
    e.search = s.search;
}
MRGFE_HD void fill_eval(const NdtCtlState& s, NdtEvalDev& e)
{
    if (done(s)) { e.active = 0; return; }
    double cs[6];
    for (int k = 0; k < 6; ++k) cs[k] = angle_trig(s.req_p, k);
    fill_eval_from(s, cs, e);
}

MRGFE_HD void store_result(NdtCtlState& s, const double r[44], bool with_score_grad, bool with_hessian, bool counted = true)
{
    if (with_score_grad) {
        s.score = r[0];
        for (int k = 0; k < 6; ++k) s.g[k] = r[1 + k];
    }
    if (with_hessian)
        for (int k = 0; k < 36; ++k) s.H[k] = r[7 + k];
    const double nb = s.n_src ? r[kNdtNbIndex] / static_cast<double>(s.n_src) : 0.0;
    if (counted) s.nb_sum += nb;  // mean neighbours per evaluation of the reference's control flow
    s.acct_points[s.req_mode] += static_cast<double>(s.n_src);  // what was launched (byte model)
    s.acct_nb[s.req_mode] += r[kNdtNbIndex];
    // remember the pose of evaluations whose transform was built from the pose vector (line-search trials)
    if (with_score_grad && s.phase != NDT_INIT) {
        for (int k = 0; k < 6; ++k) s.cache_p[k] = s.req_p[k];
        s.cache_nb = nb;
        s.cache_valid = 1;
    }
}

MRGFE_HD void ls_after_eval(NdtCtlState& s)
{
    s.phi_t = -s.score;
    double d = 0;
    for (int k = 0; k < 6; ++k) d += s.g[k] * s.dir[k];
    s.d_phi_t = -d;
    s.psi_t = psi_mt(s.a_t, s.phi_t, s.phi_0, s.d_phi_0, kMu);
    s.d_psi_t = dpsi_mt(s.d_phi_t, s.d_phi_0, kMu);
}

// The state machine, resumable around the 6x6 solve so that the device can run that solve on a whole wavefront:
//   resume(s, r)        takes the reduced sums r[0..43] (score, gradient, 6x6 Hessian row-major, neighbour count) of the pending
//                       request and runs until a new request is pending / the alignment is done (CTL_RETURN) or the next
//                       Newton step needs  delta = JacobiSVD(H).solve(-g)  (CTL_NEED_SOLVE: solve s.H * delta = -s.g);
//   after_solve(s, d)   continues with that delta; may again ask for a solve (a zero-length step ends an iteration at once).
enum { CTL_RETURN = 0, CTL_NEED_SOLVE = 1 };
enum CtlNext { CTL_LS_UPDATE, CTL_LS_DECIDE, CTL_FINISH_LS };

// line-search flow from `next` on (FINISH_LS uses a_fin); returns CTL_RETURN or CTL_NEED_SOLVE
MRGFE_HD int ls_flow(NdtCtlState& s, CtlNext next, double a_fin)
{
    for (;;) {
        switch (next) {
            case CTL_LS_UPDATE: {
                // bookkeeping of one line-search trial once its score / gradient are known (body of the while loop of computeStepLengthMT)
                ls_after_eval(s);
                if (s.open_interval && (s.psi_t <= 0 && s.d_psi_t >= 0)) {
                    s.open_interval = 0;
                    s.f_l = s.f_l + s.phi_0 - kMu * s.d_phi_0 * s.a_l;
                    s.g_l = s.g_l + kMu * s.d_phi_0;
                    s.f_u = s.f_u + s.phi_0 - kMu * s.d_phi_0 * s.a_u;
                    s.g_u = s.g_u + kMu * s.d_phi_0;
                }
                if (s.open_interval) s.interval_converged = update_interval(s.a_l, s.f_l, s.g_l, s.a_u, s.f_u, s.g_u, s.a_t, s.psi_t, s.d_psi_t) ? 1 : 0;
                else                 s.interval_converged = update_interval(s.a_l, s.f_l, s.g_l, s.a_u, s.f_u, s.g_u, s.a_t, s.phi_t, s.d_phi_t) ? 1 : 0;
                s.step_iterations++;
                next = CTL_LS_DECIDE;
                break;
            }
            case CTL_LS_DECIDE: {
                if (!s.interval_converged && s.step_iterations < kMaxStepIterations && !(s.psi_t <= 0 && s.d_phi_t <= -kNu * s.d_phi_0)) {
                    if (s.open_interval) s.a_t = trial_value(s.a_l, s.f_l, s.g_l, s.a_u, s.f_u, s.g_u, s.a_t, s.psi_t, s.d_psi_t);
                    else                 s.a_t = trial_value(s.a_l, s.f_l, s.g_l, s.a_u, s.f_u, s.g_u, s.a_t, s.phi_t, s.d_phi_t);
                    s.a_t = dmin(s.a_t, s.step_size);
                    s.a_t = dmax(s.a_t, s.trans_eps / 2);
                    for (int k = 0; k < 6; ++k) s.x_t[k] = s.x[k] + s.dir[k] * s.a_t;
                    pose_to_matrix(s.x_t, s.final_);
                    bool same = s.reuse && s.cache_valid;  // bitwise: the same doubles give the same float transform
                    for (int k = 0; k < 6 && same; ++k) same = __builtin_bit_cast(uint64_t, s.x_t[k]) == __builtin_bit_cast(uint64_t, s.cache_p[k]);
                    if (same) {
                        // The clamped trial step often repeats the previous one (a_t pinned at step_min / step_max): same pose vector
                        // -> same float transform -> the evaluation would reproduce score / g bit for bit.  The reference
                        // recomputes it; here the held values are reused and no GPU round is spent (the evaluation is still counted).
                        ++s.n_evals;
                        ++s.n_reused;
                        s.nb_sum += s.cache_nb;
                        next = CTL_LS_UPDATE;
                        break;
                    }
                    make_request(s, 1, s.x_t);
                    s.phase = NDT_LS_ITER;
                    return CTL_RETURN;
                }
                if (s.step_iterations) {
                    make_request(s, 2, s.x_t);  // computeHessian at x_t (final_ already holds its matrix)
                    s.phase = NDT_LS_HESS;
                    return CTL_RETURN;
                }
                if (s.split_first) {
                    // The first trial was accepted, so the Hessian of ITS evaluation drives the next Newton step — and in split mode
                    // that evaluation ran without it.  The reference computes score, gradient and Hessian of the first trial in one
                    // pass (computeDerivatives(.., true)) and throws the Hessian away whenever the search goes on (computeHessian then
                    // recomputes it at the final step): 46 % of the iterations on the bench workload.  Here the trial costs a
                    // score+gradient pass (1/5 of the full one) and the full pass runs at the same pose only when its Hessian is used:
                    // same per-pair float terms, same sums up to the order of the f64 additions.  Not counted as an evaluation.
                    s.req_mode = 0;
                    for (int k = 0; k < 6; ++k) s.req_p[k] = s.x_t[k];
                    s.phase = NDT_LS_FIRST_HESS;
                    return CTL_RETURN;
                }
                a_fin = s.a_t;
                next = CTL_FINISH_LS;
                break;
            }
            case CTL_FINISH_LS: {
                // back in computeTransformation's loop body
                double delta_p[6];
                for (int k = 0; k < 6; ++k) delta_p[k] = s.dir[k] * a_fin;
                pose_to_matrix(delta_p, s.transformation_);
                for (int k = 0; k < 6; ++k) s.p[k] = s.p[k] + delta_p[k];
                if (s.formulation == 1) {
                    // PCL >= 1.11.1 (ndt.hpp computeTransformation): nr_iterations_++ first; then the step's float matrix against the epsilons —
                    // the SQUARED translation against the un-squared transformation_epsilon_ (squaredNorm of a 3-vector: Eigen adds e0 + (e1 + e2));
                    // transformation_rotation_epsilon_ stays 0 in the reference (registrations.cpp:125-127), which leaves
                    //   nr_iterations_ >= max_iterations_ || (transformation_epsilon_ > 0 && translation_sqr <= transformation_epsilon_)
                    const float tx = s.transformation_[3], ty = s.transformation_[7], tz = s.transformation_[11];
                    const float yz = ty * ty + tz * tz;
                    const double translation_sqr = static_cast<double>(tx * tx + yz);
                    s.nr_iterations++;
                    if (s.nr_iterations >= s.max_iterations || (s.trans_eps > 0 && translation_sqr <= s.trans_eps)) s.converged = 1;
                } else {
                    if (s.nr_iterations > s.max_iterations || (s.nr_iterations && (fabs(a_fin) < s.trans_eps))) s.converged = 1;
                    s.nr_iterations++;
                }
                if (s.converged) {
                    s.trans_probability = s.score / static_cast<double>(s.n_src);
                    s.phase = NDT_DONE;
                    return CTL_RETURN;
                }
                return CTL_NEED_SOLVE;  // next Newton iteration from the derivatives already held for p == x_t
            }
        }
    }
}

MRGFE_HD int resume(NdtCtlState& s, const double r[44])
{
    switch (s.phase) {
        case NDT_INIT:     store_result(s, r, true, true);  return CTL_NEED_SOLVE;
        case NDT_LS_FIRST: store_result(s, r, true, !s.split_first); ls_after_eval(s); return ls_flow(s, CTL_LS_DECIDE, 0.0);
        case NDT_LS_FIRST_HESS: store_result(s, r, false, true, false); return ls_flow(s, CTL_FINISH_LS, s.a_t);
        case NDT_LS_ITER:  store_result(s, r, true, false); return ls_flow(s, CTL_LS_UPDATE, 0.0);
        case NDT_LS_HESS:  store_result(s, r, false, true); return ls_flow(s, CTL_FINISH_LS, s.a_t);
        default: return CTL_RETURN;
    }
}

// computeTransformation loop body after  delta = JacobiSVD(H).solve(-g):  computeStepLengthMT's prologue and first trial
MRGFE_HD int after_solve(NdtCtlState& s, const double delta[6])
{
    for (int k = 0; k < 16; ++k) s.previous_[k] = s.transformation_[k];
    double n2 = 0;
    for (int k = 0; k < 6; ++k) n2 += delta[k] * delta[k];
    const double norm = sqrt(n2);
    if (norm == 0 || norm != norm) {
        s.trans_probability = s.score / static_cast<double>(s.n_src);
        s.converged = s.formulation == 1 ? (norm == 0 ? 1 : 0) : ((norm == norm) ? 1 : 0);  // PCL >= 1.11.1: converged_ = delta_norm == 0
        s.phase = NDT_DONE;
        return CTL_RETURN;
    }
    for (int k = 0; k < 6; ++k) s.dir[k] = delta[k] / norm;
    for (int k = 0; k < 6; ++k) s.x[k] = s.p[k];
    s.phi_0 = -s.score;
    double d = 0;
    for (int k = 0; k < 6; ++k) d += s.g[k] * s.dir[k];
    s.d_phi_0 = -d;
    if (s.d_phi_0 >= 0) {
        if (s.d_phi_0 == 0) return ls_flow(s, CTL_FINISH_LS, 0.0);  // "return 0": a zero-length step; the outer loop converges on its second pass (|0| < eps)
        s.d_phi_0 *= -1;
        for (int k = 0; k < 6; ++k) s.dir[k] *= -1;
    }
    s.step_iterations = 0;
    s.a_l = 0; s.a_u = 0;
    s.f_l = psi_mt(s.a_l, s.phi_0, s.phi_0, s.d_phi_0, kMu);
    s.g_l = dpsi_mt(s.d_phi_0, s.d_phi_0, kMu);
    s.f_u = psi_mt(s.a_u, s.phi_0, s.phi_0, s.d_phi_0, kMu);
    s.g_u = dpsi_mt(s.d_phi_0, s.d_phi_0, kMu);
    const double step_max = s.step_size, step_min = s.trans_eps / 2;
    s.interval_converged = (step_max - step_min) < 0 ? 1 : 0;
    s.open_interval = 1;
    s.a_t = norm;
    s.a_t = dmin(s.a_t, step_max);
    s.a_t = dmax(s.a_t, step_min);
    for (int k = 0; k < 6; ++k) s.x_t[k] = s.x[k] + s.dir[k] * s.a_t;
    pose_to_matrix(s.x_t, s.final_);
    make_request(s, s.split_first ? 1 : 0, s.x_t);
    s.phase = NDT_LS_FIRST;
    return CTL_RETURN;
}

// single-lane / host form: on return either done(s) or a new request is pending (s.req_mode / s.req_p / s.final_)
MRGFE_HD void on_result(NdtCtlState& s, const double r[44])
{
    int next = resume(s, r);
    while (next == CTL_NEED_SOLVE) {
        double neg_g[6], delta[6];
        for (int k = 0; k < 6; ++k) neg_g[k] = -s.g[k];
        newton_solve6(s.H, neg_g, delta, s.svd_only);
        next = after_solve(s, delta);
    }
}

}  // namespace ctl
}  // namespace mrgfe
