// csrc/ndt_controller.cpp — resumable Newton / More-Thuente controller of NDT_HIP (see ndt_controller.h).
// Mirrors pclomp::NormalDistributionsTransform::computeTransformation / computeStepLengthMT /
// trialValueSelectionMT / updateIntervalMT / computeAngleDerivatives as the reference reaches them through
// registration_->align() (/root/reference/apps/scan_matching_odometry_component.cpp:265-266;
// src/mrg_slam/loop_detector.cpp:134).  Compiled with -ffp-contract=off: the float expressions below execute in
// the written order.
#include "ndt_controller.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>

namespace mrgfe {

namespace {

// ---- 6x6 solve through a one-sided (Hestenes) Jacobi SVD: x = V diag(1/s) U^T b over singular values above
// 6*eps*s_max, i.e. the minimum-norm solution Eigen::JacobiSVD<Matrix6d>::solve returns (no PD fix-up). -----------
void svd_solve6(const double A[36], const double b[6], double x[6])
{
    for (int i = 0; i < 36; ++i)
        if (!std::isfinite(A[i])) { for (int k = 0; k < 6; ++k) x[k] = std::numeric_limits<double>::quiet_NaN(); return; }
    double U[6][6], V[6][6];
    double scale = 0;
    for (int i = 0; i < 36; ++i) scale = std::max(scale, std::fabs(A[i]));
    if (scale == 0) { for (int k = 0; k < 6; ++k) x[k] = 0; return; }
    for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) { U[r][c] = A[r * 6 + c] / scale; V[r][c] = r == c ? 1.0 : 0.0; }
    for (int sweep = 0; sweep < 60; ++sweep) {
        bool rotated = false;
        for (int p = 0; p < 5; ++p)
            for (int q = p + 1; q < 6; ++q) {
                double alpha = 0, beta = 0, gamma = 0;
                for (int k = 0; k < 6; ++k) { alpha += U[k][p] * U[k][p]; beta += U[k][q] * U[k][q]; gamma += U[k][p] * U[k][q]; }
                if (gamma == 0.0 || std::fabs(gamma) <= 1e-15 * std::sqrt(alpha * beta)) continue;
                rotated = true;
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / std::sqrt(1.0 + t * t), s = c * t;
                for (int k = 0; k < 6; ++k) {
                    const double up = U[k][p], uq = U[k][q];
                    U[k][p] = c * up - s * uq; U[k][q] = s * up + c * uq;
                    const double vp = V[k][p], vq = V[k][q];
                    V[k][p] = c * vp - s * vq; V[k][q] = s * vp + c * vq;
                }
            }
        if (!rotated) break;
    }
    double sig[6], smax = 0;
    for (int j = 0; j < 6; ++j) {
        double n2 = 0;
        for (int k = 0; k < 6; ++k) n2 += U[k][j] * U[k][j];
        sig[j] = std::sqrt(n2);
        smax = std::max(smax, sig[j]);
    }
    const double thr = std::max(smax * 6.0 * DBL_EPSILON, std::numeric_limits<double>::min() / scale);
    for (int k = 0; k < 6; ++k) x[k] = 0;
    for (int j = 0; j < 6; ++j) {
        if (!(sig[j] > thr)) continue;
        double ub = 0;
        for (int k = 0; k < 6; ++k) ub += U[k][j] * b[k];
        const double coef = ub / (sig[j] * sig[j] * scale);  // U[:,j] is sig_j * u_j
        for (int k = 0; k < 6; ++k) x[k] += V[k][j] * coef;
    }
}

inline double psi_mt(double a, double f_a, double f_0, double g_0, double mu) { return f_a - f_0 - mu * g_0 * a; }
inline double dpsi_mt(double g_a, double g_0, double mu) { return g_a - mu * g_0; }
constexpr double kMu = 1.e-4, kNu = 0.9;
constexpr int    kMaxStepIterations = 10;

bool update_interval(double& a_l, double& f_l, double& g_l, double& a_u, double& f_u, double& g_u, double a_t, double f_t, double g_t)
{
    if (f_t > f_l) { a_u = a_t; f_u = f_t; g_u = g_t; return false; }
    if (g_t * (a_l - a_t) > 0) { a_l = a_t; f_l = f_t; g_l = g_t; return false; }
    if (g_t * (a_l - a_t) < 0) { a_u = a_l; f_u = f_l; g_u = g_l; a_l = a_t; f_l = f_t; g_l = g_t; return false; }
    return true;
}

double trial_value(double a_l, double f_l, double g_l, double a_u, double f_u, double g_u, double a_t, double f_t, double g_t)
{
    auto cubic = [](double a0, double f0, double g0, double a1, double f1, double g1) {
        // minimiser of the cubic through (a0,f0,g0) and (a1,f1,g1), Sun & Yuan eq. 2.4.52/2.4.56
        const double z = 3 * (f1 - f0) / (a1 - a0) - g1 - g0;
        const double w = std::sqrt(z * z - g1 * g0);
        return a0 + (a1 - a0) * (w - g0 - z) / (g1 - g0 + 2 * w);
    };
    if (f_t > f_l) {
        const double a_c = cubic(a_l, f_l, g_l, a_t, f_t, g_t);
        const double a_q = a_l - 0.5 * (a_l - a_t) * g_l / (g_l - (f_l - f_t) / (a_l - a_t));
        return (std::fabs(a_c - a_l) < std::fabs(a_q - a_l)) ? a_c : 0.5 * (a_q + a_c);
    }
    if (g_t * g_l < 0) {
        const double a_c = cubic(a_l, f_l, g_l, a_t, f_t, g_t);
        const double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
        return (std::fabs(a_c - a_t) >= std::fabs(a_s - a_t)) ? a_c : a_s;
    }
    if (std::fabs(g_t) <= std::fabs(g_l)) {
        const double a_c = cubic(a_l, f_l, g_l, a_t, f_t, g_t);
        const double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
        const double a_next = (std::fabs(a_c - a_t) < std::fabs(a_s - a_t)) ? a_c : a_s;
        return (a_t > a_l) ? std::min(a_t + 0.66 * (a_u - a_t), a_next) : std::max(a_t + 0.66 * (a_u - a_t), a_next);
    }
    return cubic(a_u, f_u, g_u, a_t, f_t, g_t);
}

void identity16(float M[16]) { for (int i = 0; i < 16; ++i) M[i] = (i % 5 == 0) ? 1.0f : 0.0f; }
bool is_identity16(const float M[16])
{
    for (int i = 0; i < 16; ++i) if (M[i] != ((i % 5 == 0) ? 1.0f : 0.0f)) return false;
    return true;
}

// Eigen::AngleAxisf(angle, e_axis).toRotationMatrix()
void axis_rotation(float angle, int axis, float R[9])
{
    float ax[3] = {0, 0, 0};
    ax[axis] = 1.0f;
    const float sn = std::sin(angle), c = std::cos(angle);
    const float sa[3] = {sn * ax[0], sn * ax[1], sn * ax[2]};
    const float ca[3] = {(1.0f - c) * ax[0], (1.0f - c) * ax[1], (1.0f - c) * ax[2]};
    float tmp;
    tmp = ca[0] * ax[1]; R[1] = tmp - sa[2]; R[3] = tmp + sa[2];
    tmp = ca[0] * ax[2]; R[2] = tmp + sa[1]; R[6] = tmp - sa[1];
    tmp = ca[1] * ax[2]; R[5] = tmp - sa[0]; R[7] = tmp + sa[0];
    R[0] = ca[0] * ax[0] + c; R[4] = ca[1] * ax[1] + c; R[8] = ca[2] * ax[2] + c;
}
void matmul3f(const float a[9], const float b[9], float o[9])
{
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            const float p0 = a[r * 3] * b[c], p1 = a[r * 3 + 1] * b[3 + c], p2 = a[r * 3 + 2] * b[6 + c];
            const float s = p0 + p1;
            o[r * 3 + c] = s + p2;
        }
}

}  // namespace

static bool spec_enabled()
{
    // Off by default: measured on MI355X, only ~40 % of the first trials end in computeHessian at the same pose, so the
    // speculative f64 pass costs more GPU time than the host round trips it saves (7.4k -> 5.3k alignments/s at 32 pairs).
    static const bool on = [] { const char* e = std::getenv("MRGFE_SPEC"); return e && e[0] == '1'; }();
    return on;
}

static bool reuse_enabled()
{
    static const bool on = [] { const char* e = std::getenv("MRGFE_NO_REUSE"); return !(e && e[0] == '1'); }();
    return on;
}

void NdtController::pose_to_matrix(const double p[6], float M[16])
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

void NdtController::euler_xyz(const float M[16], float out[3])
{
    // Eigen 3.4 Matrix3f::eulerAngles(0, 1, 2)
    const float kPi = 3.14159265358979323846f;
    float r0 = std::atan2(M[1 * 4 + 2], M[2 * 4 + 2]);
    const float c2 = std::sqrt(M[0] * M[0] + M[1] * M[1]);
    float r1;
    if (r0 > 0.0f) { r0 -= kPi; r1 = std::atan2(-M[2], -c2); }
    else           { r1 = std::atan2(-M[2], c2); }
    const float s1 = std::sin(r0), c1 = std::cos(r0);
    const float r2 = std::atan2(s1 * M[2 * 4 + 0] - c1 * M[1 * 4 + 0], c1 * M[1 * 4 + 1] - s1 * M[2 * 4 + 1]);
    out[0] = -r0; out[1] = -r1; out[2] = -r2;
}

void NdtController::angle_tables(const double p[6], double j[8][3], double h[15][3])
{
    double cx, cy, cz, sx, sy, sz;
    if (std::fabs(p[3]) < 10e-5) { cx = 1.0; sx = 0.0; } else { cx = std::cos(p[3]); sx = std::sin(p[3]); }
    if (std::fabs(p[4]) < 10e-5) { cy = 1.0; sy = 0.0; } else { cy = std::cos(p[4]); sy = std::sin(p[4]); }
    if (std::fabs(p[5]) < 10e-5) { cz = 1.0; sz = 0.0; } else { cz = std::cos(p[5]); sz = std::sin(p[5]); }
    const double jj[8][3] = {{(-sx * sz + cx * sy * cz), (-sx * cz - cx * sy * sz), (-cx * cy)},
                             {(cx * sz + sx * sy * cz), (cx * cz - sx * sy * sz), (-sx * cy)},
                             {(-sy * cz), sy * sz, cy},
                             {sx * cy * cz, (-sx * cy * sz), sx * sy},
                             {(-cx * cy * cz), cx * cy * sz, (-cx * sy)},
                             {(-cy * sz), (-cy * cz), 0},
                             {(cx * cz - sx * sy * sz), (-cx * sz - sx * sy * cz), 0},
                             {(sx * cz + cx * sy * sz), (cx * sy * cz - sx * sz), 0}};
    const double hh[15][3] = {{(-cx * sz - sx * sy * cz), (-cx * cz + sx * sy * sz), sx * cy},
                              {(-sx * sz + cx * sy * cz), (-cx * sy * sz - sx * cz), (-cx * cy)},
                              {(cx * cy * cz), (-cx * cy * sz), (cx * sy)},
                              {(sx * cy * cz), (-sx * cy * sz), (sx * sy)},
                              {(-sx * cz - cx * sy * sz), (sx * sz - cx * sy * cz), 0},
                              {(cx * cz - sx * sy * sz), (-sx * sy * cz - cx * sz), 0},
                              {(-cy * cz), (cy * sz), (sy)},
                              {(-sx * sy * cz), (sx * sy * sz), (sx * cy)},
                              {(cx * sy * cz), (-cx * sy * sz), (-cx * cy)},
                              {(sy * sz), (sy * cz), 0},
                              {(-sx * cy * sz), (-sx * cy * cz), 0},
                              {(cx * cy * sz), (cx * cy * cz), 0},
                              {(-cy * cz), (cy * sz), 0},
                              {(-cx * sz - sx * sy * cz), (-cx * cz + sx * sy * sz), 0},
                              {(-sx * sz + cx * sy * cz), (-cx * sy * sz - sx * cz), 0}};
    std::memcpy(j, jj, sizeof(jj));
    std::memcpy(h, hh, sizeof(hh));
}

void NdtController::make_request(int mode, const double p[6])
{
    req_.mode = mode;
    req_.spec_hessian = false;
    std::memcpy(req_.T, final_, sizeof(final_));
    std::memcpy(req_.p, p, sizeof(double) * 6);
    angle_tables(p, req_.j_ang, req_.h_ang);
    ++n_evals_;
}

void NdtController::start(const NdtParams& prm, const float guess[16], uint32_t n_src)
{
    prm_ = prm;
    n_src_ = n_src;
    // pcl::Registration::align
    converged_ = false;
    nr_iterations_ = 0;
    n_evals_ = 0;
    n_reused_ = 0;
    n_spec_used_ = 0;
    cache_valid_ = false;
    spec_valid_ = false;
    nb_sum_ = 0;
    trans_probability_ = 0;
    identity16(final_); identity16(transformation_); identity16(previous_);
    for (int k = 0; k < 36; ++k) H_[k] = 0;
    for (int k = 0; k < 6; ++k) g_[k] = 0;
    score_ = 0;
    // computeTransformation prologue
    const double c1 = 10 * (1 - prm_.outlier_ratio);
    const double c2 = prm_.outlier_ratio / std::pow(static_cast<double>(prm_.resolution), 3);
    gauss_d3_ = -std::log(c2);
    gauss_d1_ = -std::log(c1 + c2) - gauss_d3_;
    gauss_d2_ = -2 * std::log((-std::log(c1 * std::exp(-0.5) + c2) - gauss_d3_) / gauss_d1_);
    if (n_src_ == 0) { phase_ = DONE; return; }
    if (!is_identity16(guess)) std::memcpy(final_, guess, sizeof(final_));
    float eul[3];
    euler_xyz(final_, eul);
    p_[0] = final_[3]; p_[1] = final_[7]; p_[2] = final_[11];
    p_[3] = eul[0]; p_[4] = eul[1]; p_[5] = eul[2];
    make_request(0, p_);
    phase_ = INIT;
}

void NdtController::abort_no_target()
{
    identity16(final_);
    converged_ = false;
    phase_ = DONE;
}

void NdtController::store_result(const double r[44], bool with_score_grad, bool with_hessian)
{
    if (with_score_grad) {
        score_ = r[0];
        for (int k = 0; k < 6; ++k) g_[k] = r[1 + k];
    }
    if (with_hessian)
        for (int k = 0; k < 36; ++k) H_[k] = r[7 + k];
    const double nb = n_src_ ? r[43] / static_cast<double>(n_src_) : 0.0;
    nb_sum_ += nb;
    // remember the pose of evaluations whose transform was built from the pose vector (line-search trials)
    if (with_score_grad && phase_ != INIT) {
        std::memcpy(cache_p_, req_.p, sizeof(cache_p_));
        cache_nb_ = nb;
        cache_valid_ = true;
    }
}

void NdtController::on_result(const double r[44], const double* r_spec)
{
    switch (phase_) {
        case INIT:
            store_result(r, true, true);
            newton_step();
            break;
        case LS_FIRST:
            store_result(r, true, true);
            spec_valid_ = false;
            if (r_spec && req_.spec_hessian) {
                for (int k = 0; k < 36; ++k) H_spec_[k] = r_spec[7 + k];
                std::memcpy(spec_p_, req_.p, sizeof(spec_p_));
                spec_nb_ = n_src_ ? r_spec[43] / static_cast<double>(n_src_) : 0.0;
                spec_valid_ = true;
            }
            ls_after_eval();
            ls_continue_or_finish();
            break;
        case LS_ITER:
            store_result(r, true, false);
            ls_iter_update();
            break;
        case LS_HESS:
            store_result(r, false, true);
            finish_line_search(a_t_);
            break;
        default:
            break;
    }
}

void NdtController::newton_step()
{
    std::memcpy(previous_, transformation_, sizeof(previous_));
    double neg_g[6], delta[6];
    for (int k = 0; k < 6; ++k) neg_g[k] = -g_[k];
    svd_solve6(H_, neg_g, delta);
    double n2 = 0;
    for (int k = 0; k < 6; ++k) n2 += delta[k] * delta[k];
    const double norm = std::sqrt(n2);
    if (norm == 0 || norm != norm) {
        trans_probability_ = score_ / static_cast<double>(n_src_);
        converged_ = (norm == norm);
        phase_ = DONE;
        return;
    }
    for (int k = 0; k < 6; ++k) dir_[k] = delta[k] / norm;
    // computeStepLengthMT prologue
    std::memcpy(x_, p_, sizeof(x_));
    phi_0_ = -score_;
    double d = 0;
    for (int k = 0; k < 6; ++k) d += g_[k] * dir_[k];
    d_phi_0_ = -d;
    if (d_phi_0_ >= 0) {
        if (d_phi_0_ == 0) {
            // "return 0": a zero-length step; the outer loop then converges on its second pass (|0| < eps)
            finish_line_search(0.0);
            return;
        }
        d_phi_0_ *= -1;
        for (int k = 0; k < 6; ++k) dir_[k] *= -1;
    }
    step_iterations_ = 0;
    a_l_ = 0; a_u_ = 0;
    f_l_ = psi_mt(a_l_, phi_0_, phi_0_, d_phi_0_, kMu);
    g_l_ = dpsi_mt(d_phi_0_, d_phi_0_, kMu);
    f_u_ = psi_mt(a_u_, phi_0_, phi_0_, d_phi_0_, kMu);
    g_u_ = dpsi_mt(d_phi_0_, d_phi_0_, kMu);
    const double step_max = prm_.step_size, step_min = prm_.trans_eps / 2;
    interval_converged_ = (step_max - step_min) < 0;
    open_interval_ = true;
    a_t_ = norm;
    a_t_ = std::min(a_t_, step_max);
    a_t_ = std::max(a_t_, step_min);
    for (int k = 0; k < 6; ++k) x_t_[k] = x_[k] + dir_[k] * a_t_;
    pose_to_matrix(x_t_, final_);
    make_request(0, x_t_);
    // When the first trial is rejected the search almost always ends on a repeated (clamped) step and the reference then
    // calls computeHessian at that same pose: ask for that f64 Hessian in the same round (dropped if it is not needed).
    req_.spec_hessian = spec_enabled();
    phase_ = LS_FIRST;
}

void NdtController::ls_after_eval()
{
    phi_t_ = -score_;
    double d = 0;
    for (int k = 0; k < 6; ++k) d += g_[k] * dir_[k];
    d_phi_t_ = -d;
    psi_t_ = psi_mt(a_t_, phi_t_, phi_0_, d_phi_0_, kMu);
    d_psi_t_ = dpsi_mt(d_phi_t_, d_phi_0_, kMu);
}

// bookkeeping of one line-search trial after its score / gradient are known (body of the while loop of computeStepLengthMT)
void NdtController::ls_iter_update()
{
    ls_after_eval();
    if (open_interval_ && (psi_t_ <= 0 && d_psi_t_ >= 0)) {
        open_interval_ = false;
        f_l_ = f_l_ + phi_0_ - kMu * d_phi_0_ * a_l_;
        g_l_ = g_l_ + kMu * d_phi_0_;
        f_u_ = f_u_ + phi_0_ - kMu * d_phi_0_ * a_u_;
        g_u_ = g_u_ + kMu * d_phi_0_;
    }
    if (open_interval_) interval_converged_ = update_interval(a_l_, f_l_, g_l_, a_u_, f_u_, g_u_, a_t_, psi_t_, d_psi_t_);
    else                interval_converged_ = update_interval(a_l_, f_l_, g_l_, a_u_, f_u_, g_u_, a_t_, phi_t_, d_phi_t_);
    step_iterations_++;
    ls_continue_or_finish();
}

void NdtController::ls_continue_or_finish()
{
    if (!interval_converged_ && step_iterations_ < kMaxStepIterations && !(psi_t_ <= 0 && d_phi_t_ <= -kNu * d_phi_0_)) {
        if (open_interval_) a_t_ = trial_value(a_l_, f_l_, g_l_, a_u_, f_u_, g_u_, a_t_, psi_t_, d_psi_t_);
        else                a_t_ = trial_value(a_l_, f_l_, g_l_, a_u_, f_u_, g_u_, a_t_, phi_t_, d_phi_t_);
        a_t_ = std::min(a_t_, prm_.step_size);
        a_t_ = std::max(a_t_, prm_.trans_eps / 2);
        for (int k = 0; k < 6; ++k) x_t_[k] = x_[k] + dir_[k] * a_t_;
        pose_to_matrix(x_t_, final_);
        if (reuse_enabled() && cache_valid_ && std::memcmp(x_t_, cache_p_, sizeof(cache_p_)) == 0) {
            // The clamped trial step often repeats the previous one (a_t pinned at step_min / step_max): same pose vector
            // -> same float transform -> the evaluation would reproduce score_ / g_ bit for bit.  The reference
            // recomputes it; here the held values are reused and no GPU round is spent (the evaluation is still counted).
            ++n_evals_;
            ++n_reused_;
            nb_sum_ += cache_nb_;
            ls_iter_update();
            return;
        }
        make_request(1, x_t_);
        phase_ = LS_ITER;
        return;
    }
    if (step_iterations_) {
        if (spec_valid_ && std::memcmp(x_t_, spec_p_, sizeof(spec_p_)) == 0) {
            // computeHessian at the pose whose f64 Hessian was evaluated speculatively with the first trial
            for (int k = 0; k < 36; ++k) H_[k] = H_spec_[k];
            ++n_evals_;
            ++n_spec_used_;
            nb_sum_ += spec_nb_;
            finish_line_search(a_t_);
            return;
        }
        make_request(2, x_t_);  // computeHessian at x_t (final_ already holds its matrix)
        phase_ = LS_HESS;
        return;
    }
    finish_line_search(a_t_);
}

void NdtController::finish_line_search(double a_t)
{
    // back in computeTransformation's loop body
    double delta_p[6];
    for (int k = 0; k < 6; ++k) delta_p[k] = dir_[k] * a_t;
    pose_to_matrix(delta_p, transformation_);
    for (int k = 0; k < 6; ++k) p_[k] = p_[k] + delta_p[k];
    if (nr_iterations_ > prm_.max_iterations || (nr_iterations_ && (std::fabs(a_t) < prm_.trans_eps))) converged_ = true;
    nr_iterations_++;
    if (converged_) {
        trans_probability_ = score_ / static_cast<double>(n_src_);
        phase_ = DONE;
        return;
    }
    newton_step();  // next Newton iteration from the derivatives already held for p_ == x_t
}

}  // namespace mrgfe
