// oracle/bfgs.h — TEST INFRASTRUCTURE (CPU oracle). Not part of the shipped product path.
//
// The minimiser behind pcl::GeneralizedIterativeClosestPoint::estimateRigidTransformationBFGS (PCL 1.12 <pcl/registration/bfgs.h>, itself a
// C++ port of GSL's vector_bfgs2 minimiser and its Fletcher line search, multimin/vector_bfgs2.c + linear_minimize.c), restated from the
// published algorithm — PARITY UNPINNED like everything under oracle/ (no PCL or GSL in this image).  Written in GSL's shape: a wrapper that
// caches f / gradient / slope per step length, `minimize` = bracketing + sectioning with cubic (order 3) or quadratic interpolation.
// `Functor` supplies  double f(const double x[6]);  void df(const double x[6], double g[6]);  void fdf(const double x[6], double& f, double g[6]).
#pragma once
#include <algorithm>
#include <cmath>
#include <limits>

namespace orc {

enum BfgsStatus { BFGS_SUCCESS = 0, BFGS_RUNNING = -2, BFGS_NOT_STARTED = -1, BFGS_NO_PROGRESS = 1 };  // pcl::BFGSSpace::Status values used by GICP

struct BfgsParams {  // estimateRigidTransformationBFGS sets these six; step_size is the BFGS class default
    double sigma = 0.01, rho = 0.01, tau1 = 9, tau2 = 0.05, tau3 = 0.5, step_size = 0.01;
    int    order = 3, bracket_iters = 100, section_iters = 100;
};

template <class Functor>
class Bfgs {
   public:
    explicit Bfgs(Functor& fn) : fn_(fn) {}
    BfgsParams parameters;
    double     f = 0, gradient[6] = {0, 0, 0, 0, 0, 0};
    int        evaluations = 0;  // functor calls (f, df and fdf alike)

    int minimize_init(double x[6])
    {
        iter_ = 0;
        delta_f_ = 0;
        for (int k = 0; k < 6; ++k) dx_[k] = 0;
        fn_.fdf(x, f, gradient);
        ++evaluations;
        for (int k = 0; k < 6; ++k) { x0_[k] = x[k]; g0_[k] = gradient[k]; }
        g0norm_ = norm(g0_);
        for (int k = 0; k < 6; ++k) p_[k] = gradient[k] * (-1 / g0norm_);
        pnorm_ = norm(p_);
        fp0_ = -g0norm_;
        for (int k = 0; k < 6; ++k) { x_alpha_[k] = x0_[k]; g_alpha_[k] = g0_[k]; }
        x_key_ = f_key_ = g_key_ = df_key_ = 0;
        f_alpha_ = f;
        df_alpha_ = slope();
        return BFGS_NOT_STARTED;
    }

    int minimize_one_step(double x[6])
    {
        double       alpha = 0.0, alpha1;
        const double f0 = f;
        if (pnorm_ == 0.0 || g0norm_ == 0.0 || fp0_ == 0) {
            for (int k = 0; k < 6; ++k) dx_[k] = 0;
            return BFGS_NO_PROGRESS;
        }
        if (delta_f_ < 0) {
            const double del = std::max(-delta_f_, 10 * std::numeric_limits<double>::epsilon() * std::fabs(f0));
            alpha1 = std::min(1.0, 2.0 * del / (-fp0_));
        } else {
            alpha1 = std::fabs(parameters.step_size);
        }
        const int status = line_search(alpha1, alpha);
        if (status != BFGS_SUCCESS) return status;
        update_position(alpha, x);
        delta_f_ = f - f0;
        {  // the BFGS update: p' = g1 - A dx - B dg,  B = dx.g / dx.dg,  A = -(1 + dg.dg / dx.dg) B + dg.g / dx.dg
            double dx0[6], dg0[6];
            for (int k = 0; k < 6; ++k) { dx0[k] = x[k] - x0_[k]; dx_[k] = dx0[k]; dg0[k] = gradient[k] - g0_[k]; }
            const double dxg = dot(dx0, gradient), dgg = dot(dg0, gradient), dxdg = dot(dx0, dg0), dgnorm = norm(dg0);
            double A = 0, B = 0;
            if (dxdg != 0) { B = dxg / dxdg; A = -(1.0 + dgnorm * dgnorm / dxdg) * B + dgg / dxdg; }
            for (int k = 0; k < 6; ++k) { p_[k] = -A * dx0[k]; p_[k] += gradient[k]; p_[k] += -B * dg0[k]; }
        }
        for (int k = 0; k < 6; ++k) { g0_[k] = gradient[k]; x0_[k] = x[k]; }
        g0norm_ = norm(g0_);
        pnorm_ = norm(p_);
        const double dir = (dot(p_, gradient) > 0) ? -1.0 : 1.0;
        for (int k = 0; k < 6; ++k) p_[k] *= dir / pnorm_;
        pnorm_ = norm(p_);
        fp0_ = dot(p_, g0_);
        change_direction();
        ++iter_;
        return BFGS_SUCCESS;
    }

   private:
    Functor& fn_;
    int      iter_ = 0;
    double   delta_f_ = 0, fp0_ = 0, g0norm_ = 0, pnorm_ = 0;
    double   x0_[6], dx_[6], g0_[6], p_[6];
    // the wrapper: position, value, gradient and slope at the step length they were last computed for
    double   x_alpha_[6], g_alpha_[6], f_alpha_ = 0, df_alpha_ = 0, x_key_ = 0, f_key_ = 0, g_key_ = 0, df_key_ = 0;

    static double dot(const double a[6], const double b[6]) { double s = 0; for (int k = 0; k < 6; ++k) s += a[k] * b[k]; return s; }
    static double norm(const double a[6]) { return std::sqrt(dot(a, a)); }
    double slope() const { return dot(g_alpha_, p_); }
    void   move_to(double alpha)
    {
        if (alpha == x_key_) return;
        for (int k = 0; k < 6; ++k) x_alpha_[k] = x0_[k] + alpha * p_[k];
        x_key_ = alpha;
    }
    double apply_f(double alpha)
    {
        if (alpha == f_key_) return f_alpha_;
        move_to(alpha);
        f_alpha_ = fn_.f(x_alpha_);
        ++evaluations;
        f_key_ = alpha;
        return f_alpha_;
    }
    double apply_df(double alpha)
    {
        if (alpha == df_key_) return df_alpha_;
        move_to(alpha);
        if (alpha != g_key_) { fn_.df(x_alpha_, g_alpha_); ++evaluations; g_key_ = alpha; }
        df_alpha_ = slope();
        df_key_ = alpha;
        return df_alpha_;
    }
    void apply_fdf(double alpha, double& fv, double& dfv)
    {
        if (alpha == f_key_ && alpha == df_key_) { fv = f_alpha_; dfv = df_alpha_; return; }
        if (alpha == f_key_ || alpha == g_key_ || alpha == df_key_) { fv = apply_f(alpha); dfv = apply_df(alpha); return; }
        move_to(alpha);
        fn_.fdf(x_alpha_, f_alpha_, g_alpha_);
        ++evaluations;
        f_key_ = g_key_ = alpha;
        df_alpha_ = slope();
        df_key_ = alpha;
        fv = f_alpha_;
        dfv = df_alpha_;
    }
    void update_position(double alpha, double x[6])
    {
        double fa, dfa;
        apply_fdf(alpha, fa, dfa);
        f = fa;
        for (int k = 0; k < 6; ++k) { x[k] = x_alpha_[k]; gradient[k] = g_alpha_[k]; }
    }
    void change_direction()
    {
        for (int k = 0; k < 6; ++k) { x_alpha_[k] = x0_[k]; g_alpha_[k] = g0_[k]; }
        x_key_ = f_key_ = g_key_ = 0.0;
        df_alpha_ = slope();
        df_key_ = 0.0;
    }

    // ---- interpolation (linear_minimize.c) ----
    static double cubic(double c0, double c1, double c2, double c3, double z) { return c0 + z * (c1 + z * (c2 + z * c3)); }
    static void   check_extremum(double c0, double c1, double c2, double c3, double z, double* zmin, double* fmin)
    {
        const double y = cubic(c0, c1, c2, c3, z);
        if (y < *fmin) { *zmin = z; *fmin = y; }
    }
    static int solve_quadratic(double a, double b, double c, double* x0, double* x1)  // gsl_poly_solve_quadratic
    {
        if (a == 0) {
            if (b == 0) return 0;
            *x0 = -c / b;
            return 1;
        }
        const double disc = b * b - 4 * a * c;
        if (disc > 0) {
            if (b == 0) {
                const double r = std::sqrt(-c / a);
                *x0 = -r;
                *x1 = r;
            } else {
                const double sgnb = (b > 0 ? 1 : -1), temp = -0.5 * (b + sgnb * std::sqrt(disc)), r1 = temp / a, r2 = c / temp;
                if (r1 < r2) { *x0 = r1; *x1 = r2; } else { *x0 = r2; *x1 = r1; }
            }
            return 2;
        }
        if (disc == 0) { *x0 = -0.5 * b / a; *x1 = -0.5 * b / a; return 2; }
        return 0;
    }
    static double interp_quad(double f0, double fp0, double f1, double zl, double zh)
    {
        const double fl = f0 + zl * (fp0 + zl * (f1 - f0 - fp0)), fh = f0 + zh * (fp0 + zh * (f1 - f0 - fp0));
        const double c = 2 * (f1 - f0 - fp0);  // curvature
        double zmin = zl, fmin = fl;
        if (fh < fmin) { zmin = zh; fmin = fh; }
        if (c > 0) {  // positive curvature required for a minimum
            const double z = -fp0 / c;
            if (z > zl && z < zh) {
                const double fz = f0 + z * (fp0 + z * (f1 - f0 - fp0));
                if (fz < fmin) { zmin = z; fmin = fz; }
            }
        }
        return zmin;
    }
    static double interp_cubic(double f0, double fp0, double f1, double fp1, double zl, double zh)
    {
        const double eta = 3 * (f1 - f0) - 2 * fp0 - fp1, xi = fp0 + fp1 - 2 * (f1 - f0);
        const double c0 = f0, c1 = fp0, c2 = eta, c3 = xi;
        double zmin = zl, fmin = cubic(c0, c1, c2, c3, zl), z0 = 0, z1 = 0;
        check_extremum(c0, c1, c2, c3, zh, &zmin, &fmin);
        const int n = solve_quadratic(3 * c3, 2 * c2, c1, &z0, &z1);
        if (n == 2) {
            if (z0 > zl && z0 < zh) check_extremum(c0, c1, c2, c3, z0, &zmin, &fmin);
            if (z1 > zl && z1 < zh) check_extremum(c0, c1, c2, c3, z1, &zmin, &fmin);
        } else if (n == 1) {
            if (z0 > zl && z0 < zh) check_extremum(c0, c1, c2, c3, z0, &zmin, &fmin);
        }
        return zmin;
    }
    double interpolate(double a, double fa, double fpa, double b, double fb, double fpb, double xmin, double xmax) const
    {
        double zmin = (xmin - a) / (b - a), zmax = (xmax - a) / (b - a), z;  // map [a, b] to [0, 1]
        if (zmin > zmax) std::swap(zmin, zmax);
        if (parameters.order > 2 && std::isfinite(fpb)) z = interp_cubic(fa, fpa * (b - a), fb, fpb * (b - a), zmin, zmax);
        else z = interp_quad(fa, fpa * (b - a), fb, zmin, zmax);
        return a + z * (b - a);
    }
    // Fletcher's line search (Practical Methods of Optimization, algorithms 2.6.2 and 2.6.4): bracketing, then sectioning
    int line_search(double alpha1, double& alpha_new)
    {
        const double rho = parameters.rho, sigma = parameters.sigma, tau1 = parameters.tau1, tau2 = parameters.tau2, tau3 = parameters.tau3;
        const double kNaN = std::numeric_limits<double>::quiet_NaN();
        double f0, fp0, falpha, falpha_prev, fpalpha = 0, fpalpha_prev, delta, alpha_next;
        double alpha = alpha1, alpha_prev = 0.0;
        double a = 0.0, b = alpha, fa, fb = 0.0, fpa, fpb = 0.0;
        int    i = 0;
        apply_fdf(0.0, f0, fp0);
        falpha_prev = f0;
        fpalpha_prev = fp0;
        fa = f0;
        fpa = fp0;
        while (i++ < parameters.bracket_iters) {  // bracketing
            falpha = apply_f(alpha);
            if (falpha > f0 + alpha * rho * fp0 || falpha >= falpha_prev) {  // Fletcher's rho test
                a = alpha_prev; fa = falpha_prev; fpa = fpalpha_prev;
                b = alpha; fb = falpha; fpb = kNaN;
                break;
            }
            fpalpha = apply_df(alpha);
            if (std::fabs(fpalpha) <= -sigma * fp0) { alpha_new = alpha; return BFGS_SUCCESS; }  // Fletcher's sigma test
            if (fpalpha >= 0) {
                a = alpha; fa = falpha; fpa = fpalpha;
                b = alpha_prev; fb = falpha_prev; fpb = fpalpha_prev;
                break;
            }
            delta = alpha - alpha_prev;
            alpha_next = interpolate(alpha_prev, falpha_prev, fpalpha_prev, alpha, falpha, fpalpha, alpha + delta, alpha + tau1 * delta);
            alpha_prev = alpha; falpha_prev = falpha; fpalpha_prev = fpalpha;
            alpha = alpha_next;
        }
        while (i++ < parameters.section_iters) {  // sectioning of the bracket [a, b]
            delta = b - a;
            alpha = interpolate(a, fa, fpa, b, fb, fpb, a + tau2 * delta, b - tau3 * delta);
            falpha = apply_f(alpha);
            if ((a - alpha) * fpa <= std::numeric_limits<double>::epsilon()) return BFGS_NO_PROGRESS;  // roundoff prevents progress
            if (falpha > f0 + rho * alpha * fp0 || falpha >= fa) {
                b = alpha; fb = falpha; fpb = kNaN;  // a_next = a
            } else {
                fpalpha = apply_df(alpha);
                if (std::fabs(fpalpha) <= -sigma * fp0) { alpha_new = alpha; return BFGS_SUCCESS; }
                if (((b - a) >= 0 && fpalpha >= 0) || ((b - a) <= 0 && fpalpha <= 0)) { b = a; fb = fa; fpb = fpa; a = alpha; fa = falpha; fpa = fpalpha; }
                else { a = alpha; fa = falpha; fpa = fpalpha; }
            }
        }
        return BFGS_SUCCESS;
    }
};

}  // namespace orc
