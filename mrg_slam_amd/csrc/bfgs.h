// csrc/bfgs.h — the minimiser of pcl::GeneralizedIterativeClosestPoint::estimateRigidTransformationBFGS (PCL 1.12 <pcl/registration/bfgs.h>:
// a C++ port of GSL's vector_bfgs2 and its Fletcher line search) for PCL_GICP_HIP, host side.  The functor's values come from the GPU
// (gicp.hip: pclgicp_fdf_kernel, 13 sums per evaluation); this class only decides where to evaluate next.  Six parameters, double.
// Restated from the published algorithm (no PCL / GSL in the build image; the CPU oracle holds an independently written copy in GSL's shape and
// tests/test_gpu_pclgicp.py runs the two against each other).
#pragma once
#include <algorithm>
#include <cmath>
#include <limits>

namespace mrgfe {

namespace BFGSSpace {
enum Status { NegativeGradientEpsilon = -3, NotStarted = -2, Running = -1, Success = 0, NoProgress = 1 };
}

template <class FunctorType>
class BFGS {
   public:
    struct Parameters {
        int    max_iters = 400, bracket_iters = 100, section_iters = 100, order = 3;
        double rho = 0.01, sigma = 0.01, tau1 = 9, tau2 = 0.05, tau3 = 0.5, step_size = 0.01;
    } parameters;
    explicit BFGS(FunctorType& f) : functor(f) {}
    double f = 0;
    double gradient[6] = {0, 0, 0, 0, 0, 0};

    BFGSSpace::Status minimizeInit(double x[6])
    {
        iter = 0;
        delta_f = 0;
        setZero(dx);
        functor.fdf(x, f, gradient);
        copy(x0, x);
        copy(g0, gradient);
        g0norm = norm(g0);
        for (int k = 0; k < 6; ++k) p[k] = gradient[k] * (-1 / g0norm);
        pnorm = norm(p);
        fp0 = -g0norm;
        copy(x_alpha, x0);
        x_cache_key = 0;
        f_alpha = f;
        f_cache_key = 0;
        copy(g_alpha, g0);
        g_cache_key = 0;
        df_alpha = slope();
        df_cache_key = 0;
        return BFGSSpace::NotStarted;
    }

    BFGSSpace::Status minimizeOneStep(double x[6])
    {
        double       alpha = 0.0, alpha1;
        const double f0 = f;
        if (pnorm == 0.0 || g0norm == 0.0 || fp0 == 0) {
            setZero(dx);
            return BFGSSpace::NoProgress;
        }
        if (delta_f < 0) {
            const double del = std::max(-delta_f, 10 * std::numeric_limits<double>::epsilon() * std::fabs(f0));
            alpha1 = std::min(1.0, 2.0 * del / (-fp0));
        } else {
            alpha1 = std::fabs(parameters.step_size);
        }
        const BFGSSpace::Status status = lineSearch(parameters.rho, parameters.sigma, parameters.tau1, parameters.tau2, parameters.tau3, parameters.order, alpha1, alpha);
        if (status != BFGSSpace::Success) return status;
        updatePosition(alpha, x, f, gradient);
        delta_f = f - f0;
        // choose a new direction for the next step: p' = g1 - A dx - B dg, A = -(1 + dg.dg / dx.dg) B + dg.g / dx.dg, B = dx.g / dx.dg
        {
            double dx0[6], dg0[6];
            for (int k = 0; k < 6; ++k) { dx0[k] = x[k] - x0[k]; dx[k] = dx0[k]; dg0[k] = gradient[k] - g0[k]; }
            const double dxg = dot(dx0, gradient), dgg = dot(dg0, gradient), dxdg = dot(dx0, dg0), dgnorm = norm(dg0);
            double A, B;
            if (dxdg != 0) { B = dxg / dxdg; A = -(1.0 + dgnorm * dgnorm / dxdg) * B + dgg / dxdg; }
            else { B = 0; A = 0; }
            for (int k = 0; k < 6; ++k) { p[k] = -A * dx0[k]; p[k] += gradient[k]; p[k] += -B * dg0[k]; }
        }
        copy(g0, gradient);
        copy(x0, x);
        g0norm = norm(g0);
        pnorm = norm(p);
        const double dir = (dot(p, gradient) > 0) ? -1.0 : 1.0;  // update direction and fp0
        for (int k = 0; k < 6; ++k) p[k] *= dir / pnorm;
        pnorm = norm(p);
        fp0 = dot(p, g0);
        changeDirection();
        return BFGSSpace::Success;
    }

   private:
    FunctorType& functor;
    int    iter = 0;
    double delta_f = 0, fp0 = 0, g0norm = 0, pnorm = 0;
    double x0[6], dx[6], g0[6], p[6];
    double x_alpha[6], g_alpha[6], f_alpha = 0, df_alpha = 0;  // the position, gradient, value and slope last computed ...
    double x_cache_key = 0, f_cache_key = 0, g_cache_key = 0, df_cache_key = 0;  // ... and the step lengths they belong to

    static void   setZero(double v[6]) { for (int k = 0; k < 6; ++k) v[k] = 0; }
    static void   copy(double d[6], const double s[6]) { for (int k = 0; k < 6; ++k) d[k] = s[k]; }
    static double dot(const double a[6], const double b[6]) { double s = 0; for (int k = 0; k < 6; ++k) s += a[k] * b[k]; return s; }
    static double norm(const double a[6]) { return std::sqrt(dot(a, a)); }

    void moveTo(double alpha)
    {
        if (alpha == x_cache_key) return;  // using previously cached position
        for (int k = 0; k < 6; ++k) x_alpha[k] = x0[k] + alpha * p[k];
        x_cache_key = alpha;
    }
    double slope() const { return dot(g_alpha, p); }
    double applyF(double alpha)
    {
        if (alpha == f_cache_key) return f_alpha;
        moveTo(alpha);
        f_alpha = functor(x_alpha);
        f_cache_key = alpha;
        return f_alpha;
    }
    double applyDF(double alpha)
    {
        if (alpha == df_cache_key) return df_alpha;
        moveTo(alpha);
        if (alpha != g_cache_key) { functor.df(x_alpha, g_alpha); g_cache_key = alpha; }
        df_alpha = slope();
        df_cache_key = alpha;
        return df_alpha;
    }
    void applyFDF(double alpha, double& fv, double& dfv)
    {
        if (alpha == f_cache_key && alpha == df_cache_key) { fv = f_alpha; dfv = df_alpha; return; }
        if (alpha == f_cache_key || alpha == g_cache_key || alpha == df_cache_key) { fv = applyF(alpha); dfv = applyDF(alpha); return; }
        moveTo(alpha);
        functor.fdf(x_alpha, f_alpha, g_alpha);
        f_cache_key = alpha;
        g_cache_key = alpha;
        df_alpha = slope();
        df_cache_key = alpha;
        fv = f_alpha;
        dfv = df_alpha;
    }
    void updatePosition(double alpha, double x[6], double& fv, double g[6])
    {
        double fa, dfa;
        applyFDF(alpha, fa, dfa);
        fv = fa;
        copy(x, x_alpha);
        copy(g, g_alpha);
    }
    void changeDirection()
    {
        copy(x_alpha, x0);
        x_cache_key = 0.0;
        f_cache_key = 0.0;
        copy(g_alpha, g0);
        g_cache_key = 0.0;
        df_alpha = slope();
        df_cache_key = 0.0;
    }

    // cubic / quadratic minimiser of the interpolant over [xmin, xmax], coordinates mapped so that [a, b] is [0, 1]
    double interpolate(double a, double fa, double fpa, double b, double fb, double fpb, double xmin, double xmax, int order) const
    {
        double y, ymin = (xmin - a) / (b - a), ymax = (xmax - a) / (b - a);
        if (ymin > ymax) std::swap(ymin, ymax);  // ensure ymin <= ymax
        if (order > 2 && !(fpb != fpb) && fpb != std::numeric_limits<double>::infinity() && fpb != -std::numeric_limits<double>::infinity()) {
            fpa = fpa * (b - a);
            fpb = fpb * (b - a);
            const double eta = 3 * (fb - fa) - 2 * fpa - fpb, xi = fpa + fpb - 2 * (fb - fa);
            const double c0 = fa, c1 = fpa, c2 = eta, c3 = xi;
            auto cubic = [&](double z) { return c0 + z * (c1 + z * (c2 + z * c3)); };
            double zmin = ymin, fmin = cubic(ymin);
            auto checkExtremum = [&](double z) { const double v = cubic(z); if (v < fmin) { zmin = z; fmin = v; } };
            checkExtremum(ymax);
            // roots of the derivative 3 c3 z^2 + 2 c2 z + c1
            const double qa = 3 * c3, qb = 2 * c2, qc = c1;
            double z0 = 0, z1 = 0;
            int    n = 0;
            if (qa == 0) {
                if (qb != 0) { z0 = -qc / qb; n = 1; }
            } else {
                const double disc = qb * qb - 4 * qa * qc;
                if (disc > 0) {
                    if (qb == 0) { const double r = std::sqrt(-qc / qa); z0 = -r; z1 = r; }
                    else {
                        const double sgnb = (qb > 0 ? 1 : -1), temp = -0.5 * (qb + sgnb * std::sqrt(disc)), r1 = temp / qa, r2 = qc / temp;
                        if (r1 < r2) { z0 = r1; z1 = r2; } else { z0 = r2; z1 = r1; }
                    }
                    n = 2;
                } else if (disc == 0) { z0 = -0.5 * qb / qa; z1 = -0.5 * qb / qa; n = 2; }
            }
            if (n == 2) {
                if (z0 > ymin && z0 < ymax) checkExtremum(z0);
                if (z1 > ymin && z1 < ymax) checkExtremum(z1);
            } else if (n == 1) {
                if (z0 > ymin && z0 < ymax) checkExtremum(z0);
            }
            y = zmin;
        } else {
            fpa = fpa * (b - a);
            const double fl = fa + ymin * (fpa + ymin * (fb - fa - fpa)), fh = fa + ymax * (fpa + ymax * (fb - fa - fpa));
            const double c = 2 * (fb - fa - fpa);  // curvature
            double zmin = ymin, fmin = fl;
            if (fh < fmin) { zmin = ymax; fmin = fh; }
            if (c > 0) {  // positive curvature required for a minimum
                const double z = -fpa / c;
                if (z > ymin && z < ymax) {
                    const double fz = fa + z * (fpa + z * (fb - fa - fpa));
                    if (fz < fmin) { zmin = z; fmin = fz; }
                }
            }
            y = zmin;
        }
        return a + y * (b - a);
    }

    BFGSSpace::Status lineSearch(double rho, double sigma, double tau1, double tau2, double tau3, int order, double alpha1, double& alpha_new)
    {
        const double NaN = std::numeric_limits<double>::quiet_NaN();
        double f0, fp0l, falpha, falpha_prev, fpalpha, fpalpha_prev, delta, alpha_next;
        double alpha = alpha1, alpha_prev = 0.0;
        double a, b, fa, fb, fpa, fpb;
        int    i = 0;
        applyFDF(0.0, f0, fp0l);
        falpha_prev = f0;
        fpalpha_prev = fp0l;
        a = 0.0; b = alpha; fa = f0; fb = 0.0; fpa = fp0l; fpb = 0.0;  // avoid uninitialised variables
        while (i++ < parameters.bracket_iters) {  // begin bracketing
            falpha = applyF(alpha);
            if (falpha > f0 + alpha * rho * fp0l || falpha >= falpha_prev) {  // Fletcher's rho test
                a = alpha_prev; fa = falpha_prev; fpa = fpalpha_prev;
                b = alpha; fb = falpha; fpb = NaN;
                break;  // goto sectioning
            }
            fpalpha = applyDF(alpha);
            if (std::fabs(fpalpha) <= -sigma * fp0l) { alpha_new = alpha; return BFGSSpace::Success; }  // Fletcher's sigma test
            if (fpalpha >= 0) {
                a = alpha; fa = falpha; fpa = fpalpha;
                b = alpha_prev; fb = falpha_prev; fpb = fpalpha_prev;
                break;  // goto sectioning
            }
            delta = alpha - alpha_prev;
            alpha_next = interpolate(alpha_prev, falpha_prev, fpalpha_prev, alpha, falpha, fpalpha, alpha + delta, alpha + tau1 * delta, order);
            alpha_prev = alpha;
            falpha_prev = falpha;
            fpalpha_prev = fpalpha;
            alpha = alpha_next;
        }
        while (i++ < parameters.section_iters) {  // sectioning of bracket [a, b]
            delta = b - a;
            alpha = interpolate(a, fa, fpa, b, fb, fpb, a + tau2 * delta, b - tau3 * delta, order);
            falpha = applyF(alpha);
            if ((a - alpha) * fpa <= std::numeric_limits<double>::epsilon()) return BFGSSpace::NoProgress;  // roundoff prevents progress
            if (falpha > f0 + rho * alpha * fp0l || falpha >= fa) {
                b = alpha; fb = falpha; fpb = NaN;  // a_next = a
            } else {
                fpalpha = applyDF(alpha);
                if (std::fabs(fpalpha) <= -sigma * fp0l) { alpha_new = alpha; return BFGSSpace::Success; }  // terminate
                if (((b - a) >= 0 && fpalpha >= 0) || ((b - a) <= 0 && fpalpha <= 0)) { b = a; fb = fa; fpb = fpa; a = alpha; fa = falpha; fpa = fpalpha; }
                else { a = alpha; fa = falpha; fpa = fpalpha; }
            }
        }
        return BFGSSpace::Success;
    }
};

}  // namespace mrgfe
