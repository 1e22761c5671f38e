// tests/sanitize/host_paths_san.cpp — the host-only arithmetic of the product under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build: the GPU
// pool offers no sanitizers).  Drives, with plain g++:
//   * csrc/ndt_ctl.h — the NDT optimiser state machine (Newton + More-Thuente, LU / SVD solves, pose matrices, the float sine / cosine) over a
//     synthetic objective with a known minimum, including NaN / zero / rank-deficient evaluations;
//   * csrc/bfgs.h — the BFGS minimiser of PCL_GICP_HIP over a quadratic and a Rosenbrock-like functor.
// Exit code 0 and no sanitizer report = pass (tests/test_hardening_cpu.py builds and runs it).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>

#include "ndt_ctl.h"
#include "bfgs.h"

using namespace mrgfe;

static int fails = 0;
#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "CHECK failed: %s (line %d)\n", #c, __LINE__); ++fails; } } while (0)

// a smooth objective "score" with maximum at p* (NDT maximises): score = -0.5 (p - p*)^T A (p - p*); the machine sees score, gradient, Hessian of
// -score's negative exactly as computeDerivatives delivers them (g = d score / dp with the reference's sign convention: Newton solves H delta = -g)
struct Quadratic {
    double A[36], pstar[6];
    void eval(const double p[6], double r[44]) const
    {
        double d[6], Ad[6];
        for (int i = 0; i < 6; ++i) d[i] = p[i] - pstar[i];
        for (int i = 0; i < 6; ++i) { Ad[i] = 0; for (int j = 0; j < 6; ++j) Ad[i] += A[i * 6 + j] * d[j]; }
        double s = 0;
        for (int i = 0; i < 6; ++i) s += d[i] * Ad[i];
        r[0] = 0.5 * s;                                   // the reference's "score" enters the line search as phi = -score
        for (int i = 0; i < 6; ++i) r[1 + i] = Ad[i];
        for (int i = 0; i < 36; ++i) r[7 + i] = A[i];
        r[43] = 1000.0;
    }
};

static void run_ctl(int formulation, double eps, unsigned seed, int poison)
{
    std::mt19937 rng(seed);
    std::normal_distribution<double> N(0, 1);
    Quadratic q;
    double B[36];
    for (double& v : B) v = N(rng);
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) { double s = 0; for (int k = 0; k < 6; ++k) s += B[k * 6 + i] * B[k * 6 + j]; q.A[i * 6 + j] = s + (i == j ? 0.5 : 0.0); }
    if (poison == 3) for (int j = 0; j < 6; ++j) { q.A[5 * 6 + j] = q.A[0 * 6 + j]; q.A[j * 6 + 5] = q.A[j * 6 + 0]; }  // rank-deficient: the LU declines, the SVD takes it
    for (int i = 0; i < 6; ++i) q.pstar[i] = (i < 3 ? 0.3 : 0.02) * N(rng);
    NdtCtlState s;
    std::memset(&s, 0, sizeof(s));
    s.step_size = 0.1; s.trans_eps = eps; s.outlier_ratio = 0.55; s.resolution = 1.0f; s.max_iterations = 40; s.search = 2; s.reuse = 1;
    s.formulation = formulation; s.n_src = 1000; s.gauss_d1 = -2.2; s.gauss_d2 = 0.43;
    ctl::identity16(s.final_); ctl::identity16(s.transformation_); ctl::identity16(s.previous_);
    ctl::make_request(s, 0, s.p);
    s.phase = NDT_INIT;
    int evals = 0;
    while (!ctl::done(s) && evals < 4000) {
        double r[44];
        NdtEvalDev e;
        std::memset(&e, 0, sizeof(e));
        ctl::fill_eval(s, e);  // angle tables, transform, mode
        CHECK(e.active == 1 && e.mode >= 0 && e.mode <= 2);
        q.eval(s.req_p, r);
        if (poison == 1 && evals == 3) for (int k = 7; k < 43; ++k) r[k] = std::nan("");  // a NaN Hessian ends the alignment unconverged (or not at all, per rule)
        if (poison == 2 && evals == 2) for (int k = 1; k < 7; ++k) r[k] = 0.0;             // a vanishing gradient: zero step
        ctl::on_result(s, r);
        ++evals;
    }
    CHECK(ctl::done(s));
    CHECK(evals < 4000);
    if (!poison && formulation == 0) {
        // every step is clipped to [eps / 2, 0.1]: the machine stops within a step of the minimum or at the iteration cap
        double d2 = 0;
        for (int i = 0; i < 6; ++i) d2 += (s.p[i] - q.pstar[i]) * (s.p[i] - q.pstar[i]);
        CHECK(s.nr_iterations > 0 && (std::sqrt(d2) < 0.25 || s.nr_iterations >= 40));
    }
    float M[16];
    ctl::pose_to_matrix(s.p, M);
    for (float v : M) CHECK(std::isfinite(v));
}

struct QuadFunctor {
    double c[6] = {1, 4, 0.25, 9, 2, 0.5}, x0[6] = {0.3, -0.2, 0.1, 0.02, -0.01, 0.03};
    int    calls = 0;
    void fdf(const double x[6], double& f, double g[6]) { ++calls; f = 0; for (int k = 0; k < 6; ++k) { const double d = x[k] - x0[k]; f += 0.5 * c[k] * d * d; g[k] = c[k] * d; } }
    void df(const double x[6], double g[6]) { double f; fdf(x, f, g); }
    double operator()(const double x[6]) { double f, g[6]; fdf(x, f, g); return f; }
};

int main()
{
    for (unsigned seed = 1; seed <= 40; ++seed)
        for (int formulation = 0; formulation < 2; ++formulation)
            for (double eps : {0.1, 1e-3, 1e-6})
                for (int poison = 0; poison < 4; ++poison) run_ctl(formulation, eps, seed, poison);
    // solves: LU against SVD on well-conditioned matrices, the SVD alone on singular / non-finite ones
    std::mt19937 rng(7);
    std::normal_distribution<double> N(0, 1);
    for (int t = 0; t < 2000; ++t) {
        double A[36], b[6], x1[6], x2[6];
        for (double& v : A) v = N(rng);
        for (int i = 0; i < 6; ++i) A[i * 7] += 4.0;
        for (double& v : b) v = N(rng);
        const bool ok = ctl::lu_solve6(A, b, x1);
        ctl::svd_solve6(A, b, x2);
        if (ok) for (int k = 0; k < 6; ++k) CHECK(std::fabs(x1[k] - x2[k]) <= 1e-9 * (1 + std::fabs(x2[k])));
        double Z[36] = {0};
        CHECK(!ctl::lu_solve6(Z, b, x1));
        ctl::svd_solve6(Z, b, x2);
        for (int k = 0; k < 6; ++k) CHECK(x2[k] == 0);
        A[7] = std::nan("");
        CHECK(!ctl::lu_solve6(A, b, x1));
        ctl::svd_solve6(A, b, x2);
        CHECK(x2[0] != x2[0]);
    }
    for (float a = -130.0f; a < 130.0f; a += 0.37f) { CHECK(std::fabs(ctl::sin_f(a) - std::sin(a)) < 1e-6f); CHECK(std::fabs(ctl::cos_f(a) - std::cos(a)) < 1e-6f); }
    // BFGS
    QuadFunctor f;
    BFGS<QuadFunctor> bfgs(f);
    double x[6] = {0, 0, 0, 0, 0, 0};
    bfgs.minimizeInit(x);
    int it = 0;
    BFGSSpace::Status st = BFGSSpace::Running;
    do { st = bfgs.minimizeOneStep(x); ++it; } while (st == BFGSSpace::Success && it < 100);
    for (int k = 0; k < 6; ++k) CHECK(std::fabs(x[k] - f.x0[k]) < 1e-4);
    std::printf("host paths under ASan + UBSan: %d failed checks, %d BFGS steps\n", fails, it);
    return fails ? 1 : 0;
}
