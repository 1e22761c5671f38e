// tests/sanitize/oracle_san.cpp — the CPU oracle (oracle/*.cpp, test infrastructure) under AddressSanitizer + UndefinedBehaviorSanitizer: every
// restated algorithm once on a small structured cloud, including the empty / degenerate inputs the GPU tests feed it.  tests/test_hardening_cpu.py
// builds it with the oracle's sources and runs it.
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

#include "filters.h"
#include "gicp.h"
#include "mapcloud.h"
#include "ndt.h"
#include "pcl_gicp.h"
#include "pcl_ndt.h"

using namespace orc;

static std::vector<float> cloud(int n, unsigned seed)
{
    std::mt19937 rng(seed);
    std::uniform_real_distribution<float> U(-1, 1);
    std::normal_distribution<float> N(0, 0.02f);
    std::vector<float> c(static_cast<size_t>(n) * 4);
    for (int i = 0; i < n; ++i) {
        const int kind = i % 4;
        float x = 15 * U(rng), y = 10 * U(rng), z = 2 * U(rng);
        if (kind < 2) z = -1.73f + N(rng);
        else if (kind == 2) y = 9.6f + N(rng);
        c[4 * i] = x; c[4 * i + 1] = y; c[4 * i + 2] = z; c[4 * i + 3] = 0.5f * (1 + U(rng));
    }
    return c;
}

int main()
{
    const int n = 3000;
    std::vector<float> tgt = cloud(n, 1), src(tgt.begin(), tgt.begin() + 4 * 2200);
    for (size_t i = 0; i < src.size(); i += 4) { src[i] += 0.2f; src[i + 1] -= 0.1f; }
    float I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    std::vector<float> out(tgt.size()), out2(tgt.size());
    std::vector<unsigned char> keep(n);
    int m = 0;
    // filters
    int k = distance_filter(tgt.data(), n, 0.1, 35.0, out.data());
    voxelgrid(out.data(), k, 0.1f, 1, 0, out2.data(), &m);
    radius_outlier(out2.data(), m, 0.5, 2, out.data(), keep.data());
    statistical_outlier(out2.data(), m, 30, 1.2, out.data(), keep.data());
    approx_voxelgrid(tgt.data(), n, 0.5f, out.data());
    voxelgrid(tgt.data(), 0, 0.1f, 1, 0, out2.data(), &m);
    // NDT, every neighbourhood, and the f64 PCL class
    for (int s = 0; s < 4; ++s) {
        Ndt a;
        a.search = static_cast<NdtSearch>(s);
        a.trans_eps = 0.01; a.max_iterations = 30; a.num_threads = 2;
        a.set_target(tgt.data(), n);
        a.set_source(src.data(), 2200);
        a.align(I, out.data());
        (void)a.fitness(1e300);
        a.gpu_order_ppt = 1;
        a.align(I, nullptr);
    }
    {
        PclNdt p;
        p.trans_eps = 1e-5; p.max_iterations = 20;
        p.set_target(tgt.data(), n);
        p.set_source(src.data(), 2200);
        p.align(I, out.data());
        p.gpu_order = 2; p.num_threads = 2;
        p.align(I, nullptr);
        PclNdt e;  // no target, empty source
        e.set_source(src.data(), 0);
        e.align(I, nullptr);
        e.set_target(tgt.data(), 0);
    }
    // GICP family
    for (int v = 0; v < 4; ++v) {
        FastGicp g;
        g.variant = v; g.trans_eps = 0.01; g.num_threads = 2; g.max_iterations = 12;
        g.set_target(tgt.data(), n);
        g.set_source(src.data(), 2200);
        g.align(I, nullptr);
        (void)g.fitness(1e300);
    }
    for (int omp = 0; omp < 2; ++omp) {
        PclGicp g;
        g.whole_gradient_norm = omp; g.max_iterations = 6; g.num_threads = 2;
        g.set_target(tgt.data(), n);
        g.set_source(src.data(), 2200);
        g.align(I, nullptr);
    }
    // map cloud, other-robot removal, deskewing
    const float* clouds[2] = {tgt.data(), src.data()};
    const int counts[2] = {n, 2200};
    const double poses[32] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0.5, -0.2, 0, 1};
    const unsigned char first[2] = {1, 0};
    std::vector<float> map(static_cast<size_t>(n + 2200) * 4);
    map_cloud_generate(2, clouds, counts, poses, first, 0.25f, 1, 100.0f, 0, map.data(), &m);
    const float centres[3] = {0.5f, 0.5f, 0.0f};
    int n_removed = 0;
    remove_points_near(tgt.data(), n, centres, 1, 4.0f, out.data(), out2.data(), &n_removed);
    const float w[3] = {0.1f, -0.2f, 0.3f};
    deskew(tgt.data(), n, w, 0.1, out.data());
    std::printf("oracle under ASan + UBSan: done (%d map points, %d removed)\n", m, n_removed);
    return 0;
}
