// tests/adapter_stub/adapter_main.cpp — the reference's call sequences made through a pcl::Registration BASE pointer against
// mrgfe_pcl::HipRegistration (built against the stand-in headers of this directory, linked with libmrgfe.so, run on the GPU):
//   LoopDetector::matching            /root/reference/src/mrg_slam/loop_detector.cpp:104,127-144
//   publish_scan_matching_status      /root/reference/apps/scan_matching_odometry_component.cpp:403-417
// Prints one line per check; exit code 0 iff all hold.  Clouds: two perturbed copies of a seeded random "room".
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>

#include <mrgfe_pcl_adapter.hpp>

using PointT = pcl::PointXYZI;
using Cloud = pcl::PointCloud<PointT>;

static Cloud::Ptr make_room(unsigned seed, int n, float shift)
{
    std::mt19937 rng(seed);
    std::uniform_real_distribution<float> u(-1.f, 1.f);
    Cloud::Ptr c(new Cloud);
    c->resize(n);
    for (int i = 0; i < n; ++i) {
        PointT& p = (*c)[i];
        const int face = i % 5;  // floor and four walls of a 20 x 14 x 3 m room, 1 cm of noise
        const float a = u(rng), b = u(rng), e = 0.01f * u(rng);
        if (face == 0)      { p.x = 10 * a; p.y = 7 * b; p.z = -1.5f + e; }
        else if (face == 1) { p.x = 10 * a; p.y = 7 + e; p.z = 1.5f * b; }
        else if (face == 2) { p.x = 10 * a; p.y = -7 + e; p.z = 1.5f * b; }
        else if (face == 3) { p.x = 10 + e; p.y = 7 * a; p.z = 1.5f * b; }
        else                { p.x = -10 + e; p.y = 7 * a; p.z = 1.5f * b; }
        p.x += shift;
        p.intensity = 0.5f + 0.4f * u(rng);
    }
    return c;
}

static int failures = 0;
static void check(bool ok, const char* what)
{
    std::printf("%s %s\n", ok ? "ok  " : "FAIL", what);
    if (!ok) ++failures;
}

int main(int argc, char** argv)
{
    const int n = argc > 1 ? std::atoi(argv[1]) : 6000;
    mrgfe_reg_params prm;
    mrgfe_reg_default_params(MRGFE_NDT_HIP, &prm);
    prm.transformation_epsilon = 0.01;
    prm.resolution = 1.0;
    // the factory hands out the BASE pointer (registrations.hpp:20)
    pcl::Registration<PointT, PointT>::Ptr registration(new mrgfe_pcl::HipRegistration<PointT, PointT>(prm));
    auto* hip = static_cast<mrgfe_pcl::HipRegistration<PointT, PointT>*>(registration.get());

    Cloud::Ptr target = make_room(1, n, 0.0f), source = make_room(2, n, 0.3f);
    const int builds0 = pcl::search::KdTree<PointT>::builds();
    registration->setInputTarget(target);                      // loop_detector.cpp:104
    registration->setInputSource(source);                      // :127
    Cloud::Ptr aligned(new Cloud);
    registration->align(*aligned, Eigen::Matrix4f::Identity());  // :134
    check(pcl::search::KdTree<PointT>::builds() == builds0, "initCompute() built no CPU kd-tree over the target");
    const double score = registration->getFitnessScore(std::numeric_limits<double>::max());  // :137, PCL's non-virtual method
    double direct = -1;
    mrgfe_reg_fitness(hip->handle(), std::numeric_limits<double>::max(), &direct);
    std::printf("     fitness through the base pointer %.17g, mrgfe_reg_fitness %.17g, converged %d\n", score, direct, int(registration->hasConverged()));
    check(score == direct, "base-class getFitnessScore == mrgfe_reg_fitness bit for bit");
    check(hip->gpuSearch().batched_answers() == std::size_t(n) && hip->gpuSearch().single_queries() == 0, "all N queries served by the one batched GPU pass");
    check(!hip->gpuSearch().flann_built(), "no lazy FLANN fallback was needed");
    const double bounded = registration->getFitnessScore(0.01);
    mrgfe_reg_fitness(hip->handle(), 0.01, &direct);
    check(bounded == direct, "getFitnessScore(max_range = 0.01) agrees as well");
    float Tf[16];
    mrgfe_reg_final_transformation(hip->handle(), Tf);
    const Eigen::Matrix4f T = registration->getFinalTransformation();  // :144
    bool same = true;
    for (int i = 0; i < 16; ++i) same = same && T.data()[i] == Tf[i];
    check(same, "getFinalTransformation == mrgfe_reg_final_transformation");
    check(std::fabs(T(0, 3) + 0.3f) < 0.05f, "the 0.3 m shift was recovered");
    // intensity of the 32-byte records reached the device: the aligned cloud carries it
    check((*aligned)[n / 2].intensity == (*source)[n / 2].intensity, "aligned cloud keeps the source intensities");

    // scan_matching_odometry_component.cpp:405-417: inlier fraction through getSearchMethodTarget()
    const std::size_t before = hip->gpuSearch().batched_answers();
    int                num_inliers = 0;
    pcl::Indices       k_indices;
    std::vector<float> k_sq_dists;
    for (std::size_t i = 0; i < aligned->size(); i++) {
        const auto& pt = aligned->at(i);
        registration->getSearchMethodTarget()->nearestKSearch(pt, 1, k_indices, k_sq_dists);
        if (k_sq_dists[0] < 0.5 * 0.5) num_inliers++;
    }
    std::vector<int32_t> idx(n);
    std::vector<float>   sqd(n), q(4 * n);
    for (int i = 0; i < n; ++i) { q[4 * i] = (*aligned)[i].x; q[4 * i + 1] = (*aligned)[i].y; q[4 * i + 2] = (*aligned)[i].z; q[4 * i + 3] = 0; }
    mrgfe_reg_nn1_target(hip->handle(), q.data(), n, 16, idx.data(), sqd.data());
    int exp_inliers = 0;
    for (int i = 0; i < n; ++i) exp_inliers += sqd[i] < 0.25f;
    check(num_inliers == exp_inliers && hip->gpuSearch().batched_answers() == before + n, "status loop: inliers from the batched pass");
    // a query nobody predicted still gets the exact answer (single GPU query), checked against brute force
    PointT odd;
    odd.x = 1.234f; odd.y = -2.5f; odd.z = 0.7f;
    registration->getSearchMethodTarget()->nearestKSearch(odd, 1, k_indices, k_sq_dists);
    pcl::search::KdTree<PointT> brute;
    brute.setInputCloud(target);
    pcl::Indices bi; std::vector<float> bd;
    brute.nearestKSearch(odd, 1, bi, bd);
    check(k_sq_dists[0] == bd[0] && hip->gpuSearch().single_queries() == 1, "unpredicted query: exact 1-NN distance from a single GPU query");
    // second target: the search object follows setInputTarget
    registration->setInputTarget(source);
    registration->align(*aligned, Eigen::Matrix4f::Identity());
    const double self = registration->getFitnessScore();
    check(self < 1e-6, "source aligned onto itself after setInputTarget(source): fitness ~ 0");
    std::printf("%s\n", failures ? "ADAPTER CHECK FAILED" : "adapter check passed");
    return failures ? 1 : 0;
}
