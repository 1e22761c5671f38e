// tests/adapter_stub/adapter_main.cpp — the reference's call sequences made through a pcl::Registration BASE pointer against
// mrgfe_pcl::HipRegistration (built against the stand-in headers of this directory, linked with libmrgfe.so, run on the GPU):
//   LoopDetector::matching            /root/reference/src/mrg_slam/loop_detector.cpp:104,127-144
//   publish_scan_matching_status      /root/reference/apps/scan_matching_odometry_component.cpp:403-417
//   PrefilteringComponent::downsample / outlier_removal   /root/reference/apps/prefiltering_component.cpp:37,54-55,168-171,189-199
//   (mrgfe_pcl::HipVoxelGrid / HipRadiusOutlierRemoval / HipStatisticalOutlierRemoval held in the reference's own member types)
// Prints one line per check; exit code 0 iff all hold.  Clouds: two perturbed copies of a seeded random "room".
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <vector>
#include <random>

#include <mrgfe_pcl_adapter.hpp>
#include <mrgfe_pcl_filters.hpp>

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

    // the two components' registrations on their own contexts (the odometry one at the highest stream priority, the loop detector's
    // confined to a part of the chip): distinct HIP streams, the same transformation bit for bit
    {
        pcl::Registration<PointT, PointT>::Ptr odo(new mrgfe_pcl::HipRegistration<PointT, PointT>(prm, 0, mrgfe_pcl::ContextRole::Odometry));
        pcl::Registration<PointT, PointT>::Ptr lc(new mrgfe_pcl::HipRegistration<PointT, PointT>(prm, 0, mrgfe_pcl::ContextRole::LoopClosure));
        check(mrgfe_pcl::shared_context(0, mrgfe_pcl::ContextRole::Odometry) != mrgfe_pcl::shared_context(0, mrgfe_pcl::ContextRole::LoopClosure) &&
                  mrgfe_pcl::shared_context(0, mrgfe_pcl::ContextRole::Odometry) != mrgfe_pcl::shared_context(),
              "one context per role");
        bool same_T = true;
        for (auto& r : {odo, lc}) {
            Cloud::Ptr out(new Cloud);
            r->setInputTarget(target);
            r->setInputSource(source);
            r->align(*out, Eigen::Matrix4f::Identity());
            const Eigen::Matrix4f Tr = r->getFinalTransformation(), T0 = registration->getFinalTransformation();
            for (int i = 0; i < 16; ++i) same_T = same_T && Tr.data()[i] == T0.data()[i];
        }
        check(same_T, "odometry-role and loop-closure-role registrations return the general one's transformation");
    }

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
    // a source that is not dense: PCL's getFitnessScore skips its non-finite points and never asks the search object about them
    {
        Cloud::Ptr holes(new Cloud(*make_room(2, n, 0.3f)));
        holes->is_dense = false;
        for (int i : {0, 7, 8, n / 2, n - 1}) (*holes)[i].x = std::numeric_limits<float>::quiet_NaN();
        registration->setInputTarget(target);
        registration->setInputSource(holes);
        registration->align(*aligned, Eigen::Matrix4f::Identity());
        const std::size_t b0 = hip->gpuSearch().batched_answers(), s0 = hip->gpuSearch().single_queries();
        const double sc = registration->getFitnessScore();
        double       dir = -1;
        mrgfe_reg_fitness(hip->handle(), std::numeric_limits<double>::max(), &dir);
        check(sc == dir, "non-dense source: base-class getFitnessScore == mrgfe_reg_fitness");
        check(hip->gpuSearch().batched_answers() == b0 + std::size_t(n - 5) && hip->gpuSearch().single_queries() == s0, "non-dense source: the skipped points cost no single-point round trips");
    }

    // ---- prefiltering_component.cpp: the filters, held in the reference's member types and driven by its call sequences ----
    {
        using Filter = pcl::Filter<PointT>;
        std::shared_ptr<pcl::VoxelGrid<PointT>>                 voxelgrid_filter_ = std::make_shared<mrgfe_pcl::HipVoxelGrid<PointT>>();                                   // :37
        std::shared_ptr<pcl::StatisticalOutlierRemoval<PointT>> statistical_outlier_removal_filter_ = std::make_shared<mrgfe_pcl::HipStatisticalOutlierRemoval<PointT>>();  // :54
        std::shared_ptr<pcl::RadiusOutlierRemoval<PointT>>      radius_outlier_removal_filter_ = std::make_shared<mrgfe_pcl::HipRadiusOutlierRemoval<PointT>>();            // :55
        const int  cpu0 = Filter::cpu_calls();
        Cloud::ConstPtr cloud = make_room(5, n, 0.0f);
        auto packed = [](const Cloud& c) {
            std::vector<float> v(4 * c.size());
            for (std::size_t i = 0; i < c.size(); ++i) { v[4 * i] = c[i].x; v[4 * i + 1] = c[i].y; v[4 * i + 2] = c[i].z; v[4 * i + 3] = c[i].intensity; }
            return v;
        };
        auto same_as = [&](const Cloud& got, const std::vector<float>& exp, std::size_t m) {
            if (got.size() != m || got.width != m || got.height != 1) return false;
            const std::vector<float> g = packed(got);
            return std::memcmp(g.data(), exp.data(), 16 * m) == 0;
        };
        mrgfe_ctx* ctx = mrgfe_pcl::shared_context();
        const std::vector<float> in = packed(*cloud);
        std::vector<float> exp(in.size());
        std::size_t m = 0;
        int         overflow = 0;
        // downsample(), :168-171
        Cloud::Ptr filtered(new Cloud());
        const double downsample_resolution = 0.25;
        voxelgrid_filter_->setLeafSize(downsample_resolution, downsample_resolution, downsample_resolution);
        voxelgrid_filter_->setMinimumPointsNumberPerVoxel(2);
        voxelgrid_filter_->setInputCloud(cloud);
        voxelgrid_filter_->filter(*filtered);
        mrgfe_voxelgrid(ctx, in.data(), cloud->size(), 16, 0.25f, 2, exp.data(), &m, &overflow);
        check(m > 0 && m < cloud->size() && same_as(*filtered, exp, m), "pcl::VoxelGrid::filter through HipVoxelGrid == mrgfe_voxelgrid on the packed cloud");
        check((*filtered)[0].data3 == 1.0f && filtered->is_dense, "voxel grid output: PCL's padding word and is_dense");
        // outlier_removal(), RADIUS :195-198
        Cloud::Ptr kept(new Cloud());
        radius_outlier_removal_filter_->setRadiusSearch(0.4);
        radius_outlier_removal_filter_->setMinNeighborsInRadius(3);
        radius_outlier_removal_filter_->setInputCloud(filtered);
        radius_outlier_removal_filter_->filter(*kept);
        const std::vector<float> fin = packed(*filtered);
        std::size_t mr = 0;
        mrgfe_radius_outlier(ctx, fin.data(), filtered->size(), 16, 0.4, 3, exp.data(), &mr);
        check(mr > 0 && same_as(*kept, exp, mr), "pcl::RadiusOutlierRemoval::filter through HipRadiusOutlierRemoval == mrgfe_radius_outlier");
        // outlier_removal(), STATISTICAL :189-192
        Cloud::Ptr kept2(new Cloud());
        statistical_outlier_removal_filter_->setMeanK(12);
        statistical_outlier_removal_filter_->setStddevMulThresh(1.0);
        statistical_outlier_removal_filter_->setInputCloud(filtered);
        statistical_outlier_removal_filter_->filter(*kept2);
        std::size_t ms = 0;
        mrgfe_statistical_outlier(ctx, fin.data(), filtered->size(), 16, 12, 1.0, exp.data(), &ms);
        check(ms > 0 && ms < filtered->size() && same_as(*kept2, exp, ms), "pcl::StatisticalOutlierRemoval::filter through HipStatisticalOutlierRemoval == mrgfe_statistical_outlier");
        check(Filter::cpu_calls() == cpu0, "none of the three filter() calls ran PCL's CPU code");
        // filtering in place (output is the input) and through the pcl::Filter base pointer
        std::shared_ptr<Filter> base = voxelgrid_filter_;
        Cloud::Ptr inplace(new Cloud(*cloud));
        base->setInputCloud(inplace);
        base->filter(*inplace);
        check(same_as(*inplace, packed(*filtered), filtered->size()), "pcl::Filter base pointer, in place: same voxel grid output");
        // downsample(), APPROX_VOXELGRID :173-175
        {
            std::shared_ptr<pcl::ApproximateVoxelGrid<PointT>> approx_voxelgrid_filter_ = std::make_shared<mrgfe_pcl::HipApproximateVoxelGrid<PointT>>();  // :38
            Cloud::Ptr approx(new Cloud());
            approx_voxelgrid_filter_->setLeafSize(downsample_resolution, downsample_resolution, downsample_resolution);
            approx_voxelgrid_filter_->setInputCloud(cloud);
            approx_voxelgrid_filter_->filter(*approx);
            std::size_t ma = 0;
            mrgfe_approx_voxelgrid(ctx, in.data(), cloud->size(), 16, 0.25f, exp.data(), &ma);
            check(ma > 0 && ma < cloud->size() && same_as(*approx, exp, ma), "pcl::ApproximateVoxelGrid::filter through HipApproximateVoxelGrid == mrgfe_approx_voxelgrid on the packed cloud");
            check(!approx->is_dense && Filter::cpu_calls() == cpu0, "approximate voxel grid output: is_dense = false, PCL's CPU loop did not run");
        }
        // what the GPU path does not offer stays PCL's: unequal leaf sizes
        voxelgrid_filter_->setLeafSize(0.25f, 0.5f, 0.25f);
        voxelgrid_filter_->setInputCloud(cloud);
        voxelgrid_filter_->filter(*filtered);
        check(Filter::cpu_calls() == cpu0 + 1, "unequal leaf sizes fall back to PCL's own applyFilter");
    }
    std::printf("%s\n", failures ? "ADAPTER CHECK FAILED" : "adapter check passed");
    return failures ? 1 : 0;
}
