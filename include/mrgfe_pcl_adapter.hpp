// include/mrgfe_pcl_adapter.hpp — header-only adapter that makes libmrgfe.so a pcl::Registration, i.e. the thing
// mrg_slam::select_registration_method returns (/root/reference/include/mrg_slam/registrations.hpp:20).
//
// It occupies the slot the reference already has for a GPU back end (FAST_VGICP_CUDA:
// /root/reference/src/mrg_slam/registrations.cpp:19-21,65-75; CMakeLists.txt:45-49).  PCL is not installed in the
// build container, so this file ships as source and is compiled only where <pcl/registration/registration.h> exists
// (INTEGRATION.md shows the factory patch).  It uses nothing but the C ABI of include/mrgfe.h.
//
// What the reference calls through the BASE pointer, and how each call reaches the GPU:
//   setInputTarget / setInputSource   virtual in pcl::Registration -> overridden here (clouds go up in their 32-byte
//                                     pcl::PointXYZI layout, gathered on the device: no host repacking)
//   align(output, guess)              non-virtual; calls initCompute() and the virtual computeTransformation() -> overridden
//   hasConverged / getFinalTransformation   read converged_ / final_transformation_, which computeTransformation fills
//   getFitnessScore(max_range)        NON-virtual in PCL (loop_detector.cpp:137, scan_matching_odometry_component.cpp:403):
//                                     it transforms the source by final_transformation_ and asks tree_ for the 1-NN of every
//                                     point in turn.  tree_ is replaced (setSearchMethodTarget(tree, force_no_recompute =
//                                     true)) by GpuTargetSearch below, whose nearestKSearch answers those N sequential queries
//                                     from ONE batched GPU pass (mrgfe_reg_nn1_target over final_transformation_ * source), so
//                                     PCL's own loop sums GPU-computed distances — and initCompute() never builds a FLANN
//                                     kd-tree over the target.
//   getSearchMethodTarget()->nearestKSearch(pt, 1, ..)   the status loop (scan_matching_odometry_component.cpp:405-417)
//                                     walks the aligned cloud in order: served by the same batch.
// Callers that hold the derived pointer can use getFitnessScore() of this class (it hides the base method): one GPU call, no
// per-point host loop at all.
#pragma once
#if __has_include(<pcl/registration/registration.h>)

#include <pcl/point_types.h>
#include <pcl/registration/registration.h>
#include <pcl/search/kdtree.h>

#include <array>
#include <cmath>
#include <cstddef>
#include <cstring>
#include <limits>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "mrgfe.h"

namespace mrgfe_pcl {

#ifndef MRGFE_PCL_SHARED_CONTEXT
#define MRGFE_PCL_SHARED_CONTEXT
// One context (HIP stream + workspaces) per process, GPU and ROLE, shared by the adapter objects of that role: calls on one context are
// serialised, so the odometry registration (scan_matching_odometry_component, every scan) must not share one with the loop-closure
// registration (mrg_slam_component's LoopDetector, batches of candidates) when both components live in one container process.
//   General     : mrgfe_ctx_create
//   Odometry    : mrgfe_ctx_create_priority — its small launches are dispatched ahead of a batch's as slots free up
//   LoopClosure : mrgfe_ctx_create_reserving(MRGFE_RESERVE_AUTO) — its kernels leave a quarter of the chip (at most 64 compute units) to the
//                 odometry contexts; on a device too small to split (or if the masked stream cannot be made) a plain context: the node keeps running
enum class ContextRole { General = 0, Odometry = 1, LoopClosure = 2 };
inline mrgfe_ctx* shared_context(int device = 0, ContextRole role = ContextRole::General)
{
    static std::mutex                              mu;
    static std::map<std::pair<int, int>, mrgfe_ctx*> ctxs;
    std::lock_guard<std::mutex> lock(mu);
    const std::pair<int, int> key(device, static_cast<int>(role));
    auto it = ctxs.find(key);
    if (it != ctxs.end()) return it->second;
    mrgfe_ctx* ctx = nullptr;
    int st = role == ContextRole::Odometry ? mrgfe_ctx_create_priority(device, 1, &ctx)
             : role == ContextRole::LoopClosure ? mrgfe_ctx_create_reserving(device, MRGFE_RESERVE_AUTO, &ctx) : mrgfe_ctx_create(device, &ctx);
    if (st != MRGFE_OK && role == ContextRole::LoopClosure) st = mrgfe_ctx_create(device, &ctx);  // degrade to an unmasked context rather than stop the node
    if (st != MRGFE_OK) throw std::runtime_error(std::string("mrgfe: ") + mrgfe_last_error());
    ctxs[key] = ctx;
    return ctx;
}
// stride_bytes descriptor of a PCL point type with x, y, z and intensity members (pcl::PointXYZI: 32 bytes, intensity at 16);
// member addresses of an object instead of offsetof: PCL's point types are not standard-layout (-Winvalid-offsetof)
template <typename PointT>
inline std::size_t point_layout()
{
    static const PointT p{};
    const char* base = reinterpret_cast<const char*>(&p);
    return MRGFE_LAYOUT(sizeof(PointT), static_cast<std::size_t>(reinterpret_cast<const char*>(&p.x) - base), static_cast<std::size_t>(reinterpret_cast<const char*>(&p.intensity) - base));
}
#endif

// The target "kd-tree" of a HipRegistration: a pcl::search::KdTree whose FLANN index is never built.  1-NN queries that walk
// final_transformation * source in order (pcl::Registration::getFitnessScore, the scan-matching status loop) are answered from
// one batched GPU pass; any other use (k > 1, radius search, 1-NN queries of unrelated points) falls back to a single GPU query
// or — for the searches the ABI does not offer — to the real FLANN tree, built lazily on first use.
template <typename PointT>
class GpuTargetSearch : public pcl::search::KdTree<PointT> {
   public:
    using Base = pcl::search::KdTree<PointT>;
    using PointCloudConstPtr = typename Base::PointCloudConstPtr;
    using IndicesConstPtr = typename Base::IndicesConstPtr;

    explicit GpuTargetSearch(mrgfe_reg* reg) : reg_(reg) {}

    // initCompute() / setInputTarget never reach FLANN: remember the cloud for the lazy fallback only.
    // (pcl::search::KdTree::setInputCloud returns void up to PCL 1.12 — ROS 2 Humble — and bool from 1.13 on — Jazzy.)
#if defined(PCL_VERSION_COMPARE)
#if PCL_VERSION_COMPARE(>=, 1, 13, 0)
#define MRGFE_PCL_SET_INPUT_RETURNS_BOOL 1
#endif
#endif
#ifdef MRGFE_PCL_SET_INPUT_RETURNS_BOOL
    bool setInputCloud(const PointCloudConstPtr& cloud, const IndicesConstPtr& indices = IndicesConstPtr()) override
    {
        cloud_ = cloud;
        indices_ = indices;
        flann_built_ = false;
        return true;
    }
#else
    void setInputCloud(const PointCloudConstPtr& cloud, const IndicesConstPtr& indices = IndicesConstPtr()) override
    {
        cloud_ = cloud;
        indices_ = indices;
        flann_built_ = false;
    }
#endif
    PointCloudConstPtr getInputCloud() const override { return cloud_; }

    // the registration hands over final_transformation * source (packed xyzi, n points) after every align()
    void expect_queries(const float* xyzi, std::size_t n)
    {
        if (n) expected_.assign(xyzi, xyzi + 4 * n);
        else   expected_.clear();
        answered_ = false;
        cursor_ = 0;
    }
    std::size_t batched_answers() const { return n_batched_; }
    std::size_t single_queries() const { return n_single_; }
    bool        flann_built() const { return flann_built_; }

    int nearestKSearch(const PointT& point, int k, pcl::Indices& k_indices, std::vector<float>& k_sqr_distances) const override
    {
        if (k != 1) return fallback().nearestKSearch(point, k, k_indices, k_sqr_distances);
        k_indices.resize(1);
        k_sqr_distances.resize(1);
        const std::size_t n = expected_.size() / 4;
        // pcl::Registration::getFitnessScore skips the non-finite points of a source that is not dense (`if (!input_->is_dense &&
        // !pcl::isXYZFinite(point)) continue;`): it never asks about them, so neither do we wait for them
        auto finite_at = [&](std::size_t c) { return std::isfinite(expected_[4 * c]) && std::isfinite(expected_[4 * c + 1]) && std::isfinite(expected_[4 * c + 2]); };
        if (std::isfinite(point.x) && std::isfinite(point.y) && std::isfinite(point.z))
            while (cursor_ < n && !finite_at(cursor_)) ++cursor_;
        // sequential walk over the expected queries: position `cursor_`, else the start (a second pass over the same cloud)
        for (int attempt = 0; attempt < 2 && n; ++attempt) {
            std::size_t c = attempt == 0 ? cursor_ : 0;
            if (attempt == 1)
                while (c < n && !finite_at(c) && std::isfinite(point.x)) ++c;  // a second pass starts at the first point PCL asks about
            if (c < n && std::memcmp(&expected_[4 * c], &point.x, 12) == 0) {
                if (!answered_) answer_batch();
                cursor_ = c + 1;
                ++n_batched_;
                k_indices[0] = idx_[c];
                k_sqr_distances[0] = sqd_[c];
                return idx_[c] >= 0 ? 1 : 0;
            }
        }
        // Not bit-equal to the point we predicted.  PCL's getFitnessScore transforms the source on the host
        // (pcl::transformPointCloud: SSE, AVX or scalar code depending on how PCL was built), so its queries may differ from
        // the GPU's transform in the last bits.  Such a query keeps the batched neighbour; only its distance is recomputed
        // here against the actual query point, in FLANN's L2_Simple order (x, y, z; float).
        if (cursor_ < n && cloud_) {
            const float* e = &expected_[4 * cursor_];
            const float  tol = 2e-6f * (std::fabs(e[0]) + std::fabs(e[1]) + std::fabs(e[2]) + 1.0f);
            if (std::fabs(point.x - e[0]) <= tol && std::fabs(point.y - e[1]) <= tol && std::fabs(point.z - e[2]) <= tol) {
                if (!answered_) answer_batch();
                const std::size_t c = cursor_++;
                ++n_batched_;
                k_indices[0] = idx_[c];
                if (idx_[c] < 0) { k_sqr_distances[0] = std::numeric_limits<float>::quiet_NaN(); return 0; }
                const PointT& t = (*cloud_)[static_cast<std::size_t>(idx_[c])];
                const float   dx = t.x - point.x, dy = t.y - point.y, dz = t.z - point.z;
                float         d = dx * dx;
                d += dy * dy;
                d += dz * dz;
                k_sqr_distances[0] = d;
                return 1;
            }
        }
        // an unrelated point: one GPU query (exact, but a round trip per point)
        const float q[4] = {point.x, point.y, point.z, 0.0f};
        int32_t     i = -1;
        float       d = std::numeric_limits<float>::quiet_NaN();
        ++n_single_;
        if (mrgfe_reg_nn1_target(reg_, q, 1, 16, &i, &d) != MRGFE_OK) i = -1;
        k_indices[0] = i;
        k_sqr_distances[0] = i >= 0 ? d : std::numeric_limits<float>::quiet_NaN();
        return i >= 0 ? 1 : 0;
    }

    int radiusSearch(const PointT& point, double radius, pcl::Indices& k_indices, std::vector<float>& k_sqr_distances, unsigned int max_nn = 0) const override
    {
        return fallback().radiusSearch(point, radius, k_indices, k_sqr_distances, max_nn);
    }

   private:
    void answer_batch() const
    {
        const std::size_t n = expected_.size() / 4;
        idx_.assign(n, -1);
        sqd_.assign(n, std::numeric_limits<float>::quiet_NaN());
        if (n && mrgfe_reg_nn1_target(reg_, expected_.data(), n, 16, idx_.data(), sqd_.data()) != MRGFE_OK)
            PCL_ERROR("[mrgfe_pcl::GpuTargetSearch] %s\n", mrgfe_last_error());
        for (std::size_t i = 0; i < n; ++i)
            if (idx_[i] < 0) sqd_[i] = std::numeric_limits<float>::quiet_NaN();  // non-finite query: compares false against any max_range
        answered_ = true;
    }
    // searches the C ABI has no entry point for: the real pcl::search::KdTree over the same cloud, built on first use
    const Base& fallback() const
    {
        if (!flann_built_) {
            const_cast<GpuTargetSearch*>(this)->Base::setInputCloud(cloud_, indices_);
            flann_built_ = true;
        }
        return *this;
    }

    mrgfe_reg*         reg_;
    PointCloudConstPtr cloud_;
    IndicesConstPtr    indices_;
    std::vector<float> expected_;
    mutable std::vector<int32_t> idx_;
    mutable std::vector<float>   sqd_;
    mutable bool        answered_ = false, flann_built_ = false;
    mutable std::size_t cursor_ = 0, n_batched_ = 0, n_single_ = 0;
};

template <typename PointSource = pcl::PointXYZI, typename PointTarget = pcl::PointXYZI>
class HipRegistration : public pcl::Registration<PointSource, PointTarget, float> {
   public:
    using Base = pcl::Registration<PointSource, PointTarget, float>;
    using Ptr = pcl::shared_ptr<HipRegistration<PointSource, PointTarget>>;
    using PointCloudSource = typename Base::PointCloudSource;
    using PointCloudSourceConstPtr = typename Base::PointCloudSourceConstPtr;
    using PointCloudTargetConstPtr = typename Base::PointCloudTargetConstPtr;
    using Matrix4 = typename Base::Matrix4;
    using Search = GpuTargetSearch<PointTarget>;

    explicit HipRegistration(const mrgfe_reg_params& params, int device = 0, ContextRole role = ContextRole::General)
    {
        static const char* const names[] = {"NDT_HIP", "GICP_HIP", "SMALL_GICP_HIP", "VGICP_HIP", "ICP_HIP", "PCL_GICP_HIP", "PCL_GICP_OMP_HIP", "PCL_NDT_HIP"};
        this->reg_name_ = (params.method >= 0 && params.method <= MRGFE_PCL_NDT_HIP) ? names[params.method] : "MRGFE";
        if (mrgfe_reg_create(shared_context(device, role), &params, &reg_) != MRGFE_OK) throw std::runtime_error(std::string("mrgfe: ") + mrgfe_last_error());
        this->max_iterations_ = params.maximum_iterations;
        this->transformation_epsilon_ = params.transformation_epsilon;
        // tree_ = our search object; force_no_recompute: initCompute() must not call tree_->setInputCloud(target_) (a FLANN build
        // over the whole target on the CPU at every target change) — setInputTarget below keeps the search object current itself
        search_.reset(new Search(reg_));
        Base::setSearchMethodTarget(search_, true);
    }
    ~HipRegistration() override { mrgfe_reg_destroy(reg_); }

    // registration_->setInputTarget(cloud): the 32-byte pcl::PointXYZI records go to the device as they are and x, y, z,
    // intensity are gathered there (MRGFE_LAYOUT)
    void setInputTarget(const PointCloudTargetConstPtr& cloud) override
    {
        Base::setInputTarget(cloud);
        search_->setInputCloud(cloud);
        search_->expect_queries(nullptr, 0);
        // the odometry's keyframe update hands over the scan it has just aligned (scan_matching_odometry_component.cpp:333: the same cloud object as the
        // source): the library takes it over with what it computed for it as a source, and does not upload it again
        if (source_set_ && static_cast<const void*>(cloud.get()) == static_cast<const void*>(this->input_.get())) {
            mrgfe_reg_source_becomes_target(reg_);
            return;
        }
        // overflow -> no target, like PCL's "Leaf size is too small" warning
        mrgfe_reg_set_target(reg_, cloud->empty() ? nullptr : &cloud->points[0].x, cloud->size(), point_layout<PointTarget>());
    }
    void setInputSource(const PointCloudSourceConstPtr& cloud) override
    {
        Base::setInputSource(cloud);
        search_->expect_queries(nullptr, 0);
        source_set_ = mrgfe_reg_set_source(reg_, cloud->empty() ? nullptr : &cloud->points[0].x, cloud->size(), point_layout<PointSource>()) == MRGFE_OK;
        if (!source_set_) PCL_ERROR("[%s::setInputSource] %s\n", this->reg_name_.c_str(), mrgfe_last_error());
    }

    // Fast path for holders of the DERIVED pointer (this hides the non-virtual base method): one GPU call, no per-point host
    // loop.  Through the base pointer PCL's own getFitnessScore runs and gets the same distances from GpuTargetSearch.
    double getFitnessScore(double max_range = std::numeric_limits<double>::max())
    {
        double out = std::numeric_limits<double>::max();
        mrgfe_reg_fitness(reg_, max_range, &out);
        return out;
    }
    int getFinalNumIteration() const { return mrgfe_reg_iterations(reg_); }
    std::array<double, 36> getFinalHessian() const  // 6x6, row-major
    {
        std::array<double, 36> h{};
        mrgfe_reg_hessian(reg_, h.data());
        return h;
    }
    const Search& gpuSearch() const { return *search_; }
    mrgfe_reg*    handle() const { return reg_; }

   protected:
    // called by pcl::Registration::align(output, guess) after it copied the source into `output`
    void computeTransformation(PointCloudSource& output, const Matrix4& guess) override
    {
        aligned_.resize(output.size() * 4);
        if (mrgfe_reg_align(reg_, guess.data() /* column-major */, aligned_.data()) != MRGFE_OK) {
            PCL_ERROR("[%s::computeTransformation] %s\n", this->reg_name_.c_str(), mrgfe_last_error());
            this->converged_ = false;
            search_->expect_queries(nullptr, 0);
            return;
        }
        for (std::size_t i = 0; i < output.size(); ++i) {
            output[i].x = aligned_[4 * i];
            output[i].y = aligned_[4 * i + 1];
            output[i].z = aligned_[4 * i + 2];
        }
        mrgfe_reg_final_transformation(reg_, this->final_transformation_.data());
        this->converged_ = mrgfe_reg_has_converged(reg_) != 0;
        this->nr_iterations_ = mrgfe_reg_iterations(reg_);
        // what getFitnessScore / the status loop will ask next: final_transformation * source, point by point
        search_->expect_queries(aligned_.data(), output.size());
    }

   private:
    mrgfe_reg*                 reg_ = nullptr;
    pcl::shared_ptr<Search>    search_;
    std::vector<float>         aligned_;
    bool                       source_set_ = false;
};

}  // namespace mrgfe_pcl

#endif  // __has_include(<pcl/registration/registration.h>)
