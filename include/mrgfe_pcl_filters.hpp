// include/mrgfe_pcl_filters.hpp — header-only adapters that make libmrgfe.so the pcl::Filter objects the reference's prefiltering and
// odometry components hold:
//   voxelgrid_filter_                    /root/reference/apps/prefiltering_component.cpp:37 (used :168-171);
//                                        apps/scan_matching_odometry_component.cpp:176-179 (pcl::VoxelGrid built in place)
//   approx_voxelgrid_filter_             prefiltering_component.cpp:38 (used :173-175); approx_voxel_grid_filter_
//                                        scan_matching_odometry_component.cpp:39 (used :181-183)
//   statistical_outlier_removal_filter_  prefiltering_component.cpp:54  (used :189-192)
//   radius_outlier_removal_filter_       prefiltering_component.cpp:55  (used :195-198)
// The reference configures them through PCL's own setters (setLeafSize, setMinimumPointsNumberPerVoxel, setRadiusSearch,
// setMinNeighborsInRadius, setMeanK, setStddevMulThresh), hands over the cloud with setInputCloud and calls the NON-virtual
// pcl::Filter::filter(output), which ends in the protected virtual applyFilter(output).  The classes below derive from the PCL
// filters and override only that: the setters, getters and member variables stay PCL's, so the swap is the `make_shared` line
// (INTEGRATION.md §2).  The 32-byte pcl::PointXYZI records go to the device as they are (MRGFE_LAYOUT: gathered there).
//
// Anything the C ABI does not offer is left to the PCL implementation the classes inherit (it runs on the CPU, as before):
// unequal leaf sizes, a filter field / filter limits, downsample_all_data = false, save_leaf_layout, negative / keep_organized /
// extract_removed_indices of the outlier filters, and their filter(Indices&) overloads.
// PCL is not installed in the build container: this file ships as source (tests compile it against tests/adapter_stub).
#pragma once
#if __has_include(<pcl/filters/voxel_grid.h>)

#include <pcl/filters/approximate_voxel_grid.h>
#include <pcl/filters/radius_outlier_removal.h>
#include <pcl/filters/statistical_outlier_removal.h>
#include <pcl/filters/voxel_grid.h>
#include <pcl/point_types.h>

#include <cstddef>
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

namespace detail {
// packed x, y, z, intensity records -> PCL points (the padding word of a pcl::PointXYZ* stays at its default 1.0f)
template <typename PointT>
inline void unpack(const std::vector<float>& xyzi, std::size_t n, pcl::PointCloud<PointT>& output)
{
    output.points.assign(n, PointT());
    for (std::size_t i = 0; i < n; ++i) {
        PointT& p = output.points[i];
        p.x = xyzi[4 * i];
        p.y = xyzi[4 * i + 1];
        p.z = xyzi[4 * i + 2];
        p.intensity = xyzi[4 * i + 3];
    }
    output.width = static_cast<std::uint32_t>(n);
    output.height = 1;
}
}  // namespace detail

// pcl::VoxelGrid<PointT> whose applyFilter runs mrgfe_voxelgrid: centroids of x, y, z, intensity per occupied voxel, output in
// ascending voxel index (PCL's order), "Leaf size is too small" passes the cloud through like PCL does
template <typename PointT = pcl::PointXYZI>
class HipVoxelGrid : public pcl::VoxelGrid<PointT> {
   public:
    using Base = pcl::VoxelGrid<PointT>;
    using PointCloud = typename Base::PointCloud;
    explicit HipVoxelGrid(int device = 0) : ctx_(shared_context(device)) {}
    std::size_t gpu_calls() const { return gpu_calls_; }

   protected:
    void applyFilter(PointCloud& output) override
    {
        const bool cubic = this->leaf_size_[0] == this->leaf_size_[1] && this->leaf_size_[0] == this->leaf_size_[2];
        if (!cubic || !this->filter_field_name_.empty() || !this->downsample_all_data_ || this->save_leaf_layout_) { Base::applyFilter(output); return; }  // not offered: PCL's own code
        const auto& in = *this->input_;
        if (in.empty()) { output.points.clear(); output.width = 0; output.height = 1; output.is_dense = true; return; }
        buf_.resize(in.size() * 4);
        std::size_t m = 0;
        int         overflow = 0;
        if (mrgfe_voxelgrid(ctx_, &in.points[0].x, in.size(), point_layout<PointT>(), this->leaf_size_[0], static_cast<int>(this->min_points_per_voxel_), buf_.data(), &m, &overflow) != MRGFE_OK) {
            PCL_ERROR("[mrgfe_pcl::HipVoxelGrid::applyFilter] %s\n", mrgfe_last_error());
            output.points.clear();
            output.width = 0;
            output.height = 1;
            return;
        }
        ++gpu_calls_;
        if (overflow) {  // pcl::VoxelGrid: "Leaf size is too small for the input dataset. Integer indices would overflow." -> output = *input_
            output = in;
            return;
        }
        detail::unpack(buf_, m, output);
        output.is_dense = true;  // we filter out invalid points (pcl::VoxelGrid::applyFilter)
    }

   private:
    mrgfe_ctx*         ctx_;
    std::vector<float> buf_;
    std::size_t        gpu_calls_ = 0;
};

// pcl::ApproximateVoxelGrid<PointT> whose applyFilter runs mrgfe_approx_voxelgrid (downsample_method APPROX_VOXELGRID): the 512-entry
// history of cells flushed in arrival order — the same points in the same order as PCL's sequential loop, from 512 independent sequences on the
// GPU (csrc/filters.hip).  A history of another size (a PCL built with another histsize_), unequal leaf sizes or downsample_all_data = false
// stay PCL's.
template <typename PointT = pcl::PointXYZI>
class HipApproximateVoxelGrid : public pcl::ApproximateVoxelGrid<PointT> {
   public:
    using Base = pcl::ApproximateVoxelGrid<PointT>;
    using PointCloud = typename Base::PointCloud;
    explicit HipApproximateVoxelGrid(int device = 0) : ctx_(shared_context(device)) {}
    std::size_t gpu_calls() const { return gpu_calls_; }

   protected:
    void applyFilter(PointCloud& output) override
    {
        const bool cubic = this->leaf_size_[0] == this->leaf_size_[1] && this->leaf_size_[0] == this->leaf_size_[2];
        if (!cubic || !this->downsample_all_data_ || this->histsize_ != 512) { Base::applyFilter(output); return; }  // not offered: PCL's own code
        const auto& in = *this->input_;
        if (in.empty()) { output.points.clear(); output.width = 0; output.height = 1; output.is_dense = false; return; }
        buf_.resize(in.size() * 4);
        std::size_t m = 0;
        if (mrgfe_approx_voxelgrid(ctx_, &in.points[0].x, in.size(), point_layout<PointT>(), this->leaf_size_[0], buf_.data(), &m) != MRGFE_OK) {
            PCL_ERROR("[mrgfe_pcl::HipApproximateVoxelGrid::applyFilter] %s\n", mrgfe_last_error());
            output.points.clear();
            output.width = 0;
            output.height = 1;
            return;
        }
        ++gpu_calls_;
        detail::unpack(buf_, m, output);
        output.is_dense = false;  // "we filter out invalid points" is NOT what pcl::ApproximateVoxelGrid::applyFilter says: it sets is_dense = false
    }

   private:
    mrgfe_ctx*         ctx_;
    std::vector<float> buf_;
    std::size_t        gpu_calls_ = 0;
};

// pcl::RadiusOutlierRemoval<PointT> whose applyFilter(PointCloud&) runs mrgfe_radius_outlier: a point stays iff at least
// min_pts_radius_ neighbours (itself included in the search, as in PCL) lie within search_radius_; order preserved
template <typename PointT = pcl::PointXYZI>
class HipRadiusOutlierRemoval : public pcl::RadiusOutlierRemoval<PointT> {
   public:
    using Base = pcl::RadiusOutlierRemoval<PointT>;
    using PointCloud = typename Base::PointCloud;
    explicit HipRadiusOutlierRemoval(int device = 0) : ctx_(shared_context(device)) {}
    std::size_t gpu_calls() const { return gpu_calls_; }

   protected:
    using Base::applyFilter;  // the Indices overload stays PCL's
    void applyFilter(PointCloud& output) override
    {
        if (this->negative_ || this->keep_organized_ || this->extract_removed_indices_) { Base::applyFilter(output); return; }
        const auto& in = *this->input_;
        if (in.empty()) { output.points.clear(); output.width = 0; output.height = 1; return; }
        buf_.resize(in.size() * 4);
        std::size_t m = 0;
        if (mrgfe_radius_outlier(ctx_, &in.points[0].x, in.size(), point_layout<PointT>(), this->search_radius_, this->min_pts_radius_, buf_.data(), &m) != MRGFE_OK) {
            PCL_ERROR("[mrgfe_pcl::HipRadiusOutlierRemoval::applyFilter] %s\n", mrgfe_last_error());
            m = 0;
        }
        ++gpu_calls_;
        detail::unpack(buf_, m, output);
        output.is_dense = true;  // points with a non-finite coordinate are never inliers
    }

   private:
    mrgfe_ctx*         ctx_;
    std::vector<float> buf_;
    std::size_t        gpu_calls_ = 0;
};

// pcl::StatisticalOutlierRemoval<PointT> whose applyFilter(PointCloud&) runs mrgfe_statistical_outlier: mean distance to the
// mean_k_ nearest neighbours against the global mean + std_mul_ * stddev (f64 statistics, as in PCL); order preserved
template <typename PointT = pcl::PointXYZI>
class HipStatisticalOutlierRemoval : public pcl::StatisticalOutlierRemoval<PointT> {
   public:
    using Base = pcl::StatisticalOutlierRemoval<PointT>;
    using PointCloud = typename Base::PointCloud;
    explicit HipStatisticalOutlierRemoval(int device = 0) : ctx_(shared_context(device)) {}
    std::size_t gpu_calls() const { return gpu_calls_; }

   protected:
    using Base::applyFilter;
    void applyFilter(PointCloud& output) override
    {
        if (this->negative_ || this->keep_organized_ || this->extract_removed_indices_) { Base::applyFilter(output); return; }
        const auto& in = *this->input_;
        if (in.empty()) { output.points.clear(); output.width = 0; output.height = 1; return; }
        buf_.resize(in.size() * 4);
        std::size_t m = 0;
        if (mrgfe_statistical_outlier(ctx_, &in.points[0].x, in.size(), point_layout<PointT>(), this->mean_k_, this->std_mul_, buf_.data(), &m) != MRGFE_OK) {
            PCL_ERROR("[mrgfe_pcl::HipStatisticalOutlierRemoval::applyFilter] %s\n", mrgfe_last_error());
            m = 0;
        }
        ++gpu_calls_;
        detail::unpack(buf_, m, output);
        output.is_dense = true;
    }

   private:
    mrgfe_ctx*         ctx_;
    std::vector<float> buf_;
    std::size_t        gpu_calls_ = 0;
};

}  // namespace mrgfe_pcl

#endif  // __has_include(<pcl/filters/voxel_grid.h>)
