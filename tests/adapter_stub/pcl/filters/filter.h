// STAND-IN for <pcl/filters/filter.h> and <pcl/filters/filter_indices.h> (tests/adapter_stub/README.md): pcl::Filter with its NON-virtual
// filter(output) -> protected virtual applyFilter(output), and pcl::FilterIndices' flags (PCL 1.12 interface).  Not PCL.
#pragma once
#include <cstdio>
#include <string>

#include <pcl/point_cloud.h>

#ifndef PCL_ERROR
#define PCL_ERROR(...) std::fprintf(stderr, __VA_ARGS__)
#endif

namespace pcl {

template <typename PointT>
class Filter {
   public:
    using PointCloud = pcl::PointCloud<PointT>;
    using PointCloudConstPtr = typename PointCloud::ConstPtr;
    using Ptr = shared_ptr<Filter<PointT>>;
    virtual ~Filter() = default;
    virtual void setInputCloud(const PointCloudConstPtr& cloud) { input_ = cloud; }
    PointCloudConstPtr getInputCloud() const { return input_; }
    void filter(PointCloud& output)  // NOT virtual
    {
        if (!input_) return;
        if (&output == input_.get()) {  // filtering in place: work on a copy
            PointCloud tmp;
            applyFilter(tmp);
            output = tmp;
        } else {
            output.is_dense = input_->is_dense;
            applyFilter(output);
        }
    }
    static int& cpu_calls() { static int n = 0; return n; }  // how often a stand-in CPU applyFilter ran

   protected:
    PointCloudConstPtr input_;
    std::string        filter_name_;
    virtual void applyFilter(PointCloud& output) = 0;
};

template <typename PointT>
class FilterIndices : public Filter<PointT> {
   public:
    using PointCloud = typename Filter<PointT>::PointCloud;
    void setNegative(bool negative) { negative_ = negative; }
    void setKeepOrganized(bool keep) { keep_organized_ = keep; }
    void filter(Indices& indices) { applyFilter(indices); }  // NOT virtual
    using Filter<PointT>::filter;

   protected:
    bool negative_ = false, keep_organized_ = false, extract_removed_indices_ = false;
    using Filter<PointT>::applyFilter;
    virtual void applyFilter(Indices& indices) = 0;
};

}  // namespace pcl
