// STAND-IN for <pcl/filters/statistical_outlier_removal.h> (tests/adapter_stub/README.md): setters and protected members of
// pcl::StatisticalOutlierRemoval (PCL 1.12; prefiltering_component.cpp:189-192).  The stand-in's CPU code only counts its calls.  Not PCL.
#pragma once
#include <pcl/filters/filter.h>

namespace pcl {
template <typename PointT>
class StatisticalOutlierRemoval : public FilterIndices<PointT> {
   public:
    using PointCloud = typename FilterIndices<PointT>::PointCloud;
    StatisticalOutlierRemoval() { this->filter_name_ = "StatisticalOutlierRemoval"; }
    void   setMeanK(int nr_k) { mean_k_ = nr_k; }
    int    getMeanK() const { return mean_k_; }
    void   setStddevMulThresh(double stddev_mult) { std_mul_ = stddev_mult; }
    double getStddevMulThresh() const { return std_mul_; }

   protected:
    int    mean_k_ = 1;
    double std_mul_ = 0.0;
    void applyFilter(PointCloud& output) override { ++Filter<PointT>::cpu_calls(); output = *this->input_; }
    void applyFilter(Indices& indices) override { ++Filter<PointT>::cpu_calls(); indices.clear(); }
};
}  // namespace pcl
