// STAND-IN for <pcl/filters/radius_outlier_removal.h> (tests/adapter_stub/README.md): setters and protected members of
// pcl::RadiusOutlierRemoval (PCL 1.12; prefiltering_component.cpp:195-198).  The stand-in's CPU code only counts its calls.  Not PCL.
#pragma once
#include <pcl/filters/filter.h>

namespace pcl {
template <typename PointT>
class RadiusOutlierRemoval : public FilterIndices<PointT> {
   public:
    using PointCloud = typename FilterIndices<PointT>::PointCloud;
    RadiusOutlierRemoval() { this->filter_name_ = "RadiusOutlierRemoval"; }
    void   setRadiusSearch(double radius) { search_radius_ = radius; }
    double getRadiusSearch() const { return search_radius_; }
    void   setMinNeighborsInRadius(int min_pts) { min_pts_radius_ = min_pts; }
    int    getMinNeighborsInRadius() const { return min_pts_radius_; }

   protected:
    double search_radius_ = 0.0;
    int    min_pts_radius_ = 1;
    void applyFilter(PointCloud& output) override { ++Filter<PointT>::cpu_calls(); output = *this->input_; }
    void applyFilter(Indices& indices) override { ++Filter<PointT>::cpu_calls(); indices.clear(); }
};
}  // namespace pcl
