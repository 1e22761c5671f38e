// STAND-IN for <pcl/filters/approximate_voxel_grid.h> (tests/adapter_stub/README.md): the setters and protected members of
// pcl::ApproximateVoxelGrid (PCL 1.12) that the reference's call sites (prefiltering_component.cpp:173-175,
// scan_matching_odometry_component.cpp:181-183) and include/mrgfe_pcl_filters.hpp touch.  The CPU applyFilter of the stand-in only counts its
// calls and passes the cloud through.  Not PCL.
#pragma once
#include <pcl/filters/filter.h>

namespace Eigen {
struct Vector3f {
    float v[3] = {0, 0, 0};
    float&       operator[](int i) { return v[i]; }
    const float& operator[](int i) const { return v[i]; }
};
}  // namespace Eigen

namespace pcl {
template <typename PointT>
class ApproximateVoxelGrid : public Filter<PointT> {
   public:
    using PointCloud = typename Filter<PointT>::PointCloud;
    ApproximateVoxelGrid() { this->filter_name_ = "ApproximateVoxelGrid"; }
    void setLeafSize(float lx, float ly, float lz) { leaf_size_[0] = lx; leaf_size_[1] = ly; leaf_size_[2] = lz; }
    Eigen::Vector3f getLeafSize() const { return leaf_size_; }
    void setDownsampleAllData(bool d) { downsample_all_data_ = d; }
    bool getDownsampleAllData() const { return downsample_all_data_; }

   protected:
    Eigen::Vector3f leaf_size_;
    bool            downsample_all_data_ = true;
    std::size_t     histsize_ = 512;
    void applyFilter(PointCloud& output) override { ++Filter<PointT>::cpu_calls(); output = *this->input_; }
};
}  // namespace pcl
