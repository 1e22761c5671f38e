// STAND-IN for <pcl/filters/voxel_grid.h> (tests/adapter_stub/README.md): the setters, getters and protected members of pcl::VoxelGrid
// (PCL 1.12) that the reference's call sites (prefiltering_component.cpp:168-171) and include/mrgfe_pcl_filters.hpp touch.  The CPU
// applyFilter of the stand-in only counts its calls and passes the cloud through.  Not PCL.
#pragma once
#include <pcl/filters/filter.h>

namespace Eigen {
struct Vector4f {
    float v[4] = {0, 0, 0, 0};
    float&       operator[](int i) { return v[i]; }
    const float& operator[](int i) const { return v[i]; }
};
}  // namespace Eigen

namespace pcl {
template <typename PointT>
class VoxelGrid : public Filter<PointT> {
   public:
    using PointCloud = typename Filter<PointT>::PointCloud;
    VoxelGrid() { this->filter_name_ = "VoxelGrid"; }
    void setLeafSize(float lx, float ly, float lz) { leaf_size_[0] = lx; leaf_size_[1] = ly; leaf_size_[2] = lz; leaf_size_[3] = 1; }
    Eigen::Vector4f getLeafSize() const { return leaf_size_; }
    void setMinimumPointsNumberPerVoxel(unsigned int n) { min_points_per_voxel_ = n; }
    unsigned int getMinimumPointsNumberPerVoxel() const { return min_points_per_voxel_; }
    void setDownsampleAllData(bool d) { downsample_all_data_ = d; }
    void setSaveLeafLayout(bool s) { save_leaf_layout_ = s; }
    void setFilterFieldName(const std::string& name) { filter_field_name_ = name; }

   protected:
    Eigen::Vector4f leaf_size_;
    bool            downsample_all_data_ = true, save_leaf_layout_ = false;
    unsigned int    min_points_per_voxel_ = 0;
    std::string     filter_field_name_;
    void applyFilter(PointCloud& output) override { ++Filter<PointT>::cpu_calls(); output = *this->input_; }
};
}  // namespace pcl
