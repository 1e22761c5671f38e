// STAND-IN for <pcl/point_cloud.h> (tests/adapter_stub/README.md).  Not PCL.
#pragma once
#include <cstddef>
#include <vector>

#include <pcl/point_types.h>

namespace pcl {
template <typename PointT>
class PointCloud {
   public:
    using Ptr = shared_ptr<PointCloud<PointT>>;
    using ConstPtr = shared_ptr<const PointCloud<PointT>>;
    std::vector<PointT> points;
    std::uint32_t width = 0, height = 1;
    bool          is_dense = true;
    std::size_t   size() const { return points.size(); }
    bool          empty() const { return points.empty(); }
    void          resize(std::size_t n) { points.resize(n); width = static_cast<std::uint32_t>(n); height = 1; }
    PointT&       operator[](std::size_t i) { return points[i]; }
    const PointT& operator[](std::size_t i) const { return points[i]; }
    PointT&       at(std::size_t i) { return points.at(i); }
    const PointT& at(std::size_t i) const { return points.at(i); }
    auto begin() { return points.begin(); }
    auto end() { return points.end(); }
    auto begin() const { return points.begin(); }
    auto end() const { return points.end(); }
};

// pcl::transformPointCloud(in, out, 4x4): out = R * p + t per point, other fields copied (scalar form)
template <typename PointT>
void transformPointCloud(const PointCloud<PointT>& in, PointCloud<PointT>& out, const Eigen::Matrix4f& T)
{
    if (&in != &out) out = in;
    for (std::size_t i = 0; i < in.size(); ++i) {
        const PointT& p = in[i];
        const float   x = p.x, y = p.y, z = p.z;
        out[i].x = T(0, 0) * x + (T(0, 1) * y + (T(0, 2) * z + T(0, 3)));
        out[i].y = T(1, 0) * x + (T(1, 1) * y + (T(1, 2) * z + T(1, 3)));
        out[i].z = T(2, 0) * x + (T(2, 1) * y + (T(2, 2) * z + T(2, 3)));
    }
}
}  // namespace pcl
