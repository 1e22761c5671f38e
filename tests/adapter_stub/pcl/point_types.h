// STAND-IN for <pcl/point_types.h> (tests/adapter_stub/README.md): the memory layout of pcl::PointXYZI and the few
// typedefs the adapter needs.  Not PCL.
#pragma once
#include <cstdint>
#include <memory>
#include <vector>

namespace Eigen {
// column-major fixed matrix with the handful of members the adapter and the stub base class use
template <typename S, int R, int C>
struct Matrix {
    S m[R * C];
    S*       data() { return m; }
    const S* data() const { return m; }
    S&       operator()(int r, int c) { return m[c * R + r]; }
    const S& operator()(int r, int c) const { return m[c * R + r]; }
    static Matrix Identity()
    {
        Matrix I;
        for (int i = 0; i < R * C; ++i) I.m[i] = S(0);
        for (int i = 0; i < (R < C ? R : C); ++i) I(i, i) = S(1);
        return I;
    }
};
using Matrix4f = Matrix<float, 4, 4>;
}  // namespace Eigen

// pcl_config.h: the version macros the adapter keys the pcl::search::KdTree::setInputCloud signature on (this stand-in restates 1.12.1)
#define PCL_VERSION_CALC(MAJ, MIN, PATCH) (MAJ * 100000 + MIN * 100 + PATCH)
#define PCL_VERSION PCL_VERSION_CALC(1, 12, 1)
#define PCL_VERSION_COMPARE(OP, MAJ, MIN, PATCH) (PCL_VERSION OP PCL_VERSION_CALC(MAJ, MIN, PATCH))

namespace pcl {
template <typename T> using shared_ptr = std::shared_ptr<T>;
using Indices = std::vector<int>;
using IndicesConstPtr = std::shared_ptr<const Indices>;

// 32 bytes: x, y, z and a padding word that PCL initialises to 1.0f, then intensity and 12 bytes of padding
struct alignas(16) PointXYZI {
    float x = 0.f, y = 0.f, z = 0.f, data3 = 1.f;
    float intensity = 0.f, pad_[3] = {0.f, 0.f, 0.f};
};
static_assert(sizeof(PointXYZI) == 32, "pcl::PointXYZI is 32 bytes");
template <typename PointT>
inline bool isXYZFinite(const PointT& p) { return p.x - p.x == 0.f && p.y - p.y == 0.f && p.z - p.z == 0.f; }
}  // namespace pcl
