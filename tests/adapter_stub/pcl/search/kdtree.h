// STAND-IN for <pcl/search/kdtree.h> (tests/adapter_stub/README.md): the virtual surface of pcl::search::Search /
// pcl::search::KdTree; the "tree" is a brute-force scan that counts its builds.  Not PCL.
#pragma once
#include <cmath>
#include <limits>

#include <pcl/point_cloud.h>

namespace pcl {
namespace search {

template <typename PointT>
class Search {
   public:
    using PointCloudConstPtr = typename PointCloud<PointT>::ConstPtr;
    using IndicesConstPtr = pcl::IndicesConstPtr;
    virtual ~Search() = default;
    virtual void setInputCloud(const PointCloudConstPtr& cloud, const IndicesConstPtr& indices = IndicesConstPtr()) = 0;
    virtual PointCloudConstPtr getInputCloud() const = 0;
    virtual int nearestKSearch(const PointT& point, int k, Indices& k_indices, std::vector<float>& k_sqr_distances) const = 0;
    virtual int radiusSearch(const PointT& point, double radius, Indices& k_indices, std::vector<float>& k_sqr_distances, unsigned int max_nn = 0) const = 0;
};

template <typename PointT>
class KdTree : public Search<PointT> {
   public:
    using Ptr = shared_ptr<KdTree<PointT>>;
    using PointCloudConstPtr = typename Search<PointT>::PointCloudConstPtr;
    using IndicesConstPtr = typename Search<PointT>::IndicesConstPtr;
    static int& builds() { static int n = 0; return n; }  // how many times a (CPU) tree was built over a cloud

    void setInputCloud(const PointCloudConstPtr& cloud, const IndicesConstPtr& = IndicesConstPtr()) override
    {
        input_ = cloud;
        ++builds();
    }
    PointCloudConstPtr getInputCloud() const override { return input_; }
    int nearestKSearch(const PointT& p, int k, Indices& k_indices, std::vector<float>& k_sqr_distances) const override
    {
        k_indices.assign(k, -1);
        k_sqr_distances.assign(k, std::numeric_limits<float>::max());
        if (!input_ || k != 1) return 0;
        for (std::size_t i = 0; i < input_->size(); ++i) {
            const PointT& t = (*input_)[i];
            const float   dx = t.x - p.x, dy = t.y - p.y, dz = t.z - p.z;
            float         d = dx * dx;
            d += dy * dy;
            d += dz * dz;
            if (d < k_sqr_distances[0]) { k_sqr_distances[0] = d; k_indices[0] = static_cast<int>(i); }
        }
        return k_indices[0] >= 0 ? 1 : 0;
    }
    int radiusSearch(const PointT&, double, Indices& k_indices, std::vector<float>& k_sqr_distances, unsigned int = 0) const override
    {
        k_indices.clear();
        k_sqr_distances.clear();
        return 0;
    }

   protected:
    PointCloudConstPtr input_;
};

}  // namespace search
}  // namespace pcl
