// STAND-IN for <pcl/registration/registration.h> (tests/adapter_stub/README.md): the members, virtuals and NON-virtual
// methods of pcl::Registration (PCL 1.12 interface) that include/mrgfe_pcl_adapter.hpp and the reference's call sites
// touch.  align(), initCompute() and getFitnessScore() restate the base-class behaviour the adapter relies on:
//   - initCompute() rebuilds tree_ over the target only if the target changed AND force_no_recompute_ is false,
//   - getFitnessScore() is not virtual: it transforms the source by final_transformation_ on the host and calls
//     tree_->nearestKSearch(point, 1, ...) for every point in order.
// Not PCL.
#pragma once
#include <cstdio>
#include <limits>
#include <string>

#include <pcl/point_cloud.h>
#include <pcl/search/kdtree.h>

#define PCL_ERROR(...) std::fprintf(stderr, __VA_ARGS__)

namespace pcl {

template <typename PointT>
class PCLBase {
   public:
    using PointCloudConstPtr = typename PointCloud<PointT>::ConstPtr;
    virtual ~PCLBase() = default;
    virtual void setInputCloud(const PointCloudConstPtr& cloud) { input_ = cloud; }

   protected:
    PointCloudConstPtr input_;
    bool initCompute() { return static_cast<bool>(input_); }
};

template <typename PointSource, typename PointTarget, typename Scalar = float>
class Registration : public PCLBase<PointSource> {
   public:
    using Matrix4 = Eigen::Matrix<Scalar, 4, 4>;
    using Ptr = shared_ptr<Registration<PointSource, PointTarget, Scalar>>;
    using KdTree = pcl::search::KdTree<PointTarget>;
    using KdTreePtr = typename KdTree::Ptr;
    using PointCloudSource = pcl::PointCloud<PointSource>;
    using PointCloudSourceConstPtr = typename PointCloudSource::ConstPtr;
    using PointCloudTarget = pcl::PointCloud<PointTarget>;
    using PointCloudTargetConstPtr = typename PointCloudTarget::ConstPtr;

    Registration() : tree_(new KdTree) {}
    ~Registration() override = default;

    virtual void setInputSource(const PointCloudSourceConstPtr& cloud)
    {
        source_cloud_updated_ = true;
        PCLBase<PointSource>::setInputCloud(cloud);
    }
    virtual void setInputTarget(const PointCloudTargetConstPtr& cloud)
    {
        target_ = cloud;
        target_cloud_updated_ = true;
    }
    void setSearchMethodTarget(const KdTreePtr& tree, bool force_no_recompute = false)
    {
        tree_ = tree;
        force_no_recompute_ = force_no_recompute;
        target_cloud_updated_ = true;
    }
    KdTreePtr getSearchMethodTarget() const { return tree_; }
    Matrix4   getFinalTransformation() { return final_transformation_; }
    bool      hasConverged() const { return converged_; }

    double getFitnessScore(double max_range = std::numeric_limits<double>::max())  // NOT virtual
    {
        double           fitness_score = 0.0;
        PointCloudSource input_transformed;
        transformPointCloud(*this->input_, input_transformed, final_transformation_);
        Indices            nn_indices(1);
        std::vector<float> nn_dists(1);
        int                nr = 0;
        for (const auto& point : input_transformed) {
            if (!this->input_->is_dense && !pcl::isXYZFinite(point)) continue;  // (PCL 1.12: non-finite points of a non-dense source are skipped)
            tree_->nearestKSearch(point, 1, nn_indices, nn_dists);
            if (nn_dists[0] <= max_range) {
                fitness_score += nn_dists[0];
                nr++;
            }
        }
        if (nr > 0) return fitness_score / nr;
        return std::numeric_limits<double>::max();
    }

    void align(PointCloudSource& output) { align(output, Matrix4::Identity()); }
    void align(PointCloudSource& output, const Matrix4& guess)  // NOT virtual
    {
        if (!initCompute()) return;
        output = *this->input_;
        converged_ = false;
        final_transformation_ = transformation_ = previous_transformation_ = Matrix4::Identity();
        computeTransformation(output, guess);
    }

   protected:
    std::string reg_name_;
    KdTreePtr   tree_;
    int         nr_iterations_ = 0, max_iterations_ = 10;
    PointCloudTargetConstPtr target_;
    Matrix4     final_transformation_ = Matrix4::Identity(), transformation_ = Matrix4::Identity(), previous_transformation_ = Matrix4::Identity();
    double      transformation_epsilon_ = 0.0;
    bool        converged_ = false, target_cloud_updated_ = true, source_cloud_updated_ = true, force_no_recompute_ = false;

    bool initCompute()
    {
        if (!target_) { PCL_ERROR("[pcl::registration::%s::compute] No input target dataset was given!\n", reg_name_.c_str()); return false; }
        if (target_cloud_updated_ && !force_no_recompute_) {  // only update the target kd-tree if a new target cloud was set
            tree_->setInputCloud(target_);
            target_cloud_updated_ = false;
        }
        return PCLBase<PointSource>::initCompute();
    }
    virtual void computeTransformation(PointCloudSource& output, const Matrix4& guess) = 0;
};

}  // namespace pcl
