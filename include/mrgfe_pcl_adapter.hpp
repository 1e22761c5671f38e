// include/mrgfe_pcl_adapter.hpp — header-only adapter that makes libmrgfe.so a pcl::Registration, i.e. the thing
// mrg_slam::select_registration_method returns (/root/reference/include/mrg_slam/registrations.hpp:20).
//
// It occupies the slot the reference already has for a GPU back end (FAST_VGICP_CUDA:
// /root/reference/src/mrg_slam/registrations.cpp:19-21,65-75; CMakeLists.txt:45-49).  PCL is not installed in the
// build container, so this file ships as source and is compiled only where <pcl/registration/registration.h> exists
// (INTEGRATION.md shows the ten-line factory patch).  It uses nothing but the C ABI of include/mrgfe.h.
#pragma once
#if __has_include(<pcl/registration/registration.h>)

#include <pcl/point_types.h>
#include <pcl/registration/registration.h>

#include <limits>
#include <stdexcept>
#include <string>

#include "mrgfe.h"

namespace mrgfe_pcl {

// One context per process and GPU (stream + workspaces), shared by every registration object of that process.
inline mrgfe_ctx* shared_context(int device = 0)
{
    static mrgfe_ctx* ctx = nullptr;
    if (!ctx && mrgfe_ctx_create(device, &ctx) != MRGFE_OK) throw std::runtime_error(std::string("mrgfe: ") + mrgfe_last_error());
    return ctx;
}

template <typename PointSource = pcl::PointXYZI, typename PointTarget = pcl::PointXYZI>
class HipRegistration : public pcl::Registration<PointSource, PointTarget, float> {
   public:
    using Base = pcl::Registration<PointSource, PointTarget, float>;
    using Ptr = pcl::shared_ptr<HipRegistration<PointSource, PointTarget>>;
    using PointCloudSource = typename Base::PointCloudSource;
    using PointCloudSourceConstPtr = typename Base::PointCloudSourceConstPtr;
    using PointCloudTargetConstPtr = typename Base::PointCloudTargetConstPtr;
    using Matrix4 = typename Base::Matrix4;

    explicit HipRegistration(const mrgfe_reg_params& params, int device = 0)
    {
        this->reg_name_ = params.method == MRGFE_NDT_HIP ? "NDT_HIP" : params.method == MRGFE_GICP_HIP ? "GICP_HIP" : params.method == MRGFE_VGICP_HIP ? "VGICP_HIP" : params.method == MRGFE_ICP_HIP ? "ICP_HIP" : "SMALL_GICP_HIP";
        if (mrgfe_reg_create(shared_context(device), &params, &reg_) != MRGFE_OK) throw std::runtime_error(std::string("mrgfe: ") + mrgfe_last_error());
        this->max_iterations_ = params.maximum_iterations;
        this->transformation_epsilon_ = params.transformation_epsilon;
    }
    ~HipRegistration() override { mrgfe_reg_destroy(reg_); }

    // registration_->setInputTarget(cloud): pcl::PointXYZI is x,y,z,pad,intensity,pad...: 32-byte stride; the intensity sits
    // at float offset 4, so the cloud is repacked to xyzi once per call on the host.
    void setInputTarget(const PointCloudTargetConstPtr& cloud) override
    {
        Base::setInputTarget(cloud);
        pack(*cloud);
        mrgfe_reg_set_target(reg_, packed_.data(), cloud->size(), 16);  // overflow -> no target, like PCL's warning
    }
    void setInputSource(const PointCloudSourceConstPtr& cloud) override
    {
        Base::setInputSource(cloud);
        pack(*cloud);
        if (mrgfe_reg_set_source(reg_, packed_.data(), cloud->size(), 16) != MRGFE_OK) PCL_ERROR("[%s::setInputSource] %s\n", this->reg_name_.c_str(), mrgfe_last_error());
    }

    // getFitnessScore(max_range): same semantics as the PCL base (squared distance against max_range)
    double getFitnessScore(double max_range = std::numeric_limits<double>::max())
    {
        double out = std::numeric_limits<double>::max();
        mrgfe_reg_fitness(reg_, max_range, &out);
        return out;
    }
    int getFinalNumIteration() const { return mrgfe_reg_iterations(reg_); }
    Eigen::Matrix<double, 6, 6> getFinalHessian() const
    {
        double h[36];
        mrgfe_reg_hessian(reg_, h);
        return Eigen::Map<Eigen::Matrix<double, 6, 6, Eigen::RowMajor>>(h);
    }

   protected:
    // called by pcl::Registration::align(output, guess) after it copied the source into `output`
    void computeTransformation(PointCloudSource& output, const Matrix4& guess) override
    {
        aligned_.resize(output.size() * 4);
        if (mrgfe_reg_align(reg_, guess.data() /* column-major */, aligned_.data()) != MRGFE_OK) {
            PCL_ERROR("[%s::computeTransformation] %s\n", this->reg_name_.c_str(), mrgfe_last_error());
            this->converged_ = false;
            return;
        }
        for (std::size_t i = 0; i < output.size(); ++i) {
            output[i].x = aligned_[4 * i];
            output[i].y = aligned_[4 * i + 1];
            output[i].z = aligned_[4 * i + 2];
        }
        mrgfe_reg_final_transformation(reg_, this->final_transformation_.data());
        this->converged_ = mrgfe_reg_has_converged(reg_) != 0;
        this->nr_iterations_ = mrgfe_reg_iterations(reg_);
    }

   private:
    template <class Cloud>
    void pack(const Cloud& c)
    {
        packed_.resize(c.size() * 4);
        for (std::size_t i = 0; i < c.size(); ++i) {
            packed_[4 * i] = c[i].x;
            packed_[4 * i + 1] = c[i].y;
            packed_[4 * i + 2] = c[i].z;
            packed_[4 * i + 3] = c[i].intensity;
        }
    }
    mrgfe_reg*         reg_ = nullptr;
    std::vector<float> packed_, aligned_;
};

}  // namespace mrgfe_pcl

#endif  // __has_include(<pcl/registration/registration.h>)
