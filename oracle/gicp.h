// oracle/gicp.h — TEST INFRASTRUCTURE (CPU oracle). Not part of the shipped product path.
//
// CPU restatement of fast_gicp::FastGICP<PointXYZI,PointXYZI> (SMRT-AIST/fast_gicp, un-vendored dependency
// of the reference: /root/reference/CMakeLists.txt:85, package.xml:19; instantiated by
// src/mrg_slam/registrations.cpp:55-63 with num_threads / transformation_epsilon / maximum_iterations /
// max_correspondence_distance / correspondence_randomness) on top of fast_gicp::LsqRegistration
// (Levenberg-Marquardt) — SURVEY.md Appendix A.6.  PARITY UNPINNED (see quirks.h).
#pragma once
#include <vector>

#include "nn.h"

namespace orc {

struct FastGicp {
    int    k_correspondences = 20;
    double max_corr_dist     = 2.0;   // setMaxCorrespondenceDistance
    double trans_eps         = 5e-4;  // LsqRegistration default; mrg_slam passes reg_transformation_epsilon
    double rot_eps           = 2e-3;
    int    max_iterations    = 64;
    int    num_threads       = 1;
    int    lm_max_iterations = 10;
    double lm_init_lambda_factor = 1e-9;

    std::vector<float>  target, source;        // xyzi
    std::vector<double> target_covs, source_covs;  // 9 doubles (3x3 block of the 4x4) per point
    bool target_covs_valid = false, source_covs_valid = false;

    float  final_[16];  // row-major
    double final_hessian[36];
    bool   converged = false;
    int    nr_iterations = 0;
    int    n_linearize = 0, n_error_evals = 0;

    void   set_target(const float* xyzi, int n);
    void   set_source(const float* xyzi, int n);
    void   align(const float guess_rowmajor[16], float* aligned);
    double fitness(double max_range) const;
    void   get_covariances(int which, double* out) const;  // 0 = source, 1 = target (computes if needed)
    // update_correspondences + linearize at T (row-major 4x4 double). returns sum of errors.
    double linearize(const double T[16], double H[36], double b[6], int* n_corr);
    double compute_error(const double T[16]) const;

   private:
    std::vector<int>    correspondences_;
    std::vector<double> mahalanobis_;  // 9 doubles per source point
    void calculate_covariances(const std::vector<float>& cloud, std::vector<double>& covs) const;
    void ensure_covs();
    NnGrid target_grid_;
    bool   target_grid_valid_ = false;
};

}  // namespace orc
