// oracle/gicp.h — TEST INFRASTRUCTURE (CPU oracle). Not part of the shipped product path.
//
// CPU restatement of fast_gicp::FastGICP<PointXYZI,PointXYZI> (SMRT-AIST/fast_gicp, un-vendored dependency
// of the reference: /root/reference/CMakeLists.txt:85, package.xml:19; instantiated by
// src/mrg_slam/registrations.cpp:55-63 with num_threads / transformation_epsilon / maximum_iterations /
// max_correspondence_distance / correspondence_randomness) on top of fast_gicp::LsqRegistration
// (Levenberg-Marquardt) — SURVEY.md Appendix A.6.  PARITY UNPINNED (see quirks.h).
//
// `variant = 1` restates small_gicp::RegistrationPCL<PointXYZI,PointXYZI> (koide3/small_gicp, un-vendored, no version
// pinned: CMakeLists.txt:91, package.xml:21; the YAML default "SMALL_GICP", registrations.cpp:46-54) as published:
// the same GICP factor (k-NN covariances regularised to eigenvalues (1e-3, 1, 1), 1-NN correspondences rejected beyond
// max_correspondence_distance, Mahalanobis (C_B + T C_A T^T)^-1) linearised for a RIGHT perturbation T <- T exp(d),
// J = [ R skew(a) | -R ], error 0.5 r^T M r, and its LevenbergMarquardtOptimizer (lambda 1e-3, factor 10, at most 10
// inner trials, a trial is accepted iff its error does not exceed the current one; converged iff |d_rot| <= rotation_eps
// and |d_trans| <= translation_eps, RegistrationPCL sets rotation_eps 2e-3 and translation_eps = transformation_epsilon).
// Deviation kept from the fast_gicp path: correspondences are searched with the float-cast transform (small_gicp's own
// kd-tree works on the double coordinates).
//
// `variant = 2` restates fast_gicp::FastVGICP<PointXYZI,PointXYZI> (registrations.cpp:76-84; its CUDA sibling
// FAST_VGICP_CUDA, :65-75, minimises the same voxelised cost on the device and is not restated separately): the target becomes a GaussianVoxelMap (resolution = reg_resolution, ADDITIVE
// accumulation: per voxel the mean of its points and the mean of their regularised covariances; voxel of x =
// floor(x / resolution - 0.5)), a source point corresponds to the voxel its transformed position falls in (DIRECT1),
// every term carries the weight sqrt(points in the voxel); optimiser, Jacobian and convergence test are fast_gicp's.
//
// `variant = 3` restates pcl::IterativeClosestPoint<PointXYZI,PointXYZI> (PCL 1.12, registrations.cpp:85-92: setTransformationEpsilon,
// setMaximumIterations, setMaxCorrespondenceDistance, setUseReciprocalCorrespondences): per iteration the cumulatively
// transformed source is matched to its nearest target points within the distance limit, TransformationEstimationSVD
// (Umeyama without scaling) gives the increment, DefaultConvergenceCriteria decides (iterations >= max; or rotation cosine >=
// 1 - epsilon and squared translation <= epsilon; or |mse - previous mse| < 1e-12).  With reciprocal correspondences a pair (i, j) counts
// only if the nearest point of target j among the TRANSFORMED source points is i again, within the distance limit
// (CorrespondenceEstimation::determineReciprocalCorrespondences; ties at equal distance go to the lowest index here).  Documented deviations: Eigen's float
// reductions have no fixed order, so the moment sums are f64 and the 3x3 SVD is a one-sided Jacobi in f64 before the result
// is cast to the float matrices PCL works with.
#pragma once
#include <map>
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
    // variant 1 only, diagnostic: small_gicp's OWN correspondence search — the source point transformed in double (Isometry3d * Vector4d), the nearest
    // target point by double-precision distance, rejected iff that exceeds max_dist_sq — instead of fast_gicp's float transform and float distances that
    // the restatement (and the HIP path) keeps for both formulations ("deviations").  The candidates are the 8 float-nearest target points.
    bool   double_search = false;
    int    variant = 0;                 // 0: fast_gicp::FastGICP, 1: small_gicp::RegistrationPCL (GICP), 2: fast_gicp::FastVGICP, 3: pcl::IterativeClosestPoint
    double voxel_resolution = 1.0;      // variant 2: setResolution(reg_resolution)
    bool   use_reciprocal = false;      // variant 3: setUseReciprocalCorrespondences (registrations.cpp:91): CorrespondenceEstimation::determineReciprocalCorrespondences
    double sg_init_lambda = 1e-3, sg_lambda_factor = 10.0;  // small_gicp::LevenbergMarquardtOptimizer defaults
    int    sg_max_inner_iterations = 10;

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
    void   align_small_gicp(const float guess_rowmajor[16]);  // variant 1: leaves the result in final_
    void   align_icp(const float guess_rowmajor[16]);         // variant 3
    double fitness(double max_range) const;
    void   get_covariances(int which, double* out) const;  // 0 = source, 1 = target (computes if needed)
    // update_correspondences + linearize at T (row-major 4x4 double). returns sum of errors.
    double linearize(const double T[16], double H[36], double b[6], int* n_corr);
    double compute_error(const double T[16]) const;
    int    num_voxels() const { return static_cast<int>(voxels_.size()); }

   private:
    std::vector<int>    correspondences_;
    std::vector<double> mahalanobis_;  // 9 doubles per source point
    void calculate_covariances(const std::vector<float>& cloud, std::vector<double>& covs) const;
    void ensure_covs();
    NnGrid target_grid_;
    bool   target_grid_valid_ = false;
    // variant 2: GaussianVoxelMap of the target
    struct Voxel { int num_points = 0; double mean[3] = {0, 0, 0}; double cov[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}; };
    struct Coord { int c[3]; bool operator<(const Coord& o) const { return c[0] != o.c[0] ? c[0] < o.c[0] : c[1] != o.c[1] ? c[1] < o.c[1] : c[2] < o.c[2]; } };
    std::map<Coord, int> voxel_index_;
    std::vector<Voxel>   voxels_;
    bool voxelmap_valid_ = false;
    void build_voxelmap();
    Coord voxel_coord(const double x[3]) const;
};

}  // namespace orc
