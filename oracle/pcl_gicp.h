// oracle/pcl_gicp.h — TEST INFRASTRUCTURE (CPU oracle). Not part of the shipped product path.
//
// CPU restatement of pcl::GeneralizedIterativeClosestPoint<PointXYZI,PointXYZI> (PCL 1.12 <pcl/registration/impl/gicp.hpp>) as the reference
// configures it (/root/reference/src/mrg_slam/registrations.cpp:93-103: "GICP": setTransformationEpsilon, setMaximumIterations,
// setUseReciprocalCorrespondences, setMaxCorrespondenceDistance, setCorrespondenceRandomness, setMaximumOptimizerIterations) and of
// pclomp::GeneralizedIterativeClosestPoint (koide3/ndt_omp gicp_omp, :104-114: "GICP_OMP", the same algorithm with OpenMP loops and the older
// whole-gradient-norm stopping rule) — PARITY UNPINNED (SURVEY.md Appendix A.6 names the family; the text is recalled upstream source).
//
//   covariances   k = correspondence_randomness nearest neighbours of every point (source: its own cloud, target: its own cloud); raw second
//                 moments of the float coordinates summed in double, / k, minus mean mean^T; SVD; singular values replaced by (1, 1, gicp_epsilon
//                 = 1e-3): cov = sum_k v_k u_k u_k^T
//   outer loop    the source is moved by the guess once (float), then per iteration: query = transformation_ * point (float), nearest target
//                 point, kept iff squared distance < max_correspondence_distance^2; Mahalanobis M_i = (R C1_i R^T + C2_j)^-1 with R the rotation
//                 of transformation_ * guess in double; estimateRigidTransformationBFGS; delta = max over the 4 x 4 entries of |change| /
//                 (rotation_epsilon for the 3 x 3 block, transformation_epsilon elsewhere); converged iff iterations >= max or delta < 1
//   inner         x = (t, euler ZYX of transformation_), f(x) = 1/m sum d^T M d with d = T(x) p_src - p_tgt in float, gradient through
//                 computeRDerivative; BFGS (bfgs.h) until its gradient test passes or max_optimizer_iterations steps
//   result        final = previous_transformation_ * guess (float)
// setUseReciprocalCorrespondences has no effect: GICP's computeTransformation runs its own search loop.
#pragma once
#include <vector>

#include "nn.h"

namespace orc {

struct PclGicp {
    int    k_correspondences = 20;       // setCorrespondenceRandomness
    double max_corr_dist = 5.0;          // corr_dist_threshold_ (ctor default 5), setMaxCorrespondenceDistance
    double trans_eps = 5e-4;             // transformation_epsilon_ (GICP ctor), setTransformationEpsilon
    double rot_eps = 2e-3;               // rotation_epsilon_
    double gicp_epsilon = 1e-3;
    int    max_iterations = 200;         // max_iterations_ (GICP ctor), setMaximumIterations
    int    max_inner_iterations = 20;    // setMaximumOptimizerIterations
    double translation_gradient_tolerance = 1e-2, rotation_gradient_tolerance = 1e-2;  // PCL >= 1.11
    int    whole_gradient_norm = 0;      // 1: pclomp / PCL <= 1.10: |g| < 1e-2 over all six components
    int    num_threads = 1;
    int    sum_threads = 1;              // > 1: pclomp's accumulation for that many OpenMP threads — f_array / g_array / R_array[omp_get_thread_num()] over the static
                                         // chunks of the correspondence list (libgomp: the first m mod T threads take one more), the per-thread partials
                                         // then added in thread order (pcl_gicp.cpp Functor::terms).  1: one chain over all correspondences (serial pcl::GICP)
    int    gpu_order = 0;                // diagnostic: the cost sums in the HIP kernels' order of additions (pcl_gicp.cpp Functor::terms_gpu_order)

    std::vector<float>  target, source;  // xyzi
    std::vector<double> target_covs, source_covs;  // 9 per point
    bool   target_covs_valid = false, source_covs_valid = false;

    float  final_[16];  // row-major
    bool   converged = false;
    int    nr_iterations = 0, n_evaluations = 0, n_inner_steps = 0;

    void   set_target(const float* xyzi, int n);
    void   set_source(const float* xyzi, int n);
    void   align(const float guess_rowmajor[16], float* aligned);
    double fitness(double max_range) const;
    void   get_covariances(int which, double* out9);
    // the cost of estimateRigidTransformationBFGS at x for the correspondences of `T_rowmajor` (= transformation_, guess identity): f and g[6]; tests
    double evaluate(const float T_rowmajor[16], const double x[6], double g[6], int* n_corr);

   private:
    NnGrid target_grid_;
    bool   target_grid_valid_ = false;
    void   compute_covariances(const std::vector<float>& cloud, std::vector<double>& covs) const;
    void   ensure();
};

}  // namespace orc
