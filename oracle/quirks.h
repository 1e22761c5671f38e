// oracle/quirks.h — TEST INFRASTRUCTURE (CPU oracle). Not part of the shipped product path.
//
// Every constant and behavioural quirk that the CPU restatement inherits from the un-vendored upstream
// libraries the reference links against (PCL 1.12, koide3/ndt_omp, SMRT-AIST/fast_gicp; see
// /root/reference/CMakeLists.txt:26,84-85,91 and src/mrg_slam/registrations.cpp:4-24).
//
// PARITY UNPINNED: none of those libraries (nor Eigen/FLANN) exist in this container and the reference
// ships no tests or golden vectors, so these values are restated from the published upstream sources
//  and are kept in ONE place so they can be corrected in one edit.
#pragma once

namespace orc {
namespace quirks {

// ---- pclomp::VoxelGridCovariance (ndt_omp voxel_grid_covariance_omp.h) ------------------------------
constexpr int    kNdtMinPointsPerVoxel   = 6;     // min_points_per_voxel_
constexpr double kNdtMinCovarEigvalMult  = 0.01;  // min_covar_eigvalue_mult_
// A leaf is invalid when an eigenvalue of its covariance is negative.  ndt_omp's copy of the filter (forked from PCL 1.8) tests `< 0`; pcl::VoxelGridCovariance of
// PCL >= 1.11 — what pcl::NormalDistributionsTransform (registration_method "NDT") builds its target with — tests `< -Eigen::NumTraits<double>::dummy_precision()`
// (= -1e-12) for the two smaller ones, so that an exactly planar voxel (smallest eigenvalue ~ -1e-18 by rounding) is inflated and kept instead of dropped.
// [UPSTREAM-RECALL, ADVICE r5: PCL is not in the image] — one edit here and in csrc/ndt_types.h (kPclVgcNegativeEigenTolerance) to correct.
constexpr double kNdtNegativeEigenTolerance    = 0.0;    // pclomp::VoxelGridCovariance
constexpr double kPclVgcNegativeEigenTolerance = 1e-12;  // pcl::VoxelGridCovariance (PCL 1.12)

// ---- pclomp::NormalDistributionsTransform defaults (ndt_omp_impl.hpp ctor) ---------------------------
constexpr float  kNdtResolution          = 1.0f;
constexpr double kNdtStepSize            = 0.1;
constexpr double kNdtOutlierRatio        = 0.55;
constexpr double kNdtTransformationEps   = 0.1;
constexpr int    kNdtMaxIterations       = 35;
// computeAngleDerivatives: angles with |a| < 10e-5 use cos=1, sin=0 exactly.
constexpr double kNdtSmallAngle          = 10e-5;
// computeAngleDerivatives, second-derivative row d1 (the x row of d^2 R / d ry^2 for R = Rx Ry Rz): upstream (PCL ndt.hpp, copied by
// ndt_omp) has  h_ang_d1 = (-cy cz, cy sz, +sy);  the true second derivative ends in -sy (tests/ndt_analytic.py builds it from
// rotation-matrix derivative products).  The restatement KEEPS upstream's +sy: the Hessian entry (ry, ry) of every registration of
// the reference carries it, and parity is with the reference, not with the thesis.  tests/test_oracle_ndt.py::
// test_derivatives_match_the_first_principles_model pins it: first principles + this one sign = the oracle.
constexpr double kNdtHAngD1ZSign         = +1.0;
// updateDerivatives evaluates the exponent in FLOAT with a float copy of gauss_d2 ("float gauss_d2 = gauss_d2_;" at the top of
// ndt_omp's updateDerivatives): e_x_cov_x = exp(-gauss_d2f * x_trans4.dot(x_trans4 * c_inv4) * 0.5f) with the unqualified ::exp, i.e.
// the float argument widened to double, exp in double, the result narrowed to float.  Audited in round 2: the
// HIP kernel (ndt_derivatives.hip pair_float: arg0 = -gauss_d2f * qCq; arg = arg0 * 0.5f; e = float(exp(double(arg)))) and
// oracle/ndt.cpp pair_terms_f execute exactly that sequence; computeHessian (updateHessian) keeps the double gauss_d2_.
// This is recalled upstream text, like everything in this file: no copy of ndt_omp exists here to confirm it.
// computeStepLengthMT (More-Thuente) constants.
constexpr int    kMtMaxStepIterations    = 10;
constexpr double kMtMu                   = 1.e-4;
constexpr double kMtNu                   = 0.9;

// ---- pcl::NormalDistributionsTransform (PCL 1.12 registration/impl/ndt.hpp; oracle/pcl_ndt.cpp) — what differs from the pclomp fork ----------
// Same class defaults (resolution 1.0, step_size 0.1, outlier_ratio 0.55, transformation_epsilon 0.1, max_iterations 35), same gauss constants,
// same small-angle rule and the same +sy in angular_hessian_ row d1, same More-Thuente constants; min_points_per_voxel_ 6 and
// min_covar_eigvalue_mult_ 0.01 in pcl::VoxelGridCovariance.  All per-pair arithmetic is f64.
// kPclNdtIterationRule 1 (PCL >= 1.11.1, the assumed 1.12.1): after each Newton step  nr_iterations_++  and then converged_ when
//   nr_iterations_ >= max_iterations_  or  |t_step|^2 <= transformation_epsilon_ (the SQUARED translation of the step's float matrix against the
//   un-squared epsilon; the rotation test cos_angle >= transformation_rotation_epsilon_ joins only when that epsilon is > 0, and the reference
//   never sets it: registrations.cpp:125-127).  With step_size 0.1 every step is at most 0.1 long, so for any epsilon >= 0.01 — mrg_slam's YAML has
//   0.1 — the first step already satisfies the test: "NDT" in the reference runs ONE Newton iteration.  0: PCL <= 1.11.0, the rule pclomp kept.
constexpr int    kPclNdtIterationRule    = 1;
// a vanishing or NaN Newton step ends computeTransformation with converged_ = (delta_norm == 0) in PCL >= 1.11.1 (true); before: = !isnan (false)
constexpr bool   kPclNdtZeroStepConverges = true;
// three-term f64 products are written left to right here; Eigen's unrolled reduction adds e0 + (e1 + e2): 1e-16 relative, not part of the contract

// ---- mrg_slam overrides (config/mrg_slam.yaml:100-109, registrations.cpp:130-148) -------------------
constexpr double kMrgTransformationEps   = 0.1;
constexpr int    kMrgMaxIterations       = 64;
constexpr double kMrgResolution          = 1.0;

// ---- pcl::VoxelGrid -----------------------------------------------------------------------------------
// In-voxel accumulation order. PCL sorts (idx, point) pairs with std::sort (unstable introsort): the order
// inside a voxel is whatever libstdc++ produces. ORDER_STABLE sums in ascending point index instead.
enum VoxelOrder { ORDER_STABLE = 0, ORDER_STD_SORT = 1 };

// ---- pcl::RadiusOutlierRemoval ------------------------------------------------------------------------
// Dense-cloud path (VoxelGrid output is_dense=true): nearestKSearch(min_pts+1) and the k-th neighbour is an
// inlier iff (double)sqdist <= radius*radius  (inclusive). The query point itself is among the neighbours.

// ---- pcl::Registration::getFitnessScore(max_range) ----------------------------------------------------
// Squared NN distance is compared against the UN-squared max_range (same quirk in
// /root/reference/src/mrg_slam/information_matrix_calculator.cpp:70).

// ---- fast_gicp::FastGICP defaults ---------------------------------------------------------------------
constexpr int    kGicpKCorrespondences   = 20;
constexpr double kGicpRotationEps        = 2e-3;
constexpr double kGicpLmInitLambdaFactor = 1e-9;
constexpr int    kGicpLmMaxIterations    = 10;
constexpr double kGicpPlaneEps           = 1e-3;  // RegularizationMethod::PLANE singular values (1,1,1e-3)

// ---- pcl::GeneralizedIterativeClosestPoint / pclomp::GeneralizedIterativeClosestPoint (pcl_gicp.cpp, bfgs.h) ----------------------------
// Recalled upstream text like everything here: gicp_epsilon_ = 0.001, k_correspondences_ = 20, rotation_epsilon_ = 2e-3, max_inner_iterations_ = 20,
// min_number_correspondences_ = 4; the BFGS parameters estimateRigidTransformationBFGS sets (sigma = rho = 0.01, tau1 = 9, tau2 = 0.05, tau3 = 0.5,
// order = 3) and the class defaults it leaves (step_size = 0.01, bracket / section iteration caps of 100); the inner loop's stopping rule:
// PCL >= 1.11 checks the translation and the rotation part of the gradient apart (each norm < 1e-2), PCL <= 1.10 and pclomp the whole norm < 1e-2.
// The initial Euler angles mix float and double library calls: x[3] and x[5] through std::atan2 on floats, x[4] through the C asin on a double.
constexpr double kPclGicpEpsilon         = 1e-3;
constexpr double kPclGicpGradientTol     = 1e-2;

}  // namespace quirks
}  // namespace orc
