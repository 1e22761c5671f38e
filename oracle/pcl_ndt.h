// oracle/pcl_ndt.h — TEST INFRASTRUCTURE (CPU oracle). Not part of the shipped product path.
//
// CPU restatement of pcl::NormalDistributionsTransform<PointXYZI, PointXYZI> (PCL 1.12, registration/impl/ndt.hpp) over
// pcl::VoxelGridCovariance (filters/impl/voxel_grid_covariance.hpp) — the class the reference's factory returns for
// registration_method "NDT" and for every name it does not know (/root/reference/src/mrg_slam/registrations.cpp:115-129; the
// option is listed in config/mrg_slam.yaml:98).  PCL is an un-vendored, un-pinned dependency (SURVEY.md §8c assumes 1.12.1):
// PARITY UNPINNED, every recalled constant and rule sits in quirks.h.
//
// What differs from pclomp's class (oracle/ndt.h), which is a fork of an older PCL:
//   * all per-pair arithmetic is f64 (Eigen::Vector3d / Matrix3d throughout; pclomp evaluates the pair terms in float);
//   * one neighbourhood: target_cells_.radiusSearch(x_trans_pt, resolution_) — the kd-tree over the voxel centroids, every hit used,
//     including leaves the eigenvalue check rejected after their centroid went into the tree (VoxelGridCovariance::radius_neighbours);
//   * the iteration test of computeTransformation (PCL >= 1.11.1: nr_iterations_ >= max_iterations_, or the SQUARED translation of the
//     last step <= transformation_epsilon_ [and the rotation test when transformation_rotation_epsilon_ > 0, which mrg_slam never sets]),
//     and converged_ = (delta_norm == 0) when the Newton step vanishes or is NaN;
//   * single thread, one running sum over points in order and neighbours in distance order.
#pragma once
#include <vector>

#include "ndt.h"

namespace orc {

struct PclNdt {
    // parameters: class defaults (ndt.hpp ctor); mrg_slam sets epsilon, iterations and resolution (registrations.cpp:125-127)
    float  resolution        = 1.0f;
    double step_size         = 0.1;
    double outlier_ratio     = 0.55;
    double trans_eps         = 0.1;
    double rot_eps           = 0.0;   // pcl::Registration::transformation_rotation_epsilon_ (cosine of an angle); never set by the reference
    int    max_iterations    = 35;
    int    gpu_order         = 0;     // diagnostic: != 0 adds the per-POINT factorised sums the HIP kernel forms (see pcl_ndt.cpp) instead of the per-pair ones
    int    num_threads       = 1;     // only the gpu_order diagnostic uses more than one

    VoxelGridCovariance cells;
    std::vector<float> target, source;  // xyzi
    int target_status = -2;

    // results
    float  final_[16];  // row-major
    bool   converged   = false;
    int    nr_iterations = 0;
    double trans_likelihood = 0;
    double hessian[36];
    int    n_evals = 0;
    double neighbours_sum = 0;

    int  set_target(const float* xyzi, int n);
    void set_source(const float* xyzi, int n);
    void align(const float guess[16], float* aligned);
    double fitness(double max_range) const;
    // mode 0: score + gradient + Hessian, 1: score + gradient, 2: Hessian only (computeHessian)
    double evaluate(const float T[16], const double p[6], int mode, double grad[6], double hess[36]);

   private:
    double gauss_d1 = 0, gauss_d2 = 0;
    double j_ang[8][3], h_ang[15][3];
    std::vector<float> trans_;
    void   init_gauss();
    void   angle_derivatives(const double p[6]);
    void   transform_cloud(const float T[16]);
    double compute_derivatives(double grad[6], double hess[36], const double p[6], bool compute_hessian);
    void   compute_hessian(double hess[36]);
    double derivatives_point_order(double grad[6], double hess[36], bool with_score_grad, bool with_hessian);
    double step_length_mt(const double x[6], double step_dir[6], double step_init, double step_max, double step_min, double& score, double grad[6], double hess[36]);
};

}  // namespace orc
