// oracle/ndt.h — TEST INFRASTRUCTURE (CPU oracle). Not part of the shipped product path.
//
// CPU restatement of pclomp::VoxelGridCovariance + pclomp::NormalDistributionsTransform (koide3/ndt_omp,
// un-vendored dependency of the reference: /root/reference/CMakeLists.txt:84, package.xml:18,
// src/mrg_slam/registrations.cpp:130-148) and of pcl::Registration::align/getFitnessScore as the
// reference calls them (apps/scan_matching_odometry_component.cpp:203,208,265-276;
// src/mrg_slam/loop_detector.cpp:104,127,134-144).  PARITY UNPINNED (see quirks.h): follows SURVEY.md
// Appendix A.2-A.5.
#pragma once
#include <cstddef>
#include <cstdint>
#include <unordered_map>
#include <vector>

namespace orc {

enum NdtSearch { NDT_KDTREE = 0, NDT_DIRECT26 = 1, NDT_DIRECT7 = 2, NDT_DIRECT1 = 3 };

struct NdtLeaf {
    int    key;          // linear voxel index (ijk - min_b) . divb_mul
    int    nr_points;    // >= 6 valid; -1 invalidated by the eigenvalue / inf checks; < 6 unused
    double mean[3];
    double cov[9];
    double icov[9];
    float  centroid[4];  // float accumulation of x,y,z,intensity (KDTREE search uses xyz)
    int    in_search;    // 1: the leaf had >= 6 points in the second pass, so its centroid went into voxel_centroids_ (and the kd-tree) BEFORE the
                         // eigenvalue / inf checks could set nr_points = -1: radiusSearch still returns it (no nr_points test there), with the
                         // zero inverse covariance of the Leaf constructor (eigenvalue check) or the non-finite one (inf check)
};

// pclomp::VoxelGridCovariance<PointXYZI>::applyFilter + getNeighborhoodAtPoint{,7,1}
struct VoxelGridCovariance {
    float leaf_size = 1.0f, inv_leaf = 1.0f;
    int   min_b[3] = {0, 0, 0}, max_b[3] = {0, 0, 0}, div_b[3] = {0, 0, 0}, divb_mul[3] = {0, 0, 0};
    std::vector<NdtLeaf>         leaves;  // ascending key (std::map iteration order)
    std::unordered_map<int, int> index;   // key -> leaves[] position (all leaves, also < 6 points)
    int n_valid = 0;
    double negative_eigen_tolerance = 0.0;  // quirks.h: 0 for pclomp's filter, 1e-12 for PCL 1.12's (set by PclNdt)

    // returns 0 ok, -1 index overflow ("Leaf size is too small"), -2 empty input
    int build(const float* xyzi, int n, float leaf);
    // neighbours (leaf positions) of an already transformed point; returns count (<= 27)
    int neighbours(float x, float y, float z, NdtSearch method, int out[27]) const;
    // VoxelGridCovariance::radiusSearch(point, leaf size) as upstream has it: every leaf whose centroid went into the kd-tree (in_search),
    // sorted by (distance, leaf position); leaves the later checks rejected are among them
    int radius_neighbours(float x, float y, float z, int out[27]) const;
};

// computeStepLengthMT's helpers (identical text in PCL's ndt.hpp and in ndt_omp): shared by ndt.cpp and pcl_ndt.cpp
bool   mt_update_interval(double& a_l, double& f_l, double& g_l, double& a_u, double& f_u, double& g_u, double a_t, double f_t, double g_t);
double mt_trial_value_selection(double a_l, double f_l, double g_l, double a_u, double f_u, double g_u, double a_t, double f_t, double g_t);
// the HIP derivative kernels' reduction tree (diagnostic "GPU order" modes of ndt.cpp and pcl_ndt.cpp): 256 lanes x 48 slots -> one item record
// (64-lane shuffle tree, offsets 32..1; the four waves as ((w0 + w1) + w2) + w3), and the item records of an evaluation in four interleaved slices
void gpu_tree_reduce(std::vector<double>& acc /* [256][48] */, double out[48]);
void gpu_slice_reduce(const std::vector<double>& partials /* [nblk][48] */, size_t nblk, double out[48]);

struct Ndt {
    // parameters (defaults = ndt_omp ctor; mrg_slam overrides through registrations.cpp:134-146)
    float  resolution        = 1.0f;
    double step_size         = 0.1;
    double outlier_ratio     = 0.55;
    double trans_eps         = 0.1;
    int    max_iterations    = 35;
    int    num_threads       = 1;
    NdtSearch search         = NDT_DIRECT7;
    bool   fused             = true;   // float/double three-term products accumulated with FMA (see ndt.cpp dot3f)
    int    gpu_order_ppt     = 0;      // > 0: diagnostic — add the terms in the HIP kernels' order, items of this many 256-point tiles (ndt.cpp)
    bool   thread_sums       = false;  // true: upstream's accumulation — one score / gradient / Hessian accumulator per OpenMP thread, added up in
                                       // thread order (ndt_omp computeDerivatives): the sums depend on the schedule in the last bits, which is why the
                                       // checker keeps per-point records and adds them in point order; this mode is what bench.py TIMES (cpu_baseline)

    VoxelGridCovariance cells;
    std::vector<float> target, source;  // xyzi
    int target_status = -2;

    // results
    float  final_[16];  // row-major
    bool   converged   = false;
    int    nr_iterations = 0;
    double trans_probability = 0;
    double hessian[36];
    int    n_evals = 0;          // derivative evaluations executed (all modes)
    double last_p[6] = {0, 0, 0, 0, 0, 0};  // pose vector of the last derivative evaluation (diagnostic: the optimiser's trajectory in double)
    double neighbours_sum = 0;   // sum over evals of mean valid neighbour voxels per point (k-bar numerator)

    int  set_target(const float* xyzi, int n);
    void set_source(const float* xyzi, int n);
    // pcl::Registration::align + computeTransformation. guess row-major. aligned (n*4 floats xyzi) optional.
    void align(const float guess[16], float* aligned);
    // pcl::Registration::getFitnessScore(max_range) on the last final transformation
    double fitness(double max_range) const;

    // one derivative evaluation at pose vector p with the points transformed by T (row-major):
    // mode 0: score+gradient+hessian (float path), 1: score+gradient only, 2: hessian only (double path,
    // computeHessian). Exposed for kernel-level parity tests.
    double evaluate(const float T[16], const double p[6], int mode, double grad[6], double hess[36]);

   private:
    double gauss_d1 = 0, gauss_d2 = 0, gauss_d3 = 0;
    double j_ang_d[8][3], h_ang_d[15][3];
    float  j_ang_f[8][3], h_ang_f[15][3];
    std::vector<double> scores_, grads_, hessians_;  // per-point results, summed in index order
    std::vector<float>  trans_;                      // transformed cloud xyz (stride 3)
    void   init_gauss();
    void   angle_derivatives(const double p[6], bool compute_hessian = true);
    void   transform_cloud(const float T[16]);
    double compute_derivatives(double grad[6], double hess[36], const double p[6], bool compute_hessian);
    void   compute_hessian(double hess[36], const double p[6]);
    double compute_derivatives_gpu_order(double grad[6], double hess[36], const double p[6], bool compute_hessian);
    void   compute_hessian_gpu_order(double hess[36], const double p[6]);
    int    neighbours_probe_order(float x, float y, float z, int out[27]) const;
    template <bool FUSED> double compute_derivatives_impl(double grad[6], double hess[36], const double p[6], bool compute_hessian);
    template <bool FUSED> void   compute_hessian_impl(double hess[36], const double p[6]);
    double step_length_mt(const double x[6], double step_dir[6], double step_init, double step_max, double step_min, double& score,
                          double grad[6], double hess[36]);
};

}  // namespace orc
