// csrc/gicp_engine.h — GICP engines on MI355X: GICP_HIP = fast_gicp::FastGICP (registrations.cpp:55-63) and
// SMALL_GICP_HIP = small_gicp::RegistrationPCL (registrations.cpp:46-54, the YAML default).  Both share the k-NN
// covariances, the 1-NN correspondence search and the Mahalanobis weights; they differ in the side the pose is
// perturbed on (Jacobian) and in the Levenberg-Marquardt schedule, which is the `variant` of GicpParams.
#pragma once
#include "common.h"
#include <vector>

#include "nn_grid.h"

namespace mrgfe {

struct GicpParams {
    int    k_correspondences = 20;
    double max_corr_dist = 2.0;
    double trans_eps = 5e-4;
    double rot_eps = 2e-3;
    int    max_iterations = 64;
    int    lm_max_iterations = 10;
    double lm_init_lambda_factor = 1e-9;
    // variant 1, small_gicp: T <- T exp(d), J = [R skew(a) | -R], error 0.5 r^T M r; LevenbergMarquardtOptimizer with
    // lambda 1e-3, factor 10, at most 10 inner trials, a trial accepted iff its error does not exceed the current one;
    // converged iff |d_rot| <= rot_eps and |d_trans| <= trans_eps
    int    variant = 0;
    // variant 2, fast_gicp::FastVGICP (registrations.cpp:76-84; the algorithm of the FAST_VGICP_CUDA slot :65-75): the target is
    // a GaussianVoxelMap of edge voxel_resolution (mean of the points and of their covariances per voxel), a source point
    // corresponds to the voxel its transformed position falls in, terms weighted by sqrt(points in the voxel); fast_gicp's LM
    double voxel_resolution = 1.0;
    // variant 3, pcl::IterativeClosestPoint (registrations.cpp:85-92): nearest target point within max_corr_dist of the cumulatively
    // transformed source, TransformationEstimationSVD (Umeyama, no scaling) per iteration, DefaultConvergenceCriteria with
    // translation threshold trans_eps (on the squared translation) and rotation threshold 1 - trans_eps; no covariances
    // ... with use_reciprocal (setUseReciprocalCorrespondences, registrations.cpp:91): a pair (i, j) counts only if the nearest point of
    // target j among the source points AS TRANSFORMED SO FAR is i again, within max_corr_dist (determineReciprocalCorrespondences):
    // a second exact-NN grid, over the working copy of the source, rebuilt every iteration
    bool   use_reciprocal = false;
    // variant 4, pcl::GeneralizedIterativeClosestPoint (registrations.cpp:93-103; pclomp::GICP :104-114 with pcl_whole_gradient_norm): PCL's own
    // covariances (raw float moments, singular values (1, 1, 1e-3)), per outer iteration nearest target points within max_corr_dist and
    // Mahalanobis matrices (R C1 R^T + C2)^-1, inner BFGS (csrc/bfgs.h) over (t, euler ZYX) for at most max_inner_iterations steps,
    // converged iff max entry change / (rot_eps | trans_eps) < 1 or the iteration limit is reached
    int    max_inner_iterations = 20;
    bool   pcl_whole_gradient_norm = false;
    // serial pcl::GICP (registration_method "GICP") is deterministic: its cost and gradient sums are added in point order, and so are ours
    // (pclgicp_seqsum_kernel) unless this is false (pclomp::GICP: per-thread sums without a fixed order -> the block tree)
    bool   pcl_reference_order_sums = false;
    // pclomp::GICP (PCL_GICP_OMP_HIP): T > 1 = the sums as T OpenMP threads form them — T chains over the static chunks of the correspondence list, the
    // partials added in thread order (gicp.hip pclgicp_chunksum_kernel); 0 / 1: the block tree (matches no reference, 5 x faster per evaluation)
    int    pcl_omp_sum_threads = 0;
    double sg_init_lambda = 1e-3, sg_lambda_factor = 10.0;
    int    sg_max_inner_iterations = 10;
};

class GicpEngine {
   public:
    GicpEngine(mrgfe_ctx* ctx, const GicpParams& prm) : ctx_(ctx), prm_(prm) {}
    ~GicpEngine();
    int set_target(const void* d_xyzi, size_t n);
    int set_source(const void* d_xyzi, size_t n, const float* enclosing_box = nullptr);  // enclosing_box (min xyz, max xyz; all points finite): NnGrid::build's known_box
    // the keyframe update of the odometry (scan_matching_odometry_component.cpp:326-339: keyframe = the scan just aligned): the source cloud becomes the
    // target WITH what was computed for it as a source — its k-NN covariances and the grid they were found through, which is the grid the correspondence
    // search runs on (prepare_target builds exactly these for a new target): no k-NN search, no grid build.  The source stays set, without covariances.
    int source_becomes_target();
    int align(const float guess_rowmajor[16]);
    int aligned_cloud(float* out_xyzi_host);
    // update_correspondences + linearize at T (row-major double 4x4): tests
    int linearize(const double T[16], double H[36], double b[6], double* err, int* n_corr);
    int covariances(int which, double* out9);  // 0 source, 1 target

    // k-NN covariances of a packed device cloud into `out` (6 doubles per point), through `grid` (rebuilt over the cloud)
    int compute_covariances(const float4* d_pts, size_t n, DevBuf& out, NnGrid& grid);
    // batch access (GicpBatch): the target's covariances and correspondence grid, built on demand
    int prepare_target();
    const float4*     target_points() const { return d_tgt_; }
    size_t            target_size() const { return n_tgt_; }
    const double*     target_covariances() const { return d_tgt_cov_.as<double>(); }
    const NnGrid2Dev& target_grid() const { return tgt_grid_.dev2(); }
    // VGICP: the voxel records (10 doubles per voxel) and the voxel grid geometry, valid after prepare_target()
    const double*     voxel_records() const { return d_vox_.as<double>(); }
    void              voxel_grid(double* res, int32_t cmin[3], int32_t dim[3], uint32_t* n_cells) const;
    uint32_t          voxels_occupied() const { return vox_occupied_; }
    NnGrid&           scratch_grid() { return cov_grid_; }
    const GicpParams& params() const { return prm_; }

    const float* final_transformation() const { return final_; }
    bool converged() const { return converged_; }
    int  iterations() const { return nr_iterations_; }
    int  evaluations() const { return n_linearize_ + n_error_; }
    const double* hessian() const { return final_hessian_; }

    double  kernel_ms = 0;
    int64_t kernel_launches = 0;
    double  kernel_alg_bytes = 0;

   private:
    mrgfe_ctx* ctx_;
    GicpParams prm_;
    const float4* d_tgt_ = nullptr;
    const float4* d_src_ = nullptr;
    size_t n_tgt_ = 0, n_src_ = 0;
    NnGrid tgt_grid_;
    NnGrid cov_grid_;        // k-NN grid of the source cloud (covariances only; buffers reused)
    DevBuf d_knn_i_, d_knn_d_;
    bool   tgt_grid_valid_ = false, tgt_cov_valid_ = false, src_cov_valid_ = false;
    bool   src_box_valid_ = false;
    float  src_box_[6] = {0, 0, 0, 0, 0, 0};
    DevBuf d_tgt_cov_, d_src_cov_, d_corr_, d_mahal_, d_partial_, d_T_;
    DevBuf d_vox_, d_vox_runs_;  // VGICP voxel records; first / last run positions (build scratch)
    bool     vox_valid_ = false;
    double   vox_res_ = 1.0;
    int32_t  vox_cmin_[3] = {0, 0, 0}, vox_dim_[3] = {1, 1, 1};
    uint32_t vox_cells_ = 0, vox_occupied_ = 0;
    int build_voxelmap();
    int align_icp(const float guess_rowmajor[16]);
    int align_pcl_gicp(const float guess_rowmajor[16]);
   public:
    // PCL_GICP_HIP: (search) the correspondences + Mahalanobis matrices at transformation_ = T with `guess`, and / or (x != NULL) the functor of
    // estimateRigidTransformationBFGS at x over the stored correspondences of the points d_pts: f, g[6], number of correspondences
    int pcl_evaluate(const float T_rowmajor[16], const float guess_rowmajor[16], const float4* d_pts, bool search, const double x[6], double* f, double g[6], int* n_corr);
    const float4* source_points() const { return d_src_; }
   private:
    DevBuf   d_chunk_bounds_;  // PCL_GICP_OMP_HIP: the chunk boundaries (source-point indices) of pclomp's per-thread sums for the current correspondences
    DevBuf   d_terms_;      // PCL_GICP_HIP: the per-point terms of one cost / gradient evaluation, a column per sum
    PinBuf   h_rec_;        // the reduced record of a linearisation / error evaluation, written by the device (gicp_reduce_host_kernel)
    uint64_t rec_tag_ = 0;  // ... and the tag of the last one asked for
    DevBuf d_cur_;  // ICP: the source as transformed so far
    NnGrid cur_grid_;  // ICP with reciprocal correspondences: exact-NN grid over d_cur_
    float  final_[16];
    double final_hessian_[36];
    bool   converged_ = false;
    int    nr_iterations_ = 0, n_linearize_ = 0, n_error_ = 0;
    int ensure_ready();

    int run_linearize(const double T[16], bool with_jacobian, double H[36], double b[6], double* err, int* n_corr);
    int run_error(const double T[16], double* err);
};

// One Levenberg-Marquardt loop of fast_gicp::LsqRegistration as a resumable state machine: the same decisions, in the
// same order, as GicpEngine::align — which is this controller driven one request at a time.
struct GicpRequest {
    int    type;   // 0: update_correspondences + linearize at T, 1: compute_error at T
    double T[16];  // row-major
};
class GicpLmController {
   public:
    void start(const GicpParams& prm, const float guess_rowmajor[16], uint32_t n_src);
    bool done() const { return done_; }
    const GicpRequest& request() const { return req_; }
    void on_result(const double r[32]);  // record of gicp_reduce: err, b[6], H upper[21], n_corr
    bool converged() const { return converged_; }
    int  iterations() const { return nr_iterations_; }
    int  evaluations() const { return n_linearize_ + n_error_; }
    const double* hessian() const { return final_hessian_; }
    void final_transformation(float out_rowmajor[16]) const { for (int i = 0; i < 16; ++i) out_rowmajor[i] = static_cast<float>(x0_[i]); }

   private:
    GicpParams  prm_;
    GicpRequest req_;
    bool   done_ = true, converged_ = false;
    int    nr_iterations_ = 0, n_linearize_ = 0, n_error_ = 0;
    int    outer_ = 0, inner_ = 0;
    double x0_[16], xi_[16], delta_[16], H_[36], b_[6], d_[6], y0_ = 0, lambda_ = -1, nu_ = 2;
    double final_hessian_[36];
    void propose();       // next LM trial from (H_, b_, lambda_): request compute_error at xi_
    void end_outer(bool ok);
    void on_result_small(const double r[32]);  // variant 1
};

// Batched GICP_HIP: the candidates of a batch advance through their LM loops together, one launch per kernel per round
// (blockIdx.y = busy pair), like the NDT rounds.  Source covariances are computed cloud by cloud beforehand.
struct GicpBatchPair {
    int           target = -1;
    const float4* d_src = nullptr;
    uint32_t      n = 0;
    float         guess[16];
    DevBuf        cov, corr, mahal;
    // keyframe store (api.cpp): covariances kept with the cloud; *ext_cov_k == k_correspondences means they are valid
    DevBuf*       ext_cov = nullptr;
    int*          ext_cov_k = nullptr;
    GicpLmController ctl;
};
class GicpBatch {
   public:
    explicit GicpBatch(mrgfe_ctx* ctx) : ctx_(ctx) {}
    ~GicpBatch();
    // engines[t] holds target t (set_target done); pairs: target index, device source cloud, guess (row-major)
    int align_all(std::vector<GicpEngine*>& engines, std::vector<GicpBatchPair>& pairs);

   private:
    mrgfe_ctx* ctx_;
    // helper streams + workspaces for the source covariances (created on first use, kept)
    struct Lane { mrgfe_ctx* ctx = nullptr; NnGridSet set; std::vector<NnGrid> views; DevBuf knn_i, knn_d; };
    std::vector<Lane*> lanes_;
    DevBuf d_pairs_, d_evals_, d_grids_, d_partials_;
    PinBuf h_evals_, h_results_;
    hipEvent_t done_ = nullptr;
};

// PCL_GICP_HIP: 1 the cost / gradient sums in the reference's (point) order, 0 in a tree; other values query
int gicp_set_pcl_reference_order(int mode);
// correspondence search of GICP_HIP / SMALL_GICP_HIP: 1 the passes of nn_nearest_batch for large batches (default), 0 one lane group per query always, 2 the passes always
int gicp_set_corr_passes(int mode);

}  // namespace mrgfe
