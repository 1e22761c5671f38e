// csrc/gicp_engine.h — GICP_HIP engine: fast_gicp::FastGICP (registrations.cpp:55-63) on MI355X.
#pragma once
#include "common.h"
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
};

class GicpEngine {
   public:
    GicpEngine(mrgfe_ctx* ctx, const GicpParams& prm) : ctx_(ctx), prm_(prm) {}
    ~GicpEngine();
    int set_target(const void* d_xyzi, size_t n);
    int set_source(const void* d_xyzi, size_t n);
    int align(const float guess_rowmajor[16]);
    int aligned_cloud(float* out_xyzi_host);
    // update_correspondences + linearize at T (row-major double 4x4): tests
    int linearize(const double T[16], double H[36], double b[6], double* err, int* n_corr);
    int covariances(int which, double* out9);  // 0 source, 1 target

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
    DevBuf d_tgt_cov_, d_src_cov_, d_corr_, d_mahal_, d_partial_, d_T_;
    float  final_[16];
    double final_hessian_[36];
    bool   converged_ = false;
    int    nr_iterations_ = 0, n_linearize_ = 0, n_error_ = 0;
    int ensure_ready();
    int compute_covariances(const float4* d_pts, size_t n, DevBuf& out, NnGrid& grid);
    int run_linearize(const double T[16], bool with_jacobian, double H[36], double b[6], double* err, int* n_corr);
    int run_error(const double T[16], double* err);
};

}  // namespace mrgfe
