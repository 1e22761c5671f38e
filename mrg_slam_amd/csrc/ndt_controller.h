// csrc/ndt_controller.h — host-side optimiser of NDT_HIP: pclomp::NormalDistributionsTransform::
// computeTransformation + computeStepLengthMT unrolled into a resumable state machine.
//
// The reference runs Newton iterations with a More-Thuente line search whose every trial calls
// computeDerivatives over all source points.  Here each such call is one request to the GPU (`request()`), and the
// controller resumes when the 28 reduced sums come back (`on_result()`).  Many controllers advance in lock-step in a
// batch: one kernel launch serves the pending request of every alignment still running.
#pragma once
#include <cstdint>

namespace mrgfe {

struct NdtParams {
    float  resolution = 1.0f;
    double step_size = 0.1;
    double outlier_ratio = 0.55;
    double trans_eps = 0.1;
    int    max_iterations = 35;
    int    search = 2;  // MRGFE_DIRECT7
};

// request for one derivative evaluation (what NdtEvalDev carries to the device)
struct NdtRequest {
    int    mode;          // 0 score+grad+hess, 1 score+grad, 2 hessian only (double)
    bool   spec_hessian;  // with mode 0: also evaluate the f64 Hessian (computeHessian) at the same pose in this round
    float  T[16];         // row-major final_transformation_
    double p[6];          // pose vector the angular derivative tables are built for
    double j_ang[8][3];
    double h_ang[15][3];
};

class NdtController {
   public:
    enum Phase { IDLE, INIT, LS_FIRST, LS_ITER, LS_HESS, DONE };

    void start(const NdtParams& prm, const float guess_rowmajor[16], uint32_t n_src);
    bool done() const { return phase_ == DONE || phase_ == IDLE; }
    const NdtRequest& request() const { return req_; }
    // reduced sums of the requested evaluation: r[0] score, r[1..6] gradient, r[7..42] Hessian (row-major), r[43] neighbours
    // r_spec: the f64 Hessian record when the request asked for it (spec_hessian), else nullptr
    void on_result(const double r[44], const double* r_spec = nullptr);
    // finish immediately without target (no usable grid): align() leaves final = guess semantics of an empty run
    void abort_no_target();

    // pcl::Registration read-outs
    const float* final_transformation() const { return final_; }  // row-major
    bool   converged() const { return converged_; }
    int    iterations() const { return nr_iterations_; }
    int    evaluations() const { return n_evals_; }          // derivative evaluations of the reference's control flow
    int    reused_evaluations() const { return n_reused_; }
    int    speculative_hessians_used() const { return n_spec_used_; }  // of those, served from the previous identical trial
    double trans_probability() const { return trans_probability_; }
    const double* hessian() const { return H_; }  // 6x6 row-major, symmetric
    double neighbours_sum() const { return nb_sum_; }
    double gauss_d1() const { return gauss_d1_; }
    double gauss_d2() const { return gauss_d2_; }

    // helpers shared with the API layer
    static void angle_tables(const double p[6], double j_ang[8][3], double h_ang[15][3]);
    static void pose_to_matrix(const double p[6], float M[16]);
    static void euler_xyz(const float M[16], float out[3]);

   private:
    NdtParams prm_;
    Phase     phase_ = IDLE;
    uint32_t  n_src_ = 0;
    NdtRequest req_;
    float  final_[16];
    float  transformation_[16], previous_[16];
    bool   converged_ = false;
    int    nr_iterations_ = 0, n_evals_ = 0, n_reused_ = 0;
    double cache_p_[6], cache_nb_ = 0;
    bool   cache_valid_ = false;
    double H_spec_[36], spec_p_[6], spec_nb_ = 0;  // speculative f64 Hessian of the last first trial and its pose
    bool   spec_valid_ = false;
    int    n_spec_used_ = 0;
    double trans_probability_ = 0, nb_sum_ = 0;
    double gauss_d1_ = 0, gauss_d2_ = 0, gauss_d3_ = 0;
    double p_[6], score_ = 0, g_[6], H_[36];
    // More-Thuente state
    double x_[6], x_t_[6], dir_[6];
    double phi_0_, d_phi_0_, a_l_, f_l_, g_l_, a_u_, f_u_, g_u_, a_t_, phi_t_, d_phi_t_, psi_t_, d_psi_t_;
    bool   interval_converged_, open_interval_;
    int    step_iterations_;

    void make_request(int mode, const double p[6]);
    void store_result(const double r[44], bool with_score_grad, bool with_hessian);
    void newton_step();
    void ls_after_eval();
    void ls_iter_update();
    void ls_continue_or_finish();
    void finish_line_search(double a_t);
};

}  // namespace mrgfe
