// csrc/ndt_controller.h — host-side handle of one NDT_HIP alignment: a thin class around the host/device state machine
// of ndt_ctl.h (pclomp::NormalDistributionsTransform::computeTransformation + computeStepLengthMT).
//
// start() runs on the host in every mode (pcl::Registration::align's prologue, the gauss constants, the Euler angles of
// the guess).  After that either the host steps the machine (on_result, the host-controlled path) or the state is copied
// to the device, advanced there by ndt_reduce_control_kernel for as many rounds as it takes, and copied back (adopt()).
#pragma once
#include <cstdint>

#include "ndt_ctl.h"

namespace mrgfe {

struct NdtParams {
    float  resolution = 1.0f;
    double step_size = 0.1;
    double outlier_ratio = 0.55;
    double trans_eps = 0.1;
    int    max_iterations = 35;
    int    search = 2;  // MRGFE_DIRECT7
    int    formulation = 0;  // 0: pclomp (NDT_HIP); 1: pcl::NormalDistributionsTransform, PCL 1.12 (PCL_NDT_HIP): f64 terms, radius search (search is
                             // MRGFE_KDTREE then), PCL's iteration test
};

class NdtController {
   public:
    // split_first: evaluate the first trial of every line search without its Hessian and fetch that only when it is used (ndt_ctl.h)
    void start(const NdtParams& prm, const float guess_rowmajor[16], uint32_t n_src, bool split_first = false);
    bool done() const { return ctl::done(s_); }
    // pending request: kernel variant (0 score+grad+hess, 1 score+grad, 2 f64 hessian) and the record the kernels read
    int  request_mode() const { return s_.req_mode; }
    void fill_eval(NdtEvalDev& e) const { ctl::fill_eval(s_, e); }
    // reduced sums of the requested evaluation: r[0] score, r[1..6] gradient, r[7..42] Hessian (row-major), r[43] neighbours
    void on_result(const double r[44]) { ctl::on_result(s_, r); }
    // finish immediately without target (no usable grid): align() leaves final = guess semantics of an empty run
    void abort_no_target();

    // device-controlled path: the state goes to the device as it is and comes back when the batch is done
    const NdtCtlState& state() const { return s_; }
    void adopt(const NdtCtlState& s) { s_ = s; }

    // pcl::Registration read-outs
    const float* final_transformation() const { return s_.final_; }  // row-major
    bool   converged() const { return s_.converged != 0; }
    int    iterations() const { return s_.nr_iterations; }
    int    evaluations() const { return s_.n_evals; }          // derivative evaluations of the reference's control flow
    int    reused_evaluations() const { return s_.n_reused; }  // of those, served from the previous identical trial
    double trans_probability() const { return s_.trans_probability; }
    const double* hessian() const { return s_.H; }  // 6x6 row-major
    void force_reference_solve() { s_.svd_only = 2; }  // every Newton solve through Eigen's two-sided JacobiSVD, operation for operation (reference-order mode)
    double neighbours_sum() const { return s_.nb_sum; }
    double gauss_d1() const { return s_.gauss_d1; }
    double gauss_d2() const { return s_.gauss_d2; }

    // helpers shared with the API layer
    static void angle_tables(const double p[6], double j_ang[8][3], double h_ang[15][3]) { ctl::angle_tables(p, j_ang, h_ang); }
    static void pose_to_matrix(const double p[6], float M[16]) { ctl::pose_to_matrix(p, M); }
    static void euler_xyz(const float M[16], float out[3]);

   private:
    NdtCtlState s_{};
};

}  // namespace mrgfe
