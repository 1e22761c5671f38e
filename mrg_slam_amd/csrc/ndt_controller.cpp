// csrc/ndt_controller.cpp — host prologue of an NDT_HIP alignment (see ndt_controller.h; the state machine itself is
// ndt_ctl.h): pcl::Registration::align's resets and pclomp::NormalDistributionsTransform::computeTransformation's prologue as
// the reference reaches them through registration_->align() (/root/reference/apps/scan_matching_odometry_component.cpp:265-266;
// src/mrg_slam/loop_detector.cpp:134).  Compiled with -ffp-contract=off: the float expressions execute in the written order.
#include "ndt_controller.h"

#include <cmath>
#include <cstdlib>
#include <cstring>

namespace mrgfe {

static bool reuse_enabled()
{
    static const bool on = [] { const char* e = std::getenv("MRGFE_NO_REUSE"); return !(e && e[0] == '1'); }();
    return on;
}

static bool newton_svd_only()
{
    static const bool on = [] { const char* e = std::getenv("MRGFE_NEWTON_SVD"); return e && e[0] == '1'; }();
    return on;
}

void NdtController::euler_xyz(const float M[16], float out[3])
{
    // Eigen 3.4 Matrix3f::eulerAngles(0, 1, 2)
    const float kPi = 3.14159265358979323846f;
    float r0 = std::atan2(M[1 * 4 + 2], M[2 * 4 + 2]);
    const float c2 = std::sqrt(M[0] * M[0] + M[1] * M[1]);
    float r1;
    if (r0 > 0.0f) { r0 -= kPi; r1 = std::atan2(-M[2], -c2); }
    else           { r1 = std::atan2(-M[2], c2); }
    const float s1 = std::sin(r0), c1 = std::cos(r0);
    const float r2 = std::atan2(s1 * M[2 * 4 + 0] - c1 * M[1 * 4 + 0], c1 * M[1 * 4 + 1] - s1 * M[2 * 4 + 1]);
    out[0] = -r0; out[1] = -r1; out[2] = -r2;
}

void NdtController::start(const NdtParams& prm, const float guess[16], uint32_t n_src, bool split_first)
{
    std::memset(&s_, 0, sizeof(s_));
    s_.step_size = prm.step_size;
    s_.trans_eps = prm.trans_eps;
    s_.outlier_ratio = prm.outlier_ratio;
    s_.resolution = prm.resolution;
    s_.max_iterations = prm.max_iterations;
    s_.search = prm.search;
    s_.reuse = reuse_enabled() ? 1 : 0;
    s_.split_first = (split_first && prm.formulation == 0) ? 1 : 0;
    s_.formulation = prm.formulation;
    s_.svd_only = newton_svd_only() ? 1 : 0;
    s_.n_src = n_src;
    // pcl::Registration::align
    ctl::identity16(s_.final_); ctl::identity16(s_.transformation_); ctl::identity16(s_.previous_);
    // computeTransformation prologue
    const double c1 = 10 * (1 - prm.outlier_ratio);
    const double c2 = prm.outlier_ratio / std::pow(static_cast<double>(prm.resolution), 3);
    s_.gauss_d3 = -std::log(c2);
    s_.gauss_d1 = -std::log(c1 + c2) - s_.gauss_d3;
    s_.gauss_d2 = -2 * std::log((-std::log(c1 * std::exp(-0.5) + c2) - s_.gauss_d3) / s_.gauss_d1);
    if (n_src == 0) { s_.phase = NDT_DONE; return; }
    if (!ctl::is_identity16(guess)) std::memcpy(s_.final_, guess, sizeof(s_.final_));
    float eul[3];
    euler_xyz(s_.final_, eul);
    s_.p[0] = s_.final_[3]; s_.p[1] = s_.final_[7]; s_.p[2] = s_.final_[11];
    s_.p[3] = eul[0]; s_.p[4] = eul[1]; s_.p[5] = eul[2];
    ctl::make_request(s_, 0, s_.p);
    s_.phase = NDT_INIT;
}

void NdtController::abort_no_target()
{
    ctl::identity16(s_.final_);
    s_.converged = 0;
    s_.phase = NDT_DONE;
}

}  // namespace mrgfe
