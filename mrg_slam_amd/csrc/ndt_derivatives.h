// csrc/ndt_derivatives.h — launchers of the NDT plan / derivative / reduce(+control) kernels (ndt_derivatives.hip).
#pragma once
#include "common.h"
#include "ndt_ctl.h"
#include "ndt_types.h"

namespace mrgfe {

// the round's plan from the pending requests of the P pairs (ndt_types.h: NdtPlanHead); h_info[round] (pinned, may be NULL)
// receives the number of pairs still running
int ndt_launch_plan(mrgfe_ctx* ctx, const NdtPairDev* d_pairs, const NdtEvalDev* d_evals, uint32_t P, uint32_t* d_plan, uint32_t wg_target, uint32_t max_ppt, uint32_t forced_ppt,
                    uint32_t round, NdtRoundInfo* h_info);
// between two rounds: which pairs have finished and their final transformations, into pinned host memory (h_head->tag = tag, written last)
int ndt_launch_snapshot(mrgfe_ctx* ctx, const NdtCtlState* d_states, uint32_t P, uint32_t tag, NdtSnapshotHead* h_head, NdtSnapshotRec* h_recs);
// one derivative evaluation for the pairs the plan lists under `mode`: `grid` workgroups walk the plan's items
int ndt_launch_derivatives(mrgfe_ctx* ctx, int mode, int search, uint32_t grid, const NdtGridDev* d_grids, const NdtPairDev* d_pairs, const NdtEvalDev* d_evals,
                           const uint32_t* d_plan, uint32_t P, double* d_partials, int formulation = 0);
// the same for all three variants in ONE launch (items interleaved in proportion to their counts)
int ndt_launch_derivatives_all(mrgfe_ctx* ctx, int search, uint32_t grid, const NdtGridDev* d_grids, const NdtPairDev* d_pairs, const NdtEvalDev* d_evals, const uint32_t* d_plan,
                               uint32_t P, double* d_partials, int formulation = 0);  // formulation 1: the f64 items of PCL_NDT_HIP (radius search, 27 probes)
// fixed-order sum of the item partials of every pending evaluation.  d_states != NULL: followed by the controller step on the
// device (next request written to d_evals).  d_states == NULL: results[pair][48] = {score, g[6], H[36] row-major, neighbours, pad}
// ONE host-stepped registration's round in one launch: the items of its pending evaluation, then — in the workgroup that finishes last — the sum of the item
// records into h_results (pinned) and the tag behind them (ndt_launch_reduce without states, P == 1).  *d_ticket must be 0 before the first launch; the kernel leaves it 0.
int ndt_launch_single_round(mrgfe_ctx* ctx, int search, uint32_t grid, const NdtGridDev* d_grids, const NdtPairDev* d_pairs, const NdtEvalDev* d_evals, const uint32_t* d_plan,
                            double* d_partials, uint32_t* d_ticket, double* h_results, double tag);
int ndt_launch_reduce(mrgfe_ctx* ctx, uint32_t P, const NdtPairDev* d_pairs, NdtEvalDev* d_evals, const double* d_partials, const uint32_t* d_plan, double* d_results,
                      NdtCtlState* d_states, double tag = 0.0, uint32_t* d_ticket = nullptr);  // d_ticket (a zeroed word the kernel leaves zero): P > 1 records + ONE tag  // tag != 0 and P == 1: stored at d_results[48] once the record is visible to the host
// NDT_OMP in the reference's summation order (opt-in): per job the records kernel (what every step of the chain adds) and the chain kernel (one lane per
// accumulator, in order); results[pair][48] like ndt_launch_reduce without states.  max_tiles = ceil(max n_src of the jobs / 256).
size_t ndt_ref_record_doubles(int mode, size_t n_src, int nnb);  // workspace of one job (tile-major record layout, ndt_derivatives.hip)
int ndt_launch_ref_round(mrgfe_ctx* ctx, int search, const NdtGridDev* d_grids, const NdtPairDev* d_pairs, const NdtEvalDev* d_evals, const NdtRefJob* d_jobs, uint32_t n_jobs,
                         uint32_t max_tiles, double* d_rec, uint8_t* d_cnt, double* results, bool any_mode01, bool any_mode2);
// diagnostic: ctl::pose_to_matrix / angle_tables / svd_solve6 for n cases of 48 doubles (p[6], A[36], b[6]) on the device
int ndt_ctl_math_device(mrgfe_ctx* ctx, const double* d_in, int n, float* d_M, double* d_tables, double* d_x);
// diagnostic: n_vals (44, 37 or 1) doubles per lane and wavefront, summed by wave_sum_fold and by wave_sum (dev_utils.h)
int wave_fold_check_device(mrgfe_ctx* ctx, int n_vals, const double* d_in, int cases, double* d_fold, double* d_plain);
int ndt_ctl_svd_wave_device(mrgfe_ctx* ctx, const double* d_in, int n, double* d_x);  // the wavefront form of the solve, one case per workgroup
int glibc_exp_device(mrgfe_ctx* ctx, const double* d_x, size_t n, double* d_out);  // diagnostic: glibc_exp (glibc_exp.h) on n doubles
// diagnostic builds (-DNDT_PHASE_CLOCK): prints the phase clocks of the derivative kernel to stderr; a no-op otherwise
void ndt_phase_dump();
// dst = T * src (row-major 3x4 float T in device memory)
int launch_transform_cloud(mrgfe_ctx* ctx, const float4* d_src, float4* d_dst, uint32_t n, const float* d_T12);

}  // namespace mrgfe
