// csrc/ndt_derivatives.h — launchers of the NDT derivative / reduction kernels (ndt_derivatives.hip).
#pragma once
#include "common.h"
#include "ndt_types.h"

namespace mrgfe {

// one derivative evaluation for the `npairs` pairs listed in NdtEvalDev::order[mode]; every workgroup takes `ppt` tiles of
// 256 points, grid = (max_nblk x npairs) with max_nblk = ceil(largest source cloud / (256 * ppt))
int ndt_launch_derivatives(mrgfe_ctx* ctx, int mode, int search, uint32_t max_nblk, int npairs, const NdtGridDev* d_grids, const NdtPairDev* d_pairs,
                           const NdtEvalDev* d_evals, double* d_partials, int ppt, uint32_t spec_part_base);
// fixed-order sum of the block partials -> results[pair][48] = {score, g[6], H[36] row-major, neighbours, pad}
// with_spec: also reduce the speculative f64 Hessian records (second half of the partial buffer) into
// results[spec_result_base + pair]
int ndt_launch_reduce(mrgfe_ctx* ctx, int npairs, const NdtPairDev* d_pairs, const NdtEvalDev* d_evals, const double* d_partials, double* d_results, bool with_spec,
                      uint32_t spec_part_base, uint32_t spec_result_base, const int ppt[3] /* tiles per workgroup of the three variants' launches */);
// dst = T * src (row-major 3x4 float T in device memory)
int launch_transform_cloud(mrgfe_ctx* ctx, const float4* d_src, float4* d_dst, uint32_t n, const float* d_T12);

}  // namespace mrgfe
