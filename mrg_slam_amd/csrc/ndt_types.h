// csrc/ndt_types.h — device-visible data layout of the NDT target grid and of one derivative evaluation.
//
// HBM layout: a target grid is
//   leaves   : NdtLeafRec[V]   48 B each: mean as 3 x f64 (the reference subtracts the f64 mean from the f32 point,
//                              pclomp computeDerivatives), inverse covariance as 6 x f32 upper triangle (it is cast
//                              to float in updateDerivatives anyway)
//   icov64   : double[V][9]    only read by the rare double-precision Hessian pass (pclomp computeHessian)
//   lookup   : dense  int32[D]          voxel key -> leaf id (-1 = no usable voxel) when D = div_b product <= 2^22
//              hashed uint2[capacity]   (key, leaf id) open addressing, linear probing, when the box is larger
//   nr_points, keys : int32[V] bookkeeping (ascending key = std::map iteration order of the reference)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mrgfe {

constexpr int      kNdtMinPointsPerVoxel = 6;      // pclomp::VoxelGridCovariance::min_points_per_voxel_
constexpr double   kNdtMinCovarEigMult   = 0.01;   // min_covar_eigvalue_mult_
// a leaf is invalid when one of its two smaller covariance eigenvalues is below minus this: 0 in ndt_omp's filter (forked from PCL 1.8), dummy_precision() = 1e-12
// in pcl::VoxelGridCovariance of PCL >= 1.11, which PCL_NDT_HIP's targets follow: an exactly planar voxel (eigenvalue ~ -1e-18) is inflated and kept there
constexpr double   kPclVgcNegativeEigenTolerance = 1e-12;
constexpr uint32_t kHashEmpty            = 0xFFFFFFFFu;
constexpr uint32_t kDenseLookupMaxCells  = 1u << 22;

struct __attribute__((aligned(16))) NdtLeafRec {
    double mean[3];
    float  icov[6];  // xx, xy, xz, yy, yz, zz
};
static_assert(sizeof(NdtLeafRec) == 48, "leaf record is 48 bytes (SURVEY.md §8d byte model)");

struct NdtGridDev {
    int32_t  min_b[3];
    int32_t  max_b[3];
    int32_t  divb_mul[3];
    float    leaf_size;
    float    inv_leaf;
    uint32_t n_cells;      // D = div_b product
    uint32_t dense;        // 1: lookup is int32[D]; 0: uint2 hash slots
    uint32_t hash_shift;   // 32 - log2(capacity)
    uint32_t hash_mask;    // capacity - 1
    uint32_t n_leaves;     // V
    const void*       lookup;
    const NdtLeafRec* leaves;
    const double*     icov64;
    const float4*     centroid;  // float centroid of every leaf (KDTREE search)
    const int32_t*    nr_points;
};

__device__ __forceinline__ uint32_t ndt_hash(uint32_t key, uint32_t shift) { return (key * 0x9E3779B1u) >> shift; }

// leaf id of voxel `key`, or -1
__device__ __forceinline__ int32_t ndt_lookup(const NdtGridDev& g, uint32_t key)
{
    if (g.dense) return static_cast<const int32_t*>(g.lookup)[key];
    const uint2* slots = static_cast<const uint2*>(g.lookup);
    uint32_t h = ndt_hash(key, g.hash_shift);
    for (uint32_t probe = 0; probe <= g.hash_mask; ++probe) {
        uint2 s = slots[h];
        if (s.x == key) return static_cast<int32_t>(s.y);
        if (s.x == kHashEmpty) return -1;
        h = (h + 1) & g.hash_mask;
    }
    return -1;
}

// one alignment in flight: static part
struct NdtPairDev {
    const float4* src;     // packed xyzi
    uint32_t      n_src;
    uint32_t      grid;    // index into the NdtGridDev array
    uint32_t      part_off;  // first block-partial record of this pair
    uint32_t      nblk;      // block-partial records reserved for this pair: ceil(n_src / 256), the workgroups of a launch with one tile per workgroup
};

// per-evaluation part: the pending request of one alignment, written by its controller step (ctl::fill_eval in ndt_ctl.h —
// on the device by ndt_reduce_control_kernel, or by the host in the host-stepped path) and read by the derivative kernels
struct NdtEvalDev {
    float    T[12];        // row-major 3x4 of final_transformation_ (pcl::transformPointCloud operand)
    float    j_ang[8][3];  // computeAngleDerivatives rows a..h (double products cast to float)
    float    h_ang[15][3]; // rows a2,a3,b2,b3,c2,c3,d1,d2,d3,e1,e2,e3,f1,f2,f3
    double   j_ang_d[8][3];
    double   h_ang_d[15][3];
    double   gauss_d1, gauss_d2;
    int32_t  mode;         // 0: score+grad+hess (float path), 1: score+grad, 2: hessian only (double path)
    int32_t  active;       // 0: this pair is finished, no kernel touches it
    int32_t  search;       // mrgfe_ndt_search
    int32_t  pad;
};

// Plan of one round, built on the device by ndt_plan_kernel from the pending requests: which pairs each kernel variant has
// work for and how that work is cut into items of `ppt` tiles (256 points each).  A derivative launch has a FIXED grid; its
// workgroups walk the items of their variant (item = blockIdx.x, += gridDim.x), so the host needs to know neither how many
// pairs are still running nor which variant they want: dispatching one y-slice per pair of the batch instead would cost
// ~100 us per launch in workgroups that exit at once (measured on MI355X), and asking the host would cost a round trip.
// Layout (uint32 words): NdtPlanHead (16 words), then for m = 0..2: pair_of[m][P] | item_start[m][P + 1].
struct NdtPlanHead {
    uint32_t n_pairs[3];   // busy pairs per variant
    uint32_t n_items[3];   // work items per variant
    uint32_t ppt[3];       // tiles per item: clamp(tiles of the variant / (4 * CUs), 1, 8) — full batches amortise the 44 wave
                           // reductions of an item's epilogue over 8 tiles, a lone straggler spreads over the whole chip
    uint32_t n_active;     // pairs still running
    uint32_t round;
    uint32_t pad[5];
};
constexpr uint32_t kNdtPlanHeadWords = 16;
static_assert(sizeof(NdtPlanHead) == 4 * kNdtPlanHeadWords, "plan head is 16 words");
__host__ __device__ inline size_t   ndt_plan_words(uint32_t P) { return kNdtPlanHeadWords + 3 * (2 * size_t(P) + 1); }
__host__ __device__ inline uint32_t ndt_plan_pair_off(uint32_t P, int m) { return kNdtPlanHeadWords + uint32_t(m) * (2 * P + 1); }
__host__ __device__ inline uint32_t ndt_plan_start_off(uint32_t P, int m) { return ndt_plan_pair_off(P, m) + P; }

// what the plan kernel tells the host about a round (pinned host memory, polled: the host only decides when to stop enqueueing)
struct NdtRoundInfo {
    uint32_t tag;          // round number + 1, written last
    uint32_t n_active;
    uint32_t n_pairs[3];
    uint32_t n_items[3];
};

// Snapshot of a batch between two rounds (ndt_snapshot_kernel, pinned host memory): which alignments have finished, and their final
// transformations — what the fitness passes of finished pairs need while the stragglers still iterate (mrgfe_batch_align).
struct NdtSnapshotRec {
    uint32_t done;     // 1: the optimiser has finished, T12 is final
    float    T12[12];  // row-major 3 x 4
};
struct NdtSnapshotHead {
    uint32_t tag;  // request number, written last
    uint32_t n_done;
    uint32_t pad[2];
};

// one evaluation in the reference's summation order (ndt_ref_records_kernel / ndt_ref_chain_kernel): which pair, which kind, where its records start
struct NdtRefJob {
    uint32_t pair;
    uint32_t mode;
    uint64_t rec_off;  // doubles into the record workspace
    uint64_t cnt_off;  // bytes into the per-point pair-count workspace (mode 2)
};

// block partial / final result of one evaluation: score, gradient(6), full 6x6 Hessian(36, row-major), neighbour count.
// All 36 Hessian entries are kept: the reference fills H(i,j) and H(j,i) with differently rounded float terms, and an
// ill-conditioned Newton solve amplifies that 1e-7 asymmetry far above the 1e-4 parity bar if it is mirrored away.
constexpr int kNdtAccum = 44;  // 1 + 6 + 36 + 1
constexpr int kNdtNbIndex = 43;
constexpr int kNdtPartialStride = 48;

}  // namespace mrgfe
