// csrc/ndt_build.h — launchers of the target voxelisation kernels (ndt_build.hip).
#pragma once
#include "cellsort.h"
#include "ndt_types.h"

namespace mrgfe {

// voxel indexing parameters of one cloud (host computes them from the bounding box exactly as PCL does)
struct VoxelParams {
    int32_t  min_b[3];
    int32_t  divb_mul[3];
    float    inv_leaf;
    uint32_t n_cells;  // div_b product; also the sort key given to non-finite points
};

// where the leaves of one target live in the packed per-leaf arrays
struct LeafSlice {
    uint32_t n_leaves;   // V: voxels holding >= 1 point
    uint32_t leaf_off;   // first leaf in the per-leaf arrays
    uint32_t seg_off;    // first entry in seg_start (V + 1 entries per target)
    uint32_t n_valid;    // finite points
    uint32_t dense;      // lookup kind
    uint32_t hash_shift, hash_mask;
    uint32_t keep_rejected;  // 1: voxels with >= 6 points that the eigenvalue / inf checks rejected (nr_points -1) stay in the lookup — upstream's
                             // radiusSearch returns them (their centroid entered the kd-tree before the checks); the DIRECT searches test nr_points
    uint32_t pcl_eigen_rule;   // 1: pcl::VoxelGridCovariance's validity test (eigenvalues below -1e-12 invalidate), 0: pclomp's (below 0)
    uint32_t pad_;
    uint64_t lookup_byte_off;  // byte offset of this target's lookup table in the lookup arena
};

// voxel keys of every point and, in d_hist, the radix sort's first per-tile digit histograms (radix_sort_pairs(..., iota_vals, first_hist_ready))
int ndt_launch_cellkeys(mrgfe_ctx* ctx, const float4* const* d_clouds, const Slice* d_slices, const SliceTable& t, const VoxelParams* d_vp, uint32_t* d_keys, uint32_t* d_hist);
// seg_start / seg_key of every run head of the sorted keys; d_blk = the tiles' head-count prefixes that exclusive_scan_run_heads(..., d_out = nullptr, ...) left
int ndt_launch_segments(mrgfe_ctx* ctx, const uint32_t* d_sorted_keys, const Slice* d_slices, const SliceTable& t, const uint32_t* d_n_valid, const uint32_t* d_blk,
                        const LeafSlice* d_leaf_slices, uint32_t* d_seg_start, int32_t* d_seg_key);
int ndt_launch_leaves(mrgfe_ctx* ctx, const float4* const* d_clouds, const uint32_t* d_sorted_vals, const Slice* d_slices, const SliceTable& t, const LeafSlice* d_leaf_slices,
                      const VoxelParams* d_vp, uint32_t max_leaves, const uint32_t* d_seg_start, uint32_t* d_big_cnt, uint32_t* d_big_list, const int32_t* d_seg_key, double* d_sums, NdtLeafRec* d_leaves,
                      double* d_icov64, float4* d_centroid, int32_t* d_nr_points, void* d_lookup_base);

// A single target's voxel parameters made on the device (NdtEngine::build_targets with one host wait): the tile boxes merged, voxel_params_from_bbox's
// arithmetic float for float, the parameters and the finite-point count written where the key / run-head kernels read them, and a copy of everything the
// host wants to see behind its one wait.  A cloud without a finite point or with PCL's index overflow gets zeroed parameters and n_valid = 0.
struct DdTargetOut {
    BBox        bb;
    VoxelParams vp;
    uint32_t    n_valid;
    uint32_t    n_runs;  // written later by the run-head scan (exclusive_scan_run_heads' d_totals points here)
};
int ndt_launch_dd_voxel_params(mrgfe_ctx* ctx, const BBox* d_partial, uint32_t n_partial, float leaf, VoxelParams* d_vp, uint32_t* d_n_valid, DdTargetOut* d_out);

// host: PCL's bounding-box -> (min_b, max_b, div_b, divb_mul) arithmetic. Returns MRGFE_ERR_OVERFLOW when
// dx*dy*dz > INT32_MAX ("Leaf size is too small for the input dataset").
int voxel_params_from_bbox(const BBox& bb, float leaf, VoxelParams* vp, int32_t max_b[3], int32_t div_b[3]);

}  // namespace mrgfe
