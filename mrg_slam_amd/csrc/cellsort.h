// csrc/cellsort.h — batched device primitives under every grid structure of the front end: bounding box,
// stable LSD radix sort of (cell key, point index) pairs, exclusive scan, segment (run) detection.
//
// "Batched" = several independent problems per launch: blockIdx.y selects the problem, blockIdx.x a 2048-element
// tile of it.  Problem p owns elements [off, off+n) of every per-element array and tiles [blk_off, blk_off+nblk)
// of every per-tile array.
#pragma once
#include "common.h"

namespace mrgfe {

constexpr int kTile = 2048;  // elements per workgroup tile (256 threads x 8)

struct Slice {
    uint32_t n;        // elements of this problem
    uint32_t off;      // first element in the packed per-element arrays
    uint32_t blk_off;  // first tile in the packed per-tile arrays
    uint32_t nblk;     // ceil(n / kTile)
};

// host-side helper: lays problems of sizes n[0..P) out back to back (offsets aligned to 4 elements)
struct SliceTable {
    std::vector<Slice> h;
    uint32_t total_elems = 0, total_blks = 0, max_blks = 0;
    void build(const uint32_t* n, int nprob);
    int  nprob() const { return static_cast<int>(h.size()); }
};

struct BBox {  // result of bounding_boxes, one per problem (32 bytes)
    float    mn[3];
    uint32_t n_finite;
    float    mx[3];
    uint32_t pad;
};

// d_slices: device copy of SliceTable::h.  All functions enqueue on ctx->stream and do not synchronise.

// min / max over finite points (pcl::getMinMax3D) of each problem. d_clouds[p] = packed float4 cloud of problem p.
// d_partial needs total_blks BBox entries; d_out nprob entries.
int bounding_boxes(mrgfe_ctx* ctx, const float4* const* d_clouds, const Slice* d_slices, const SliceTable& t, BBox* d_partial, BBox* d_out);

// the first half of bounding_boxes only: one box per 2048-point tile in d_partial (a consumer that merges them itself: bbox_device.h)
int bounding_box_partials(mrgfe_ctx* ctx, const float4* const* d_clouds, const Slice* d_slices, const SliceTable& t, BBox* d_partial);

// stable LSD radix sort by the low `key_bits` bits. Buffers ping-pong; *out_keys/*out_vals point at the sorted data
// (either the input or the tmp buffers). d_hist needs (total_blks + nprob) * 256 words.
// `iota_vals`: the values are the element indices 0 .. n-1 of each problem and d_vals need not be filled (the first pass makes them up);
// `first_hist_ready`: d_hist already holds the per-tile histograms of the lowest digit in rs_hist_kernel's layout, [tile][256] (the kernel
// that made the keys counted them while it had them in registers).
int radix_sort_pairs(mrgfe_ctx* ctx, uint32_t* d_keys, uint32_t* d_vals, uint32_t* d_keys_tmp, uint32_t* d_vals_tmp, const Slice* d_slices,
                     const SliceTable& t, int key_bits, uint32_t* d_hist, uint32_t** out_keys, uint32_t** out_vals, bool iota_vals = false, bool first_hist_ready = false);

// exclusive prefix sum of uint32 per problem; d_totals[p] = sum. d_blk needs total_blks words. in == out allowed.
int exclusive_scan(mrgfe_ctx* ctx, const uint32_t* d_in, uint32_t* d_out, const Slice* d_slices, const SliceTable& t, uint32_t* d_blk, uint32_t* d_totals);

// the first step of exclusive_scan only: d_blk[tile] = sum of the tile's 2048 elements (a consumer that adds up the tiles before its own itself)
int tile_sums(mrgfe_ctx* ctx, const uint32_t* d_in, const Slice* d_slices, const SliceTable& t, uint32_t* d_blk);
// the first step of exclusive_scan_run_heads only: d_blk[tile] = run heads in the tile
int run_head_tile_counts(mrgfe_ctx* ctx, const uint32_t* d_sorted_keys, const Slice* d_slices, const SliceTable& t, const uint32_t* d_n_valid, uint32_t* d_blk);

// run heads of a sorted key array: element i starts a run of equal keys among the first d_n_valid[p] elements of problem p (the invalid
// keys sort behind them).  d_out[i] = number of run heads before element i (the ordinal of i's run when i is a head), d_totals[p] = runs.
// d_out == nullptr: only d_totals and the exclusive prefix of the tiles' head counts in d_blk are produced (what ndt_launch_segments needs).
int exclusive_scan_run_heads(mrgfe_ctx* ctx, const uint32_t* d_sorted_keys, uint32_t* d_out, const Slice* d_slices, const SliceTable& t, const uint32_t* d_n_valid, uint32_t* d_blk,
                             uint32_t* d_totals);

}  // namespace mrgfe
