// csrc/nn_grid.h — exact nearest-neighbour queries on a radix-sorted uniform grid: the MI355X stand-in for the
// pcl::search::KdTree / FLANN exact searches on the hot path
//   pcl::Registration::getFitnessScore        (/root/reference/src/mrg_slam/loop_detector.cpp:137)
//   InformationMatrixCalculator::calc_fitness_score (src/mrg_slam/information_matrix_calculator.cpp:46-81)
//   getSearchMethodTarget()->nearestKSearch   (apps/scan_matching_odometry_component.cpp:405-417)
//   RadiusOutlierRemoval / StatisticalOutlierRemoval (apps/prefiltering_component.cpp:182-204)
//   fast_gicp kNN covariances and 1-NN correspondences (src/mrg_slam/registrations.cpp:55-63)
// Points are sorted by cell (stable radix sort, so ascending index inside a cell); cell rows along x are contiguous,
// so a query walks a few contiguous ranges per ring.  Ties at equal distance resolve to the lowest index.
#pragma once
#include <vector>

#include "cellsort.h"
#include "common.h"

namespace mrgfe {

struct NnGridDev {  // one resolution level
    float           origin[3];
    float           cell;
    float           slack;       // absolute bound on the binning error of floorf((p - origin) / cell) * cell, metres
    int32_t         dim[3];
    uint32_t        n;           // points in `sorted`
    int32_t         bdim[3];     // bricks of 4 x 4 x 4 cells per axis
    const uint32_t* cell_start;  // dim product + 1
    const float4*   sorted;      // xyz + original index (bit pattern) in w
    // occupancy: one 64-bit word per brick, bit x + 4 y + 16 z (cell offsets inside the brick) set iff the cell holds a
    // point.  A query whose 3x3x3 block is empty finds the occupied cells around it from a few of these words instead
    // of probing cell_start row by row.
    const unsigned long long* occ;
    const unsigned long long* occ1;  // per super-brick (4^3 bricks): which bricks are occupied
    const unsigned long long* occ2;  // per block (4^3 super-bricks)
};

// Up to three levels over the same points (cell edge x kLevelRatio per level).  k-NN queries whose neighbourhood is
// sparse at the finest scale climb to the coarser ones (nn_knn_kernel); 1-NN queries stay on the finest level and reach
// far neighbours through its occupancy pyramid (nn_device.h), so grids that only serve 1-NN are built with one level.
constexpr int kNnMaxLevels = 3;
struct NnGrid2Dev {
    NnGridDev level[kNnMaxLevels];
    int32_t   n_levels;  // >= 1
    int32_t   pad;
};

// one getFitnessScore evaluation: source cloud `src` (device) moved by the row-major 3x4 float transform, matched
// against `grid`
struct NnFitnessJob {
    NnGrid2Dev    grid;
    const float4* src;
    uint32_t      n;
    uint32_t      gicp_order;  // how T12 is applied: 0 pcl::transformPointCloud's order (getFitnessScore), 1 fast_gicp's trans_f * Vector4f (correspondence search)
    float         T12[12];
    int32_t*      idx_out;     // nn_nearest_batch: the index of the nearest target point per query (-1: none within the range); null in fitness jobs
};

class NnGrid {
   public:
    // (re)build over a packed float4 device cloud. `cell_size` is the largest cell edge; with `crowding_target` > 0 the
    // edge is halved (at most four times) while the population of the cell an average point sits in exceeds the target,
    // so a walk over the 27 cells around a query touches tens, not thousands, of candidates on dense clouds.
    static constexpr double kCrowding1nn = 12.0;  // lane-group 1-NN / radius walks: short cell runs
    static constexpr double kCrowdingKnn = 32.0;  // wave-per-query k-NN: cell rows of about one wavefront
    static constexpr float  kLevelRatio = 4.0f;
    // known_box (min xyz, max xyz): a box the caller KNOWS to enclose the cloud, all of whose points are finite — the bounding-box pass and its stream wait are
    // skipped (search results do not depend on the box, only on its enclosing the points)
    int build(mrgfe_ctx* ctx, const float4* d_pts, size_t n, float cell_size, double crowding_target = kCrowding1nn, int max_levels = kNnMaxLevels, const float* known_box = nullptr);
    void release();
    bool valid() const { return built_; }
    const NnGridDev&  dev() const { return h_.level[0]; }  // radius queries walk the finest level only
    const NnGrid2Dev& dev2() const { return h_; }
    size_t size() const { return n_; }

    // mean squared 1-NN distance of T*src over points whose squared distance <= max_range (PCL getFitnessScore)
    int fitness(mrgfe_ctx* ctx, const float4* d_src, size_t n_src, const float T_rowmajor[16], double max_range, double* out);
    NnFitnessJob make_fitness_job(const float4* d_src, size_t n_src, const float T_rowmajor[16]) const;
    // 1-NN of host queries
    int nearest_host(mrgfe_ctx* ctx, const float* q, size_t n, size_t stride, int32_t* idx, float* sqd);
    // 1-NN of device queries, optionally transformed by a row-major 3x4 float matrix in device memory (may be null)
    int nearest_device(mrgfe_ctx* ctx, const float4* d_q, size_t n, const float* d_T12, int32_t* d_idx, float* d_sqd);
    // flags[i] = 1 iff #{j : sqdist(q_i, p_j) <= r2} >= need   (queries are the device cloud d_q)
    int radius_count_flags(mrgfe_ctx* ctx, const float4* d_q, size_t n, double r2, int need, uint32_t* d_flags);
    // k nearest neighbours (ascending by (sqdist, index)) of every query: d_idx / d_sqd are [n][k]; missing -> -1
    int knn_device(mrgfe_ctx* ctx, const float4* d_q, size_t n, int k, int32_t* d_idx, float* d_sqd);
    // become a view of a grid whose device arrays another object owns (NnGridSet); release() then only forgets it
    void adopt(const NnGrid2Dev& h, size_t n) { release(); h_ = h; n_ = n; built_ = true; }

   private:
    bool      built_ = false;
    size_t    n_ = 0;
    float     hint_cell_ = 0, hint_cell_size_ = 0;  // the finest edge the adaptive passes chose last time, for which (cell_size, crowding target)
    double    hint_target_ = 0;
    NnGrid2Dev h_;
    DevBuf     d_cell_start_[kNnMaxLevels], d_sorted_[kNnMaxLevels];
    // the crowding of a build on the hinted edge is read back WITHOUT a wait and looked at by the next build of this object (below its own first wait):
    // search results do not depend on the edge, only the time does, so a cloud that has outgrown the hint costs one slower search, not a stream wait per build
    PinBuf      crowd_box_;
    bool        crowd_pending_ = false;
    hipEvent_t  crowd_event_ = nullptr;
    uint32_t    crowd_n_finite_ = 0;
    int build_levels_together(mrgfe_ctx* ctx, const float4* d_pts, uint32_t nn, const BBox& bb, const float* cell, int n_levels, double* crowding, unsigned long long* h_slots_async = nullptr);
    int count_level(mrgfe_ctx* ctx, const float4* d_pts, uint32_t nn, const BBox& bb, float cell, DevBuf& d_cells, double* crowding);
};

// Grids over several clouds built TOGETHER: every step of the build (bounding boxes, binning, the radix sort, the scan of the cell
// counts, the occupancy pyramid, the gather) is one launch over all members instead of one per cloud, and the host waits three times
// per set instead of three times per cloud.  One grid over a 130k-point scan is ~20 launches of 5-10 us each and was bound by their
// issue, not by their work: 64 of them (the targets of a loop-closure batch) took 17 ms of wall time on four streams.
// The members' device arrays live packed in the set's buffers; `out[m]` become views (NnGrid::adopt) that stay valid until the set is
// built again or released.  Search results do not depend on how a grid was built.
class NnGridSet {
   public:
    int  build(mrgfe_ctx* ctx, const float4* const* d_clouds, const uint32_t* n, int count, float cell_size, double crowding_target, int max_levels, NnGrid* const* out);
    void release();

   private:
    DevBuf d_cells_[kNnMaxLevels], d_sorted_[kNnMaxLevels];
    std::vector<float> hint_cell_;  // per member: the edge the adaptive passes settled on last time
    float  hint_cell_size_ = 0;
    double hint_target_ = 0;
};

// A single-level grid over a cloud whose SIZE THE HOST DOES NOT KNOW (round 4: the prefilter chain keeps its point counts on the device and
// waits once, at its end).  The point count comes from a Slice in device memory, the bounding box from bounding_boxes' device output; one
// single-thread kernel derives the geometry exactly as NnGrid::build_level does on the host and writes the build descriptor the *_many
// kernels read; launches are sized for `n_cap` points.  The cell table is written entry by entry from the sorted keys (nn_fill_body), so
// its capacity `cells_cap` only has to be allocated, never cleared; a cloud that needs more cells sets bit kNnAnomalyCells in *d_anomaly
// (the caller then falls back to the host-driven build).  No pyramid: the grid serves ring walks (radius counts) only.
constexpr uint32_t kNnAnomalyCells = 1u << 8;
struct NnDeviceDrivenGrid {
    DevBuf cells, sorted, desc;  // cell table (cells_cap + 8 words), cell-major points (n_cap), the NnBuildDev record
};
// d_n_boxes == NULL: d_bbox is the cloud's bounding box; else d_bbox is a list of *d_n_boxes partial boxes that the geometry kernel merges itself
int nn_build_device_driven(mrgfe_ctx* ctx, const float4* const* d_cloud_ptr, const Slice* d_slice, uint32_t n_cap, const BBox* d_bbox, const uint32_t* d_n_boxes, float cell,
                           uint32_t cells_cap, NnDeviceDrivenGrid& g, uint32_t* d_anomaly, BBox* h_box_out = nullptr);  // h_box_out: pinned; receives the (merged) box
// flags[i] = 1 iff #{j : sqdist(q_i, p_j) <= r2} >= need for the first d_slice->n points of d_q (the grid's own cloud): nn_radius_flags_kernel with
// the grid and the count read from device memory
int nn_radius_flags_device_driven(mrgfe_ctx* ctx, const NnDeviceDrivenGrid& g, const float4* d_q, const Slice* d_slice, uint32_t n_cap, double r2, int need, float cell, uint32_t* d_flags);

// the context's reusable grid (created on first use; buffers grow only) and its disposal in mrgfe_ctx_destroy
NnGrid& ctx_tmp_grid(mrgfe_ctx* ctx);

// all jobs in one launch (blockIdx.y = job); out[j] = mean squared distance or DBL_MAX when nothing is in range
int nn_fitness_batch(mrgfe_ctx* ctx, const NnFitnessJob* jobs, size_t count, double max_range, double* out);

// The same passes as a correspondence search (fast_gicp / small_gicp update_correspondences): jobs[j].idx_out[i] = index of the target point
// nearest to T * src[i] (ties: the lowest index) if its squared distance is < max_sq, else -1.  Enqueued on ctx->stream, no host wait.
int nn_nearest_batch(mrgfe_ctx* ctx, const NnFitnessJob* jobs, size_t count, double max_sq);

// far pass of the fitness score: 1 = seed + sweep (nn_fit_sweep_kernel), 0 = the pyramid walk for every queued query
int nn_set_fit_sweep(int mode);
// diagnostic counters of the seed + sweep pass: 0 off, 1 counters (FitStats::words ...), 2 also phase clocks and a line on stderr
int nn_set_fit_stats(int mode);

}  // namespace mrgfe
