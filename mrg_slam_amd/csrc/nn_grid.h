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
#include "cellsort.h"
#include "common.h"

namespace mrgfe {

struct NnGridDev {
    float           origin[3];
    float           cell;
    int32_t         dim[3];
    uint32_t        n;           // points in `sorted`
    const uint32_t* cell_start;  // dim product + 1
    const float4*   sorted;      // xyz + original index (bit pattern) in w
};

class NnGrid {
   public:
    // (re)build over a packed float4 device cloud
    int build(mrgfe_ctx* ctx, const float4* d_pts, size_t n, float cell_size);
    void release();
    bool valid() const { return built_; }
    const NnGridDev& dev() const { return h_; }
    size_t size() const { return n_; }

    // mean squared 1-NN distance of T*src over points whose squared distance <= max_range (PCL getFitnessScore)
    int fitness(mrgfe_ctx* ctx, const float4* d_src, size_t n_src, const float T_rowmajor[16], double max_range, double* out);
    // 1-NN of host queries
    int nearest_host(mrgfe_ctx* ctx, const float* q, size_t n, size_t stride, int32_t* idx, float* sqd);
    // 1-NN of device queries, optionally transformed by a row-major 3x4 float matrix in device memory (may be null)
    int nearest_device(mrgfe_ctx* ctx, const float4* d_q, size_t n, const float* d_T12, int32_t* d_idx, float* d_sqd);
    // flags[i] = 1 iff #{j : sqdist(q_i, p_j) <= r2} >= need   (queries are the device cloud d_q)
    int radius_count_flags(mrgfe_ctx* ctx, const float4* d_q, size_t n, double r2, int need, uint32_t* d_flags);
    // k nearest neighbours (ascending by (sqdist, index)) of every query: d_idx / d_sqd are [n][k]; missing -> -1
    int knn_device(mrgfe_ctx* ctx, const float4* d_q, size_t n, int k, int32_t* d_idx, float* d_sqd);

   private:
    bool      built_ = false;
    size_t    n_ = 0;
    NnGridDev h_;
    DevBuf    d_cell_start_, d_sorted_;
};

}  // namespace mrgfe
