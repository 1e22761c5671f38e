// csrc/ndt_engine.h — batched NDT_HIP engine: owns the device-resident clouds and target grids of a batch of
// (target, source, guess) alignments and advances all of them together, one derivative launch per round.
// A single pcl::Registration-style object (mrgfe_reg) is a batch with one target and one pair.
#pragma once
#include <atomic>
#include <vector>

#include "cellsort.h"
#include "common.h"
#include "ndt_build.h"
#include "ndt_controller.h"
#include "ndt_types.h"

namespace mrgfe {

struct NdtTargetInfo {
    const float4* d_pts = nullptr;
    uint32_t      n = 0;
    int           status = MRGFE_ERR_STATE;  // MRGFE_OK once the grid is built
    int32_t       min_b[3] = {0, 0, 0}, max_b[3] = {0, 0, 0}, div_b[3] = {0, 0, 0};
    uint32_t      n_leaves = 0;
    uint32_t      leaf_off = 0;
    bool          built = false;
};

struct NdtPairInfo {
    int           target = -1;
    const float4* d_src = nullptr;
    uint32_t      n = 0;
    float         guess[16];  // row-major
    NdtController ctl;
};

// who steps the optimiser of the following alignments: -1 automatic (default; MRGFE_HOST_CONTROL overrides), 0 device, 1 host
void ndt_set_host_control(int mode);
// derivative launches of the following alignments: 1 = one launch per round for all kernel variants (default; MRGFE_FUSED
// overrides), 0 = one launch per variant; any other value only asks.  Returns the setting in effect.
int ndt_set_fused_launch(int mode);
// NDT_OMP sums of the following alignments: 1 = the reference's own order (per-point sums, then point-order chains: ndt_ref_*_kernel; host-stepped, several
// times slower), 0 = the tree (default; MRGFE_NDT_REFERENCE_ORDER sets the initial value); any other value only asks.  Returns the setting in effect.
int ndt_set_reference_order(int mode);

// A second host thread may ask a running device-controlled align_all() for snapshots (which pairs have finished, their final
// transformations): it raises `want`; the aligning thread enqueues ndt_snapshot_kernel between two rounds and counts `issued` up; the
// kernel writes the records and then head->tag = issued.  `finished` is set when align_all() returns (no more snapshots).
struct NdtSnapshotPort {
    std::atomic<int>      want{0};
    std::atomic<uint32_t> issued{0};
    std::atomic<int>      finished{0};
    std::atomic<uint32_t> n_active{0xFFFFFFFFu};  // pairs still running as of the last round plan the aligning thread has seen (published for the other thread)
    PinBuf                buf;  // NdtSnapshotHead, then NdtSnapshotRec[P]
    NdtSnapshotHead*      head() const { return buf.as<NdtSnapshotHead>(); }
    NdtSnapshotRec*       recs() const { return reinterpret_cast<NdtSnapshotRec*>(buf.as<char>() + sizeof(NdtSnapshotHead)); }
};

class NdtEngine {
   public:
    NdtEngine(mrgfe_ctx* ctx, const NdtParams& prm) : ctx_(ctx), prm_(prm) {}
    ~NdtEngine();

    void clear();                 // forget targets and pairs (device memory is kept for reuse)
    void clear_pairs();
    // clouds: host (strided) or device (packed float4). device clouds are referenced, not copied.
    int add_target_host(const float* xyzi, size_t n, size_t stride);
    int add_target_device(const void* d_xyzi, size_t n);
    int add_pair_host(int target, const float* xyzi, size_t n, size_t stride, const float guess_rowmajor[16]);
    int add_pair_device(int target, const void* d_xyzi, size_t n, const float guess_rowmajor[16]);
    int set_guess(int pair, const float guess_rowmajor[16]);

    int build_targets(bool wait = true);  // voxelise every target not yet built; `wait`: return with the stream drained
    int align_all(NdtSnapshotPort* port = nullptr);  // run every pair to completion (port: see NdtSnapshotPort; batches under device control only)
    // one derivative evaluation of pair `pair` (tests): mode 0/1/2
    int evaluate(int pair, const float T_rowmajor[16], const double p[6], int mode, double* score, double grad[6], double hess[36]);
    int aligned_cloud(int pair, float* out_xyzi_host);  // final_transformation * source

    int n_targets() const { return static_cast<int>(targets_.size()); }
    int n_pairs() const { return static_cast<int>(pairs_.size()); }
    const NdtTargetInfo& target(int i) const { return targets_[i]; }
    const NdtPairInfo&   pair(int i) const { return pairs_[i]; }
    int read_leaves(int target, int32_t* keys, int32_t* nr_points, double* mean3, double* icov9);

    // derivative-kernel accounting of the last align_all(), per kernel variant (mode 0 / 1 / 2): device time from HIP
    // events around each launch, launch count, algorithmic bytes (SURVEY.md §8d model)
    double  mode_ms[3] = {0, 0, 0};
    int64_t mode_launches[3] = {0, 0, 0};
    double   largest_ms = 0;                 // longest timed derivative launch of the last align_all and the pairs of each kind its round had busy
    uint32_t largest_pairs[3] = {0, 0, 0};
    double  mode_alg_bytes[3] = {0, 0, 0};
    double  mode_points[3] = {0, 0, 0};      // source points of the evaluations actually launched (reused trials are not)
    double  mode_neighbours[3] = {0, 0, 0};  // valid (point, voxel) pairs they found
    void kernel_stats(int mode, double* ms, int64_t* launches, double* bytes) const
    {
        double m = 0, b = 0;
        int64_t l = 0;
        for (int k = 0; k < 3; ++k)
            if (mode < 0 || mode == k) { m += mode_ms[k]; l += mode_launches[k]; b += mode_alg_bytes[k]; }
        if (ms) *ms = m;
        if (launches) *launches = l;
        if (bytes) *bytes = b;
    }

    const NdtParams& params() const { return prm_; }
    void set_force_hash(bool f) { force_hash_ = f; }  // tests: exercise the hashed lookup on small grids
    mrgfe_ctx* ctx() const { return ctx_; }
    int rounds() const { return rounds_; }  // rounds of the last align_all()

   private:
    mrgfe_ctx* ctx_;
    NdtParams  prm_;
    std::vector<NdtTargetInfo> targets_;
    std::vector<NdtPairInfo>   pairs_;
    Arena cloud_arena_;  // host-supplied clouds copied to the device
    Arena grid_arena_;   // leaves, lookups, ...
    // packed per-leaf arrays of the built targets (grid_arena_)
    std::vector<NdtGridDev> h_grids_;
    DevBuf d_grids_, d_pairs_, d_evals_, d_partials_, d_T12_, d_aligned_;
    DevBuf d_ticket_;              // ndt_derivatives_single_kernel's workgroup counter: 0 between launches (the last workgroup clears it)
    bool   ticket_dirty_ = false;  // a round was enqueued whose record has not been seen: the counter is cleared before the next one
    PinBuf h_evals_, h_results_;
    bool   pairs_dirty_ = true;
    bool   force_hash_ = false;
    int    forced_ppt_ = 0;  // MRGFE_PPT tuning hook (0: chosen per launch)
    uint32_t max_nblk_ = 0;
    uint32_t total_part_blocks_ = 0;
    std::vector<NdtPairDev> h_pairs_;
    // per-target arrays kept for read_leaves
    struct LeafArrays { int32_t* keys; int32_t* nr_points; NdtLeafRec* leaves; double* icov64; };
    std::vector<LeafArrays> leaf_arrays_;

    // rounds (see align_all)
    DevBuf d_states_;                     // NdtCtlState[P] (device control)
    PinBuf h_states_, h_info_;            // staging of the states, NdtRoundInfo per round (written by the plan kernel)
    size_t evals_bytes_ = 0;              // the round's plan lives behind the requests in d_evals_ / h_evals_ (one copy per host-stepped round)
    uint32_t* d_plan() const { return reinterpret_cast<uint32_t*>(d_evals_.as<char>() + evals_bytes_); }
    std::vector<uint32_t> plan_scratch_;
    void host_plan(std::vector<uint32_t>& plan, uint32_t wg_target, uint32_t max_ppt) const;
    std::vector<hipEvent_t> ev_pool_;     // [round][variant][begin, end]
    int    rounds_ = 0;
    int      key_bits_hint_ = 0;   // key width of this engine's last single-target build + 1 (0: none yet): lets the next one sort before the host has seen its box
    uint64_t result_tag_ = 0;  // host-stepped single registration: the value its next reduction stores behind the record
    std::vector<NdtRoundInfo> round_info_;  // busy pairs per kind of every round of the last align_all
    int upload_pairs();
    int ensure_events(size_t rounds);
    uint32_t derivative_grid(int mode) const;
    int enqueue_round(uint32_t round, bool device_control, const bool want_mode[3], NdtRoundInfo* h_info, double result_tag = 0.0);
    int reference_round();                // the same round in the reference's summation order (ndt_set_reference_order)
    DevBuf d_ref_rec_, d_ref_cnt_, d_ref_jobs_;
    void account(const std::vector<NdtRoundInfo>& info, size_t rounds);
};

}  // namespace mrgfe
