// csrc/ndt_engine.cpp — batched NDT_HIP engine (see ndt_engine.h): target voxelisation pipeline and the lock-step
// alignment rounds.  Replaces, for a whole batch at once, what the reference does per object through
// setInputTarget / setInputSource / align (/root/reference/src/mrg_slam/loop_detector.cpp:104,126-145).
#include "ndt_engine.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <chrono>
#include <cstdio>
#include <cstring>

#include "ndt_derivatives.h"

namespace mrgfe {

int voxel_params_from_bbox(const BBox& bb, float leaf, VoxelParams* vp, int32_t max_b[3], int32_t div_b[3])
{
    const float inv_leaf = 1.0f / leaf;
    // "Check that the leaf size is not too small, given the size of the data"
    const int64_t dx = static_cast<int64_t>((bb.mx[0] - bb.mn[0]) * inv_leaf) + 1;
    const int64_t dy = static_cast<int64_t>((bb.mx[1] - bb.mn[1]) * inv_leaf) + 1;
    const int64_t dz = static_cast<int64_t>((bb.mx[2] - bb.mn[2]) * inv_leaf) + 1;
    if (dx * dy * dz > static_cast<int64_t>(INT32_MAX)) return MRGFE_ERR_OVERFLOW;
    for (int a = 0; a < 3; ++a) {
        vp->min_b[a] = static_cast<int32_t>(std::floor(bb.mn[a] * inv_leaf));
        max_b[a] = static_cast<int32_t>(std::floor(bb.mx[a] * inv_leaf));
        div_b[a] = max_b[a] - vp->min_b[a] + 1;
    }
    vp->divb_mul[0] = 1;
    vp->divb_mul[1] = div_b[0];
    vp->divb_mul[2] = div_b[0] * div_b[1];
    vp->inv_leaf = inv_leaf;
    const int64_t cells = static_cast<int64_t>(div_b[0]) * div_b[1] * div_b[2];
    if (cells > static_cast<int64_t>(INT32_MAX)) return MRGFE_ERR_OVERFLOW;
    vp->n_cells = static_cast<uint32_t>(cells);
    return MRGFE_OK;
}

NdtEngine::~NdtEngine()
{
    if (ctx_) (void)hipSetDevice(ctx_->device);
    cloud_arena_.release();
    grid_arena_.release();
    for (auto& g : groups_) {
        if (g.done) (void)hipEventDestroy(g.done);
        for (auto& pr : g.ev) for (auto& e : pr) if (e) (void)hipEventDestroy(e);
    }
    d_grids_.release(); d_pairs_.release(); d_evals_.release(); d_partials_.release(); d_T12_.release(); d_aligned_.release();
    h_evals_.release(); h_results_.release();
}

void NdtEngine::clear()
{
    targets_.clear();
    pairs_.clear();
    h_grids_.clear();
    leaf_arrays_.clear();
    cloud_arena_.reset();
    grid_arena_.reset();
    pairs_dirty_ = true;
}

void NdtEngine::clear_pairs()
{
    pairs_.clear();
    pairs_dirty_ = true;
}

int NdtEngine::add_target_device(const void* d_xyzi, size_t n)
{
    if (n > 0 && !d_xyzi) { set_error("add_target: NULL cloud"); return MRGFE_ERR_INVALID; }
    if (n > 0x7fffffffu) { set_error("add_target: cloud too large"); return MRGFE_ERR_INVALID; }
    NdtTargetInfo t;
    t.d_pts = static_cast<const float4*>(d_xyzi);
    t.n = static_cast<uint32_t>(n);
    targets_.push_back(t);
    return static_cast<int>(targets_.size()) - 1;
}

int NdtEngine::add_target_host(const float* xyzi, size_t n, size_t stride)
{
    if (n > 0 && !xyzi) { set_error("add_target: NULL cloud"); return MRGFE_ERR_INVALID; }
    MRGFE_TRY(ctx_->bind());
    void* d = nullptr;
    MRGFE_TRY(cloud_arena_.alloc(n * 16, &d));
    MRGFE_TRY(upload_cloud(ctx_, xyzi, n, stride, d));
    return add_target_device(d, n);
}

int NdtEngine::add_pair_device(int target, const void* d_xyzi, size_t n, const float guess[16])
{
    if (target < 0 || target >= n_targets()) { set_error("add_pair: target index %d out of range", target); return MRGFE_ERR_INVALID; }
    if (n > 0 && !d_xyzi) { set_error("add_pair: NULL cloud"); return MRGFE_ERR_INVALID; }
    if (n > 0x7fffffffu) { set_error("add_pair: cloud too large"); return MRGFE_ERR_INVALID; }
    NdtPairInfo p;
    p.target = target;
    p.d_src = static_cast<const float4*>(d_xyzi);
    p.n = static_cast<uint32_t>(n);
    std::memcpy(p.guess, guess, sizeof(p.guess));
    pairs_.push_back(p);
    pairs_dirty_ = true;
    return static_cast<int>(pairs_.size()) - 1;
}

int NdtEngine::add_pair_host(int target, const float* xyzi, size_t n, size_t stride, const float guess[16])
{
    if (n > 0 && !xyzi) { set_error("add_pair: NULL cloud"); return MRGFE_ERR_INVALID; }
    MRGFE_TRY(ctx_->bind());
    void* d = nullptr;
    MRGFE_TRY(cloud_arena_.alloc(n * 16, &d));
    MRGFE_TRY(upload_cloud(ctx_, xyzi, n, stride, d));
    return add_pair_device(target, d, n, guess);
}

int NdtEngine::set_guess(int pair, const float guess[16])
{
    if (pair < 0 || pair >= n_pairs()) { set_error("set_guess: pair index %d out of range", pair); return MRGFE_ERR_INVALID; }
    std::memcpy(pairs_[pair].guess, guess, sizeof(float) * 16);
    return MRGFE_OK;
}

// ------------------------------------------------------------------------------------------------------
// target voxelisation
// ------------------------------------------------------------------------------------------------------
int NdtEngine::build_targets()
{
    MRGFE_TRY(ctx_->bind());
    std::vector<int> todo;
    for (int i = 0; i < n_targets(); ++i) if (!targets_[i].built) todo.push_back(i);
    if (todo.empty()) return MRGFE_OK;
    const int P = static_cast<int>(todo.size());
    hipStream_t st = ctx_->stream;

    std::vector<uint32_t> sizes(P);
    for (int k = 0; k < P; ++k) sizes[k] = targets_[todo[k]].n;
    SliceTable tab;
    tab.build(sizes.data(), P);

    // descriptor block in pinned memory: slices | cloud pointers | n_valid | voxel params | leaf slices
    const size_t o_sl = 0;
    const size_t o_cp = o_sl + sizeof(Slice) * P;
    const size_t o_nv = o_cp + sizeof(void*) * P;
    const size_t o_vp = (o_nv + sizeof(uint32_t) * P + 15) & ~size_t(15);
    const size_t o_ls = (o_vp + sizeof(VoxelParams) * P + 15) & ~size_t(15);
    const size_t desc_bytes = o_ls + sizeof(LeafSlice) * P;
    PinBuf& hdesc = ctx_->pin[1];
    MRGFE_TRY(hdesc.ensure(desc_bytes + sizeof(BBox) * P + sizeof(uint32_t) * P));
    char* hd = hdesc.as<char>();
    Slice*         h_sl = reinterpret_cast<Slice*>(hd + o_sl);
    const float4** h_cp = reinterpret_cast<const float4**>(hd + o_cp);
    uint32_t*      h_nv = reinterpret_cast<uint32_t*>(hd + o_nv);
    VoxelParams*   h_vp = reinterpret_cast<VoxelParams*>(hd + o_vp);
    LeafSlice*     h_ls = reinterpret_cast<LeafSlice*>(hd + o_ls);
    BBox*          h_bb = reinterpret_cast<BBox*>(hd + desc_bytes);
    uint32_t*      h_tot = reinterpret_cast<uint32_t*>(hd + desc_bytes + sizeof(BBox) * P);

    DevBuf& ddesc = ctx_->scratch[0];
    MRGFE_TRY(ddesc.ensure(desc_bytes));
    char* dd = ddesc.as<char>();
    const Slice*         d_sl = reinterpret_cast<const Slice*>(dd + o_sl);
    const float4* const* d_cp = reinterpret_cast<const float4* const*>(dd + o_cp);
    const uint32_t*      d_nv = reinterpret_cast<const uint32_t*>(dd + o_nv);
    const VoxelParams*   d_vp = reinterpret_cast<const VoxelParams*>(dd + o_vp);
    const LeafSlice*     d_ls = reinterpret_cast<const LeafSlice*>(dd + o_ls);

    for (int k = 0; k < P; ++k) { h_sl[k] = tab.h[k]; h_cp[k] = targets_[todo[k]].d_pts; h_nv[k] = 0; }
    std::memset(h_vp, 0, sizeof(VoxelParams) * P);
    std::memset(h_ls, 0, sizeof(LeafSlice) * P);
    MRGFE_HIP_CHECK(hipMemcpyAsync(dd, hd, desc_bytes, hipMemcpyHostToDevice, st));

    // 1. bounding boxes
    DevBuf& dbb = ctx_->scratch[1];
    MRGFE_TRY(dbb.ensure(sizeof(BBox) * (tab.total_blks + P)));
    BBox* d_bb_part = dbb.as<BBox>();
    BBox* d_bb_out = d_bb_part + tab.total_blks;
    MRGFE_TRY(bounding_boxes(ctx_, d_cp, d_sl, tab, d_bb_part, d_bb_out));
    MRGFE_HIP_CHECK(hipMemcpyAsync(h_bb, d_bb_out, sizeof(BBox) * P, hipMemcpyDeviceToHost, st));
    MRGFE_HIP_CHECK(hipStreamSynchronize(st));

    // 2. voxel parameters (host, PCL arithmetic); failed targets become empty problems
    uint32_t max_cells = 1;
    for (int k = 0; k < P; ++k) {
        NdtTargetInfo& T = targets_[todo[k]];
        T.built = true;
        if (h_bb[k].n_finite == 0) { T.status = MRGFE_ERR_EMPTY; h_sl[k].n = 0; h_sl[k].nblk = 0; continue; }
        int st_vp = voxel_params_from_bbox(h_bb[k], prm_.resolution, &h_vp[k], T.max_b, T.div_b);
        if (st_vp != MRGFE_OK) { T.status = st_vp; h_sl[k].n = 0; h_sl[k].nblk = 0; std::memset(&h_vp[k], 0, sizeof(VoxelParams)); continue; }
        for (int a = 0; a < 3; ++a) T.min_b[a] = h_vp[k].min_b[a];
        T.status = MRGFE_OK;
        h_nv[k] = h_bb[k].n_finite;
        max_cells = std::max(max_cells, h_vp[k].n_cells);
    }
    // rebuild the tile bookkeeping for the (possibly emptied) problems
    for (int k = 0; k < P; ++k) sizes[k] = h_sl[k].n;
    tab.build(sizes.data(), P);
    for (int k = 0; k < P; ++k) h_sl[k] = tab.h[k];
    MRGFE_HIP_CHECK(hipMemcpyAsync(dd, hd, desc_bytes, hipMemcpyHostToDevice, st));
    int key_bits = 1;
    while (key_bits < 32 && (uint64_t(1) << key_bits) <= max_cells) ++key_bits;  // key == n_cells marks non-finite points

    // 3. keys, stable sort, run heads, ordinals
    const size_t ne = std::max<size_t>(tab.total_elems, 4);
    DevBuf &dk = ctx_->scratch[2], &dv = ctx_->scratch[3], &dkt = ctx_->scratch[4], &dvt = ctx_->scratch[5], &dh = ctx_->scratch[6], &dfl = ctx_->scratch[7], &dblk = ctx_->scratch[8];
    MRGFE_TRY(dk.ensure(ne * 4)); MRGFE_TRY(dv.ensure(ne * 4)); MRGFE_TRY(dkt.ensure(ne * 4)); MRGFE_TRY(dvt.ensure(ne * 4));
    MRGFE_TRY(dh.ensure(sizeof(uint32_t) * 256 * (tab.total_blks + P)));
    MRGFE_TRY(dfl.ensure(ne * 4));
    MRGFE_TRY(dblk.ensure(sizeof(uint32_t) * (tab.total_blks + P + 4)));
    MRGFE_TRY(ndt_launch_cellkeys(ctx_, d_cp, d_sl, tab, d_vp, dk.as<uint32_t>(), dv.as<uint32_t>()));
    uint32_t *sk = nullptr, *sv = nullptr;
    MRGFE_TRY(radix_sort_pairs(ctx_, dk.as<uint32_t>(), dv.as<uint32_t>(), dkt.as<uint32_t>(), dvt.as<uint32_t>(), d_sl, tab, key_bits, dh.as<uint32_t>(), &sk, &sv));
    MRGFE_TRY(mark_run_heads(ctx_, sk, dfl.as<uint32_t>(), d_sl, tab, d_nv));
    // flags stay in dfl; ordinals go to the unused sort buffer
    uint32_t* d_flags = dfl.as<uint32_t>();
    uint32_t* d_ord = (sk == dk.as<uint32_t>()) ? dkt.as<uint32_t>() : dk.as<uint32_t>();
    uint32_t* d_tot = dblk.as<uint32_t>() + tab.total_blks;
    MRGFE_TRY(exclusive_scan(ctx_, d_flags, d_ord, d_sl, tab, dblk.as<uint32_t>(), d_tot));
    MRGFE_HIP_CHECK(hipMemcpyAsync(h_tot, d_tot, sizeof(uint32_t) * P, hipMemcpyDeviceToHost, st));
    MRGFE_HIP_CHECK(hipStreamSynchronize(st));

    // 4. leaf storage
    uint32_t total_leaves = 0, max_leaves = 0;
    uint64_t lookup_bytes = 0;
    for (int k = 0; k < P; ++k) {
        NdtTargetInfo& T = targets_[todo[k]];
        uint32_t V = (T.status == MRGFE_OK) ? h_tot[k] : 0;
        if (V >= (1u << 24)) {  // the derivative kernel packs (tile slot, leaf id) into 8 + 24 bits
            set_error("target has %u occupied voxels; NDT_HIP supports fewer than 2^24 per target", V);
            T.status = MRGFE_ERR_INVALID;
            V = 0;
        }
        T.n_leaves = V;
        LeafSlice& ls = h_ls[k];
        ls.n_leaves = V;
        ls.leaf_off = total_leaves;
        ls.seg_off = total_leaves + k;
        ls.n_valid = h_nv[k];
        ls.lookup_byte_off = lookup_bytes;
        if (T.status == MRGFE_OK) {
            if (h_vp[k].n_cells <= kDenseLookupMaxCells && !force_hash_) {
                ls.dense = 1;
                lookup_bytes += (uint64_t(h_vp[k].n_cells) * 4 + 255) & ~uint64_t(255);
            } else {
                ls.dense = 0;
                uint32_t bits = 4;
                while ((uint64_t(1) << bits) < uint64_t(V) * 2 + 1) ++bits;
                ls.hash_shift = 32 - bits;
                ls.hash_mask = (1u << bits) - 1;
                lookup_bytes += (uint64_t(1) << bits) * 8;
            }
        }
        total_leaves += V;
        max_leaves = std::max(max_leaves, V);
    }
    void *p_keys = nullptr, *p_npts = nullptr, *p_leaves = nullptr, *p_icov = nullptr, *p_cent = nullptr, *p_lookup = nullptr;
    const size_t nl = std::max<uint32_t>(total_leaves, 1);
    MRGFE_TRY(grid_arena_.alloc(nl * 4, &p_keys));
    MRGFE_TRY(grid_arena_.alloc(nl * 4, &p_npts));
    MRGFE_TRY(grid_arena_.alloc(nl * sizeof(NdtLeafRec), &p_leaves));
    MRGFE_TRY(grid_arena_.alloc(nl * 72, &p_icov));
    MRGFE_TRY(grid_arena_.alloc(nl * 16, &p_cent));
    MRGFE_TRY(grid_arena_.alloc(std::max<uint64_t>(lookup_bytes, 256), &p_lookup));
    MRGFE_HIP_CHECK(hipMemsetAsync(p_lookup, 0xFF, std::max<uint64_t>(lookup_bytes, 256), st));
    DevBuf &dseg = ctx_->scratch[9], &dsum = ctx_->scratch[10];
    MRGFE_TRY(dseg.ensure(sizeof(uint32_t) * (total_leaves + P + 4)));
    MRGFE_TRY(dsum.ensure(sizeof(double) * 16 * nl));
    MRGFE_HIP_CHECK(hipMemcpyAsync(dd + o_ls, hd + o_ls, sizeof(LeafSlice) * P, hipMemcpyHostToDevice, st));

    // 5. segments and leaves
    MRGFE_TRY(ndt_launch_segments(ctx_, sk, d_flags, d_ord, d_sl, tab, d_ls, dseg.as<uint32_t>(), static_cast<int32_t*>(p_keys)));
    MRGFE_TRY(ndt_launch_leaves(ctx_, d_cp, sv, d_sl, tab, d_ls, d_vp, max_leaves, dseg.as<uint32_t>(), static_cast<const int32_t*>(p_keys), dsum.as<double>(),
                                static_cast<NdtLeafRec*>(p_leaves), static_cast<double*>(p_icov), static_cast<float4*>(p_cent), static_cast<int32_t*>(p_npts), p_lookup));

    // 6. device grid descriptors
    if (h_grids_.size() < targets_.size()) { h_grids_.resize(targets_.size()); leaf_arrays_.resize(targets_.size()); }
    for (int k = 0; k < P; ++k) {
        const int ti = todo[k];
        NdtTargetInfo& T = targets_[ti];
        NdtGridDev g;
        std::memset(&g, 0, sizeof(g));
        const LeafSlice& ls = h_ls[k];
        for (int a = 0; a < 3; ++a) { g.min_b[a] = T.min_b[a]; g.max_b[a] = T.max_b[a]; g.divb_mul[a] = h_vp[k].divb_mul[a]; }
        g.leaf_size = prm_.resolution;
        g.inv_leaf = 1.0f / prm_.resolution;
        g.n_cells = h_vp[k].n_cells;
        g.dense = ls.dense;
        g.hash_shift = ls.hash_shift;
        g.hash_mask = ls.hash_mask;
        g.n_leaves = ls.n_leaves;
        g.lookup = static_cast<char*>(p_lookup) + ls.lookup_byte_off;
        g.leaves = static_cast<NdtLeafRec*>(p_leaves) + ls.leaf_off;
        g.icov64 = static_cast<double*>(p_icov) + size_t(ls.leaf_off) * 9;
        g.centroid = static_cast<float4*>(p_cent) + ls.leaf_off;
        g.nr_points = static_cast<int32_t*>(p_npts) + ls.leaf_off;
        h_grids_[ti] = g;
        T.leaf_off = ls.leaf_off;
        leaf_arrays_[ti] = {static_cast<int32_t*>(p_keys) + ls.leaf_off, static_cast<int32_t*>(p_npts) + ls.leaf_off, static_cast<NdtLeafRec*>(p_leaves) + ls.leaf_off,
                            static_cast<double*>(p_icov) + size_t(ls.leaf_off) * 9};
    }
    MRGFE_TRY(d_grids_.ensure(sizeof(NdtGridDev) * h_grids_.size()));
    // pageable source: the copy is staged by the runtime before the call returns
    MRGFE_HIP_CHECK(hipMemcpyAsync(d_grids_.p, h_grids_.data(), sizeof(NdtGridDev) * h_grids_.size(), hipMemcpyHostToDevice, st));
    MRGFE_HIP_CHECK(hipStreamSynchronize(st));
    return MRGFE_OK;
}

int NdtEngine::read_leaves(int target, int32_t* keys, int32_t* nr_points, double* mean3, double* icov9)
{
    if (target < 0 || target >= n_targets() || !targets_[target].built) { set_error("read_leaves: target not built"); return MRGFE_ERR_STATE; }
    MRGFE_TRY(ctx_->bind());
    const uint32_t V = targets_[target].n_leaves;
    if (V == 0) return MRGFE_OK;
    const LeafArrays& la = leaf_arrays_[target];
    std::vector<NdtLeafRec> recs(V);
    MRGFE_HIP_CHECK(hipMemcpy(keys, la.keys, V * 4, hipMemcpyDeviceToHost));
    MRGFE_HIP_CHECK(hipMemcpy(nr_points, la.nr_points, V * 4, hipMemcpyDeviceToHost));
    MRGFE_HIP_CHECK(hipMemcpy(recs.data(), la.leaves, V * sizeof(NdtLeafRec), hipMemcpyDeviceToHost));
    MRGFE_HIP_CHECK(hipMemcpy(icov9, la.icov64, size_t(V) * 72, hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < V; ++i) { mean3[3 * i] = recs[i].mean[0]; mean3[3 * i + 1] = recs[i].mean[1]; mean3[3 * i + 2] = recs[i].mean[2]; }
    return MRGFE_OK;
}

// ------------------------------------------------------------------------------------------------------
// alignment rounds
// ------------------------------------------------------------------------------------------------------
int NdtEngine::upload_pairs()
{
    const int P = n_pairs();
    if (const char* e = std::getenv("MRGFE_PPT")) forced_ppt_ = std::max(1, std::min(std::atoi(e), 64));  // tuning experiments
    h_pairs_.resize(P);
    uint32_t part = 0;
    max_nblk_ = 0;
    for (int i = 0; i < P; ++i) {
        NdtPairDev d;
        d.src = pairs_[i].d_src;
        d.n_src = pairs_[i].n;
        d.grid = static_cast<uint32_t>(pairs_[i].target);
        d.nblk = (pairs_[i].n + 255u) / 256u;
        d.part_off = part;
        part += d.nblk;
        max_nblk_ = std::max(max_nblk_, d.nblk);
        h_pairs_[i] = d;
    }
    MRGFE_TRY(d_pairs_.ensure(sizeof(NdtPairDev) * std::max(P, 1)));
    MRGFE_TRY(d_evals_.ensure(sizeof(NdtEvalDev) * std::max(P, 1)));
    total_part_blocks_ = part;
    MRGFE_TRY(d_partials_.ensure(sizeof(double) * kNdtPartialStride * 2 * std::max<uint32_t>(part, 1)));  // second half: speculative Hessians
    MRGFE_TRY(h_evals_.ensure(sizeof(NdtEvalDev) * std::max(P, 1)));
    MRGFE_TRY(h_results_.ensure(sizeof(double) * kNdtPartialStride * 2 * std::max(P, 1)));
    if (P) MRGFE_HIP_CHECK(hipMemcpyAsync(d_pairs_.p, h_pairs_.data(), sizeof(NdtPairDev) * P, hipMemcpyHostToDevice, ctx_->stream));
    MRGFE_HIP_CHECK(hipStreamSynchronize(ctx_->stream));
    pairs_dirty_ = false;
    return MRGFE_OK;
}

static void fill_eval(NdtEvalDev& e, const NdtRequest& r, const NdtController& c, int search, bool active)
{
    for (int row = 0; row < 3; ++row) for (int col = 0; col < 4; ++col) e.T[row * 4 + col] = r.T[row * 4 + col];
    for (int a = 0; a < 8; ++a) for (int b = 0; b < 3; ++b) { e.j_ang_d[a][b] = r.j_ang[a][b]; e.j_ang[a][b] = static_cast<float>(r.j_ang[a][b]); }
    for (int a = 0; a < 15; ++a) for (int b = 0; b < 3; ++b) { e.h_ang_d[a][b] = r.h_ang[a][b]; e.h_ang[a][b] = static_cast<float>(r.h_ang[a][b]); }
    e.gauss_d1 = c.gauss_d1();
    e.gauss_d2 = c.gauss_d2();
    e.mode = r.mode;
    e.active = active ? 1 : 0;
    e.search = search;
    e.spec = r.spec_hessian ? 1 : 0;
}

// ---- rounds --------------------------------------------------------------------------------------------------------
// The pairs of a batch are split into (up to) two contiguous groups that take turns on the context stream: while the
// host steps the controllers of one group (6x6 solves, line-search logic) and uploads its next requests, the other
// group's derivative kernels are already queued, so the GPU does not idle during host turnarounds.  Both groups use
// the same stream: kernels never overlap each other and the per-launch HIP-event timings stay clean.
// batches at least this large are split into two alternating groups (each half still fills the GPU)
// HIP events around the derivative launches (what mrgfe_*_kernel_stats reports): 2 = every variant (default), 1 = only the
// dominant score+gradient+Hessian variant, 0 = none.  MRGFE_KERNEL_TIMING overrides.
static int timing_level() { static const int v = [] { const char* e = std::getenv("MRGFE_KERNEL_TIMING"); return e ? std::atoi(e) : 2; }(); return v; }

constexpr int kHostParallelMinPairs = 48;  // below this the controller steps of a round run on the calling thread

static int pipeline_min_pairs() { const char* e = std::getenv("MRGFE_PIPELINE_MIN_PAIRS"); return e ? std::atoi(e) : 1 << 30; }  // measured on MI355X: alternating half-batches lose more to smaller launches than they hide

// Tiles of 256 points per workgroup for a launch over `pts` source points: enough workgroups to fill the CUs several
// times over, few enough that the 384-byte block reduction amortises.  Chosen per launch: the late rounds of a batch
// have a few stragglers left, and those want one tile per workgroup to spread over the whole chip.
int NdtEngine::tiles_per_workgroup(uint64_t pts) const
{
    if (forced_ppt_ > 0) return forced_ppt_;
    static const int per_cu = [] { const char* e = std::getenv("MRGFE_WG_PER_CU"); return e ? std::max(1, std::atoi(e)) : 4; }();
    static const int max_ppt = [] { const char* e = std::getenv("MRGFE_MAX_PPT"); return e ? std::max(1, std::atoi(e)) : 8; }();
    const uint64_t target_blocks = uint64_t(ctx_->cu_count) * per_cu;
    return static_cast<int>(std::max<uint64_t>(1, std::min<uint64_t>(pts / (256 * target_blocks), max_ppt)));
}

int NdtEngine::launch_group(RoundGroup& g)
{
    NdtEvalDev* he = h_evals_.as<NdtEvalDev>();
    g.modes[0] = g.modes[1] = g.modes[2] = false;
    int active = 0;
    host_parallel_for(g.count, kHostParallelMinPairs, [&](int b, int e) {
        for (int i = g.first + b; i < g.first + e; ++i) {
            NdtController& c = pairs_[i].ctl;
            if (c.done()) he[i].active = 0;
            else          fill_eval(he[i], c.request(), c, prm_.search, true);
        }
    });
    g.any_spec = false;
    g.n_mode[0] = g.n_mode[1] = g.n_mode[2] = 0;
    uint64_t mode_pts[3] = {0, 0, 0};
    uint32_t mode_max_n[3] = {0, 0, 0};
    for (int i = g.first; i < g.first + g.count; ++i) {
        const NdtController& c = pairs_[i].ctl;
        if (c.done()) continue;
        const int  m = c.request().mode;
        const bool spec = m == 0 && c.request().spec_hessian;
        he[g.first + g.n_mode[m]++].order[m] = static_cast<uint32_t>(i - g.first);
        mode_pts[m] += pairs_[i].n;
        mode_max_n[m] = std::max(mode_max_n[m], pairs_[i].n);
        if (spec) {
            he[g.first + g.n_mode[2]++].order[2] = static_cast<uint32_t>(i - g.first);
            mode_pts[2] += pairs_[i].n;
            mode_max_n[2] = std::max(mode_max_n[2], pairs_[i].n);
            g.any_spec = true;
        }
        ++active;
    }
    for (int m = 0; m < 3; ++m) {
        g.modes[m] = g.n_mode[m] > 0;
        g.ppt[m] = tiles_per_workgroup(mode_pts[m]);
        g.nblk[m] = (mode_max_n[m] + 256u * g.ppt[m] - 1) / (256u * g.ppt[m]);
    }
    g.inflight = false;
    if (!active) return MRGFE_OK;
    hipStream_t st = ctx_->stream;
    NdtEvalDev*       d_ev = d_evals_.as<NdtEvalDev>() + g.first;
    const NdtPairDev* d_pr = d_pairs_.as<NdtPairDev>() + g.first;
    // the reduction kernel writes the 384-byte result records straight into pinned host memory (device-visible): no
    // device-to-host copy command, and its queue gap, in any of the ~20 rounds of a batch
    double*           d_res = h_results_.as<double>() + size_t(g.first) * kNdtPartialStride;
    MRGFE_HIP_CHECK(hipMemcpyAsync(d_ev, he + g.first, sizeof(NdtEvalDev) * g.count, hipMemcpyHostToDevice, st));
    // every kernel variant (mode) is bracketed by its own HIP events on the launch stream
    for (int m = 0; m < 3; ++m)
        if (g.modes[m]) {
            if (timing_level() > (m == 0 ? 0 : 1)) MRGFE_HIP_CHECK(hipEventRecord(g.ev[m][0], st));
            MRGFE_TRY(ndt_launch_derivatives(ctx_, m, prm_.search, g.nblk[m], g.n_mode[m], d_grids_.as<NdtGridDev>(), d_pr, d_ev, d_partials_.as<double>(), g.ppt[m], total_part_blocks_));
            if (timing_level() > (m == 0 ? 0 : 1)) MRGFE_HIP_CHECK(hipEventRecord(g.ev[m][1], st));
        }
    const uint32_t P = static_cast<uint32_t>(n_pairs());
    MRGFE_TRY(ndt_launch_reduce(ctx_, g.count, d_pr, d_ev, d_partials_.as<double>(), d_res, g.any_spec, total_part_blocks_, P, g.ppt));
    MRGFE_HIP_CHECK(hipEventRecord(g.done, st));
    g.inflight = true;
    return MRGFE_OK;
}

int NdtEngine::finish_group(RoundGroup& g)
{
    MRGFE_HIP_CHECK(hipEventSynchronize(g.done));
    g.inflight = false;
    for (int m = 0; m < 3; ++m)
        if (g.modes[m] && timing_level() > (m == 0 ? 0 : 1)) {
            float ms = 0;
            MRGFE_HIP_CHECK(hipEventElapsedTime(&ms, g.ev[m][0], g.ev[m][1]));
            mode_ms[m] += ms;
            mode_launches[m] += 1;
        }
    const double* hr = h_results_.as<double>();
    const int probes = prm_.search == MRGFE_DIRECT7 ? 7 : (prm_.search == MRGFE_DIRECT1 ? 1 : 27);
    for (int i = g.first; i < g.first + g.count; ++i) {
        const NdtController& c = pairs_[i].ctl;
        if (c.done()) continue;
        const double* r = hr + size_t(i) * kNdtPartialStride;
        // SURVEY.md §8(d) byte model: point (16) + probes (8 each) + 48 per valid neighbour voxel
        const double bytes = double(pairs_[i].n) * (16.0 + 8.0 * probes) + r[kNdtNbIndex] * 48.0;
        mode_alg_bytes[c.request().mode] += bytes;
        mode_points[c.request().mode] += double(pairs_[i].n);
        mode_neighbours[c.request().mode] += r[kNdtNbIndex];
        if (c.request().mode == 0 && c.request().spec_hessian) mode_alg_bytes[2] += bytes;  // the speculative f64 pass reads the same data
    }
    const size_t spec_base = size_t(n_pairs()) * kNdtPartialStride;
    // controller steps are independent per pair: spread them over the host worker threads for large batches
    host_parallel_for(g.count, kHostParallelMinPairs, [&](int b, int e) {
        for (int i = g.first + b; i < g.first + e; ++i) {
            NdtController& c = pairs_[i].ctl;
            if (c.done()) continue;
            const bool spec = c.request().mode == 0 && c.request().spec_hessian;
            c.on_result(hr + size_t(i) * kNdtPartialStride, spec ? hr + spec_base + size_t(i) * kNdtPartialStride : nullptr);
        }
    });
    return MRGFE_OK;
}

int NdtEngine::align_all()
{
    MRGFE_TRY(ctx_->bind());
    MRGFE_TRY(build_targets());
    if (pairs_dirty_) MRGFE_TRY(upload_pairs());
    for (int m = 0; m < 3; ++m) { mode_ms[m] = 0; mode_launches[m] = 0; mode_alg_bytes[m] = 0; mode_points[m] = 0; mode_neighbours[m] = 0; }
    for (auto& p : pairs_) {
        p.ctl.start(prm_, p.guess, p.n);
        if (targets_[p.target].status != MRGFE_OK && !p.ctl.done()) p.ctl.abort_no_target();
    }
    const int P = n_pairs();
    if (P == 0) return MRGFE_OK;
    // large batches: keep the host workers spinning between rounds (see host_parallel_for)
    struct HotGuard {
        bool on;
        explicit HotGuard(bool o) : on(o) { if (on) host_parallel_hot(true); }
        ~HotGuard() { if (on) host_parallel_hot(false); }
    } hot_guard(P >= kHostParallelMinPairs);
    // groups: two halves of (roughly) equal point count once the batch is large enough to keep the GPU busy with one
    const int n_groups = P >= pipeline_min_pairs() ? 2 : 1;
    if (groups_.size() < 2) {
        groups_.resize(2);
        for (auto& g : groups_) {
            MRGFE_HIP_CHECK(hipEventCreate(&g.done));
            for (int m = 0; m < 3; ++m) for (int k = 0; k < 2; ++k) MRGFE_HIP_CHECK(hipEventCreate(&g.ev[m][k]));
        }
    }
    int split = P;
    if (n_groups == 2) {
        uint64_t total = 0, run = 0;
        for (auto& p : pairs_) total += p.n;
        split = 0;
        while (split < P - 1 && (run + pairs_[split].n) * 2 <= total + pairs_[split].n) run += pairs_[split++].n;
        split = std::max(1, std::min(split, P - 1));
    }
    for (int k = 0; k < 2; ++k) {
        RoundGroup& g = groups_[k];
        g.first = k == 0 ? 0 : split;
        g.count = k == 0 ? split : P - split;
        g.inflight = false;
    }
    // every controller needs at most (max_iterations + 2) * (max line-search trials + 2) evaluations
    const int round_cap = (prm_.max_iterations + 3) * 13 + 8;
    static const bool trace = std::getenv("MRGFE_TRACE") != nullptr;  // per-round host timings on stderr
    for (int k = 0; k < n_groups; ++k) MRGFE_TRY(launch_group(groups_[k]));
    for (int round = 0; round < round_cap; ++round) {
        bool any = false;
        for (int k = 0; k < n_groups; ++k) {
            RoundGroup& g = groups_[k];
            if (!g.inflight) continue;
            any = true;
            const auto t0 = std::chrono::steady_clock::now();
            MRGFE_HIP_CHECK(hipEventSynchronize(g.done));
            const auto t1 = std::chrono::steady_clock::now();
            MRGFE_TRY(finish_group(g));
            const auto t2 = std::chrono::steady_clock::now();
            MRGFE_TRY(launch_group(g));  // no-op when every pair of the group is done
            const auto t3 = std::chrono::steady_clock::now();
            if (trace) std::fprintf(stderr, "[mrgfe round %d] wait %.0f us, finish %.0f us, launch %.0f us, busy pairs %d/%d/%d\n", round,
                                    std::chrono::duration<double, std::micro>(t1 - t0).count(), std::chrono::duration<double, std::micro>(t2 - t1).count(),
                                    std::chrono::duration<double, std::micro>(t3 - t2).count(), g.n_mode[0], g.n_mode[1], g.n_mode[2]);
        }
        if (!any) return MRGFE_OK;
    }
    MRGFE_HIP_CHECK(hipStreamSynchronize(ctx_->stream));
    set_error("NDT alignment did not terminate within %d rounds", round_cap);
    return MRGFE_ERR_STATE;
}

int NdtEngine::evaluate(int pair, const float T[16], const double p[6], int mode, double* score, double grad[6], double hess[36])
{
    if (pair < 0 || pair >= n_pairs()) { set_error("evaluate: pair index out of range"); return MRGFE_ERR_INVALID; }
    if (mode < 0 || mode > 2) { set_error("evaluate: mode must be 0, 1 or 2"); return MRGFE_ERR_INVALID; }
    MRGFE_TRY(ctx_->bind());
    MRGFE_TRY(build_targets());
    if (targets_[pairs_[pair].target].status != MRGFE_OK) { set_error("evaluate: target has no grid"); return MRGFE_ERR_STATE; }
    if (pairs_dirty_) MRGFE_TRY(upload_pairs());
    const int P = n_pairs();
    NdtController tmp;
    tmp.start(prm_, T, pairs_[pair].n);  // gauss constants
    NdtRequest r;
    r.mode = mode;
    std::memcpy(r.T, T, sizeof(r.T));
    std::memcpy(r.p, p, sizeof(r.p));
    NdtController::angle_tables(p, r.j_ang, r.h_ang);
    NdtEvalDev* he = h_evals_.as<NdtEvalDev>();
    for (int i = 0; i < P; ++i) he[i].active = 0;
    fill_eval(he[pair], r, tmp, prm_.search, true);
    he[0].order[mode] = static_cast<uint32_t>(pair);
    hipStream_t st = ctx_->stream;
    MRGFE_HIP_CHECK(hipMemcpyAsync(d_evals_.p, he, sizeof(NdtEvalDev) * P, hipMemcpyHostToDevice, st));
    const int ppt1[3] = {1, 1, 1};
    MRGFE_TRY(ndt_launch_derivatives(ctx_, mode, prm_.search, h_pairs_[pair].nblk, 1, d_grids_.as<NdtGridDev>(), d_pairs_.as<NdtPairDev>(), d_evals_.as<NdtEvalDev>(),
                                     d_partials_.as<double>(), 1, total_part_blocks_));
    MRGFE_TRY(ndt_launch_reduce(ctx_, P, d_pairs_.as<NdtPairDev>(), d_evals_.as<NdtEvalDev>(), d_partials_.as<double>(), h_results_.as<double>(), false, 0, 0, ppt1));
    MRGFE_HIP_CHECK(hipStreamSynchronize(st));
    const double* res = h_results_.as<double>() + size_t(pair) * kNdtPartialStride;
    *score = res[0];
    for (int k = 0; k < 6; ++k) grad[k] = res[1 + k];
    for (int k = 0; k < 36; ++k) hess[k] = res[7 + k];
    return MRGFE_OK;
}

int NdtEngine::aligned_cloud(int pair, float* out)
{
    if (pair < 0 || pair >= n_pairs()) { set_error("aligned_cloud: pair index out of range"); return MRGFE_ERR_INVALID; }
    const NdtPairInfo& p = pairs_[pair];
    if (p.n == 0) return MRGFE_OK;
    MRGFE_TRY(ctx_->bind());
    MRGFE_TRY(d_T12_.ensure(64));
    MRGFE_TRY(d_aligned_.ensure(size_t(p.n) * 16));
    MRGFE_HIP_CHECK(hipMemcpyAsync(d_T12_.p, p.ctl.final_transformation(), 48, hipMemcpyHostToDevice, ctx_->stream));
    MRGFE_TRY(launch_transform_cloud(ctx_, p.d_src, d_aligned_.as<float4>(), p.n, d_T12_.as<float>()));
    MRGFE_HIP_CHECK(hipMemcpyAsync(out, d_aligned_.p, size_t(p.n) * 16, hipMemcpyDeviceToHost, ctx_->stream));
    MRGFE_HIP_CHECK(hipStreamSynchronize(ctx_->stream));
    return MRGFE_OK;
}

}  // namespace mrgfe
