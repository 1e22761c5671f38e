// csrc/ndt_engine.cpp — batched NDT_HIP engine (see ndt_engine.h): target voxelisation pipeline and the lock-step
// alignment rounds.  Replaces, for a whole batch at once, what the reference does per object through
// setInputTarget / setInputSource / align (/root/reference/src/mrg_slam/loop_detector.cpp:104,126-145).
#include "ndt_engine.h"

#include <algorithm>
#include <atomic>
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

static bool one_wait_build_allowed();

NdtEngine::~NdtEngine()
{
    if (ctx_) (void)hipSetDevice(ctx_->device);
    cloud_arena_.release();
    grid_arena_.release();
    for (auto& e : ev_pool_) if (e) (void)hipEventDestroy(e);
    d_grids_.release(); d_pairs_.release(); d_evals_.release(); d_partials_.release(); d_T12_.release(); d_aligned_.release(); d_states_.release(); d_ticket_.release();
    d_ref_rec_.release(); d_ref_cnt_.release(); d_ref_jobs_.release();
    h_evals_.release(); h_results_.release(); h_states_.release(); h_info_.release();
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
int NdtEngine::build_targets(bool wait)
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
    MRGFE_TRY(hdesc.ensure(desc_bytes + sizeof(BBox) * P + sizeof(uint32_t) * P + 16 + sizeof(DdTargetOut)));
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

    DevBuf& dbb = ctx_->scratch[1];
    MRGFE_TRY(dbb.ensure(sizeof(BBox) * (tab.total_blks + P)));
    BBox* d_bb_part = dbb.as<BBox>();
    BBox* d_bb_out = d_bb_part + tab.total_blks;
    DevBuf &dk = ctx_->scratch[2], &dv = ctx_->scratch[3], &dkt = ctx_->scratch[4], &dvt = ctx_->scratch[5], &dh = ctx_->scratch[6], &dblk = ctx_->scratch[8];
    uint32_t *sk = nullptr, *sv = nullptr;
    uint32_t* d_tot = nullptr;
    int       key_bits = 1;
    // ONE target whose key width the previous build of this engine suggests (consecutive keyframes of one sensor): the voxel parameters are made on
    // the device and the build waits ONCE, behind the run-head count, for the bounding box, the parameters and the count together — the host
    // recomputes the parameters from the box as always and falls back to the two-wait path below if anything is off (no finite point, PCL's index
    // overflow, more cells than the guessed key width holds, parameters that differ).
    bool one_wait = P == 1 && key_bits_hint_ > 0 && one_wait_build_allowed();
    if (one_wait) {
        // (the record lives in pinned host memory: the kernels write it over PCIe and the host polls its last word — no copy command, no stream wait)
        DdTargetOut& h_out = *reinterpret_cast<DdTargetOut*>(hd + ((desc_bytes + sizeof(BBox) * P + sizeof(uint32_t) * P + 15) & ~size_t(15)));
        DdTargetOut* d_out = &h_out;
        constexpr uint32_t kNotYet = 0xFFFFFFFFu;  // (a run count is at most the point count, < 2^31)
        h_out.n_runs = kNotYet;
        MRGFE_TRY(bounding_box_partials(ctx_, d_cp, d_sl, tab, d_bb_part));
        MRGFE_TRY(ndt_launch_dd_voxel_params(ctx_, d_bb_part, tab.total_blks, prm_.resolution, const_cast<VoxelParams*>(d_vp), const_cast<uint32_t*>(d_nv), d_out));
        key_bits = key_bits_hint_;
        const size_t ne = std::max<size_t>(tab.total_elems, 4);
        MRGFE_TRY(dk.ensure(ne * 4)); MRGFE_TRY(dv.ensure(ne * 4)); MRGFE_TRY(dkt.ensure(ne * 4)); MRGFE_TRY(dvt.ensure(ne * 4));
        MRGFE_TRY(dh.ensure(sizeof(uint32_t) * 256 * (tab.total_blks + P)));
        MRGFE_TRY(dblk.ensure(sizeof(uint32_t) * (tab.total_blks + P + 4)));
        MRGFE_TRY(ndt_launch_cellkeys(ctx_, d_cp, d_sl, tab, d_vp, dk.as<uint32_t>(), dh.as<uint32_t>()));
        MRGFE_TRY(radix_sort_pairs(ctx_, dk.as<uint32_t>(), dv.as<uint32_t>(), dkt.as<uint32_t>(), dvt.as<uint32_t>(), d_sl, tab, key_bits, dh.as<uint32_t>(), &sk, &sv, true, true));
        d_tot = &d_out->n_runs;
        MRGFE_TRY(exclusive_scan_run_heads(ctx_, sk, nullptr, d_sl, tab, d_nv, dblk.as<uint32_t>(), d_tot));
        {
            const volatile uint32_t* flag = &h_out.n_runs;
            MRGFE_TRY(poll_host_record(st, [&] { return __atomic_load_n(flag, __ATOMIC_ACQUIRE) != kNotYet; }, "NDT target build"));
        }
        NdtTargetInfo& T = targets_[todo[0]];
        VoxelParams    vp_host;
        std::memset(&vp_host, 0, sizeof(vp_host));
        const bool ok = h_out.bb.n_finite != 0 && voxel_params_from_bbox(h_out.bb, prm_.resolution, &vp_host, T.max_b, T.div_b) == MRGFE_OK &&
                        std::memcmp(&vp_host, &h_out.vp, sizeof(vp_host)) == 0 && h_out.n_valid == h_out.bb.n_finite && (uint64_t(1) << key_bits) > vp_host.n_cells;
        if (ok) {
            T.built = true;
            T.status = MRGFE_OK;
            for (int a = 0; a < 3; ++a) T.min_b[a] = vp_host.min_b[a];
            h_bb[0] = h_out.bb;
            h_vp[0] = vp_host;
            h_nv[0] = h_out.n_valid;
            h_tot[0] = h_out.n_runs;
            int kb = 1;
            while (kb < 32 && (uint64_t(1) << kb) <= vp_host.n_cells) ++kb;
            key_bits_hint_ = std::min(32, kb + 1);
        } else {
            one_wait = false;  // the ordinary path decides what the reference does with this cloud
        }
    }
    if (!one_wait) {
        // 1. bounding boxes
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
        key_bits = 1;
        while (key_bits < 32 && (uint64_t(1) << key_bits) <= max_cells) ++key_bits;  // key == n_cells marks non-finite points
        if (P == 1 && targets_[todo[0]].status == MRGFE_OK) key_bits_hint_ = std::min(32, key_bits + 1);  // (a bit of room for the next keyframe's extent)

        // 3. keys, stable sort, run heads, ordinals
        const size_t ne = std::max<size_t>(tab.total_elems, 4);
        MRGFE_TRY(dk.ensure(ne * 4)); MRGFE_TRY(dv.ensure(ne * 4)); MRGFE_TRY(dkt.ensure(ne * 4)); MRGFE_TRY(dvt.ensure(ne * 4));
        MRGFE_TRY(dh.ensure(sizeof(uint32_t) * 256 * (tab.total_blks + P)));
        MRGFE_TRY(dblk.ensure(sizeof(uint32_t) * (tab.total_blks + P + 4)));
        MRGFE_TRY(ndt_launch_cellkeys(ctx_, d_cp, d_sl, tab, d_vp, dk.as<uint32_t>(), dh.as<uint32_t>()));
        MRGFE_TRY(radix_sort_pairs(ctx_, dk.as<uint32_t>(), dv.as<uint32_t>(), dkt.as<uint32_t>(), dvt.as<uint32_t>(), d_sl, tab, key_bits, dh.as<uint32_t>(), &sk, &sv, true, true));
        // run heads of the sorted keys: their number per target and the tiles' prefixes (a head's ordinal = its voxel's leaf index, made up by the segments kernel)
        d_tot = dblk.as<uint32_t>() + tab.total_blks;
        MRGFE_TRY(exclusive_scan_run_heads(ctx_, sk, nullptr, d_sl, tab, d_nv, dblk.as<uint32_t>(), d_tot));
        MRGFE_HIP_CHECK(hipMemcpyAsync(h_tot, d_tot, sizeof(uint32_t) * P, hipMemcpyDeviceToHost, st));
        MRGFE_HIP_CHECK(hipStreamSynchronize(st));
    }

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
        ls.keep_rejected = prm_.search == MRGFE_KDTREE ? 1u : 0u;  // radiusSearch has no nr_points test (ndt_build.h)
        ls.pcl_eigen_rule = prm_.formulation == 1 ? 1u : 0u;       // PCL_NDT_HIP: pcl::VoxelGridCovariance, not ndt_omp's fork of it
        ls.pad_ = 0;
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
    // seg_start (V + 1 per target), then the per-target counts and lists of big voxels (ndt_big_leaves_kernel)
    const size_t seg_words = size_t(total_leaves) + P + 4;
    MRGFE_TRY(dseg.ensure(sizeof(uint32_t) * (seg_words + P + nl)));
    MRGFE_TRY(dsum.ensure(sizeof(double) * 16 * nl));
    MRGFE_HIP_CHECK(hipMemcpyAsync(dd + o_ls, hd + o_ls, sizeof(LeafSlice) * P, hipMemcpyHostToDevice, st));

    // 5. segments and leaves
    MRGFE_TRY(ndt_launch_segments(ctx_, sk, d_sl, tab, d_nv, dblk.as<uint32_t>(), d_ls, dseg.as<uint32_t>(), static_cast<int32_t*>(p_keys)));
    MRGFE_TRY(ndt_launch_leaves(ctx_, d_cp, sv, d_sl, tab, d_ls, d_vp, max_leaves, dseg.as<uint32_t>(), dseg.as<uint32_t>() + seg_words, dseg.as<uint32_t>() + seg_words + P,
                                static_cast<const int32_t*>(p_keys), dsum.as<double>(),
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
    // align_all goes on enqueueing behind the build on the same stream (its first host wait comes with the first round's plan); the pinned
    // descriptor block is next written by the next build, which is behind that wait
    if (wait) MRGFE_HIP_CHECK(hipStreamSynchronize(st));
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
    const size_t P1 = static_cast<size_t>(std::max(P, 1));
    MRGFE_TRY(d_pairs_.ensure(sizeof(NdtPairDev) * P1));
    // the requests and the round's plan share one device buffer and one pinned staging buffer: the host-stepped path sends both with ONE copy per round
    evals_bytes_ = (sizeof(NdtEvalDev) * P1 + 255) & ~size_t(255);
    const size_t plan_bytes = sizeof(uint32_t) * ndt_plan_words(static_cast<uint32_t>(P1));
    MRGFE_TRY(d_evals_.ensure(evals_bytes_ + plan_bytes));
    MRGFE_TRY(d_states_.ensure(sizeof(NdtCtlState) * P1));
    total_part_blocks_ = part;
    MRGFE_TRY(d_partials_.ensure(sizeof(double) * kNdtPartialStride * std::max<uint32_t>(part, 1)));  // one record per tile: enough for any tiles-per-item
    MRGFE_TRY(h_evals_.ensure(evals_bytes_ + plan_bytes));
    MRGFE_TRY(h_states_.ensure(sizeof(NdtCtlState) * P1));
    MRGFE_TRY(h_results_.ensure(sizeof(double) * (kNdtPartialStride * P1 + 1)));  // (+1: the tag a single registration polls for)
    // (pageable source: staged by the runtime before the call returns — no wait, the rounds are enqueued behind the build and this copy)
    if (P) MRGFE_HIP_CHECK(hipMemcpyAsync(d_pairs_.p, h_pairs_.data(), sizeof(NdtPairDev) * P, hipMemcpyHostToDevice, ctx_->stream));
    pairs_dirty_ = false;
    return MRGFE_OK;
}

// ---- rounds --------------------------------------------------------------------------------------------------------
// A round = plan -> one derivative launch per kernel variant -> reduce (+ controller step).  Two ways to drive it:
//   device control (default for batches): the optimiser state of every pair lives in HBM and ndt_reduce_kernel<true> steps it,
//       so the host only ENQUEUES rounds — a few ahead of the GPU — and watches the count of running pairs the plan kernel
//       publishes in pinned memory to know when to stop.  No host round trip inside a batch.
//   host control (single registrations, MRGFE_HOST_CONTROL=1): the reduced sums come back through pinned memory and the host
//       steps the same state machine (ndt_ctl.h), one synchronisation per round.
// HIP events around the derivative launches (what mrgfe_*_kernel_stats reports): 2 = every variant, 1 = only the dominant
// score+gradient+Hessian variant (default: each event pair costs ~4 us of queue gap per round), 0 = none.  MRGFE_KERNEL_TIMING.
static int timing_level() { static const int v = [] { const char* e = std::getenv("MRGFE_KERNEL_TIMING"); return e ? std::atoi(e) : 1; }(); return v; }

// MRGFE_FUSED=0: one derivative launch per kernel variant and round (the layout before the fused launch; kept to hold it against)
static std::atomic<int> g_fused{-1};  // -1: not read yet
static bool fused_launch()
{
    int v = g_fused.load(std::memory_order_relaxed);
    if (v < 0) { const char* e = std::getenv("MRGFE_FUSED"); v = e ? (std::atoi(e) != 0 ? 1 : 0) : 1; g_fused.store(v, std::memory_order_relaxed); }
    return v != 0;
}
// MRGFE_NDT_BUILD_TWO_WAITS=1: a single target's build always waits for its bounding box before it sorts (the path before round 4's second half)
static bool one_wait_build_allowed() { static const bool v = std::getenv("MRGFE_NDT_BUILD_TWO_WAITS") == nullptr; return v; }
int ndt_set_fused_launch(int mode)
{
    if (mode == 0 || mode == 1) g_fused.store(mode, std::memory_order_relaxed);
    return fused_launch() ? 1 : 0;
}

// MRGFE_NDT_REFERENCE_ORDER=1 / mrgfe_dbg_set_ndt_reference_order(1): NDT_OMP's sums in the reference's own order (ndt_ref_records_kernel + ndt_ref_chain_kernel,
// ndt_derivatives.hip) — per-point sums, then a point-order chain; computeHessian pair after pair.  Host-stepped (the host cuts a round's evaluations into
// chunks that fit the record workspace).  Several times slower than the default tree; bit-identical to the reference-order oracle.
static std::atomic<int> g_ref_order{-1};  // -1: not read yet
static bool reference_order()
{
    int v = g_ref_order.load(std::memory_order_relaxed);
    if (v < 0) { const char* e = std::getenv("MRGFE_NDT_REFERENCE_ORDER"); v = (e && std::atoi(e) != 0) ? 1 : 0; g_ref_order.store(v, std::memory_order_relaxed); }
    return v != 0;
}
int ndt_set_reference_order(int mode)
{
    if (mode == 0 || mode == 1) g_ref_order.store(mode, std::memory_order_relaxed);
    return reference_order() ? 1 : 0;
}

constexpr int kHostParallelMinPairs = 48;  // below this the controller steps of a round run on the calling thread

// -1: automatic (single registrations are stepped by the host, batches on the device); 0 / 1 force device / host control
static std::atomic<int> g_host_control{-2};  // -2: not read yet
static int host_control_mode()
{
    int v = g_host_control.load(std::memory_order_relaxed);
    if (v == -2) { const char* e = std::getenv("MRGFE_HOST_CONTROL"); v = e ? std::atoi(e) : -1; g_host_control.store(v, std::memory_order_relaxed); }
    return v;
}
void ndt_set_host_control(int mode) { g_host_control.store(mode < -1 || mode > 1 ? -1 : mode, std::memory_order_relaxed); }
static int env_int(const char* name, int dflt) { const char* e = std::getenv(name); return e ? std::atoi(e) : dflt; }

// grid of a derivative launch: a few workgroups per resident slot; they walk the plan's items (item += gridDim.x)
uint32_t NdtEngine::derivative_grid(int mode) const
{
    static const int per_slot = std::max(1, env_int("MRGFE_GRID_PER_SLOT", 8));
    const bool narrow = prm_.search != MRGFE_KDTREE && prm_.search != MRGFE_DIRECT26 && prm_.formulation == 0;
    const int slots_per_cu = fused_launch() ? (narrow ? 3 : 2) : ((mode != 1 && narrow) ? 3 : 2);  // __launch_bounds__ of the kernels
    return static_cast<uint32_t>(ctx_->cu_count * slots_per_cu * per_slot);
}

int NdtEngine::ensure_events(size_t rounds)
{
    while (ev_pool_.size() < rounds * 6) {
        hipEvent_t e = nullptr;
        MRGFE_HIP_CHECK(hipEventCreate(&e));
        ev_pool_.push_back(e);
    }
    return MRGFE_OK;
}

// The plan of ndt_plan_kernel computed on the host (the host-stepped path knows every pending request): same compaction, same
// tiles-per-item rule.  Saves the plan launch of every round of a single registration.
void NdtEngine::host_plan(std::vector<uint32_t>& plan, uint32_t wg_target, uint32_t max_ppt) const
{
    const uint32_t P = static_cast<uint32_t>(n_pairs());
    plan.assign(ndt_plan_words(P), 0u);
    NdtPlanHead h{};
    uint32_t tiles[3] = {0, 0, 0};
    for (uint32_t i = 0; i < P; ++i) {
        const NdtController& c = pairs_[i].ctl;
        if (!c.done()) tiles[c.request_mode()] += (pairs_[i].n + 255u) / 256u;
    }
    for (int m = 0; m < 3; ++m) {
        const uint32_t ppt = std::max(1u, std::min(tiles[m] / wg_target, max_ppt));
        h.ppt[m] = forced_ppt_ > 0 ? static_cast<uint32_t>(forced_ppt_) : ppt;
    }
    for (uint32_t i = 0; i < P; ++i) {
        const NdtController& c = pairs_[i].ctl;
        if (c.done()) continue;
        const int m = c.request_mode();
        plan[ndt_plan_pair_off(P, m) + h.n_pairs[m]] = i;
        plan[ndt_plan_start_off(P, m) + h.n_pairs[m]] = h.n_items[m];
        h.n_pairs[m] += 1;
        h.n_items[m] += ((pairs_[i].n + 255u) / 256u + h.ppt[m] - 1) / h.ppt[m];
        h.n_active += 1;
    }
    for (int m = 0; m < 3; ++m) plan[ndt_plan_start_off(P, m) + h.n_pairs[m]] = h.n_items[m];
    std::memcpy(plan.data(), &h, sizeof(h));
}

// One round of the host-stepped path in the reference's summation order: every pending request (filled into h_evals_ by the caller) becomes a job — its
// records, then its chain — and the 384-byte result records land in pinned host memory like those of ndt_reduce_kernel<false>.  The record workspace is
// bounded (MRGFE_REF_WORKSPACE_MB, default 16 GiB of the 288): a round whose jobs need more runs in several launches, one after the other on the stream.
int NdtEngine::reference_round()
{
    hipStream_t st = ctx_->stream;
    const uint32_t P = static_cast<uint32_t>(n_pairs());
    const size_t ws_cap = static_cast<size_t>(std::max(1, env_int("MRGFE_REF_WORKSPACE_MB", 16384))) << 20;  // (read per round: tests shrink it to force several launches)
    const size_t nnb = prm_.search == MRGFE_DIRECT7 ? 7 : (prm_.search == MRGFE_DIRECT1 ? 1 : 27);
    MRGFE_HIP_CHECK(hipMemcpyAsync(d_evals_.p, h_evals_.p, sizeof(NdtEvalDev) * P, hipMemcpyHostToDevice, st));
    std::vector<NdtRefJob> jobs;
    size_t   rec_doubles = 0, cnt_bytes = 0;
    uint32_t max_tiles = 0;
    bool     any01 = false, any2 = false;
    auto flush = [&]() -> int {
        if (jobs.empty()) return MRGFE_OK;
        MRGFE_TRY(d_ref_rec_.ensure(rec_doubles * sizeof(double)));
        MRGFE_TRY(d_ref_cnt_.ensure(std::max<size_t>(cnt_bytes, 1)));
        MRGFE_TRY(d_ref_jobs_.ensure(sizeof(NdtRefJob) * jobs.size()));
        MRGFE_TRY(ctx_->stage_h2d(d_ref_jobs_.p, jobs.data(), sizeof(NdtRefJob) * jobs.size(), st));
        MRGFE_TRY(ndt_launch_ref_round(ctx_, prm_.search, d_grids_.as<NdtGridDev>(), d_pairs_.as<NdtPairDev>(), d_evals_.as<NdtEvalDev>(), d_ref_jobs_.as<NdtRefJob>(),
                                       static_cast<uint32_t>(jobs.size()), max_tiles, d_ref_rec_.as<double>(), d_ref_cnt_.as<uint8_t>(), h_results_.as<double>(), any01, any2));
        jobs.clear();
        rec_doubles = cnt_bytes = 0;
        max_tiles = 0;
        any01 = any2 = false;
        return MRGFE_OK;
    };
    for (uint32_t i = 0; i < P; ++i) {
        const NdtController& c = pairs_[i].ctl;
        if (c.done()) continue;
        const uint32_t mode = static_cast<uint32_t>(c.request_mode());
        const size_t   n = pairs_[i].n;
        const size_t   need = ndt_ref_record_doubles(static_cast<int>(mode), n, static_cast<int>(nnb));  // doubles
        if (!jobs.empty() && (rec_doubles + need) * sizeof(double) > ws_cap) MRGFE_TRY(flush());
        jobs.push_back(NdtRefJob{i, mode, rec_doubles, cnt_bytes});
        (mode == 2 ? any2 : any01) = true;
        rec_doubles += need;
        if (mode == 2) cnt_bytes += (n + 15) & ~size_t(15);
        max_tiles = std::max(max_tiles, static_cast<uint32_t>((n + 255) / 256));
    }
    return flush();
}

int NdtEngine::enqueue_round(uint32_t round, bool device_control, const bool want_mode[3], NdtRoundInfo* h_info, double result_tag)
{
    hipStream_t st = ctx_->stream;
    const uint32_t P = static_cast<uint32_t>(n_pairs());
    static const uint32_t per_cu = static_cast<uint32_t>(std::max(1, env_int("MRGFE_WG_PER_CU", 4)));
    static const uint32_t max_ppt = static_cast<uint32_t>(std::max(1, env_int("MRGFE_MAX_PPT", 8)));
    if (device_control) {
        MRGFE_TRY(ndt_launch_plan(ctx_, d_pairs_.as<NdtPairDev>(), d_evals_.as<NdtEvalDev>(), P, d_plan(), static_cast<uint32_t>(ctx_->cu_count) * per_cu, max_ppt,
                                  static_cast<uint32_t>(forced_ppt_), round, h_info));
    } else {
        // host-stepped: the requests (filled by the caller in h_evals_) and the plan go up in ONE copy out of the pinned buffer, no plan launch
        // (round 3 sent them as two copies: a copy command per round less on the single registration's critical path)
        const size_t words = ndt_plan_words(P);
        host_plan(plan_scratch_, static_cast<uint32_t>(ctx_->cu_count) * per_cu, max_ppt);
        std::memcpy(h_evals_.as<char>() + evals_bytes_, plan_scratch_.data(), words * 4);
        MRGFE_HIP_CHECK(hipMemcpyAsync(d_evals_.p, h_evals_.p, evals_bytes_ + words * 4, hipMemcpyHostToDevice, st));
    }
    // ONE host-stepped NDT_HIP registration: items and the sum of their records in one launch (ndt_derivatives_single_kernel); MRGFE_SINGLE_ROUND=0: two launches
    static const bool single_round = env_int("MRGFE_SINGLE_ROUND", 1) != 0;
    if (fused_launch() && !device_control && P == 1 && prm_.formulation == 0 && result_tag != 0.0 && single_round) {
        const NdtPlanHead* h = reinterpret_cast<const NdtPlanHead*>(plan_scratch_.data());
        const uint32_t grid = std::min(derivative_grid(0), h->n_items[0] + h->n_items[1] + h->n_items[2]);
        if (grid > 0) {
            if (!d_ticket_.p || ticket_dirty_) {  // first use, or a round that never reported (a failed launch): start from zero
                MRGFE_TRY(d_ticket_.ensure(256));
                MRGFE_HIP_CHECK(hipMemsetAsync(d_ticket_.p, 0, 256, st));
            }
            ticket_dirty_ = true;
            const bool timed = timing_level() > 0;
            if (timed) MRGFE_HIP_CHECK(hipEventRecord(ev_pool_[size_t(round) * 6], st));
            MRGFE_TRY(ndt_launch_single_round(ctx_, prm_.search, grid, d_grids_.as<NdtGridDev>(), d_pairs_.as<NdtPairDev>(), d_evals_.as<NdtEvalDev>(), d_plan(), d_partials_.as<double>(),
                                              d_ticket_.as<uint32_t>(), h_results_.as<double>(), result_tag));
            if (timed) MRGFE_HIP_CHECK(hipEventRecord(ev_pool_[size_t(round) * 6 + 1], st));
            return MRGFE_OK;
        }
    }
    if (fused_launch()) {
        // every variant's items in one launch (ndt_derivatives_all_kernel); its events sit in the slots of variant 0
        if (want_mode[0] || want_mode[1] || want_mode[2]) {
            const bool timed = timing_level() > 0;
            if (timed) MRGFE_HIP_CHECK(hipEventRecord(ev_pool_[size_t(round) * 6], st));
            uint32_t grid = derivative_grid(0);
            if (!device_control) {
                const NdtPlanHead* h = reinterpret_cast<const NdtPlanHead*>(plan_scratch_.data());
                grid = std::min(grid, h->n_items[0] + h->n_items[1] + h->n_items[2]);  // the host knows the item count
            }
            MRGFE_TRY(ndt_launch_derivatives_all(ctx_, prm_.search, grid, d_grids_.as<NdtGridDev>(), d_pairs_.as<NdtPairDev>(), d_evals_.as<NdtEvalDev>(), d_plan(), P,
                                                 d_partials_.as<double>(), prm_.formulation));
            if (timed) MRGFE_HIP_CHECK(hipEventRecord(ev_pool_[size_t(round) * 6 + 1], st));
        }
    } else
    for (int m = 0; m < 3; ++m) {
        if (!want_mode[m]) continue;
        const bool timed = timing_level() > (m == 0 ? 0 : 1);
        if (timed) MRGFE_HIP_CHECK(hipEventRecord(ev_pool_[size_t(round) * 6 + m * 2], st));
        uint32_t grid = derivative_grid(m);
        if (!device_control) grid = std::min(grid, reinterpret_cast<const NdtPlanHead*>(plan_scratch_.data())->n_items[m]);  // the host knows the item count
        MRGFE_TRY(ndt_launch_derivatives(ctx_, m, prm_.search, grid, d_grids_.as<NdtGridDev>(), d_pairs_.as<NdtPairDev>(), d_evals_.as<NdtEvalDev>(),
                                         d_plan(), P, d_partials_.as<double>(), prm_.formulation));
        if (timed) MRGFE_HIP_CHECK(hipEventRecord(ev_pool_[size_t(round) * 6 + m * 2 + 1], st));
    }
    // host control: the 384-byte result records go straight into pinned host memory (no device-to-host copy command); a host-stepped batch that polls for
    // its records (result_tag != 0, P > 1) hands the reduction the counter its last workgroup is told by
    uint32_t* d_reduce_ticket = nullptr;
    if (!device_control && P > 1 && result_tag != 0.0) {
        if (!d_ticket_.p || ticket_dirty_) {
            MRGFE_TRY(d_ticket_.ensure(256));
            MRGFE_HIP_CHECK(hipMemsetAsync(d_ticket_.p, 0, 256, st));
        }
        ticket_dirty_ = true;
        d_reduce_ticket = d_ticket_.as<uint32_t>() + 1;  // (word 0 is ndt_derivatives_single_kernel's)
    }
    MRGFE_TRY(ndt_launch_reduce(ctx_, P, d_pairs_.as<NdtPairDev>(), d_evals_.as<NdtEvalDev>(), d_partials_.as<double>(), d_plan(), h_results_.as<double>(),
                                device_control ? d_states_.as<NdtCtlState>() : nullptr, result_tag, d_reduce_ticket));
    return MRGFE_OK;
}

// the reduction's tag store is the last thing a host-stepped round of ONE registration does: seeing it = the record is complete
static int wait_result_tag(hipStream_t st, const volatile double* h_tag, double tag)
{
    uint64_t want;
    std::memcpy(&want, &tag, sizeof(want));
    const volatile uint64_t* p = reinterpret_cast<const volatile uint64_t*>(h_tag);
    return poll_host_record(st, [&] { return __atomic_load_n(p, __ATOMIC_ACQUIRE) == want; }, "NDT reduction");
}

void NdtEngine::account(const std::vector<NdtRoundInfo>& info, size_t rounds)
{
    // every launch counts, also those of a round in which the variant had no busy pair (its workgroups read the plan and exit,
    // ~4 us): the device-controlled path launches all three variants every round, and rocprofv3's per-kernel average — which
    // bench.py's HIP-event average is held against — is over all of them too
    // (the host-stepped path launches only the variants with work: `info` lists them and the other event slots are stale)
    if (fused_launch()) {
        // one launch per round; its time and count go under variant 0, the byte model below stays per variant
        for (size_t r = 0; r < rounds && timing_level() > 0; ++r) {
            if (!info.empty() && (r >= info.size() || info[r].n_active == 0)) continue;
            float ms = 0;
            if (hipEventElapsedTime(&ms, ev_pool_[r * 6], ev_pool_[r * 6 + 1]) != hipSuccess) { (void)hipGetLastError(); continue; }  // (never recorded, e.g. reference-order rounds: not a sticky error for the next call)
            mode_ms[0] += ms;
            mode_launches[0] += 1;
            if (ms > largest_ms && r < round_info_.size()) {
                largest_ms = ms;
                for (int m = 0; m < 3; ++m) largest_pairs[m] = round_info_[r].n_pairs[m];
            }
        }
    } else
    for (size_t r = 0; r < rounds; ++r)
        for (int m = 0; m < 3; ++m) {
            if (timing_level() <= (m == 0 ? 0 : 1)) continue;
            if (!info.empty() && (r >= info.size() || info[r].n_pairs[m] == 0)) continue;
            float ms = 0;
            if (hipEventElapsedTime(&ms, ev_pool_[r * 6 + m * 2], ev_pool_[r * 6 + m * 2 + 1]) != hipSuccess) { (void)hipGetLastError(); continue; }
            mode_ms[m] += ms;
            mode_launches[m] += 1;
        }
    const int probes = prm_.search == MRGFE_DIRECT7 ? 7 : (prm_.search == MRGFE_DIRECT1 ? 1 : 27);
    for (const auto& p : pairs_) {
        const NdtCtlState& s = p.ctl.state();
        for (int m = 0; m < 3; ++m) {
            // SURVEY.md §8(d) byte model: point (16) + probes (8 each) + 48 per valid neighbour voxel; the f64 formulation reads 96 per
            // neighbour (mean 3 x f64 + inverse covariance 9 x f64) and a 16-byte centroid per hit
            mode_alg_bytes[m] += s.acct_points[m] * (16.0 + 8.0 * probes) + s.acct_nb[m] * (prm_.formulation == 1 ? 112.0 : 48.0);
            mode_points[m] += s.acct_points[m];
            mode_neighbours[m] += s.acct_nb[m];
        }
    }
}

int NdtEngine::align_all(NdtSnapshotPort* port)
{
    struct PortGuard {  // whatever way this returns: no more snapshots
        NdtSnapshotPort* p;
        ~PortGuard() { if (p) p->finished.store(1, std::memory_order_release); }
    } port_guard{port};
    struct PhaseDump { ~PhaseDump() { static const bool on = std::getenv("MRGFE_PHASE") != nullptr; if (on) ndt_phase_dump(); } } phase_dump;  // diagnostic builds only
    MRGFE_TRY(ctx_->bind());
    MRGFE_TRY(build_targets(false));
    if (pairs_dirty_) MRGFE_TRY(upload_pairs());
    for (int m = 0; m < 3; ++m) { mode_ms[m] = 0; mode_launches[m] = 0; mode_alg_bytes[m] = 0; mode_points[m] = 0; mode_neighbours[m] = 0; largest_pairs[m] = 0; }
    largest_ms = 0;
    round_info_.clear();
    rounds_ = 0;
    const int P = n_pairs();
    if (P == 0) return MRGFE_OK;
    NdtEvalDev*  he = h_evals_.as<NdtEvalDev>();
    NdtCtlState* hs = h_states_.as<NdtCtlState>();
    // MRGFE_SPLIT_FIRST=1: evaluate the first trial of a line search without its Hessian and fetch that by a second pass only when
    // the trial is accepted (ndt_ctl.h).  Same results, 25 % less derivative work on the bench workload — and measured SLOWER
    // (13.6 vs 13.0 ms per 256-pair step): the fetches add rounds, and in lock-step rounds every extra launch pays its own tail.
    static const bool split_first = env_int("MRGFE_SPLIT_FIRST", 0) != 0;
    int running = 0;
    const bool ref_order = reference_order() && prm_.formulation == 0;
    for (int i = 0; i < P; ++i) {
        NdtPairInfo& p = pairs_[i];
        p.ctl.start(prm_, p.guess, p.n, split_first && !ref_order);
        // the reference's order is the reference's solve too: Eigen's two-sided JacobiSVD for every Newton step (the LU fast path and the one-sided SVD give
        // the same step to ~1e-16 — which a run to the iteration limit amplifies like it amplifies summation order)
        if (ref_order) p.ctl.force_reference_solve();
        if (targets_[p.target].status != MRGFE_OK && !p.ctl.done()) p.ctl.abort_no_target();
        he[i].active = 0;
        p.ctl.fill_eval(he[i]);
        running += p.ctl.done() ? 0 : 1;
    }
    if (!running) return MRGFE_OK;
    // every controller needs at most (max_iterations + 2) * (max line-search trials + 2) evaluations
    const size_t round_cap = size_t(prm_.max_iterations + 3) * 13 + 8;
    const int    hc = host_control_mode();
    // Who steps the optimisers (MRGFE_HOST_CONTROL: 1 the host, 0 the device; unset: by the size of the batch).  The device-stepped rounds (plan, derivatives,
    // reduce + controller, the host only enqueueing) win where the chip is full: from 128 pairs on, and from 16 on when the early fitness pass wants its
    // snapshots (`port`).  Below that the host-stepped round — one copy, derivatives, a reduction, a stream wait — is shorter than the three launches of a
    // device-stepped one and its plan (round 6, records identical either way: 2 / 8 / 32 / 64 pairs of 33k points 0.69 -> 0.56, 0.91 -> 0.74, 1.22 -> 1.06,
    // 1.57 -> 1.43 ms; of 129k points 0.64 -> 0.55, 0.98 -> 0.88, 1.95 -> 1.83, 2.96 -> 2.91; with getFitnessScore 4 / 8 / 13 pairs 3.15 -> 3.02, 4.18 -> 3.97, 4.45 -> 4.15 ms
    // but 16 / 32 / 64 pairs 4.4 / 6.1 / 8.5 against 4.7 / 6.3 / 8.7 host-stepped; 128 and 256 pairs: the device by 2 - 4 %).
    const bool   device_control = !ref_order && (hc == 0 || (hc < 0 && (P > 64 || (port != nullptr && P >= 16))));
    hipStream_t  st = ctx_->stream;
    if (device_control) MRGFE_HIP_CHECK(hipMemcpyAsync(d_evals_.p, he, sizeof(NdtEvalDev) * P, hipMemcpyHostToDevice, st));  // (host control: enqueue_round sends requests + plan together)
    static const bool trace = std::getenv("MRGFE_TRACE") != nullptr;  // per-round host timings on stderr
    std::vector<NdtRoundInfo> info;

    if (device_control) {
        // rounds enqueued beyond the last plan the host has seen.  A round's plan is published when its plan kernel runs, i.e. before its derivative launch:
        // with ONE round ahead the host enqueues round r + 1 while round r computes, and no launch is queued for a round that finds nothing left to do
        // (two ahead, the default until round 6: 256 config[1] pairs 8.98 -> 8.85 ms per step with two steps in flight, 10.13 -> 10.05 one at a time;
        // 32 pairs of 33k points 1.24 -> 1.21 ms; config[3] and its shard of 8 unchanged; same records)
        static const size_t lookahead = static_cast<size_t>(std::max(1, env_int("MRGFE_LOOKAHEAD", 1)));
        for (int i = 0; i < P; ++i) hs[i] = pairs_[i].ctl.state();
        MRGFE_HIP_CHECK(hipMemcpyAsync(d_states_.p, hs, sizeof(NdtCtlState) * P, hipMemcpyHostToDevice, st));
        MRGFE_TRY(h_info_.ensure(sizeof(NdtRoundInfo) * (round_cap + lookahead + 2)));
        volatile NdtRoundInfo* hi = h_info_.as<NdtRoundInfo>();
        const bool all_modes[3] = {true, true, true};
        size_t enq = 0, seen = 0;  // rounds enqueued / rounds whose plan the host has seen
        bool   finished = false;
        auto serve_port = [&]() -> int {  // a snapshot between two rounds, when the other thread has asked for one
            if (port && port->want.exchange(0, std::memory_order_acq_rel)) {
                const uint32_t tag = port->issued.load(std::memory_order_relaxed) + 1;
                MRGFE_TRY(ndt_launch_snapshot(ctx_, d_states_.as<NdtCtlState>(), static_cast<uint32_t>(P), tag, port->head(), port->recs()));
                port->issued.store(tag, std::memory_order_release);
            }
            return MRGFE_OK;
        };
        while (!finished) {
            // keep a few rounds queued ahead of the GPU; beyond that wait for the oldest unseen plan
            if (enq - seen >= lookahead || enq >= round_cap) {
                // (paused loads; the stream — whose query takes runtime locks other contexts' launches need — is asked every 256th time, and not at
                // all while polling is off: MRGFE_NO_POLL=1 waits for the stream instead)
                for (uint32_t spin = 0; __atomic_load_n(&hi[seen].tag, __ATOMIC_ACQUIRE) != static_cast<uint32_t>(seen + 1); ++spin) {
                    MRGFE_TRY(serve_port());
                    cpu_relax();
                    const bool ask = poll_disabled() || (spin & 0xff) == 0xff;
                    if (ask && (poll_disabled() || hipStreamQuery(st) != hipErrorNotReady) && __atomic_load_n(&hi[seen].tag, __ATOMIC_ACQUIRE) != static_cast<uint32_t>(seen + 1)) {
                        MRGFE_HIP_CHECK(hipStreamSynchronize(st));
                        if (hi[seen].tag != static_cast<uint32_t>(seen + 1)) { set_error("NDT round %zu never reported", seen); return MRGFE_ERR_HIP; }
                    }
                }
                if (port) port->n_active.store(hi[seen].n_active, std::memory_order_release);
                if (hi[seen].n_active == 0) finished = true;
                ++seen;
                if (seen >= round_cap && !finished) break;
                continue;
            }
            MRGFE_TRY(ensure_events(enq + 1));
            const_cast<NdtRoundInfo*>(hi)[enq].tag = 0;
            MRGFE_TRY(serve_port());
            MRGFE_TRY(enqueue_round(static_cast<uint32_t>(enq), true, all_modes, const_cast<NdtRoundInfo*>(hi)));
            ++enq;
        }
        MRGFE_HIP_CHECK(hipMemcpyAsync(hs, d_states_.p, sizeof(NdtCtlState) * P, hipMemcpyDeviceToHost, st));
        MRGFE_HIP_CHECK(hipStreamSynchronize(st));
        if (!finished) { set_error("NDT alignment did not terminate within %zu rounds", round_cap); return MRGFE_ERR_STATE; }
        for (int i = 0; i < P; ++i) pairs_[i].ctl.adopt(hs[i]);
        const size_t rounds = seen - 1;  // the last plan seen found nothing left to do
        info.resize(rounds);
        for (size_t r = 0; r < rounds; ++r) info[r] = const_cast<NdtRoundInfo*>(hi)[r];
        rounds_ = static_cast<int>(rounds);
        if (trace) std::fprintf(stderr, "[mrgfe] device control: %zu rounds, %zu enqueued\n", rounds, enq);
        round_info_ = info;  // (busy pairs per kind and round: names the largest launch)
        info.clear();  // device control: all three variants were launched in every enqueued round
        account(info, enq);
        return MRGFE_OK;
    }

    // ---- host control --------------------------------------------------------------------------------------------------------
    struct HotGuard {
        bool on;
        explicit HotGuard(bool o) : on(o) { if (on) host_parallel_hot(true); }
        ~HotGuard() { if (on) host_parallel_hot(false); }
    } hot_guard(P >= kHostParallelMinPairs);  // large batches: keep the host workers spinning between rounds (see host_parallel_for)
    const double* hr = h_results_.as<double>();
    // MRGFE_BATCH_POLL=0: host-stepped batches wait for the stream after every round, as before round 6
    static const bool batch_poll = env_int("MRGFE_BATCH_POLL", 1) != 0;
    const bool poll_records = P == 1 || (batch_poll && P <= 64);
    for (size_t round = 0; round < round_cap; ++round) {
        bool want[3] = {false, false, false};
        NdtRoundInfo ri{};
        for (int i = 0; i < P; ++i) {
            const NdtController& c = pairs_[i].ctl;
            if (c.done()) continue;
            want[c.request_mode()] = true;
            ri.n_pairs[c.request_mode()]++;
            ri.n_active++;
        }
        if (!ri.n_active) {
            rounds_ = static_cast<int>(round);
            round_info_ = info;
            account(info, info.size());
            return MRGFE_OK;
        }
        info.push_back(ri);
        const auto t0 = std::chrono::steady_clock::now();
        MRGFE_TRY(ensure_events(round + 1));
        if (ref_order) {
            MRGFE_TRY(reference_round());
            MRGFE_HIP_CHECK(hipStreamSynchronize(st));
        } else if (poll_records) {
            // a single registration, or a host-stepped batch: the reduction writes `tag` behind the records in pinned memory — its last workgroup, when
            // there are several — and the host polls for it (a stream wait costs ~10 us more per round than seeing the store)
            const double tag = static_cast<double>(++result_tag_);
            const_cast<double*>(hr)[size_t(kNdtPartialStride) * P] = 0.0;
            MRGFE_TRY(enqueue_round(static_cast<uint32_t>(round), false, want, nullptr, tag));
            MRGFE_TRY(wait_result_tag(st, hr + size_t(kNdtPartialStride) * P, tag));
            ticket_dirty_ = false;  // (the records are there: the last workgroup has cleared the counter)
        } else {
            MRGFE_TRY(enqueue_round(static_cast<uint32_t>(round), false, want, nullptr));
            MRGFE_HIP_CHECK(hipStreamSynchronize(st));
        }
        const auto t1 = std::chrono::steady_clock::now();
        // controller steps are independent per pair: spread them over the host worker threads for large batches
        host_parallel_for(P, kHostParallelMinPairs, [&](int b, int e) {
            for (int i = b; i < e; ++i) {
                NdtController& c = pairs_[i].ctl;
                if (c.done()) continue;
                c.on_result(hr + size_t(i) * kNdtPartialStride);
                he[i].active = 0;
                c.fill_eval(he[i]);
            }
        });
        if (trace) std::fprintf(stderr, "[mrgfe round %zu] gpu %.0f us, host %.0f us, busy pairs %u/%u/%u\n", round, std::chrono::duration<double, std::micro>(t1 - t0).count(),
                                std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t1).count(), ri.n_pairs[0], ri.n_pairs[1], ri.n_pairs[2]);
    }
    set_error("NDT alignment did not terminate within %zu rounds", round_cap);
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
    NdtCtlState s = tmp.state();
    std::memcpy(s.final_, T, sizeof(s.final_));
    s.phase = NDT_INIT;
    s.req_mode = mode;
    std::memcpy(s.req_p, p, sizeof(s.req_p));
    NdtEvalDev* he = h_evals_.as<NdtEvalDev>();
    for (int i = 0; i < P; ++i) he[i].active = 0;
    ctl::fill_eval(s, he[pair]);
    hipStream_t st = ctx_->stream;
    MRGFE_TRY(ensure_events(1));
    bool want[3] = {mode == 0, mode == 1, mode == 2};
    // the host-built plan lists the pairs whose controller has a request pending: lend this pair's controller the request
    std::vector<NdtCtlState> saved(P);
    for (int i = 0; i < P; ++i) { saved[i] = pairs_[i].ctl.state(); NdtCtlState idle = saved[i]; idle.phase = NDT_DONE; pairs_[i].ctl.adopt(idle); }
    pairs_[pair].ctl.adopt(s);
    const int keep = forced_ppt_;
    forced_ppt_ = 1;
    const int rc = (reference_order() && prm_.formulation == 0) ? reference_round() : enqueue_round(0, false, want, nullptr);
    forced_ppt_ = keep;
    for (int i = 0; i < P; ++i) pairs_[i].ctl.adopt(saved[i]);
    MRGFE_TRY(rc);
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
