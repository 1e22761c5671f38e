// csrc/ndt_build.hip — target voxelisation for NDT_HIP: the MI355X counterpart of
// pclomp::VoxelGridCovariance<PointXYZI>::applyFilter (reached through registration_->setInputTarget:
// /root/reference/apps/scan_matching_odometry_component.cpp:203,295,333; src/mrg_slam/loop_detector.cpp:104).
//
// Pipeline (all batched over targets):
//   bounding box -> voxel key per point -> stable radix sort (key, index) -> run heads -> scan = leaf ordinal
//   -> per-leaf f64 sums (one wavefront per leaf, coalesced index reads, shuffle reduction)
//   -> per-leaf mean / covariance / eigen clamp / inverse (one thread per leaf) -> dense or hashed lookup.
// The stable sort keeps points of a voxel in ascending index order and leaves in ascending key order, i.e. the
// order std::map iteration gives the reference.
#include "cellsort.h"
#include "dev_linalg.h"
#include "dev_utils.h"
#include "ndt_build.h"
#include "bbox_device.h"

namespace mrgfe {

// voxel key of every point (pcl VoxelGrid indexing: floor(p * inverse_leaf_size) - min_b, dotted with divb_mul)
__global__ __launch_bounds__(256) void ndt_cellkey_kernel(const float4* const* __restrict__ clouds, const Slice* __restrict__ slices, const VoxelParams* __restrict__ vp,
                                                           uint32_t* __restrict__ keys, uint32_t* __restrict__ hist)
{
    const Slice       s = slices[blockIdx.y];
    if (blockIdx.x >= s.nblk) return;
    const VoxelParams g = vp[blockIdx.y];
    const float4* __restrict__ pts = clouds[blockIdx.y];
    const uint32_t    base = blockIdx.x * kTile;
    // the radix sort's first per-tile digit histogram on the way (rs_hist_kernel's layout and method: the keys are in registers here, a pass
    // of its own would read them back), and no index array: the sort's first scatter makes the indices up
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    uint32_t key[kTile / 256];
#pragma unroll
    for (int k = 0; k < kTile / 256; ++k) {
        const uint32_t i = base + k * 256 + threadIdx.x;
        key[k] = 0u;
        if (i < s.n) {
            const float4 p = load_point(pts + i);
            key[k] = g.n_cells;  // non-finite points sort behind every voxel
            if (finite3(p.x, p.y, p.z)) {
                const int ijk0 = static_cast<int>(floorf(p.x * g.inv_leaf) - static_cast<float>(g.min_b[0]));
                const int ijk1 = static_cast<int>(floorf(p.y * g.inv_leaf) - static_cast<float>(g.min_b[1]));
                const int ijk2 = static_cast<int>(floorf(p.z * g.inv_leaf) - static_cast<float>(g.min_b[2]));
                key[k] = static_cast<uint32_t>(ijk0 * g.divb_mul[0] + ijk1 * g.divb_mul[1] + ijk2 * g.divb_mul[2]);
            }
            keys[s.off + i] = key[k];
        }
    }
#pragma unroll
    for (int k = 0; k < kTile / 256; ++k) wave_hist_add(h, key[k] & 255u, base + k * 256 + threadIdx.x < s.n);
    __syncthreads();
    hist[(size_t)(s.blk_off + blockIdx.x) * 256 + threadIdx.x] = h[threadIdx.x];
}

// run heads -> seg_start[leaf] (position in the sorted arrays) and seg_key[leaf]; the thread that sees the last valid
// element also writes the sentinel seg_start[V] = n_valid.  A head's leaf index is its ordinal among the run heads: the exclusive prefix of the
// tile (blk, from exclusive_scan_run_heads) + the scan of the head flags inside the tile, computed here — round 3 wrote an ordinal for EVERY
// point in one launch and read it back for the heads in the next (8 bytes per point and a launch more: 138 + 54 us per 256 targets).
__global__ __launch_bounds__(256) void ndt_segments_kernel(const uint32_t* __restrict__ sorted_keys, const Slice* __restrict__ slices, const uint32_t* __restrict__ n_valid,
                                                            const uint32_t* __restrict__ blk, const LeafSlice* __restrict__ leaf_slices, uint32_t* __restrict__ seg_start,
                                                            int32_t* __restrict__ seg_key)
{
    const Slice s = slices[blockIdx.y];
    if (blockIdx.x >= s.nblk) return;
    __shared__ uint32_t lds[8];
    const LeafSlice ls = leaf_slices[blockIdx.y];
    const uint32_t  nv = n_valid[blockIdx.y];
    const uint32_t* __restrict__ k0 = sorted_keys + s.off;
    const uint32_t first = blockIdx.x * kTile + threadIdx.x * 8;
    uint32_t kk[9];  // the thread's eight keys and the one before them
    kk[0] = (first > 0 && first - 1 < s.n) ? k0[first - 1] : 0u;
    load8_u32(k0, first, s.n, 0u, kk + 1);
    uint32_t head[8], tsum = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) { head[k] = (first + k < nv && (first + k == 0 || kk[k] != kk[k + 1])) ? 1u : 0u; tsum += head[k]; }
    uint32_t total;
    uint32_t ord = block_exclusive_scan<256>(tsum, lds, &total) + blk[s.blk_off + blockIdx.x];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint32_t i = first + k;
        if (head[k] && ord < ls.n_leaves) {  // (n_leaves = 0 for a target the host has disabled)
            seg_start[ls.seg_off + ord] = i;
            seg_key[ls.leaf_off + ord] = static_cast<int32_t>(kk[k + 1]);
        }
        ord += head[k];
        if (i < s.n && i + 1 == ls.n_valid) seg_start[ls.seg_off + ls.n_leaves] = ls.n_valid;
    }
}

// The reference accumulates each voxel's sums point by point in index order (pclomp::VoxelGridCovariance first pass); to
// reproduce those f64 sums BIT FOR BIT the additions of one accumulator stay sequential — a voxel is 13 dependent-add chains (x, y,
// z, the six products in f64; x, y, z, intensity in f32) — and the parallelism is moved elsewhere.  What the kernel waits for is
// then the LONGEST chain: a voxel next to the sensor holds thousands of points (the inner rings of a VLP-64 put ~90 points per
// metre on the ground; 5145 in the benchmark scans), and with one wavefront per voxel in launch order the 256-target build ran at 1.4
// resident wavefronts per SIMD (SQ_WAVE_CYCLES / GRBM_GUI_ACTIVE) behind the big voxels of the last targets.  So:
//   * voxels of more than kLeafBig points are listed per target (ndt_big_leaves_kernel) and taken by the FIRST workgroups of the
//     launch, one wavefront each, built for a short chain: the gathers of four 64-point steps in flight, every point's 13 terms
//     computed by its lane into LDS, then three instructions per point on the chain (8-byte LDS read, f64 add, f32 add);
//   * the rest go four at a time per wavefront, one per group of 13 lanes (group g takes the voxels g, g + 4, ... of a span of
//     kLeafSpan and moves on when one is finished, so unequal voxels even out): per round and group 64 lanes fetch the next 64
//     points (coalesced index read + 16-byte gather, the gathers of the NEXT round in flight while this round is summed) into
//     LDS as rows x, y, z, intensity; lanes past the voxel's end store zeros — the accumulators start at +0 and are never -0,
//     so adding a zero term is exact — and lane k of group g walks row(ia) * row(ib) (a row of ones standing for the factor of
//     the plain sums: x * 1.0 == x) or row(k - 9) of ITS group's rows in point order; the products are not on the chain;
//   * the target index is the fast grid dimension: the big voxels of ALL targets start at once.
// (Dropped earlier in round 2: owning the voxels that start in a 256-point window of the sorted stream and fetching 64 points
// per step across voxel boundaries, without the look-ahead: 1.04 ms against 0.87 ms for one wavefront per voxel.  This kernel:
// 0.32 ms per 256 targets of 130k points; kLeafBig 128 / 256 / 512 / 1024: 0.41 / 0.37 / 0.32 / 0.32 ms.)
// sums layout per leaf (16 doubles): [0..2] sum p, [3..8] sum xx,xy,xz,yy,yz,zz, [9] n, [10..13] float centroid sums.
constexpr int      kLeafGroups = 4;     // voxels in flight per wavefront (small voxels)
constexpr int      kLeafSpan = 16;      // voxels per wavefront (a build of a few targets takes 4: more, shorter wavefronts)
constexpr uint32_t kLeafBig = 512;      // points above which a voxel gets a wavefront of its own (256 for a build of a few targets)
constexpr uint32_t kLeafBigBlocks = 64; // workgroups per target that walk the list of big voxels

// one thread per voxel: the big ones into the target's list (the order of the list does not enter any result)
__global__ __launch_bounds__(256) void ndt_big_leaves_kernel(const LeafSlice* __restrict__ leaf_slices, const uint32_t* __restrict__ seg_start, uint32_t big_thr,
                                                              uint32_t* __restrict__ big_cnt, uint32_t* __restrict__ big_list)
{
    const LeafSlice ls = leaf_slices[blockIdx.y];
    const uint32_t  leaf = blockIdx.x * 256u + threadIdx.x;
    if (leaf >= ls.n_leaves) return;
    if (seg_start[ls.seg_off + leaf + 1] - seg_start[ls.seg_off + leaf] > big_thr) big_list[ls.leaf_off + atomicAdd(&big_cnt[blockIdx.y], 1u)] = leaf;
}

__device__ __forceinline__ void leaf_store(double* __restrict__ o, int k, double acc, float facc, uint32_t n)
{
    if (k < 9) o[k] = acc;
    else if (k < 13) o[k + 1] = static_cast<double>(facc);
    if (k == 9) o[9] = static_cast<double>(n);
}

__global__ __launch_bounds__(64, 6) void ndt_leaf_sums_kernel(const float4* const* __restrict__ clouds, const uint32_t* __restrict__ sorted_vals, const Slice* __restrict__ slices,
                                                               const LeafSlice* __restrict__ leaf_slices, const uint32_t* __restrict__ seg_start,
                                                               const uint32_t* __restrict__ big_cnt, const uint32_t* __restrict__ big_list, uint32_t big_thr,
                                                               uint32_t span_max, uint32_t big_blocks, double* __restrict__ sums)
{
    // small voxels: rows x, y, z, intensity, ones per group (+1: the rows lie in different banks); big voxels: 13 rows of 8-byte slots
    __shared__ double s_mem[13 * (kWave + 1)];
    static_assert(sizeof(double) * 13 * (kWave + 1) >= sizeof(float) * kLeafGroups * 5 * (kWave + 1), "LDS union");
    const LeafSlice ls = leaf_slices[blockIdx.x];
    const Slice     s = slices[blockIdx.x];
    const float4* __restrict__ pts = clouds[blockIdx.x];
    const uint32_t* __restrict__ order = sorted_vals + s.off;
    const int lane = lane_id();

    if (blockIdx.y < big_blocks) {
        // ---- big voxels: one at a time --------------------------------------------------------------------------------------
        double (*s_t)[kWave + 1] = reinterpret_cast<double (*)[kWave + 1]>(s_mem);
        const uint32_t n_big = big_cnt[blockIdx.x];
        const int      k = lane < 13 ? lane : 12;
        const double* __restrict__ row = s_t[k];
        for (uint32_t bi = blockIdx.y; bi < n_big; bi += big_blocks) {
            const uint32_t leaf = big_list[ls.leaf_off + bi];
            const uint32_t b = seg_start[ls.seg_off + leaf], e = seg_start[ls.seg_off + leaf + 1];
            double acc = 0.0;
            float  facc = 0.0f;
            constexpr int kAhead = 4;
            float4 pre[kAhead];
            auto fetch4 = [&](uint32_t base4) {  // unconditional loads of positions that exist: all index reads, one wait, all gathers
                uint32_t id[kAhead];
#pragma unroll
                for (int d = 0; d < kAhead; ++d) id[d] = order[min(base4 + d * kWave + lane, e - 1)];
#pragma unroll
                for (int d = 0; d < kAhead; ++d) pre[d] = load_point(pts + id[d]);
            };
            fetch4(b);
            for (uint32_t base4 = b; base4 < e; base4 += kAhead * kWave) {
                float4 cur4[kAhead];
#pragma unroll
                for (int d = 0; d < kAhead; ++d) cur4[d] = pre[d];
                if (base4 + kAhead * kWave < e) fetch4(base4 + kAhead * kWave);  // wave-uniform
#pragma unroll
                for (int d = 0; d < kAhead; ++d) {
                    const uint32_t base = base4 + d * kWave;
                    if (base >= e) break;  // wave-uniform
                    {
                        const bool   in = base + lane < e;
                        const float4 p = cur4[d];
                        const double x = p.x, y = p.y, z = p.z;
                        // a lane past the end stores zeros (exact to add, see above)
                        s_t[0][lane] = in ? x : 0.0; s_t[1][lane] = in ? y : 0.0; s_t[2][lane] = in ? z : 0.0;
                        s_t[3][lane] = in ? x * x : 0.0; s_t[4][lane] = in ? x * y : 0.0; s_t[5][lane] = in ? x * z : 0.0;
                        s_t[6][lane] = in ? y * y : 0.0; s_t[7][lane] = in ? y * z : 0.0; s_t[8][lane] = in ? z * z : 0.0;
                        // the f32 sums travel in the low half of an 8-byte slot: one read serves both kinds of chain
                        reinterpret_cast<float*>(&s_t[9][lane])[0] = in ? p.x : 0.0f; reinterpret_cast<float*>(&s_t[10][lane])[0] = in ? p.y : 0.0f;
                        reinterpret_cast<float*>(&s_t[11][lane])[0] = in ? p.z : 0.0f; reinterpret_cast<float*>(&s_t[12][lane])[0] = in ? p.w : 0.0f;
                    }
                    __builtin_amdgcn_wave_barrier();
                    __threadfence_block();
                    const uint32_t cnt = min(static_cast<uint32_t>(kWave), e - base);
                    for (uint32_t h = 0; h < (cnt + 15) / 16; ++h) {
                        double v[16];
#pragma unroll
                        for (int j = 0; j < 16; ++j) v[j] = row[h * 16 + j];
#pragma unroll
                        for (int j = 0; j < 16; ++j) {
                            acc += v[j];
                            facc += __builtin_bit_cast(float, static_cast<uint32_t>(__builtin_bit_cast(unsigned long long, v[j])));
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                    __threadfence_block();
                }
            }
            if (lane < 13) leaf_store(sums + (size_t)(ls.leaf_off + leaf) * 16, lane, acc, facc, e - b);
        }
        return;
    }

    // ---- small voxels: four at a time ---------------------------------------------------------------------------------------
    float (*s_c)[5][kWave + 1] = reinterpret_cast<float (*)[5][kWave + 1]>(s_mem);
  for (uint32_t first = (blockIdx.y - big_blocks) * span_max; first < ls.n_leaves; first += (gridDim.y - big_blocks) * span_max) {  // (one trip unless the grid was capped)
    const uint32_t span = min(span_max, ls.n_leaves - first);
    // extents of the span's voxels: lane l holds the start of voxel first + l (l <= span)
    const uint32_t seg_v = seg_start[ls.seg_off + first + min(static_cast<uint32_t>(lane), span)];
    auto seg_at = [&](uint32_t l) { return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(seg_v), static_cast<int>(l))); };
    const int g = min(lane / 13, kLeafGroups - 1), k = lane < 52 ? lane % 13 : 13;  // k == 13: spare lanes, never stored
    //   k:  0 1 2 | 3  4  5  6  7  8         (x y z | xx xy xz yy yz zz)
    const int ia = k < 3 ? k : (k < 6 ? 0 : (k < 8 ? 1 : 2));
    const int ib = k < 3 ? 4 : (k < 6 ? k - 3 : (k < 8 ? k - 5 : 2));
    const float* __restrict__ ra = s_c[g][ia];
    const float* __restrict__ rb = s_c[g][ib];
    const float* __restrict__ rf = s_c[g][(k - 9) & 3];
#pragma unroll
    for (int q = 0; q < kLeafGroups; ++q) s_c[q][4][lane] = 1.0f;
    // per group (wave-uniform): the voxel of the span it is on, the next position in it, its end
    uint32_t cur[kLeafGroups], pos[kLeafGroups], end[kLeafGroups];
    auto enter = [&](int q) {  // group q on to its next small voxel
        pos[q] = end[q] = 0;
        while (cur[q] < span) {
            pos[q] = seg_at(cur[q]);
            end[q] = seg_at(cur[q] + 1);
            if (end[q] - pos[q] <= big_thr) break;
            cur[q] += kLeafGroups;  // a big one: not ours
        }
    };
#pragma unroll
    for (int q = 0; q < kLeafGroups; ++q) { cur[q] = q; enter(q); }
    double acc = 0.0;
    float  facc = 0.0f;
    float4   pre[kLeafGroups];
    uint32_t cnt[kLeafGroups];
    auto fetch = [&]() {  // the next 64 points of every group's voxel
        // unconditional loads from positions that exist (a lane past the end re-reads a point and drops it): no branch between
        // them, so the four index reads go out together and the four gathers after ONE wait
        uint32_t id[kLeafGroups];
#pragma unroll
        for (int q = 0; q < kLeafGroups; ++q) {
            cnt[q] = cur[q] < span ? min(static_cast<uint32_t>(kWave), end[q] - pos[q]) : 0u;
            const uint32_t at = cur[q] < span ? pos[q] : seg_at(0);
            id[q] = order[at + (static_cast<uint32_t>(lane) < cnt[q] ? lane : 0)];
        }
#pragma unroll
        for (int q = 0; q < kLeafGroups; ++q) pre[q] = load_point(pts + id[q]);
#pragma unroll
        for (int q = 0; q < kLeafGroups; ++q)
            if (static_cast<uint32_t>(lane) >= cnt[q]) pre[q] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    };
    fetch();
    for (;;) {
        uint32_t now_leaf[kLeafGroups], now_n[kLeafGroups];
        bool     now_done[kLeafGroups];
        uint32_t maxcnt = 0;
#pragma unroll
        for (int q = 0; q < kLeafGroups; ++q) {
            s_c[q][0][lane] = pre[q].x; s_c[q][1][lane] = pre[q].y; s_c[q][2][lane] = pre[q].z; s_c[q][3][lane] = pre[q].w;
            now_leaf[q] = cur[q];
            maxcnt = max(maxcnt, cnt[q]);
            pos[q] += cnt[q];
            now_done[q] = cur[q] < span && pos[q] == end[q];
            now_n[q] = 0;
            if (now_done[q]) {  // on to the group's next voxel
                now_n[q] = end[q] - seg_at(cur[q]);
                cur[q] += kLeafGroups;
                enter(q);
            }
        }
        if (maxcnt == 0) break;  // wave-uniform: every group is through its voxels
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
        fetch();  // the gathers of the next round fly while this one is summed
        // the staged values into registers eight at a time, then their additions back to back (the additions of an accumulator
        // cannot be reordered; their operands can be fetched and multiplied ahead).  Rows are zero past a voxel's end, so the last
        // eight may run over it.
        for (uint32_t h = 0; h < (maxcnt + 7) / 8; ++h) {
            double t[8];
            float  f[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { t[j] = static_cast<double>(ra[h * 8 + j]) * static_cast<double>(rb[h * 8 + j]); f[j] = rf[h * 8 + j]; }
#pragma unroll
            for (int j = 0; j < 8; ++j) { acc += t[j]; facc += f[j]; }
        }
        // voxels that ended in this round: out with their sums, and the group's accumulators start over
#pragma unroll
        for (int q = 0; q < kLeafGroups; ++q) {
            if (!now_done[q]) continue;  // wave-uniform
            if (g == q && k < 13) {
                leaf_store(sums + (size_t)(ls.leaf_off + first + now_leaf[q]) * 16, k, acc, facc, now_n[q]);
                acc = 0.0;
                facc = 0.0f;
            }
        }
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
    }
  }
}

// one thread per leaf: pclomp::VoxelGridCovariance second pass + lookup insertion
__global__ __launch_bounds__(256) void ndt_leaf_finalize_kernel(const LeafSlice* __restrict__ leaf_slices, const VoxelParams* __restrict__ vp,
                                                                 const double* __restrict__ sums, const int32_t* __restrict__ seg_key,
                                                                 NdtLeafRec* __restrict__ leaves, double* __restrict__ icov64, float4* __restrict__ centroid,
                                                                 int32_t* __restrict__ nr_points, void* __restrict__ lookup_base)
{
    const LeafSlice ls = leaf_slices[blockIdx.y];
    const uint32_t  leaf = blockIdx.x * 256 + threadIdx.x;
    if (leaf >= ls.n_leaves) return;
    const VoxelParams g = vp[blockIdx.y];
    const size_t      gl = ls.leaf_off + leaf;
    const double*     a = sums + gl * 16;
    const int         n = static_cast<int>(a[9]);
    const double      dn = static_cast<double>(n);
    const double      pt_sum[3] = {a[0], a[1], a[2]};
    double            mean[3] = {a[0] / dn, a[1] / dn, a[2] / dn};
    const float       fn = static_cast<float>(n);
    centroid[gl] = make_float4(static_cast<float>(a[10]) / fn, static_cast<float>(a[11]) / fn, static_cast<float>(a[12]) / fn, static_cast<float>(a[13]) / fn);
    NdtLeafRec rec;
    rec.mean[0] = mean[0]; rec.mean[1] = mean[1]; rec.mean[2] = mean[2];
#pragma unroll
    for (int k = 0; k < 6; ++k) rec.icov[k] = 0.0f;
    double icov[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    int    npts = n;
    if (n >= kNdtMinPointsPerVoxel) {
        const double xx[9] = {a[3], a[4], a[5], a[4], a[6], a[7], a[5], a[7], a[8]};
        double cov[9];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) cov[r * 3 + c] = (xx[r * 3 + c] - 2 * (pt_sum[r] * mean[c])) / dn + mean[r] * mean[c];
        const double f = (dn - 1.0) / dn;
        for (int k = 0; k < 9; ++k) cov[k] *= f;
        double w[3], V[9];
        dl_sym_eig3(cov, w, V);
        const double neg_tol = ls.pcl_eigen_rule ? kPclVgcNegativeEigenTolerance : 0.0;
        if (w[0] < -neg_tol || w[1] < -neg_tol || w[2] <= 0) {
            npts = -1;
        } else {
            const double min_ev = kNdtMinCovarEigMult * w[2];
            if (w[0] < min_ev) {
                w[0] = min_ev;
                if (w[1] < min_ev) w[1] = min_ev;
                const double D[9] = {w[0], 0, 0, 0, w[1], 0, 0, 0, w[2]};
                double VD[9], Vinv[9];
                dl_mul3(V, D, VD);
                dl_inv3(V, Vinv);
                dl_mul3(VD, Vinv, cov);
            }
            dl_inv3(cov, icov);
            bool bad = false;
            for (int k = 0; k < 9; ++k) bad = bad || !isfinite(icov[k]);
            if (bad) npts = -1;
        }
    }
    for (int k = 0; k < 9; ++k) icov64[gl * 9 + k] = icov[k];
    rec.icov[0] = static_cast<float>(icov[0]); rec.icov[1] = static_cast<float>(icov[1]); rec.icov[2] = static_cast<float>(icov[2]);
    rec.icov[3] = static_cast<float>(icov[4]); rec.icov[4] = static_cast<float>(icov[5]); rec.icov[5] = static_cast<float>(icov[8]);
    leaves[gl] = rec;
    nr_points[gl] = npts;
    if (npts >= kNdtMinPointsPerVoxel || (ls.keep_rejected && n >= kNdtMinPointsPerVoxel)) {
        const uint32_t key = static_cast<uint32_t>(seg_key[gl]);
        if (ls.dense) {
            reinterpret_cast<int32_t*>(static_cast<char*>(lookup_base) + ls.lookup_byte_off)[key] = static_cast<int32_t>(leaf);
        } else {
            uint2*   slots = reinterpret_cast<uint2*>(static_cast<char*>(lookup_base) + ls.lookup_byte_off);
            uint32_t h = ndt_hash(key, ls.hash_shift);
            for (uint32_t probe = 0; probe <= ls.hash_mask; ++probe) {
                const uint32_t prev = atomicCAS(&slots[h].x, kHashEmpty, key);
                if (prev == kHashEmpty) { slots[h].y = leaf; break; }
                h = (h + 1) & ls.hash_mask;
            }
        }
    }
    (void)g;
}

__global__ __launch_bounds__(256) void ndt_dd_voxel_params_kernel(const BBox* __restrict__ partial, uint32_t n_partial, float leaf, VoxelParams* __restrict__ vp_out,
                                                                   uint32_t* __restrict__ nv_out, DdTargetOut* __restrict__ out)
{
#pragma clang fp contract(off)
    const BBox bb = block_merge_partials(partial, n_partial);
    if (threadIdx.x) return;
    VoxelParams vp;
    memset(&vp, 0, sizeof(vp));
    uint32_t nv = 0;
    if (bb.n_finite != 0) {
        // (voxel_params_from_bbox, ndt_engine.cpp: the same float operations in the same order)
        const float   inv_leaf = 1.0f / leaf;
        const int64_t dx = static_cast<int64_t>((bb.mx[0] - bb.mn[0]) * inv_leaf) + 1;
        const int64_t dy = static_cast<int64_t>((bb.mx[1] - bb.mn[1]) * inv_leaf) + 1;
        const int64_t dz = static_cast<int64_t>((bb.mx[2] - bb.mn[2]) * inv_leaf) + 1;
        bool over = dx * dy * dz > static_cast<int64_t>(INT32_MAX);
        if (!over) {
            int32_t div_b[3];
            for (int a = 0; a < 3; ++a) {
                vp.min_b[a] = static_cast<int32_t>(floorf(bb.mn[a] * inv_leaf));
                div_b[a] = static_cast<int32_t>(floorf(bb.mx[a] * inv_leaf)) - vp.min_b[a] + 1;
            }
            vp.divb_mul[0] = 1;
            vp.divb_mul[1] = div_b[0];
            vp.divb_mul[2] = div_b[0] * div_b[1];
            vp.inv_leaf = inv_leaf;
            const int64_t cells = static_cast<int64_t>(div_b[0]) * div_b[1] * div_b[2];
            over = cells > static_cast<int64_t>(INT32_MAX);
            vp.n_cells = static_cast<uint32_t>(cells);
        }
        if (over) memset(&vp, 0, sizeof(vp));
        else      nv = bb.n_finite;
    }
    *vp_out = vp;
    *nv_out = nv;
    out->bb = bb;
    out->vp = vp;
    out->n_valid = nv;
}

int ndt_launch_dd_voxel_params(mrgfe_ctx* ctx, const BBox* d_partial, uint32_t n_partial, float leaf, VoxelParams* d_vp, uint32_t* d_n_valid, DdTargetOut* d_out)
{
    hipLaunchKernelGGL(ndt_dd_voxel_params_kernel, dim3(1), dim3(256), 0, ctx->stream, d_partial, n_partial, leaf, d_vp, d_n_valid, d_out);
    MRGFE_HIP_CHECK(hipGetLastError());
    return MRGFE_OK;
}

int ndt_launch_cellkeys(mrgfe_ctx* ctx, const float4* const* d_clouds, const Slice* d_slices, const SliceTable& t, const VoxelParams* d_vp, uint32_t* d_keys, uint32_t* d_hist)
{
    if (t.nprob() == 0 || t.max_blks == 0) return MRGFE_OK;
    hipLaunchKernelGGL(ndt_cellkey_kernel, dim3(t.max_blks, t.nprob()), dim3(256), 0, ctx->stream, d_clouds, d_slices, d_vp, d_keys, d_hist);
    MRGFE_HIP_CHECK(hipGetLastError());
    return MRGFE_OK;
}

int ndt_launch_segments(mrgfe_ctx* ctx, const uint32_t* d_sorted_keys, const Slice* d_slices, const SliceTable& t, const uint32_t* d_n_valid, const uint32_t* d_blk,
                        const LeafSlice* d_leaf_slices, uint32_t* d_seg_start, int32_t* d_seg_key)
{
    if (t.nprob() == 0 || t.max_blks == 0) return MRGFE_OK;
    hipLaunchKernelGGL(ndt_segments_kernel, dim3(t.max_blks, t.nprob()), dim3(256), 0, ctx->stream, d_sorted_keys, d_slices, d_n_valid, d_blk, d_leaf_slices, d_seg_start, d_seg_key);
    MRGFE_HIP_CHECK(hipGetLastError());
    return MRGFE_OK;
}

int ndt_launch_leaves(mrgfe_ctx* ctx, const float4* const* d_clouds, const uint32_t* d_sorted_vals, const Slice* d_slices, const SliceTable& t, const LeafSlice* d_leaf_slices,
                      const VoxelParams* d_vp, uint32_t max_leaves, const uint32_t* d_seg_start, uint32_t* d_big_cnt, uint32_t* d_big_list, const int32_t* d_seg_key, double* d_sums,
                      NdtLeafRec* d_leaves, double* d_icov64, float4* d_centroid, int32_t* d_nr_points, void* d_lookup_base)
{
    if (t.nprob() == 0 || max_leaves == 0) return MRGFE_OK;
    MRGFE_HIP_CHECK(hipMemsetAsync(d_big_cnt, 0, sizeof(uint32_t) * t.nprob(), ctx->stream));
    // a build of a few targets does not fill the chip: more and shorter wavefronts, and more workgroups on the list of big voxels
    // (single registration: sums 72 -> 70 us; what is left is the 5145-point chain of the biggest voxel).  Measured and dropped: a big
    // voxel's nine f64 and four f32 sums on two wavefronts, two instead of three instructions per point on the chain — 64 us for one
    // target, but 335 -> 386 us for 256 (the points are gathered twice).
    const bool     few = t.nprob() <= 8;
    const uint32_t big_thr = few ? 256u : kLeafBig, span = few ? 4u : static_cast<uint32_t>(kLeafSpan), big_blocks = few ? 512u : kLeafBigBlocks;
    hipLaunchKernelGGL(ndt_big_leaves_kernel, dim3((max_leaves + 255) / 256, t.nprob()), dim3(256), 0, ctx->stream, d_leaf_slices, d_seg_start, big_thr, d_big_cnt, d_big_list);
    // x = target (fast: the big voxels of all targets are dispatched first), y = the big-voxel workgroups, then the spans
    hipLaunchKernelGGL(ndt_leaf_sums_kernel, dim3(t.nprob(), big_blocks + std::min<uint32_t>((max_leaves + span - 1) / span, 65535u - big_blocks)), dim3(64), 0, ctx->stream, d_clouds,
                       d_sorted_vals, d_slices, d_leaf_slices, d_seg_start, d_big_cnt, d_big_list, big_thr, span, big_blocks, d_sums);
    hipLaunchKernelGGL(ndt_leaf_finalize_kernel, dim3((max_leaves + 255) / 256, t.nprob()), dim3(256), 0, ctx->stream, d_leaf_slices, d_vp, d_sums, d_seg_key,
                       d_leaves, d_icov64, d_centroid, d_nr_points, d_lookup_base);
    MRGFE_HIP_CHECK(hipGetLastError());
    return MRGFE_OK;
}

}  // namespace mrgfe
