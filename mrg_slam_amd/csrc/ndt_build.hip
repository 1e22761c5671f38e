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

namespace mrgfe {

// voxel key of every point (pcl VoxelGrid indexing: floor(p * inverse_leaf_size) - min_b, dotted with divb_mul)
__global__ __launch_bounds__(256) void ndt_cellkey_kernel(const float4* const* __restrict__ clouds, const Slice* __restrict__ slices, const VoxelParams* __restrict__ vp,
                                                           uint32_t* __restrict__ keys, uint32_t* __restrict__ vals)
{
    const Slice       s = slices[blockIdx.y];
    if (blockIdx.x >= s.nblk) return;
    const VoxelParams g = vp[blockIdx.y];
    const float4* __restrict__ pts = clouds[blockIdx.y];
    const uint32_t    base = blockIdx.x * kTile;
#pragma unroll
    for (int k = 0; k < kTile / 256; ++k) {
        const uint32_t i = base + k * 256 + threadIdx.x;
        if (i < s.n) {
            const float4 p = pts[i];
            uint32_t key = g.n_cells;  // non-finite points sort behind every voxel
            if (finite3(p.x, p.y, p.z)) {
                const int ijk0 = static_cast<int>(floorf(p.x * g.inv_leaf) - static_cast<float>(g.min_b[0]));
                const int ijk1 = static_cast<int>(floorf(p.y * g.inv_leaf) - static_cast<float>(g.min_b[1]));
                const int ijk2 = static_cast<int>(floorf(p.z * g.inv_leaf) - static_cast<float>(g.min_b[2]));
                key = static_cast<uint32_t>(ijk0 * g.divb_mul[0] + ijk1 * g.divb_mul[1] + ijk2 * g.divb_mul[2]);
            }
            keys[s.off + i] = key;
            vals[s.off + i] = i;
        }
    }
}

// run heads -> seg_start[leaf] (position in the sorted arrays) and seg_key[leaf]; the thread that sees the last valid
// element also writes the sentinel seg_start[V] = n_valid.
__global__ __launch_bounds__(256) void ndt_segments_kernel(const uint32_t* __restrict__ sorted_keys, const uint32_t* __restrict__ flags,
                                                            const uint32_t* __restrict__ ordinal, const Slice* __restrict__ slices,
                                                            const LeafSlice* __restrict__ leaf_slices, uint32_t* __restrict__ seg_start, int32_t* __restrict__ seg_key)
{
    const Slice s = slices[blockIdx.y];
    if (blockIdx.x >= s.nblk) return;
    const LeafSlice ls = leaf_slices[blockIdx.y];
    const uint32_t  base = blockIdx.x * kTile;
#pragma unroll
    for (int k = 0; k < kTile / 256; ++k) {
        const uint32_t i = base + k * 256 + threadIdx.x;
        if (i < s.n) {
            if (flags[s.off + i]) {
                const uint32_t o = ordinal[s.off + i];
                seg_start[ls.seg_off + o] = i;
                seg_key[ls.leaf_off + o] = static_cast<int32_t>(sorted_keys[s.off + i]);
            }
            if (i + 1 == ls.n_valid) seg_start[ls.seg_off + ls.n_leaves] = ls.n_valid;
        }
    }
}

// (Measured and dropped in round 2: a wavefront that owns the leaves starting in a 256-point segment of the sorted stream,
// fetches 64 points per step whatever the leaf boundaries and spreads the (leaf, term) chains over all 64 lanes — every lane
// busy, bit-identical sums — took 1.04 ms per 256 targets of 130k points against 0.82 ms for the kernel below, software
// pipelined and with the leaf extents kept in LDS: the kernel is bound by the random 16-byte point gather through the sorted
// index, which a wavefront per leaf overlaps better — 5.6 M short independent wavefronts — than 130 k long ones.)
// one wavefront per leaf. The reference accumulates each voxel's sums point by point in index order
// (pclomp::VoxelGridCovariance first pass); to reproduce those f64 sums BIT FOR BIT the additions of one accumulator stay
// sequential and the parallelism is moved elsewhere:
//   step A  64 lanes fetch 64 points of the leaf (coalesced index read + 16-byte gather) and each lane computes the 13
//           terms of ITS point (x,y,z, the six products in f64, x,y,z,intensity in f32) into LDS, [term][point];
//   step B  lane k (k < 13) walks row k of the staged terms in point order and adds them to its accumulator: 13
//           independent dependent-add chains, fed by pipelined conflict-free LDS reads.
// sums layout per leaf (16 doubles): [0..2] sum p, [3..8] sum xx,xy,xz,yy,yz,zz, [9] n, [10..13] float centroid sums.
__global__ __launch_bounds__(256) void ndt_leaf_sums_kernel(const float4* const* __restrict__ clouds, const uint32_t* __restrict__ sorted_vals, const Slice* __restrict__ slices,
                                                             const LeafSlice* __restrict__ leaf_slices, const uint32_t* __restrict__ seg_start,
                                                             double* __restrict__ sums)
{
    __shared__ double s_term[4][9][kWave + 1];   // +1: the 13 row readers hit different banks
    __shared__ float  s_fterm[4][4][kWave + 1];
    const LeafSlice ls = leaf_slices[blockIdx.y];
    const uint32_t  leaf = blockIdx.x * 4 + wave_id();
    if (leaf >= ls.n_leaves) return;  // wave-uniform
    const Slice    s = slices[blockIdx.y];
    const float4* __restrict__ pts = clouds[blockIdx.y];
    const uint32_t b = seg_start[ls.seg_off + leaf], e = seg_start[ls.seg_off + leaf + 1];
    const int      lane = lane_id(), w = wave_id();
    const int      row = lane < 9 ? lane : 0, frow = (lane >= 9 && lane < 13) ? lane - 9 : 0;
    double acc = 0.0;
    float  facc = 0.0f;
    // A voxel next to the sensor holds over a thousand points (the inner rings of a VLP-64 put ~90 points per metre on the ground),
    // and its one wavefront walks them 64 at a time: with one dependent index -> point gather per step that leaf alone took 150 us,
    // which is what a single registration's setInputTarget waited for.  The gathers of four steps are issued together.
    constexpr int kAhead = 4;
    for (uint32_t base4 = b; base4 < e; base4 += kAhead * kWave) {
        float4 pre[kAhead];
#pragma unroll
        for (int d = 0; d < kAhead; ++d) {
            const uint32_t i = base4 + d * kWave + lane;
            if (i < e) pre[d] = pts[sorted_vals[s.off + i]];
        }
#pragma unroll
        for (int d = 0; d < kAhead; ++d) {
            const uint32_t base = base4 + d * kWave;
            if (base >= e) break;  // wave-uniform
            const uint32_t i = base + lane;
            if (i < e) {
                const float4 p = pre[d];
                const double x = p.x, y = p.y, z = p.z;
                s_term[w][0][lane] = x; s_term[w][1][lane] = y; s_term[w][2][lane] = z;
                s_term[w][3][lane] = x * x; s_term[w][4][lane] = x * y; s_term[w][5][lane] = x * z;
                s_term[w][6][lane] = y * y; s_term[w][7][lane] = y * z; s_term[w][8][lane] = z * z;
                s_fterm[w][0][lane] = p.x; s_fterm[w][1][lane] = p.y; s_fterm[w][2][lane] = p.z; s_fterm[w][3][lane] = p.w;
            }
            __builtin_amdgcn_wave_barrier();
            __threadfence_block();
            const uint32_t cnt = min(static_cast<uint32_t>(kWave), e - base);
            if (cnt == kWave) {
                // full step: all 64 staged terms of the row into registers first, then the 64 dependent additions back to back
                // (the additions of a row cannot be reordered; their operands can be fetched ahead)
                // (in two halves of 32: 64 values at once cost the kernel its occupancy and the 256-target build 5 %)
                if (lane < 9) {
                    const double* t = s_term[w][row];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        double v[32];
#pragma unroll
                        for (int j = 0; j < 32; ++j) v[j] = t[h * 32 + j];
#pragma unroll
                        for (int j = 0; j < 32; ++j) acc += v[j];
                    }
                } else if (lane < 13) {
                    const float* t = s_fterm[w][frow];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        float v[32];
#pragma unroll
                        for (int j = 0; j < 32; ++j) v[j] = t[h * 32 + j];
#pragma unroll
                        for (int j = 0; j < 32; ++j) facc += v[j];
                    }
                }
            } else if (lane < 9) {
                const double* t = s_term[w][row];
#pragma unroll 8
                for (uint32_t j = 0; j < cnt; ++j) acc += t[j];
            } else if (lane < 13) {
                const float* t = s_fterm[w][frow];
#pragma unroll 8
                for (uint32_t j = 0; j < cnt; ++j) facc += t[j];
            }
            __builtin_amdgcn_wave_barrier();
            __threadfence_block();
        }
    }
    double* o = sums + (size_t)(ls.leaf_off + leaf) * 16;
    if (lane < 9) o[lane] = acc;
    else if (lane == 9) o[9] = static_cast<double>(e - b);
    if (lane >= 9 && lane < 13) o[lane + 1] = static_cast<double>(facc);
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
        if (w[0] < 0 || w[1] < 0 || w[2] <= 0) {
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
    if (npts >= kNdtMinPointsPerVoxel) {
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

int ndt_launch_cellkeys(mrgfe_ctx* ctx, const float4* const* d_clouds, const Slice* d_slices, const SliceTable& t, const VoxelParams* d_vp, uint32_t* d_keys, uint32_t* d_vals)
{
    if (t.nprob() == 0 || t.max_blks == 0) return MRGFE_OK;
    hipLaunchKernelGGL(ndt_cellkey_kernel, dim3(t.max_blks, t.nprob()), dim3(256), 0, ctx->stream, d_clouds, d_slices, d_vp, d_keys, d_vals);
    MRGFE_HIP_CHECK(hipGetLastError());
    return MRGFE_OK;
}

int ndt_launch_segments(mrgfe_ctx* ctx, const uint32_t* d_sorted_keys, const uint32_t* d_flags, const uint32_t* d_ordinal, const Slice* d_slices, const SliceTable& t,
                        const LeafSlice* d_leaf_slices, uint32_t* d_seg_start, int32_t* d_seg_key)
{
    if (t.nprob() == 0 || t.max_blks == 0) return MRGFE_OK;
    hipLaunchKernelGGL(ndt_segments_kernel, dim3(t.max_blks, t.nprob()), dim3(256), 0, ctx->stream, d_sorted_keys, d_flags, d_ordinal, d_slices, d_leaf_slices,
                       d_seg_start, d_seg_key);
    MRGFE_HIP_CHECK(hipGetLastError());
    return MRGFE_OK;
}

int ndt_launch_leaves(mrgfe_ctx* ctx, const float4* const* d_clouds, const uint32_t* d_sorted_vals, const Slice* d_slices, const SliceTable& t, const LeafSlice* d_leaf_slices,
                      const VoxelParams* d_vp, uint32_t max_leaves, const uint32_t* d_seg_start, const int32_t* d_seg_key, double* d_sums, NdtLeafRec* d_leaves,
                      double* d_icov64, float4* d_centroid, int32_t* d_nr_points, void* d_lookup_base)
{
    if (t.nprob() == 0 || max_leaves == 0) return MRGFE_OK;
    hipLaunchKernelGGL(ndt_leaf_sums_kernel, dim3((max_leaves + 3) / 4, t.nprob()), dim3(256), 0, ctx->stream, d_clouds, d_sorted_vals, d_slices, d_leaf_slices,
                       d_seg_start, d_sums);
    hipLaunchKernelGGL(ndt_leaf_finalize_kernel, dim3((max_leaves + 255) / 256, t.nprob()), dim3(256), 0, ctx->stream, d_leaf_slices, d_vp, d_sums, d_seg_key,
                       d_leaves, d_icov64, d_centroid, d_nr_points, d_lookup_base);
    MRGFE_HIP_CHECK(hipGetLastError());
    return MRGFE_OK;
}

}  // namespace mrgfe
