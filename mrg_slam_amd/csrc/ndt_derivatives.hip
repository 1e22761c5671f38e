// csrc/ndt_derivatives.hip — the correspondence + Jacobian/Hessian kernel of NDT_HIP (the dominant kernel of
// registration_->align(): /root/reference/apps/scan_matching_odometry_component.cpp:265-266,
// src/mrg_slam/loop_detector.cpp:134) — MI355X counterpart of pclomp::NormalDistributionsTransform::
// computeDerivatives / updateDerivatives / computeHessian / updateHessian.
//
// Work distribution (a workgroup owns tiles of 256 source points):
//   phase 1, one lane per POINT: coalesced 16-byte point load, float rigid transform in pcl::transformPointCloud's
//            operation order (fused in: no transformed cloud is ever written), voxel coordinate, DIRECT7 (or 1 / 27)
//            lookups in the target grid, the point's angular Jacobian / second-derivative terms.  The per-point terms
//            are staged in LDS ([field][slot], conflict-free) and every occupied (point, voxel) pair is appended to an
//            LDS work queue at the offset given by a workgroup prefix sum (deterministic order, no atomics).
//   phase 2, one lane per PAIR: the queue is consumed 256 pairs at a time, so lanes stay busy although points have
//            between 0 and 7 occupied neighbours (a point-per-lane loop leaves ~1/3 of the lanes masked off).
//            Each lane gathers its point terms from LDS and the 48-byte voxel record from L2/HBM, evaluates the f32
//            per-pair terms in the reference's order and accumulates score, gradient (6) and all 36 Hessian entries in
//            f64 registers — the f32/f64 split of the reference.
//   phase 3: wavefront shuffle reduction, 4-wave LDS combine, one 384-byte partial record per workgroup; a second
//            tiny kernel adds the records of each pair in a fixed order (bitwise reproducible results).
// The contraction is 6 wide: far too thin for MFMA; the kernel is bound by VALU issue and by the
// dependent point -> lookup -> record loads, which is what the byte-model roofline in bench.py is held against.
#include <cstdio>

#include "dev_float.h"
#include "dev_utils.h"
#include "glibc_exp.h"
#include "ndt_ctl.h"
#include "ndt_derivatives.h"

namespace mrgfe {

// -DNDT_PHASE_CLOCK (diagnostic build, see profiles/README in README.md): wall-clock ticks (100 MHz) wavefront 0 of a workgroup of the
// score+gradient+Hessian variant spends in the stages of a tile — point loaded, table probes back, point terms staged, queue built
// (end of the point phase), pair phase — and in the item's epilogue, summed over all workgroups; MRGFE_PHASE=1 prints them after an alignment
#ifdef NDT_PHASE_CLOCK
__device__ unsigned long long g_phase[8];
__device__ unsigned long long g_rphase[10];  // ndt_reduce_kernel<true>: sums | state in | resume | solves | trig + request | state out, solve count, workgroups
#define NDT_CLOCK(var) long long var = 0; if (MODE == 0 && threadIdx.x == 0) var = wall_clock64()
#define NDT_CLOCK_WAIT(var) if (MODE == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); NDT_CLOCK(var)
#define NDT_CLOCK_ADD(slot, ticks) if (MODE == 0 && threadIdx.x == 0) atomicAdd(&g_phase[slot], static_cast<unsigned long long>(ticks))
#else
#define NDT_CLOCK(var)
#define NDT_CLOCK_WAIT(var)
#define NDT_CLOCK_ADD(slot, ticks)
#endif

// DIRECT7 probes: the voxel of the point, then +x, -x, +y, -y, +z, -z (pclomp getNeighborhoodAtPoint7)

constexpr int kTilePts = 256;  // points per tile == threads per workgroup

// exp of the per-pair weight.  The f64 passes (computeHessian, PCL NDT) and everything in reference-order mode call glibc's exp restated (glibc_exp.h): the
// double the reference's host computes.  The float path of the DEFAULT kernels keeps the device library's: its result differs from glibc's in the last bit of
// the double on one argument in ten, which survives the cast to float once in ~2^29 pairs (far below the default tree order's own noise), and glibc_exp costs
// the dominant kernel 5 % (0.585 -> 0.615 ms per launch of a 256-pair step, profiles/ab_libs.sh).  GLIBC = true: the reference-order record kernel.

struct Accum {
    double score;
    double g[6];
    double H[36];  // row-major, every entry (see ndt_types.h)
};

// a voxel record through a global load (dev_utils.h: pointers out of descriptors are generic to the compiler)
__device__ __forceinline__ NdtLeafRec load_leaf(const NdtLeafRec* p)
{
    const MRGFE_GLOBAL gvec4f* q = (const MRGFE_GLOBAL gvec4f*)p;  // 48 bytes, 16-byte aligned
    union { gvec4f v[3]; NdtLeafRec r; } u;
    u.v[0] = q[0]; u.v[1] = q[1]; u.v[2] = q[2];
    return u.r;
}

// float path: updateDerivatives for one (point, voxel) pair
// `mean` + the inverse covariance as the 3 x 3 float matrix the reference casts it to.  The default kernels read the 48-byte leaf record, which keeps the
// UPPER TRIANGLE (pair_float below: C symmetric); the inverse of a clamped covariance is asymmetric at ~1e-14 relative in f64, so on roughly one target in
// twenty some leaf has a float(icov(r, c)) one ulp away from float(icov(c, r)) — a 1e-8 relative difference in the sums of the points that meet that leaf, a
// documented deviation of the default mode.  The reference-order record kernel casts all nine f64 entries like the reference does.
template <bool HESS, bool GLIBC>
__device__ __forceinline__ void pair_float_c(Accum& acc, const double (&mean)[3], const float (&C)[3][3], const float xt[3], const float J3[3], const float J4[3], const float J5[3],
                                             const float (&PH)[6][3], float gauss_d2f, double gauss_d1)
{
#pragma clang fp contract(off)
    const float q[3] = {static_cast<float>(static_cast<double>(xt[0]) - mean[0]), static_cast<float>(static_cast<double>(xt[1]) - mean[1]),
                        static_cast<float>(static_cast<double>(xt[2]) - mean[2])};
    float qC[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) qC[c] = fdot3f(q[0], C[0][c], q[1], C[1][c], q[2], C[2][c]);
    const float qCq = fdot3f(q[0], qC[0], q[1], qC[1], q[2], qC[2]);
    const float arg0 = -gauss_d2f * qCq;
    const float arg = arg0 * 0.5f;
    float e = static_cast<float>(GLIBC ? glibc_exp(static_cast<double>(arg)) : exp(static_cast<double>(arg)));
    const float score_inc = static_cast<float>(-gauss_d1 * static_cast<double>(e));
    e = gauss_d2f * e;
    if (e > 1.0f || e < 0.0f || e != e) return;
    e = static_cast<float>(static_cast<double>(e) * gauss_d1);
    // CJ = C * [I | J3 J4 J5]
    float CJ[3][6];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        CJ[r][0] = C[r][0]; CJ[r][1] = C[r][1]; CJ[r][2] = C[r][2];
        CJ[r][3] = fdot3f(C[r][0], J3[0], C[r][1], J3[1], C[r][2], J3[2]);
        CJ[r][4] = fdot3f(C[r][0], J4[0], C[r][1], J4[1], C[r][2], J4[2]);
        CJ[r][5] = fdot3f(C[r][0], J5[0], C[r][1], J5[1], C[r][2], J5[2]);
    }
    float qCJ[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) qCJ[c] = fdot3f(q[0], CJ[0][c], q[1], CJ[1][c], q[2], CJ[2][c]);
#pragma unroll
    for (int c = 0; c < 6; ++c) { const float t = e * qCJ[c]; acc.g[c] += static_cast<double>(t); }
    acc.score += static_cast<double>(score_inc);
    if (!HESS) return;
    const float* Jc[6] = {nullptr, nullptr, nullptr, J3, J4, J5};
    // PH index of the second-derivative vector of (i,j), 3 <= i,j <= 5: a,b,c,d,e,f = (3,3),(3,4),(3,5),(4,4),(4,5),(5,5)
#pragma unroll
    for (int i = 0; i < 6; ++i) {
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            // point_gradient4_colj . c_inv4_x_point_gradient4_col_i  == J(:,j) . CJ(:,i)
            float jtcj;
            if (j < 3) jtcj = CJ[j][i];
            else       jtcj = fdot3f(Jc[j][0], CJ[0][i], Jc[j][1], CJ[1][i], Jc[j][2], CJ[2][i]);
            float qch = 0.0f;
            if (i >= 3 && j >= 3) {
                const int lo = i < j ? i : j, hi = i < j ? j : i;
                const int ph = (lo == 3) ? (hi - 3) : (lo == 4 ? (hi - 4 + 3) : 5);
                // (skipping the structural zeros J3[0] and PH[a..c][0] here was measured: no gain without packed f32, 8 % slower with it)
                // (a hand-laid packed-f32 form of this function — R^3 vectors as (xy pair, z), Hessian columns (0,1) and (4,5) as pairs,
                // scalars broadcast through op_sel — brings the pair loop from 358 to 334 VALU instructions (208 -> 83 scalar + 76 packed
                // f32 operations, 27 moves) but wants 205 VGPRs; under the 168 of three waves per SIMD the allocator rotates the f64
                // accumulators through 99 extra v_mov_b64 per pair.  Not kept.  Nor was, at 0.95 VALU busy in the second half of round 2,
                // packing only the last three operations of an entry — fma, + jtcj, * e — on column pairs (v_pk_fma / add / mul_f32, the rest
                // scalar, -fno-slp-vectorize): 54 packed for 108 scalar operations, bit-identical, but 21 more moves and 16 more spill
                // accesses; 10.5 -> 11.25 ms per 256-pair step.)
                qch = fdot3f(qC[0], PH[ph][0], qC[1], PH[ph][1], qC[2], PH[ph][2]);
            }
            const float t0 = -gauss_d2f * qCJ[i];
            const float t2 = __builtin_fmaf(t0, qCJ[j], qch);
            const float t3 = t2 + jtcj;
            const float t4 = e * t3;
            acc.H[i * 6 + j] += static_cast<double>(t4);
        }
    }
}

template <bool HESS, bool GLIBC = false>
__device__ __forceinline__ void pair_float(Accum& acc, const NdtLeafRec& rec, const float xt[3], const float J3[3], const float J4[3], const float J5[3],
                                           const float (&PH)[6][3], float gauss_d2f, double gauss_d1)
{
    // symmetric inverse covariance out of the record's upper triangle
    const float c00 = rec.icov[0], c01 = rec.icov[1], c02 = rec.icov[2], c11 = rec.icov[3], c12 = rec.icov[4], c22 = rec.icov[5];
    const float  C[3][3] = {{c00, c01, c02}, {c01, c11, c12}, {c02, c12, c22}};
    const double mean[3] = {rec.mean[0], rec.mean[1], rec.mean[2]};
    pair_float_c<HESS, GLIBC>(acc, mean, C, xt, J3, J4, J5, PH, gauss_d2f, gauss_d1);
}

// MODE 0: score+gradient+Hessian, 1: score+gradient, 2: Hessian only (double).  NNB: probed voxels (7, 1 or 27).
// Register budget: the full variant (43 f64 accumulators) is latency-bound at 2 waves/SIMD; capping it at 168 VGPRs
// (3 waves/SIMD, ~44 B/lane of spill) is 12 % faster on MI355X, 128 VGPRs (4 waves) spills too much (measured).
#ifndef NDT_MODE0_WAVES
#define NDT_MODE0_WAVES 3
#endif
#ifndef NDT_MODE2_WAVES
#define NDT_MODE2_WAVES 3  // the per-point f64 Hessian pass holds 21 + 15 f64 accumulators and no LDS staging
#endif
// LDS of one workgroup (all variants share it: a fused launch runs items of every variant)
template <int NNB>
struct NdtDerivShared {
    float    T[12];
    float    ja[8][3], ha[15][3];
    float    xt[3][kTilePts];
    float    xj[8][kTilePts];
    float    xh[15][kTilePts];
    uint32_t queue[kTilePts * NNB];  // (slot << 24) | leaf id
    uint32_t scan[8];
    double   red[4][kNdtPartialStride];
};

// One work item: `ppt` consecutive tiles of pair `pi`, starting at tile item_in_pair * ppt; one 384-byte partial record.
template <int MODE, int NNB>
__device__ __forceinline__ void ndt_derivatives_item(NdtDerivShared<NNB>& sh, const NdtGridDev* __restrict__ grids, const NdtPairDev* __restrict__ pairs,
                                                     const NdtEvalDev* __restrict__ evals, uint32_t pi, uint32_t item_in_pair, uint32_t ppt, double* __restrict__ partials)
{
    constexpr int kTile = kTilePts;
    float (&s_T)[12] = sh.T;
    float (&s_ja)[8][3] = sh.ja;
    float (&s_ha)[15][3] = sh.ha;
    float (&s_xt)[3][kTilePts] = sh.xt;
    float (&s_xj)[8][kTilePts] = sh.xj;
    float (&s_xh)[15][kTilePts] = sh.xh;
    uint32_t (&s_queue)[kTilePts * NNB] = sh.queue;
    uint32_t (&s_scan)[8] = sh.scan;
    double (&s_red)[4][kNdtPartialStride] = sh.red;
    {
    const NdtPairDev pr = pairs[pi];
    const NdtEvalDev& ev = evals[pi];
    const uint32_t part_off = pr.part_off;
    const NdtGridDev g = grids[pr.grid];  // by value: the grid parameters live in scalar registers for the whole item
    __syncthreads();  // the previous item's epilogue has read s_red / the staged tables
    if (threadIdx.x < 12) s_T[threadIdx.x] = ev.T[threadIdx.x];
    if (MODE == 0) {
        if (threadIdx.x >= 64 && threadIdx.x < 64 + 24) (&s_ja[0][0])[threadIdx.x - 64] = (&ev.j_ang[0][0])[threadIdx.x - 64];
        if (threadIdx.x >= 128 && threadIdx.x < 128 + 45) (&s_ha[0][0])[threadIdx.x - 128] = (&ev.h_ang[0][0])[threadIdx.x - 128];
    }
    __syncthreads();

    const float  gauss_d2f = static_cast<float>(ev.gauss_d2);
    const double gauss_d1 = ev.gauss_d1, gauss_d2 = ev.gauss_d2;
    const float  leaf = g.leaf_size;
    const bool   kdtree = (NNB == 27) && (ev.search == MRGFE_KDTREE);

    Accum acc;
    acc.score = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) acc.g[k] = 0;
#pragma unroll
    for (int k = 0; k < 36; ++k) acc.H[k] = 0;
    uint32_t nb_total = 0, nb_mine = 0;
#ifdef NDT_PHASE_CLOCK
    long long clk[6] = {0, 0, 0, 0, 0, 0};
#endif

    const uint32_t base = item_in_pair * static_cast<uint32_t>(kTilePts) * ppt;
    const uint32_t last = min(pr.n_src, base + static_cast<uint32_t>(kTilePts) * ppt);  // end of this item's points
    for (uint32_t tile0 = base; tile0 < last; tile0 += kTile) {
        // ---- phase 1: one lane per point -------------------------------------------------------------------------
        NDT_CLOCK(tc0);
        NDT_CLOCK(tc_p); NDT_CLOCK(tc_l); NDT_CLOCK(tc_s);
        const uint32_t i = tile0 + threadIdx.x;
        int32_t  ids[NNB];
        uint32_t cnt = 0;
#pragma unroll
        for (int n = 0; n < NNB; ++n) ids[n] = -1;
        if (threadIdx.x < kTile && i < last) {
            const float4 p = load_point(pr.src + i);
            float xt[3];
#ifdef NDT_PHASE_CLOCK
            if (MODE == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); if (threadIdx.x == 0) tc_p = wall_clock64(); }
#endif
            transform_point(s_T, p.x, p.y, p.z, xt[0], xt[1], xt[2]);
            // getNeighborhoodAtPoint: floor(p / leaf_size)
            int ijk[3];
            if (leaf == 1.0f) {  // uniform; x / 1.0f == x exactly, and the IEEE division sequence is ~10 instructions per axis
                ijk[0] = static_cast<int>(floorf(xt[0])); ijk[1] = static_cast<int>(floorf(xt[1])); ijk[2] = static_cast<int>(floorf(xt[2]));
            } else {
                ijk[0] = static_cast<int>(floorf(xt[0] / leaf)); ijk[1] = static_cast<int>(floorf(xt[1] / leaf)); ijk[2] = static_cast<int>(floorf(xt[2] / leaf));
            }
            // per axis: is cell (ijk - 1, ijk, ijk + 1) inside the grid box?  key of a neighbour = key of the centre cell +
            // offset . divb_mul (two's-complement arithmetic: the centre itself may lie outside while a neighbour is inside)
            bool in_box[3][3];
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int o = -1; o <= 1; ++o) in_box[a][o + 1] = ijk[a] + o >= g.min_b[a] && ijk[a] + o <= g.max_b[a];
            const uint32_t mul[3] = {static_cast<uint32_t>(g.divb_mul[0]), static_cast<uint32_t>(g.divb_mul[1]), static_cast<uint32_t>(g.divb_mul[2])};
            const uint32_t key0 = (static_cast<uint32_t>(ijk[0]) - static_cast<uint32_t>(g.min_b[0])) * mul[0] + (static_cast<uint32_t>(ijk[1]) - static_cast<uint32_t>(g.min_b[1])) * mul[1] +
                                  (static_cast<uint32_t>(ijk[2]) - static_cast<uint32_t>(g.min_b[2])) * mul[2];
            uint32_t keys[NNB];
            bool     ok[NNB];
#pragma unroll
            for (int n = 0; n < NNB; ++n) {
                int o0, o1, o2;
                if (NNB == 27) { o0 = n / 9 - 1; o1 = (n / 3) % 3 - 1; o2 = n % 3 - 1; }
                else if (NNB == 7) { o0 = (n == 1) - (n == 2); o1 = (n == 3) - (n == 4); o2 = (n == 5) - (n == 6); }  // kOff7, folded at compile time
                else           { o0 = o1 = o2 = 0; }
                ok[n] = in_box[0][o0 + 1] && in_box[1][o1 + 1] && in_box[2][o2 + 1];
                keys[n] = key0 + static_cast<uint32_t>(o0) * mul[0] + static_cast<uint32_t>(o1) * mul[1] + static_cast<uint32_t>(o2) * mul[2];
            }
            if (g.dense) {
                // all probes of the point in flight together (a probe-by-probe loop waits for each load in turn)
                const MRGFE_GLOBAL int32_t* __restrict__ table = as_global(static_cast<const int32_t*>(g.lookup));
                int32_t v[NNB];
#pragma unroll
                for (int n = 0; n < NNB; ++n) v[n] = table[ok[n] ? keys[n] : 0u];
#pragma unroll
                for (int n = 0; n < NNB; ++n) ids[n] = ok[n] ? v[n] : -1;
            } else {
#pragma unroll
                for (int n = 0; n < NNB; ++n) ids[n] = ok[n] ? ndt_lookup(g, keys[n]) : -1;
            }
            if (kdtree) {
                // radiusSearch(point, resolution) over voxel centroids: FLANN keeps dist^2 < r^2
                const float r2 = leaf * leaf;
#pragma unroll
                for (int n = 0; n < NNB; ++n)
                    if (ids[n] >= 0) {
                        const float4 c = load_point(g.centroid + ids[n]);
                        const float  dx = c.x - xt[0], dy = c.y - xt[1], dz = c.z - xt[2];
                        const float  d = dot3f(dx, dx, dy, dy, dz, dz);
                        if (!(d < r2)) ids[n] = -1;
                    }
            }
#pragma unroll
            for (int n = 0; n < NNB; ++n) cnt += ids[n] >= 0 ? 1u : 0u;
#ifdef NDT_PHASE_CLOCK
            if (MODE == 0 && threadIdx.x == 0) tc_l = wall_clock64();  // (cnt depends on every probe)
#endif
            if (MODE == 2) {
                // ---- computeHessian (pclomp: f64, PCL's 3x6 / 18x6 forms), one lane per POINT ------------------------------
                // Per pair the reference evaluates  e * ( -d2 (q.C J_i)(q.C J_j) + q.C PH_ij + J_j.C J_i ).  J and PH belong to the
                // point, so with v = C q the sum over the point's voxels factors into
                //     J_i^T [ sum e C  -  d2 sum e v v^T ] J_j  +  (sum e v) . PH_ij
                // ~30 fused multiply-adds per voxel and ~90 per point instead of ~160 per pair: 3x fewer f64 operations.  Everything is
                // f64, so this differs from the reference's association by rounding at the 1e-16 level, like the order of the sums
                // (the float path above keeps the reference's per-pair sequence, where the rounding is 1e-7 and matters).
                nb_mine += cnt;
                if (cnt) {
                    const double x[3] = {p.x, p.y, p.z};
                    double M1[6] = {0, 0, 0, 0, 0, 0}, M2[6] = {0, 0, 0, 0, 0, 0}, w[3] = {0, 0, 0};
#pragma unroll
                    for (int n = 0; n < NNB; ++n) {
                        if (ids[n] < 0) continue;
                        const uint32_t lid = static_cast<uint32_t>(ids[n]);
                        if (lid >= g.n_leaves) continue;  // cannot happen; keeps a corrupted table entry from faulting the GPU
                        const MRGFE_GLOBAL double* __restrict__ C = as_global(g.icov64 + (size_t)lid * 9);
                        const MRGFE_GLOBAL double* __restrict__ mean = as_global(g.leaves[lid].mean);
                        const double q[3] = {static_cast<double>(xt[0]) - mean[0], static_cast<double>(xt[1]) - mean[1], static_cast<double>(xt[2]) - mean[2]};
                        double v[3];
#pragma unroll
                        for (int r = 0; r < 3; ++r) v[r] = fdot3d(C[r * 3 + 0], q[0], C[r * 3 + 1], q[1], C[r * 3 + 2], q[2]);
                        double e = gauss_d2 * glibc_exp(-gauss_d2 * fdot3d(q[0], v[0], q[1], v[1], q[2], v[2]) / 2);
                        if (e > 1 || e < 0 || e != e) continue;
                        e *= gauss_d1;
                        const double ev3[3] = {e * v[0], e * v[1], e * v[2]};
                        M1[0] = __builtin_fma(e, C[0], M1[0]); M1[1] = __builtin_fma(e, C[1], M1[1]); M1[2] = __builtin_fma(e, C[2], M1[2]);
                        M1[3] = __builtin_fma(e, C[4], M1[3]); M1[4] = __builtin_fma(e, C[5], M1[4]); M1[5] = __builtin_fma(e, C[8], M1[5]);
                        M2[0] = __builtin_fma(ev3[0], v[0], M2[0]); M2[1] = __builtin_fma(ev3[0], v[1], M2[1]); M2[2] = __builtin_fma(ev3[0], v[2], M2[2]);
                        M2[3] = __builtin_fma(ev3[1], v[1], M2[3]); M2[4] = __builtin_fma(ev3[1], v[2], M2[4]); M2[5] = __builtin_fma(ev3[2], v[2], M2[5]);
                        w[0] += ev3[0]; w[1] += ev3[1]; w[2] += ev3[2];
                    }
                    // A = sum e C - d2 sum e v v^T (symmetric: xx, xy, xz, yy, yz, zz)
                    double A[3][3];
                    {
                        double a[6];
#pragma unroll
                        for (int k = 0; k < 6; ++k) a[k] = __builtin_fma(-gauss_d2, M2[k], M1[k]);
                        A[0][0] = a[0]; A[0][1] = A[1][0] = a[1]; A[0][2] = A[2][0] = a[2]; A[1][1] = a[3]; A[1][2] = A[2][1] = a[4]; A[2][2] = a[5];
                    }
                    // computePointDerivatives, double form: rotational columns of J and the second-derivative vectors
                    double xj[8], xh[15];
#pragma unroll
                    for (int r = 0; r < 8; ++r) xj[r] = fdot3d(x[0], ev.j_ang_d[r][0], x[1], ev.j_ang_d[r][1], x[2], ev.j_ang_d[r][2]);
#pragma unroll
                    for (int r = 0; r < 15; ++r) xh[r] = fdot3d(x[0], ev.h_ang_d[r][0], x[1], ev.h_ang_d[r][1], x[2], ev.h_ang_d[r][2]);
                    const double Jr[3][3] = {{0.0, xj[2], xj[5]}, {xj[0], xj[3], xj[6]}, {xj[1], xj[4], xj[7]}};  // Jr[row][c]: columns 3, 4, 5 of J
                    double AJ[3][3];  // A * Jr
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        AJ[r][0] = fdot3d_z(A[r][1], Jr[1][0], A[r][2], Jr[2][0]);  // J(0,3) is a structural zero
                        AJ[r][1] = fdot3d(A[r][0], Jr[0][1], A[r][1], Jr[1][1], A[r][2], Jr[2][1]);
                        AJ[r][2] = fdot3d(A[r][0], Jr[0][2], A[r][1], Jr[1][2], A[r][2], Jr[2][2]);
                    }
                    // translation block and translation x rotation block (upper triangle: i <= j)
                    acc.H[0 * 6 + 0] += A[0][0]; acc.H[0 * 6 + 1] += A[0][1]; acc.H[0 * 6 + 2] += A[0][2]; acc.H[1 * 6 + 1] += A[1][1]; acc.H[1 * 6 + 2] += A[1][2]; acc.H[2 * 6 + 2] += A[2][2];
#pragma unroll
                    for (int i = 0; i < 3; ++i)
#pragma unroll
                        for (int c = 0; c < 3; ++c) acc.H[i * 6 + 3 + c] += AJ[i][c];
                    // rotation block: J_i^T (A J_j) + w . PH_ij;  PH index a,b,c,d,e,f = (3,3),(3,4),(3,5),(4,4),(4,5),(5,5)
                    const double PH[6][3] = {{0, xh[0], xh[1]}, {0, xh[2], xh[3]}, {0, xh[4], xh[5]}, {xh[6], xh[7], xh[8]}, {xh[9], xh[10], xh[11]}, {xh[12], xh[13], xh[14]}};
#pragma unroll
                    for (int i = 0; i < 3; ++i)
#pragma unroll
                        for (int j = i; j < 3; ++j) {
                            const int ph = (i == 0) ? j : (i == 1 ? j + 2 : 5);
                            const double jaj = (i == 0) ? fdot3d_z(Jr[1][0], AJ[1][j], Jr[2][0], AJ[2][j]) : fdot3d(Jr[0][i], AJ[0][j], Jr[1][i], AJ[1][j], Jr[2][i], AJ[2][j]);
                            const double wph = (ph < 3) ? fdot3d_z(w[1], PH[ph][1], w[2], PH[ph][2]) : fdot3d(w[0], PH[ph][0], w[1], PH[ph][1], w[2], PH[ph][2]);
                            acc.H[(3 + i) * 6 + 3 + j] += jaj + wph;
                        }
                }
                cnt = 0;  // nothing queued: the pair phase below is the float path's
#pragma unroll
                for (int n = 0; n < NNB; ++n) ids[n] = -1;
            } else if (MODE == 1) {
                // ---- score + gradient (line-search trials), one lane per POINT ---------------------------------------------------
                // ~110 instructions per pair: less than the queue machinery around them costs (staging through LDS, the offsets, three
                // barriers per tile — this variant spent two thirds of a tile there), so the lane walks its own voxels in probe order
                // although a third of the lanes idle in every step (0 - 7 occupied neighbours, 4.4 on average).  Same per-pair float
                // terms; the f64 sums run per point instead of per queue slot (the oracle's GPU-order mode follows).
                nb_mine += cnt;
                if (cnt) {
                    float xj[8];
#pragma unroll
                    for (int r = 0; r < 8; ++r) xj[r] = fdot3f(ev.j_ang[r][0], p.x, ev.j_ang[r][1], p.y, ev.j_ang[r][2], p.z);
                    const float J3[3] = {0.0f, xj[0], xj[1]}, J4[3] = {xj[2], xj[3], xj[4]}, J5[3] = {xj[5], xj[6], xj[7]};
                    const float PH[6][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
                    // (requesting the records of four probes together — two waits per point instead of one per occupied voxel — changed
                    // nothing: 156 registers, three waves per SIMD, same time)
#pragma unroll
                    for (int n = 0; n < NNB; ++n) {
                        if (ids[n] < 0) continue;
                        const uint32_t lid = static_cast<uint32_t>(ids[n]);
                        if (lid >= g.n_leaves) continue;  // cannot happen; keeps a corrupted table entry from faulting the GPU
                        const NdtLeafRec rec = load_leaf(g.leaves + lid);
                        pair_float<false>(acc, rec, xt, J3, J4, J5, PH, gauss_d2f, gauss_d1);
                    }
                }
                cnt = 0;
            } else if (cnt) {
                s_xt[0][threadIdx.x] = xt[0]; s_xt[1][threadIdx.x] = xt[1]; s_xt[2][threadIdx.x] = xt[2];
                // computePointDerivatives, float form: x_j_ang = j_ang * x, x_h_ang = h_ang * x
#pragma unroll
                for (int r = 0; r < 8; ++r) s_xj[r][threadIdx.x] = fdot3f(s_ja[r][0], p.x, s_ja[r][1], p.y, s_ja[r][2], p.z);
                if (MODE == 0) {
#pragma unroll
                    for (int r = 0; r < 15; ++r) s_xh[r][threadIdx.x] = fdot3f(s_ha[r][0], p.x, s_ha[r][1], p.y, s_ha[r][2], p.z);
                }
            }
        }
        if (MODE != 0) continue;  // next tile: no LDS staging, no pair queue
#ifdef NDT_PHASE_CLOCK
        if (MODE == 0) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); if (threadIdx.x == 0) tc_s = wall_clock64(); }
#endif
        // queue offsets: exclusive scan of cnt over the workgroup.  ONE barrier: every lane adds up the totals of the wavefronts before
        // its own, and the four totals alternate between two LDS slots by tile parity (block_exclusive_scan costs three barriers, and
        // each is a rendezvous of four wavefronts on four SIMDs that serve two other workgroups)
        uint32_t total, off;
        {
            // within the wavefront: cnt <= NNB is a few bits wide — one ballot per bit and a count of the set bits below the lane
            // (no cross-lane round trips; a shuffle scan is six dependent ones)
            constexpr int kBits = NNB <= 1 ? 1 : (NNB <= 7 ? 3 : 5);
            uint32_t excl = 0, wave_total = 0;
#pragma unroll
            for (int b = 0; b < kBits; ++b) {
                const unsigned long long m = __ballot((cnt >> b) & 1u);
                excl += __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u)) << b;
                wave_total += static_cast<uint32_t>(__popcll(m)) << b;
            }
            uint32_t* tot = s_scan + (((tile0 - base) / kTile) & 1u) * 4u;
            if (lane_id() == 0) tot[wave_id()] = wave_total;
            __syncthreads();
            const uint32_t t0 = tot[0], t1 = tot[1], t2 = tot[2], t3 = tot[3];
            const int w = wave_id();
            off = excl + (w > 0 ? t0 : 0u) + (w > 1 ? t1 : 0u) + (w > 2 ? t2 : 0u);
            total = t0 + t1 + t2 + t3;
        }
#pragma unroll
        for (int n = 0; n < NNB; ++n)
            if (ids[n] >= 0) s_queue[off++] = (threadIdx.x << 24) | static_cast<uint32_t>(ids[n]);
        __syncthreads();
        nb_total += (threadIdx.x == 0) ? total : 0u;

        // ---- phase 2: one lane per (point, voxel) pair -----------------------------------------------------------
        NDT_CLOCK(tc1);
        // (prefetching the next pair's record was measured slower both as a register double buffer — it costs the occupancy it
        // saves — and as an LDS-DMA into per-lane slots, -6 % / -13 % for the two float variants)
        for (uint32_t qi = threadIdx.x; qi < total; qi += kTilePts) {
            const uint32_t entry = s_queue[qi];
            const uint32_t slot = entry >> 24, lid = entry & 0x00FFFFFFu;
            if (lid >= g.n_leaves) continue;  // cannot happen; keeps a corrupted queue entry from faulting the GPU
            const NdtLeafRec rec = load_leaf(g.leaves + lid);
            const float xt[3] = {s_xt[0][slot], s_xt[1][slot], s_xt[2][slot]};
            float xj[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) xj[r] = s_xj[r][slot];
            const float J3[3] = {0.0f, xj[0], xj[1]}, J4[3] = {xj[2], xj[3], xj[4]}, J5[3] = {xj[5], xj[6], xj[7]};
            float PH[6][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
            if (MODE == 0) {
                float xh[15];
#pragma unroll
                for (int r = 0; r < 15; ++r) xh[r] = s_xh[r][slot];
                PH[0][1] = xh[0];  PH[0][2] = xh[1];                     // a  (3,3)
                PH[1][1] = xh[2];  PH[1][2] = xh[3];                     // b  (3,4)
                PH[2][1] = xh[4];  PH[2][2] = xh[5];                     // c  (3,5)
                PH[3][0] = xh[6];  PH[3][1] = xh[7];  PH[3][2] = xh[8];  // d  (4,4)
                PH[4][0] = xh[9];  PH[4][1] = xh[10]; PH[4][2] = xh[11]; // e  (4,5)
                PH[5][0] = xh[12]; PH[5][1] = xh[13]; PH[5][2] = xh[14]; // f  (5,5)
            }
            pair_float<MODE == 0>(acc, rec, xt, J3, J4, J5, PH, gauss_d2f, gauss_d1);
        }
        __syncthreads();  // the next tile overwrites the staged terms and the queue
#ifdef NDT_PHASE_CLOCK
        NDT_CLOCK(tc2);
        if (MODE == 0 && threadIdx.x == 0) { clk[0] += tc_p - tc0; clk[1] += tc_l - tc_p; clk[2] += tc_s - tc_l; clk[3] += tc1 - tc_s; clk[4] += tc2 - tc1; clk[5] += 1; }  // (one atomic per item, below: per tile they held the next tile's loads back)
#endif
    }

    // ---- phase 3: workgroup reduction -----------------------------------------------------------------------------
    NDT_CLOCK(tc3);
    // the sums this variant delivers: MODE 0 all 44 (score, gradient, Hessian, neighbour count), MODE 1 score + gradient + count,
    // MODE 2 Hessian + count; folded wave reduction (dev_utils.h): the same tree as 44 separate wave_sum calls, a sixth of the traffic
    constexpr int kVals = MODE == 0 ? kNdtAccum : (MODE == 1 ? 8 : 37);
    double vals[kVals];
    int    keys[kVals];
#pragma unroll
    for (int n = 0; n < kVals; ++n) {
        const int k = (MODE == 1 && n == 7) ? kNdtNbIndex : (MODE == 2 ? 7 + n : n);  // slot of the partial record
        keys[n] = k;
        if (k == 0)                vals[n] = acc.score;
        else if (k < 7)            vals[n] = acc.g[k - 1];
        else if (k < kNdtNbIndex)  vals[n] = (MODE == 2 && (k - 7) / 6 > (k - 7) % 6) ? acc.H[((k - 7) % 6) * 6 + (k - 7) / 6] : acc.H[k - 7];  // the f64 pass fills the upper triangle
        else                       vals[n] = static_cast<double>(MODE != 0 ? nb_mine : nb_total);
    }
    double total_v;
    int    total_k;
    wave_sum_fold<kVals, 32>(vals, keys, total_v, total_k);
    s_red[wave_id()][total_k] = total_v;  // (lanes that hold the same sum write the same double)
    __syncthreads();
    if (threadIdx.x < kNdtPartialStride) {
        const int  k = threadIdx.x;
        const bool skip = k >= kNdtAccum || (MODE == 1 && k >= 7 && k < kNdtNbIndex) || (MODE == 2 && k < 7);
        double     r = 0.0;
        if (!skip) r = ((s_red[0][k] + s_red[1][k]) + s_red[2][k]) + s_red[3][k];
        partials[(size_t)(part_off + item_in_pair) * kNdtPartialStride + k] = r;
    }
#ifdef NDT_PHASE_CLOCK
    NDT_CLOCK(tc4);
    for (int c = 0; c < 6; ++c) NDT_CLOCK_ADD(c, clk[c]);
    NDT_CLOCK_ADD(6, tc4 - tc3); NDT_CLOCK_ADD(7, 1);
#endif
    }
}

// ---- PCL_NDT_HIP: pcl::NormalDistributionsTransform (PCL 1.12), the f64 formulation ---------------------------------------------------
// What registration_method "NDT" (and every unknown name) runs in the reference (/root/reference/src/mrg_slam/registrations.cpp:115-129):
// computeDerivatives / updateDerivatives / computeHessian / updateHessian with Eigen::Vector3d / Matrix3d throughout, over
// target_cells_.radiusSearch(x_trans_pt, resolution_) — the voxels whose float centroid lies within one resolution of the point (27 cells
// to probe).  One lane per POINT for all three kinds of evaluation.  J and the second-derivative vectors belong to the point, so with
// v = C q and e the pair's weight the reference's per-pair sums
//     score += -d1 exp(..);   g_i += e (q.C J_i);   H_ij += e ( -d2 (q.C J_i)(q.C J_j) + q.C PH_ij + J_j.C J_i )
// factor over the point's voxels into
//     g_i = (sum e v) . J_i;   H_ij = J_i^T [ sum e C - d2 sum e v v^T ] J_j + (sum e v) . PH_ij
// — ~45 fused multiply-adds per voxel and ~110 per point instead of ~300 per pair.  Everything is f64: the different association moves
// a sum by ~1e-16 relative, like the order of the additions (the reference adds point after point on one thread); oracle/pcl_ndt.cpp restates
// both the reference's order and, as a diagnostic, this one.  The inverse covariance is read as its upper triangle (the cofactor inverse of
// a clamped covariance is symmetric to ~1e-14 relative).
// The lane first collects its hits — occupied cell, centroid inside the radius — in its own column of the LDS queue, then walks them: a
// wavefront runs max(hits) rounds instead of 27 mostly empty ones.
template <int MODE>
__device__ __forceinline__ void ndt_derivatives_f64_item(NdtDerivShared<27>& sh, const NdtGridDev* __restrict__ grids, const NdtPairDev* __restrict__ pairs,
                                                         const NdtEvalDev* __restrict__ evals, uint32_t pi, uint32_t item_in_pair, uint32_t ppt, double* __restrict__ partials)
{
    constexpr int NNB = 27;
    float (&s_T)[12] = sh.T;
    uint32_t (&s_hits)[kTilePts * NNB] = sh.queue;  // [hit][lane]
    double (&s_red)[4][kNdtPartialStride] = sh.red;
    const NdtPairDev pr = pairs[pi];
    const NdtEvalDev& ev = evals[pi];
    const uint32_t part_off = pr.part_off;
    const NdtGridDev g = grids[pr.grid];
    __syncthreads();  // the previous item's epilogue has read s_red / the staged transform
    if (threadIdx.x < 12) s_T[threadIdx.x] = ev.T[threadIdx.x];
    __syncthreads();
    const double gauss_d1 = ev.gauss_d1, gauss_d2 = ev.gauss_d2;
    const float  leaf = g.leaf_size;
    const float  r2 = leaf * leaf;  // float(radius * radius): the product of two floats is exact in double, so one rounding either way

    Accum acc;
    acc.score = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) acc.g[k] = 0;
#pragma unroll
    for (int k = 0; k < 36; ++k) acc.H[k] = 0;
    uint32_t nb_mine = 0;

    const uint32_t base = item_in_pair * static_cast<uint32_t>(kTilePts) * ppt;
    const uint32_t last = min(pr.n_src, base + static_cast<uint32_t>(kTilePts) * ppt);
    for (uint32_t tile0 = base; tile0 < last; tile0 += kTilePts) {
        const uint32_t i = tile0 + threadIdx.x;
        if (i >= last) continue;
        const float4 p = load_point(pr.src + i);
        float xt[3];
        transform_point(s_T, p.x, p.y, p.z, xt[0], xt[1], xt[2]);
        int ijk[3];
        if (leaf == 1.0f) { ijk[0] = static_cast<int>(floorf(xt[0])); ijk[1] = static_cast<int>(floorf(xt[1])); ijk[2] = static_cast<int>(floorf(xt[2])); }
        else              { ijk[0] = static_cast<int>(floorf(xt[0] / leaf)); ijk[1] = static_cast<int>(floorf(xt[1] / leaf)); ijk[2] = static_cast<int>(floorf(xt[2] / leaf)); }
        bool in_box[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int o = -1; o <= 1; ++o) in_box[a][o + 1] = ijk[a] + o >= g.min_b[a] && ijk[a] + o <= g.max_b[a];
        const uint32_t mul[3] = {static_cast<uint32_t>(g.divb_mul[0]), static_cast<uint32_t>(g.divb_mul[1]), static_cast<uint32_t>(g.divb_mul[2])};
        const uint32_t key0 = (static_cast<uint32_t>(ijk[0]) - static_cast<uint32_t>(g.min_b[0])) * mul[0] + (static_cast<uint32_t>(ijk[1]) - static_cast<uint32_t>(g.min_b[1])) * mul[1] +
                              (static_cast<uint32_t>(ijk[2]) - static_cast<uint32_t>(g.min_b[2])) * mul[2];
        // ---- the point's hits: nine probes (one x slab) in flight at a time, then their centroids --------------------------------------
        uint32_t cnt = 0;
#pragma unroll
        for (int slab = 0; slab < 3; ++slab) {
            int32_t ids[9];
#pragma unroll
            for (int m = 0; m < 9; ++m) {
                const int o0 = slab - 1, o1 = m / 3 - 1, o2 = m % 3 - 1;
                const bool     ok = in_box[0][o0 + 1] && in_box[1][o1 + 1] && in_box[2][o2 + 1];
                const uint32_t key = key0 + static_cast<uint32_t>(o0) * mul[0] + static_cast<uint32_t>(o1) * mul[1] + static_cast<uint32_t>(o2) * mul[2];
                if (g.dense) { const int32_t v = as_global(static_cast<const int32_t*>(g.lookup))[ok ? key : 0u]; ids[m] = ok ? v : -1; }
                else         ids[m] = ok ? ndt_lookup(g, key) : -1;
            }
            float4 c[9];
#pragma unroll
            for (int m = 0; m < 9; ++m) c[m] = load_point(g.centroid + (ids[m] >= 0 && static_cast<uint32_t>(ids[m]) < g.n_leaves ? ids[m] : 0));
#pragma unroll
            for (int m = 0; m < 9; ++m) {
                if (ids[m] < 0 || static_cast<uint32_t>(ids[m]) >= g.n_leaves) continue;
                const float dx = c[m].x - xt[0], dy = c[m].y - xt[1], dz = c[m].z - xt[2];
                const float d = dot3f(dx, dx, dy, dy, dz, dz);
                if (d < r2) { s_hits[cnt * kTilePts + threadIdx.x] = static_cast<uint32_t>(ids[m]); ++cnt; }  // FLANN keeps dist^2 < r^2
            }
        }
        nb_mine += cnt;
        if (!cnt) continue;
        // ---- the point's voxels ---------------------------------------------------------------------------------------------------------
        double M1[6] = {0, 0, 0, 0, 0, 0}, M2[6] = {0, 0, 0, 0, 0, 0}, w[3] = {0, 0, 0}, sc = 0;
        for (uint32_t k = 0; k < cnt; ++k) {
            const uint32_t lid = s_hits[k * kTilePts + threadIdx.x];
            const MRGFE_GLOBAL double* __restrict__ C = as_global(g.icov64 + (size_t)lid * 9);
            const MRGFE_GLOBAL double* __restrict__ mean = as_global(g.leaves[lid].mean);
            const double c00 = C[0], c01 = C[1], c02 = C[2], c11 = C[4], c12 = C[5], c22 = C[8];
            const double q[3] = {static_cast<double>(xt[0]) - mean[0], static_cast<double>(xt[1]) - mean[1], static_cast<double>(xt[2]) - mean[2]};
            double v[3];
            v[0] = fdot3d(c00, q[0], c01, q[1], c02, q[2]);
            v[1] = fdot3d(c01, q[0], c11, q[1], c12, q[2]);
            v[2] = fdot3d(c02, q[0], c12, q[1], c22, q[2]);
            const double e_raw = glibc_exp(-gauss_d2 * fdot3d(q[0], v[0], q[1], v[1], q[2], v[2]) / 2);
            double e = gauss_d2 * e_raw;
            if (e > 1 || e < 0 || e != e) continue;  // updateDerivatives returns 0: the pair adds nothing, not even its score
            e *= gauss_d1;
            sc += -gauss_d1 * e_raw;
            const double ev3[3] = {e * v[0], e * v[1], e * v[2]};
            w[0] += ev3[0]; w[1] += ev3[1]; w[2] += ev3[2];
            if (MODE != 1) {
                M1[0] = __builtin_fma(e, c00, M1[0]); M1[1] = __builtin_fma(e, c01, M1[1]); M1[2] = __builtin_fma(e, c02, M1[2]);
                M1[3] = __builtin_fma(e, c11, M1[3]); M1[4] = __builtin_fma(e, c12, M1[4]); M1[5] = __builtin_fma(e, c22, M1[5]);
                M2[0] = __builtin_fma(ev3[0], v[0], M2[0]); M2[1] = __builtin_fma(ev3[0], v[1], M2[1]); M2[2] = __builtin_fma(ev3[0], v[2], M2[2]);
                M2[3] = __builtin_fma(ev3[1], v[1], M2[3]); M2[4] = __builtin_fma(ev3[1], v[2], M2[4]); M2[5] = __builtin_fma(ev3[2], v[2], M2[5]);
            }
        }
        // ---- the point's share: computePointDerivatives (f64) applied once to the voxel sums ---------------------------------------------
        const double x[3] = {p.x, p.y, p.z};
        double xj[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) xj[r] = fdot3d(x[0], ev.j_ang_d[r][0], x[1], ev.j_ang_d[r][1], x[2], ev.j_ang_d[r][2]);
        const double Jr[3][3] = {{0.0, xj[2], xj[5]}, {xj[0], xj[3], xj[6]}, {xj[1], xj[4], xj[7]}};  // Jr[row][c]: columns 3, 4, 5 of J
        if (MODE != 2) {
            acc.score += sc;
            acc.g[0] += w[0]; acc.g[1] += w[1]; acc.g[2] += w[2];
            acc.g[3] += fdot3d_z(w[1], Jr[1][0], w[2], Jr[2][0]);
            acc.g[4] += fdot3d(w[0], Jr[0][1], w[1], Jr[1][1], w[2], Jr[2][1]);
            acc.g[5] += fdot3d(w[0], Jr[0][2], w[1], Jr[1][2], w[2], Jr[2][2]);
        }
        if (MODE != 1) {
            double xh[15];
#pragma unroll
            for (int r = 0; r < 15; ++r) xh[r] = fdot3d(x[0], ev.h_ang_d[r][0], x[1], ev.h_ang_d[r][1], x[2], ev.h_ang_d[r][2]);
            double A[3][3];
            {
                double a[6];
#pragma unroll
                for (int k = 0; k < 6; ++k) a[k] = __builtin_fma(-gauss_d2, M2[k], M1[k]);
                A[0][0] = a[0]; A[0][1] = A[1][0] = a[1]; A[0][2] = A[2][0] = a[2]; A[1][1] = a[3]; A[1][2] = A[2][1] = a[4]; A[2][2] = a[5];
            }
            double AJ[3][3];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                AJ[r][0] = fdot3d_z(A[r][1], Jr[1][0], A[r][2], Jr[2][0]);
                AJ[r][1] = fdot3d(A[r][0], Jr[0][1], A[r][1], Jr[1][1], A[r][2], Jr[2][1]);
                AJ[r][2] = fdot3d(A[r][0], Jr[0][2], A[r][1], Jr[1][2], A[r][2], Jr[2][2]);
            }
            acc.H[0 * 6 + 0] += A[0][0]; acc.H[0 * 6 + 1] += A[0][1]; acc.H[0 * 6 + 2] += A[0][2]; acc.H[1 * 6 + 1] += A[1][1]; acc.H[1 * 6 + 2] += A[1][2]; acc.H[2 * 6 + 2] += A[2][2];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int c = 0; c < 3; ++c) acc.H[i * 6 + 3 + c] += AJ[i][c];
            const double PH[6][3] = {{0, xh[0], xh[1]}, {0, xh[2], xh[3]}, {0, xh[4], xh[5]}, {xh[6], xh[7], xh[8]}, {xh[9], xh[10], xh[11]}, {xh[12], xh[13], xh[14]}};
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = i; j < 3; ++j) {
                    const int ph = (i == 0) ? j : (i == 1 ? j + 2 : 5);
                    const double jaj = (i == 0) ? fdot3d_z(Jr[1][0], AJ[1][j], Jr[2][0], AJ[2][j]) : fdot3d(Jr[0][i], AJ[0][j], Jr[1][i], AJ[1][j], Jr[2][i], AJ[2][j]);
                    const double wph = (ph < 3) ? fdot3d_z(w[1], PH[ph][1], w[2], PH[ph][2]) : fdot3d(w[0], PH[ph][0], w[1], PH[ph][1], w[2], PH[ph][2]);
                    acc.H[(3 + i) * 6 + 3 + j] += jaj + wph;
                }
        }
    }

    // ---- workgroup reduction: the float variants' tree (the Hessian's upper triangle is mirrored on the way) ----------------------------
    constexpr int kVals = MODE == 0 ? kNdtAccum : (MODE == 1 ? 8 : 37);
    double vals[kVals];
    int    keys[kVals];
#pragma unroll
    for (int n = 0; n < kVals; ++n) {
        const int k = (MODE == 1 && n == 7) ? kNdtNbIndex : (MODE == 2 ? 7 + n : n);
        keys[n] = k;
        if (k == 0)                vals[n] = acc.score;
        else if (k < 7)            vals[n] = acc.g[k - 1];
        else if (k < kNdtNbIndex)  vals[n] = ((k - 7) / 6 > (k - 7) % 6) ? acc.H[((k - 7) % 6) * 6 + (k - 7) / 6] : acc.H[k - 7];
        else                       vals[n] = static_cast<double>(nb_mine);
    }
    double total_v;
    int    total_k;
    wave_sum_fold<kVals, 32>(vals, keys, total_v, total_k);
    s_red[wave_id()][total_k] = total_v;
    __syncthreads();
    if (threadIdx.x < kNdtPartialStride) {
        const int  k = threadIdx.x;
        const bool skip = k >= kNdtAccum || (MODE == 1 && k >= 7 && k < kNdtNbIndex) || (MODE == 2 && k < 7);
        double     r = 0.0;
        if (!skip) r = ((s_red[0][k] + s_red[1][k]) + s_red[2][k]) + s_red[3][k];
        partials[(size_t)(part_off + item_in_pair) * kNdtPartialStride + k] = r;
    }
}

// ---- NDT_OMP in the REFERENCE's summation order (opt-in: MRGFE_NDT_REFERENCE_ORDER=1 / mrgfe_dbg_set_ndt_reference_order) ------------------
// The default kernels above add the per-pair terms in a tree: every sum differs from the reference's by rounding (1e-16 relative), which an
// optimisation that does not settle can amplify past the 1e-4 bar.  ndt_omp's own order is
//   computeDerivatives:  per POINT the terms of its voxels are added from zero in neighbourhood order (scores[i], score_gradients[i],
//                        hessians[i]), then those per-point sums are added point after point ("invariant against the summing up order");
//   computeHessian:      one thread, hess(i, j) += e * (...) pair after pair, in PCL's f64 association.
// Both are chains of dependent f64 additions over the whole cloud: they cannot be split without changing the roundings.  So the work is
// cut in two kernels: ndt_ref_records_kernel computes, fully parallel, what each step of the chain adds — 44 doubles per point, or 37 per
// (point, voxel) pair for the f64 Hessian — and ndt_ref_chain_kernel walks them in order, one lane per accumulator (a wavefront per
// evaluation).  ~8 cycles per dependent f64 add: 0.45 ms per 130k-point evaluation, 1.9 ms per f64 Hessian pass (570k pairs), whatever the
// batch size; the records are 46 MB / 270 MB per evaluation.  Same per-pair float terms as everywhere else (pair_float), hence results
// bit-identical to the reference-order oracle (oracle/ndt.cpp compute_derivatives_impl<true> / compute_hessian_impl<true>).
template <int NNB>
__device__ __forceinline__ uint32_t ndt_point_neighbours(const NdtGridDev& g, bool kdtree, const float xt[3], int32_t (&ids)[NNB])
{
    const float leaf = g.leaf_size;
    int ijk[3];
    if (leaf == 1.0f) { ijk[0] = static_cast<int>(floorf(xt[0])); ijk[1] = static_cast<int>(floorf(xt[1])); ijk[2] = static_cast<int>(floorf(xt[2])); }
    else              { ijk[0] = static_cast<int>(floorf(xt[0] / leaf)); ijk[1] = static_cast<int>(floorf(xt[1] / leaf)); ijk[2] = static_cast<int>(floorf(xt[2] / leaf)); }
    bool in_box[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int o = -1; o <= 1; ++o) in_box[a][o + 1] = ijk[a] + o >= g.min_b[a] && ijk[a] + o <= g.max_b[a];
    const uint32_t mul[3] = {static_cast<uint32_t>(g.divb_mul[0]), static_cast<uint32_t>(g.divb_mul[1]), static_cast<uint32_t>(g.divb_mul[2])};
    const uint32_t key0 = (static_cast<uint32_t>(ijk[0]) - static_cast<uint32_t>(g.min_b[0])) * mul[0] + (static_cast<uint32_t>(ijk[1]) - static_cast<uint32_t>(g.min_b[1])) * mul[1] +
                          (static_cast<uint32_t>(ijk[2]) - static_cast<uint32_t>(g.min_b[2])) * mul[2];
    uint32_t cnt = 0;
    float    dist[NNB == 27 ? 27 : 1];
#pragma unroll
    for (int n = 0; n < NNB; ++n) {
        int o0, o1, o2;
        if (NNB == 27) { o0 = n / 9 - 1; o1 = (n / 3) % 3 - 1; o2 = n % 3 - 1; }
        else if (NNB == 7) { o0 = (n == 1) - (n == 2); o1 = (n == 3) - (n == 4); o2 = (n == 5) - (n == 6); }
        else           { o0 = o1 = o2 = 0; }
        const bool     ok = in_box[0][o0 + 1] && in_box[1][o1 + 1] && in_box[2][o2 + 1];
        const uint32_t key = key0 + static_cast<uint32_t>(o0) * mul[0] + static_cast<uint32_t>(o1) * mul[1] + static_cast<uint32_t>(o2) * mul[2];
        int32_t id = -1;
        if (ok) id = g.dense ? as_global(static_cast<const int32_t*>(g.lookup))[key] : ndt_lookup(g, key);
        if (id >= 0 && static_cast<uint32_t>(id) >= g.n_leaves) id = -1;  // cannot happen; keeps a corrupted table entry from faulting the GPU
        if (kdtree && id >= 0) {  // radiusSearch(point, resolution) over voxel centroids: FLANN keeps dist^2 < r^2
            const float4 c = load_point(g.centroid + id);
            const float  dx = c.x - xt[0], dy = c.y - xt[1], dz = c.z - xt[2];
            const float  d = dot3f(dx, dx, dy, dy, dz, dz);
            if (!(d < leaf * leaf)) id = -1;
            else if (NNB == 27) dist[n] = d;
        }
        ids[n] = id;
        cnt += id >= 0 ? 1u : 0u;
    }
    if (NNB == 27 && kdtree && cnt > 1) {
        // the kd-tree hands its hits over sorted by (distance, index): that is the order the reference adds a point's voxels in.  Compact, then an
        // insertion sort over the few hits (private arrays with runtime indices: scratch memory — this path is the opt-in reference-order mode only)
        uint32_t m = 0;
        for (int n = 0; n < NNB; ++n)
            if (ids[n] >= 0) { ids[m] = ids[n]; dist[m] = dist[n]; ++m; }
        for (uint32_t n = m; n < NNB; ++n) ids[n] = -1;
        for (uint32_t a = 1; a < m; ++a) {
            const float   da = dist[a];
            const int32_t ia = ids[a];
            uint32_t b = a;
            while (b > 0 && (dist[b - 1] > da || (dist[b - 1] == da && ids[b - 1] > ia))) { dist[b] = dist[b - 1]; ids[b] = ids[b - 1]; --b; }
            dist[b] = da;
            ids[b] = ia;
        }
    }
    return cnt;
}

// rec layout of one job (doubles), TILE-major so that the chain kernel streams it front to back (column-major over the whole cloud put 44 rows a megabyte
// apart into every tile: a TLB miss per row and tile):
//   mode 0 / 1:  tiles of kChainTile points;  rec[(tile * 44 + k) * kChainTile + i % kChainTile], k = 0 score, 1..6 gradient, 7..42 Hessian (mode 0 only),
//                43 the point's neighbour count
//   mode 2:      tiles of P = ref_tile_points(NNB) points, S = P * NNB slots per row;  rec[(tile * 38 + k) * S + s], k = 0 the pair's weight e, 1..36 the bracket of
//                hess(a, b); s = 0 .. tot - 1: the tile's CONTRIBUTING (point, voxel) pairs in (point, neighbourhood) order — the record kernel compacts them
//                (a prefix over the P lanes of the tile), tot goes to the tile's header word (cnt workspace, one uint32 per tile); row 37 holds the neighbour
//                count of point i at i % P
constexpr int kChainTile = 128;   // points per tile (modes 0 / 1): two LDS buffers of 44 rows = 91 KB
__host__ __device__ constexpr int ref_tile_points(int nnb) { return nnb == 7 ? 16 : (nnb == 1 ? 64 : 4); }  // 112 / 64 / 108 slots per row
size_t ndt_ref_record_doubles(int mode, size_t n, int nnb)
{
    if (mode != 2) return (n + kChainTile - 1) / kChainTile * size_t(kChainTile) * kNdtAccum;
    const size_t P = static_cast<size_t>(ref_tile_points(nnb));
    return (n + P - 1) / P * (P * nnb) * 38;
}

template <int NNB>
__global__ __launch_bounds__(256, 2) void ndt_ref_records_kernel(const NdtGridDev* __restrict__ grids, const NdtPairDev* __restrict__ pairs, const NdtEvalDev* __restrict__ evals,
                                                                  const NdtRefJob* __restrict__ jobs, double* __restrict__ rec_base, uint8_t* __restrict__ cnt_base)
{
    const NdtRefJob job = jobs[blockIdx.y];
    const NdtPairDev pr = pairs[job.pair];
    if (blockIdx.x * 256u >= pr.n_src) return;  // (uniform: the grid is as wide as the job with the most tiles)
    const NdtEvalDev& ev = evals[job.pair];
    const NdtGridDev  g = grids[pr.grid];
    __shared__ float s_T[12];
    if (threadIdx.x < 12) s_T[threadIdx.x] = ev.T[threadIdx.x];
    __syncthreads();
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const bool     live = i < pr.n_src;
    if (!live && job.mode != 2) return;  // (a lane past the end of a mode-2 job stays: its tile's lanes compact their pairs together)
    double* __restrict__ rec = rec_base + job.rec_off;
    const float4 p = live ? load_point(pr.src + i) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    float xt[3];
    transform_point(s_T, p.x, p.y, p.z, xt[0], xt[1], xt[2]);
    int32_t ids[NNB];
#pragma unroll
    for (int nb = 0; nb < NNB; ++nb) ids[nb] = -1;
    const uint32_t cnt = live ? ndt_point_neighbours<NNB>(g, (NNB == 27) && (ev.search == MRGFE_KDTREE), xt, ids) : 0u;
    const float  gauss_d2f = static_cast<float>(ev.gauss_d2);
    const double gauss_d1 = ev.gauss_d1, gauss_d2 = ev.gauss_d2;
    if (job.mode != 2) {
        Accum pt;  // the point's own sums, from zero, in neighbourhood order (scores[i], score_gradients[i], hessians[i])
        pt.score = 0;
#pragma unroll
        for (int k = 0; k < 6; ++k) pt.g[k] = 0;
#pragma unroll
        for (int k = 0; k < 36; ++k) pt.H[k] = 0;
        if (cnt) {
            float xj[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) xj[r] = fdot3f(ev.j_ang[r][0], p.x, ev.j_ang[r][1], p.y, ev.j_ang[r][2], p.z);
            const float J3[3] = {0.0f, xj[0], xj[1]}, J4[3] = {xj[2], xj[3], xj[4]}, J5[3] = {xj[5], xj[6], xj[7]};
            float PH[6][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
            if (job.mode == 0) {
                float xh[15];
#pragma unroll
                for (int r = 0; r < 15; ++r) xh[r] = fdot3f(ev.h_ang[r][0], p.x, ev.h_ang[r][1], p.y, ev.h_ang[r][2], p.z);
                PH[0][1] = xh[0];  PH[0][2] = xh[1];
                PH[1][1] = xh[2];  PH[1][2] = xh[3];
                PH[2][1] = xh[4];  PH[2][2] = xh[5];
                PH[3][0] = xh[6];  PH[3][1] = xh[7];  PH[3][2] = xh[8];
                PH[4][0] = xh[9];  PH[4][1] = xh[10]; PH[4][2] = xh[11];
                PH[5][0] = xh[12]; PH[5][1] = xh[13]; PH[5][2] = xh[14];
            }
#pragma unroll
            for (int nb = 0; nb < NNB; ++nb) {
                if (ids[nb] < 0) continue;
                // the reference's operands: the leaf's f64 mean and ALL NINE entries of its f64 inverse covariance cast to float
                const MRGFE_GLOBAL double* __restrict__ Cg = as_global(g.icov64 + (size_t)ids[nb] * 9);
                const MRGFE_GLOBAL double* __restrict__ mg = as_global(g.leaves[ids[nb]].mean);
                const float  C[3][3] = {{static_cast<float>(Cg[0]), static_cast<float>(Cg[1]), static_cast<float>(Cg[2])},
                                        {static_cast<float>(Cg[3]), static_cast<float>(Cg[4]), static_cast<float>(Cg[5])},
                                        {static_cast<float>(Cg[6]), static_cast<float>(Cg[7]), static_cast<float>(Cg[8])}};
                const double mean[3] = {mg[0], mg[1], mg[2]};
                if (job.mode == 0) pair_float_c<true, true>(pt, mean, C, xt, J3, J4, J5, PH, gauss_d2f, gauss_d1);
                else               pair_float_c<false, true>(pt, mean, C, xt, J3, J4, J5, PH, gauss_d2f, gauss_d1);
            }
        }
        double* __restrict__ tile_rec = rec + (size_t)(i / kChainTile) * (kNdtAccum * kChainTile) + i % kChainTile;
        tile_rec[0] = pt.score;
#pragma unroll
        for (int k = 0; k < 6; ++k) tile_rec[(1 + k) * kChainTile] = pt.g[k];
        if (job.mode == 0) {
#pragma unroll
            for (int k = 0; k < 36; ++k) tile_rec[(7 + k) * kChainTile] = pt.H[k];
        }
        tile_rec[kNdtNbIndex * kChainTile] = static_cast<double>(cnt);
        return;
    }
    // ---- computeHessian (f64, PCL's 3x6 / 18x6 point derivative forms), per pair the bracket the reference multiplies by e_x_cov_x ------------
    constexpr int kP = ref_tile_points(NNB), kS = kP * NNB;
    // the pair's weight, or false when the reference's range check drops the pair
    auto weight = [&](uint32_t lid, double (&C)[9], double (&q)[3], double& e) -> bool {
        const MRGFE_GLOBAL double* __restrict__ Cg = as_global(g.icov64 + (size_t)lid * 9);
        const MRGFE_GLOBAL double* __restrict__ mean = as_global(g.leaves[lid].mean);
#pragma unroll
        for (int t = 0; t < 9; ++t) C[t] = Cg[t];
        q[0] = static_cast<double>(xt[0]) - mean[0]; q[1] = static_cast<double>(xt[1]) - mean[1]; q[2] = static_cast<double>(xt[2]) - mean[2];
        double Cq[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) Cq[r] = fdot3d(C[r * 3 + 0], q[0], C[r * 3 + 1], q[1], C[r * 3 + 2], q[2]);
        e = gauss_d2 * glibc_exp(-gauss_d2 * fdot3d(q[0], Cq[0], q[1], Cq[1], q[2], Cq[2]) / 2);
        if (e > 1 || e < 0 || e != e) return false;
        e *= gauss_d1;
        return true;
    };
    // pass 1: how many of the point's pairs contribute; then the tile's P lanes agree on where each point's pairs go (inclusive scan over the lane group)
    uint32_t used = 0;
#pragma unroll 1
    for (int nb = 0; nb < NNB; ++nb) {
        if (ids[nb] < 0) continue;
        double C[9], q[3], e;
        used += weight(static_cast<uint32_t>(ids[nb]), C, q, e) ? 1u : 0u;
    }
    uint32_t incl = used;
#pragma unroll
    for (int d = 1; d < kP; d <<= 1) {
        const uint32_t up = __shfl_up(incl, d, kP);
        if ((threadIdx.x & (kP - 1)) >= static_cast<uint32_t>(d)) incl += up;
    }
    const uint32_t off = incl - used, tot = __shfl(incl, kP - 1, kP);
    double* __restrict__ tile_rec = rec + (size_t)(i / kP) * (38 * kS);
    // (a tile whose first point lies past the end of the cloud does not exist: nothing of it is written; dead lanes of the last tile write nothing either)
    if ((threadIdx.x & (kP - 1)) == 0 && live) reinterpret_cast<uint32_t*>(cnt_base + job.cnt_off)[i / kP] = tot;
    if (live) tile_rec[37 * kS + (i % kP)] = static_cast<double>(cnt);
    if (used) {
        // pass 2: the brackets, written behind the pairs of the tile's earlier points
        const double x[3] = {p.x, p.y, p.z};
        double xjd[8], xhd[15];
#pragma unroll
        for (int r = 0; r < 8; ++r) xjd[r] = fdot3d(x[0], ev.j_ang_d[r][0], x[1], ev.j_ang_d[r][1], x[2], ev.j_ang_d[r][2]);
#pragma unroll
        for (int r = 0; r < 15; ++r) xhd[r] = fdot3d(x[0], ev.h_ang_d[r][0], x[1], ev.h_ang_d[r][1], x[2], ev.h_ang_d[r][2]);
        // J: 3 x 6, identity | columns 3..5;  PH(i, j), i, j >= 3: a b c / b d e / c e f
        const double J[3][6] = {{1, 0, 0, 0, xjd[2], xjd[5]}, {0, 1, 0, xjd[0], xjd[3], xjd[6]}, {0, 0, 1, xjd[1], xjd[4], xjd[7]}};
        const double PHd[6][3] = {{0, xhd[0], xhd[1]}, {0, xhd[2], xhd[3]}, {0, xhd[4], xhd[5]}, {xhd[6], xhd[7], xhd[8]}, {xhd[9], xhd[10], xhd[11]}, {xhd[12], xhd[13], xhd[14]}};
        uint32_t slot = off;
#pragma unroll 1
        for (int nb = 0; nb < NNB; ++nb) {
            if (ids[nb] < 0) continue;
            double C[9], q[3], e;
            if (!weight(static_cast<uint32_t>(ids[nb]), C, q, e)) continue;
            tile_rec[slot] = e;
            double CJ[6][3];  // C * J(:, c)
#pragma unroll
            for (int c = 0; c < 6; ++c)
#pragma unroll
                for (int r = 0; r < 3; ++r) CJ[c][r] = fdot3d(C[r * 3 + 0], J[0][c], C[r * 3 + 1], J[1][c], C[r * 3 + 2], J[2][c]);
            double qCJ[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) qCJ[c] = fdot3d(q[0], CJ[c][0], q[1], CJ[c][1], q[2], CJ[c][2]);
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                const double t0 = -gauss_d2 * qCJ[a];
#pragma unroll
                for (int b = 0; b < 6; ++b) {
                    double qCH = 0.0;  // q . (C * 0) = +0
                    if (a >= 3 && b >= 3) {
                        const int lo = a < b ? a : b, hi = a < b ? b : a;
                        const int ph = (lo == 3) ? (hi - 3) : (lo == 4 ? (hi - 4 + 3) : 5);
                        double CH[3];
#pragma unroll
                        for (int r = 0; r < 3; ++r) CH[r] = fdot3d(C[r * 3 + 0], PHd[ph][0], C[r * 3 + 1], PHd[ph][1], C[r * 3 + 2], PHd[ph][2]);
                        qCH = fdot3d(q[0], CH[0], q[1], CH[1], q[2], CH[2]);
                    }
                    const double jtcj = fdot3d(J[0][b], CJ[a][0], J[1][b], CJ[a][1], J[2][b], CJ[a][2]);
                    tile_rec[(1 + a * 6 + b) * kS + slot] = __builtin_fma(t0, qCJ[b], qCH) + jtcj;
                }
            }
            ++slot;
        }
    }
}

// One workgroup of eight wavefronts per job.  The additions of an accumulator are a chain, so ONE wavefront adds — wavefront 0, lane k owns accumulator k —
// and it does nothing else: what it must never do is wait for memory or spend issue slots on address arithmetic (a load per step on the chain was 0.5 s per
// evaluation; a 16-deep register prefetch of the lane's own column, 44 cache lines per load instruction, 14 ms; four wavefronts that all fetched, staged AND —
// the first of them — added, 1.7 ms per 130k points: the adding wavefront spent two thirds of a tile on its share of the copying).  The other seven wavefronts
// feed it: coalesced loads of the tile-major records TWO tiles ahead (two register sets, offsets computed once), staged into the LDS buffer the adder is not
// reading.  Lane k walks row k of the buffer (odd row length in 8-byte words: conflict-free), eight values into registers ahead of their additions.
constexpr int kChainThreads = 512, kFeeders = kChainThreads - 64;
template <bool WEIGHTED>
__device__ __forceinline__ double ref_chain_row(double acc, const double* __restrict__ row, const double* __restrict__ wrow, int count8)
{
    // `count8` (a multiple of eight) values of row (and their weights) in order: eight LDS reads, then the eight dependent operations.  No predicate on the
    // chain: what lies past the real count is zero (+0 terms, or 0 * 0 under the fused multiply-add: both leave a sum that is never -0 unchanged).  Sixteen-byte
    // reads with the next group in flight behind the additions (two or four register sets) were measured: 18 ms per alignment against 14 ms for this.
#pragma unroll 2
    for (int h = 0; h < count8; h += 8) {
        double v[8], w[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            v[j] = row[h + j];
            if (WEIGHTED) w[j] = wrow[h + j];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc = WEIGHTED ? __builtin_fma(w[j], v[j], acc) : acc + v[j];
    }
    return acc;
}

__global__ __launch_bounds__(kChainThreads) void ndt_ref_chain01_kernel(const NdtPairDev* __restrict__ pairs, const NdtRefJob* __restrict__ jobs, const double* __restrict__ rec_base,
                                                                         double* __restrict__ results)
{
    constexpr int T = kChainTile, kRow = T + 1;  // (odd row length in 8-byte words: the lanes' rows start in different banks)
    static_assert(T % 8 == 0, "tile");
    constexpr int kPer = (kNdtAccum * T + kFeeders - 1) / kFeeders;  // 13 elements of a tile per feeder thread
    __shared__ double s_v[2][kNdtAccum * kRow];
    const NdtRefJob job = jobs[blockIdx.x];
    if (job.mode == 2) return;  // (the f64 Hessian jobs of the round: ndt_ref_chain2_kernel)
    const uint32_t n = pairs[job.pair].n_src;
    const MRGFE_GLOBAL double* __restrict__ rec = as_global(rec_base + job.rec_off);
    const int  tid = threadIdx.x, f = tid - 64;  // f >= 0: a feeder
    const bool adder = tid < 64;
    // columns this kind delivers: mode 0 all 44 (score, gradient, Hessian, count), mode 1 score + gradient + count (its rows 0..7 = columns 0..6, 43)
    const int      n_cols = job.mode == 0 ? kNdtAccum : 8;
    const uint32_t n_tiles = (n + T - 1) / T;
    // what a feeder fetches is the same in every tile: element e = (row c, position j) -> offset in the tile's block / in the LDS buffer
    int  goff[kPer], loff[kPer], jpos[kPer];
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
        const int e = f + u * kFeeders, c = e / T, j = e % T;
        const bool mine = !adder && c < n_cols;
        goff[u] = mine ? ((job.mode == 0) ? c : (c < 7 ? c : kNdtNbIndex)) * T + j : -1;
        loff[u] = c * kRow + j;
        jpos[u] = j;
    }
    double pa[kPer], pb[kPer];
    double acc = 0.0;
    // A position past the end of the cloud is padding the record kernel never wrote: it reads +0, which is exact to add (the sums start at +0, never -0)
    auto fetch = [&](uint32_t tile, double (&pre)[kPer]) {
        if (tile >= n_tiles) return;
        const MRGFE_GLOBAL double* __restrict__ t0 = rec + (size_t)tile * (kNdtAccum * T);
        const uint32_t left = n - tile * T;  // points of this tile that exist (>= T except in the last one)
#pragma unroll
        for (int u = 0; u < kPer; ++u) pre[u] = (goff[u] >= 0 && static_cast<uint32_t>(jpos[u]) < left) ? t0[goff[u]] : 0.0;
    };
    auto stage = [&](int buf, const double (&pre)[kPer]) {
#pragma unroll
        for (int u = 0; u < kPer; ++u)
            if (goff[u] >= 0) s_v[buf][loff[u]] = pre[u];
    };
    // LDS holds `tile`; `next` holds tile + 1 (fetched a whole step ago); `after` receives tile + 2
    auto step = [&](uint32_t tile, const double (&next)[kPer], double (&after)[kPer]) {
        const int buf = static_cast<int>(tile & 1u);
        if (adder) {
            if (tid < n_cols) acc = ref_chain_row<false>(acc, &s_v[buf][tid * kRow], nullptr, T);
        } else {
            fetch(tile + 2, after);
            if (tile + 1 < n_tiles) stage(buf ^ 1, next);
        }
        __syncthreads();
    };
    if (!adder) { fetch(0, pa); stage(0, pa); fetch(1, pa); }
    __syncthreads();
    for (uint32_t tile = 0; tile < n_tiles; tile += 2) {
        step(tile, pa, pb);
        if (tile + 1 < n_tiles) step(tile + 1, pb, pa);
    }
    // the record of ndt_reduce_kernel<false>: score, gradient, Hessian (row-major), neighbour count at slot 43 (mode 1 carries it in its eighth row)
    if (tid < kNdtPartialStride) {
        double o = 0.0;
        if (job.mode == 0) o = tid < kNdtAccum ? acc : 0.0;
        else if (tid < 7)  o = acc;
        results[(size_t)job.pair * kNdtPartialStride + tid] = o;
    }
    if (job.mode == 1) {
        __syncthreads();
        if (tid == 7) results[(size_t)job.pair * kNdtPartialStride + kNdtNbIndex] = acc;
    }
}

// hess(a, b) = fma(e, bracket, hess(a, b)) over the contributing pairs, point after point; lanes 0..35 = entry a * 6 + b, lane 36 the neighbour count
template <int NNB>
__global__ __launch_bounds__(kChainThreads) void ndt_ref_chain2_kernel(const NdtPairDev* __restrict__ pairs, const NdtRefJob* __restrict__ jobs, const double* __restrict__ rec_base,
                                                                        const uint8_t* __restrict__ cnt_base, double* __restrict__ results)
{
    constexpr int P = ref_tile_points(NNB), S = P * NNB, kRow = (S + 8) | 1;  // rows are read in whole groups of eight; odd length
    constexpr int kPer = (38 * S + kFeeders - 1) / kFeeders;
    __shared__ double   s_v[2][38 * kRow];
    __shared__ uint32_t s_tot[2];
    const NdtRefJob job = jobs[blockIdx.x];
    if (job.mode != 2) return;
    const uint32_t n = pairs[job.pair].n_src;
    const MRGFE_GLOBAL double* __restrict__ rec = as_global(rec_base + job.rec_off);
    const MRGFE_GLOBAL uint32_t* __restrict__ tile_tot = as_global(reinterpret_cast<const uint32_t*>(cnt_base + job.cnt_off));
    const int  tid = threadIdx.x, f = tid - 64;
    const bool adder = tid < 64;
    const uint32_t n_tiles = (n + P - 1) / P;
    // element e of a tile's block = (row c, slot j); rows 0..36: the tile's compacted pairs (what lies behind them was never written: zeros are staged
    // there up to the next multiple of eight); row 37: one neighbour count per point
    int goff[kPer], loff[kPer], jpos[kPer], rowk[kPer];
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
        const int e = f + u * kFeeders, c = e / S, j = e % S;
        goff[u] = (!adder && c < 38) ? e : -1;
        loff[u] = c * kRow + j;
        jpos[u] = j;
        rowk[u] = c;
    }
    double   pa[kPer], pb[kPer];
    uint32_t ta = 0, tb = 0;
    double   acc = 0.0;
    auto fetch = [&](uint32_t tile, double (&pre)[kPer], uint32_t& tot) {
        tot = 0;
        if (tile >= n_tiles) return;
        const MRGFE_GLOBAL double* __restrict__ t0 = rec + (size_t)tile * (38 * S);
        tot = tile_tot[tile];  // (every feeder reads the same word: it bounds what it fetches)
        const uint32_t left = n - tile * P;
        // predicated loads: most of a tile's S slots per row lie behind its pairs and were never written (loading all of them and selecting afterwards
        // was measured: 17.6 ms per alignment against 14.2 ms)
#pragma unroll
        for (int u = 0; u < kPer; ++u) {
            const bool want = goff[u] >= 0 && (rowk[u] < 37 ? static_cast<uint32_t>(jpos[u]) < tot : (jpos[u] < P && static_cast<uint32_t>(jpos[u]) < left));
            pre[u] = want ? t0[goff[u]] : 0.0;
        }
    };
    auto stage = [&](int buf, const double (&pre)[kPer], uint32_t tot) {
        const uint32_t tot8 = (tot + 7u) & ~7u;
#pragma unroll
        for (int u = 0; u < kPer; ++u) {
            // everything up to the next multiple of eight behind the pairs is written (the zeros fetch left there): the adder reads whole groups
            if (goff[u] >= 0 && (rowk[u] < 37 ? static_cast<uint32_t>(jpos[u]) < tot8 : jpos[u] < ((P + 7) & ~7))) s_v[buf][loff[u]] = pre[u];
        }
        if (S % 8 != 0) {  // a row length that is no multiple of eight: the last group runs past the S fetched entries
            for (int t = f; t < 38 * 8; t += kFeeders) s_v[buf][(t / 8) * kRow + S + t % 8] = 0.0;
        }
        if (f == 0) s_tot[buf] = tot;
    };
    auto step = [&](uint32_t tile, const double (&next)[kPer], uint32_t next_tot, double (&after)[kPer], uint32_t& after_tot) {
        const int buf = static_cast<int>(tile & 1u);
        if (adder) {
            if (tid < 36)       acc = ref_chain_row<true>(acc, &s_v[buf][(1 + tid) * kRow], &s_v[buf][0], static_cast<int>((s_tot[buf] + 7u) & ~7u));
            else if (tid == 36) acc = ref_chain_row<false>(acc, &s_v[buf][37 * kRow], nullptr, (P + 7) & ~7);  // neighbour counts (integers: exact)
        } else {
            fetch(tile + 2, after, after_tot);
            if (tile + 1 < n_tiles) stage(buf ^ 1, next, next_tot);
        }
        __syncthreads();
    };
    if (!adder) { fetch(0, pa, ta); stage(0, pa, ta); fetch(1, pa, ta); }
    __syncthreads();
    for (uint32_t tile = 0; tile < n_tiles; tile += 2) {
        step(tile, pa, ta, pb, tb);
        if (tile + 1 < n_tiles) step(tile + 1, pb, tb, pa, ta);
    }
    if (tid < kNdtPartialStride) results[(size_t)job.pair * kNdtPartialStride + tid] = 0.0;  // (score / gradient slots: this kind delivers none)
    __syncthreads();
    if (tid < 36) results[(size_t)job.pair * kNdtPartialStride + 7 + tid] = acc;
    if (tid == 36) results[(size_t)job.pair * kNdtPartialStride + kNdtNbIndex] = acc;
}

int ndt_launch_ref_round(mrgfe_ctx* ctx, int search, const NdtGridDev* d_grids, const NdtPairDev* d_pairs, const NdtEvalDev* d_evals, const NdtRefJob* d_jobs, uint32_t n_jobs,
                         uint32_t max_tiles, double* d_rec, uint8_t* d_cnt, double* results, bool any_mode01, bool any_mode2)
{
    if (n_jobs == 0 || max_tiles == 0) return MRGFE_OK;
    const dim3 grid(max_tiles, n_jobs);
    if (search == MRGFE_DIRECT7)      hipLaunchKernelGGL((ndt_ref_records_kernel<7>), grid, dim3(256), 0, ctx->stream, d_grids, d_pairs, d_evals, d_jobs, d_rec, d_cnt);
    else if (search == MRGFE_DIRECT1) hipLaunchKernelGGL((ndt_ref_records_kernel<1>), grid, dim3(256), 0, ctx->stream, d_grids, d_pairs, d_evals, d_jobs, d_rec, d_cnt);
    else                              hipLaunchKernelGGL((ndt_ref_records_kernel<27>), grid, dim3(256), 0, ctx->stream, d_grids, d_pairs, d_evals, d_jobs, d_rec, d_cnt);
    // a workgroup per job in both chain kernels; a job of the other kind returns at once
    if (any_mode01) hipLaunchKernelGGL(ndt_ref_chain01_kernel, dim3(n_jobs), dim3(kChainThreads), 0, ctx->stream, d_pairs, d_jobs, d_rec, results);
    if (any_mode2) {
        if (search == MRGFE_DIRECT7)      hipLaunchKernelGGL((ndt_ref_chain2_kernel<7>), dim3(n_jobs), dim3(kChainThreads), 0, ctx->stream, d_pairs, d_jobs, d_rec, d_cnt, results);
        else if (search == MRGFE_DIRECT1) hipLaunchKernelGGL((ndt_ref_chain2_kernel<1>), dim3(n_jobs), dim3(kChainThreads), 0, ctx->stream, d_pairs, d_jobs, d_rec, d_cnt, results);
        else                              hipLaunchKernelGGL((ndt_ref_chain2_kernel<27>), dim3(n_jobs), dim3(kChainThreads), 0, ctx->stream, d_pairs, d_jobs, d_rec, d_cnt, results);
    }
    MRGFE_HIP_CHECK(hipGetLastError());
    return MRGFE_OK;
}

// pair and position within the pair of item `item` of variant `mode`: the last busy pair whose first item is <= item
// (uniform over the workgroup: scalar loads)
__device__ __forceinline__ void ndt_plan_find(const uint32_t* __restrict__ plan, uint32_t n_all_pairs, int mode, uint32_t n_busy, uint32_t item, uint32_t& pi, uint32_t& item_in_pair)
{
    const uint32_t* __restrict__ pair_of = plan + ndt_plan_pair_off(n_all_pairs, mode);
    const uint32_t* __restrict__ item_start = plan + ndt_plan_start_off(n_all_pairs, mode);
    uint32_t lo = 0, hi = n_busy;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (item_start[mid] <= item) lo = mid; else hi = mid;
    }
    pi = pair_of[lo];
    item_in_pair = item - item_start[lo];
}

// One variant per launch (MRGFE_FUSED=0; kept to hold the fused launch against).
template <int MODE, int NNB>
__global__ __launch_bounds__(256, ((MODE == 0 && NNB <= 7) ? NDT_MODE0_WAVES : ((MODE == 2 && NNB <= 7) ? NDT_MODE2_WAVES : 2))) void ndt_derivatives_kernel(const NdtGridDev* __restrict__ grids, const NdtPairDev* __restrict__ pairs,
                                                               const NdtEvalDev* __restrict__ evals, const uint32_t* __restrict__ plan, uint32_t n_all_pairs,
                                                               double* __restrict__ partials)
{
    // the round's plan (ndt_plan_kernel): this variant's busy pairs and the prefix of their work items
    const NdtPlanHead& head = *reinterpret_cast<const NdtPlanHead*>(plan);
    const uint32_t n_items = head.n_items[MODE];
    if (blockIdx.x >= n_items) return;
    const uint32_t ppt = head.ppt[MODE], n_busy = head.n_pairs[MODE];
    __shared__ NdtDerivShared<NNB> sh;
    for (uint32_t item = blockIdx.x; item < n_items; item += gridDim.x) {
        uint32_t pi, item_in_pair;
        ndt_plan_find(plan, n_all_pairs, MODE, n_busy, item, pi, item_in_pair);
        ndt_derivatives_item<MODE, NNB>(sh, grids, pairs, evals, pi, item_in_pair, ppt, partials);
    }
}

// The fixed-order sum of a pair's item records into its result record (ndt_reduce_kernel below; ndt_derivatives_single_kernel): 256 threads, four interleaved
// slices per slot, ((s0 + s1) + s2) + s3.  `s`: 4 x kNdtPartialStride doubles of LDS.  CONTROL keeps the sums in s_r instead of `results`.
template <bool CONTROL = false>
__device__ __forceinline__ void ndt_sum_records(const NdtPairDev& pr, uint32_t mode, const NdtPlanHead& head, const double* __restrict__ partials, double (&s)[4][kNdtPartialStride],
                                                double* __restrict__ results, uint32_t pair_index, double* __restrict__ s_r = nullptr)
{
    // items (= partial records) the derivative launch of this pair's kernel variant used
    const uint32_t per_item = 256u * head.ppt[mode];
    const uint32_t nblk = (pr.n_src + per_item - 1) / per_item;
    const int k = threadIdx.x & 63, slice = threadIdx.x >> 6;
    if (k < kNdtPartialStride) {
        // the additions stay in order; 32 (then eight) loads are in flight ahead of them (a straggler round has one pair with 500
        // records, and a load-add-load-add chain over them took longer than the derivative kernel it follows)
        const double* col = partials + (size_t)pr.part_off * kNdtPartialStride + k;
        double   acc = 0.0;
        uint32_t b = slice;
        for (; b + 124 < nblk; b += 128) {
            double v[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) v[u] = col[(size_t)(b + 4 * u) * kNdtPartialStride];
#pragma unroll
            for (int u = 0; u < 32; ++u) acc += v[u];
        }
        for (; b + 28 < nblk; b += 32) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = col[(size_t)(b + 4 * u) * kNdtPartialStride];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
        for (; b < nblk; b += 4) acc += col[(size_t)b * kNdtPartialStride];
        s[slice][k] = acc;
    }
    __syncthreads();
    if (threadIdx.x < kNdtPartialStride) {
        const double r = ((s[0][threadIdx.x] + s[1][threadIdx.x]) + s[2][threadIdx.x]) + s[3][threadIdx.x];
        if (CONTROL) s_r[threadIdx.x] = r;
        else         results[(size_t)pair_index * kNdtPartialStride + threadIdx.x] = r;
    }
}

// All three variants in ONE launch per round.  The items of the variants are interleaved in proportion to their counts
// (two nested Bresenham splits: variant 0 against the rest, then 2 against 1), so that the f64 Hessian items — latency-bound,
// VALU busy 0.41 on their own — and the cheap score+gradient items share the CUs with the VALU-bound items of variant 0
// instead of each variant paying its own launch, its own tail and a queue gap.
// All three variants in ONE launch per round: the workgroups walk the items of variant 0, then those of variant 2, then those of
// variant 1 (heaviest first), one loop per variant — a variant's tail is filled by the next variant's items instead of idling
// until its own launch has drained, and a round costs one queue gap instead of three.
template <int MODE, int NNB>
__device__ __forceinline__ void ndt_derivatives_walk(NdtDerivShared<NNB>& sh, uint32_t first, const NdtGridDev* __restrict__ grids, const NdtPairDev* __restrict__ pairs,
                                                     const NdtEvalDev* __restrict__ evals, const uint32_t* __restrict__ plan, uint32_t n_all_pairs, double* __restrict__ partials)
{
    const NdtPlanHead& head = *reinterpret_cast<const NdtPlanHead*>(plan);
    const uint32_t n_items = head.n_items[MODE], ppt = head.ppt[MODE], n_busy = head.n_pairs[MODE];
    for (uint32_t item = first; item < n_items; item += gridDim.x) {
        uint32_t pi, item_in_pair;
        ndt_plan_find(plan, n_all_pairs, MODE, n_busy, item, pi, item_in_pair);
        ndt_derivatives_item<MODE, NNB>(sh, grids, pairs, evals, pi, item_in_pair, ppt, partials);
    }
}

template <int NNB>
__global__ __launch_bounds__(256, NNB <= 7 ? NDT_MODE0_WAVES : 2) void ndt_derivatives_all_kernel(const NdtGridDev* __restrict__ grids, const NdtPairDev* __restrict__ pairs,
                                                               const NdtEvalDev* __restrict__ evals, const uint32_t* __restrict__ plan, uint32_t n_all_pairs,
                                                               double* __restrict__ partials)
{
    __shared__ NdtDerivShared<NNB> sh;
    const NdtPlanHead& head = *reinterpret_cast<const NdtPlanHead*>(plan);
    // position of this workgroup in the concatenated item list 0 | 2 | 1; a workgroup past the end of a variant's items starts in
    // the next variant at the position the stride leaves it
    const uint32_t n0 = head.n_items[0], n2 = head.n_items[2], g = gridDim.x;
    uint32_t at = blockIdx.x;
    ndt_derivatives_walk<0, NNB>(sh, at, grids, pairs, evals, plan, n_all_pairs, partials);
    if (at < n0) at += (n0 - at + g - 1) / g * g;  // first position >= n0 on this workgroup's stride
    at -= n0;
    ndt_derivatives_walk<2, NNB>(sh, at, grids, pairs, evals, plan, n_all_pairs, partials);
    if (at < n2) at += (n2 - at + g - 1) / g * g;
    at -= n2;
    ndt_derivatives_walk<1, NNB>(sh, at, grids, pairs, evals, plan, n_all_pairs, partials);
}

// The round of ONE host-stepped registration in one launch: the same items, and the workgroup that finishes LAST sums the item records (ndt_reduce_kernel<false>'s
// order, its code below: ndt_sum_records) and writes record + tag into pinned host memory — a 33k-point frame's round is ~41 us of which the reduction's own
// launch and the gap in front of it were ~6.  Every workgroup's record stores are released (agent scope) by the lanes that made them before the workgroup
// takes its ticket; the last one acquires before it reads them.  No workgroup waits for another: the grid cannot hang.  (A twin of the kernel above and
// not a flag in it: the batch launches keep their registers.)
template <int NNB>
__global__ __launch_bounds__(256, NNB <= 7 ? NDT_MODE0_WAVES : 2) void ndt_derivatives_single_kernel(const NdtGridDev* __restrict__ grids, const NdtPairDev* __restrict__ pairs,
                                                                  const NdtEvalDev* __restrict__ evals, const uint32_t* __restrict__ plan, double* __restrict__ partials,
                                                                  uint32_t* __restrict__ ticket, double* __restrict__ results, double tag)
{
    __shared__ NdtDerivShared<NNB> sh;
    __shared__ uint32_t s_last;
    const NdtPlanHead& head = *reinterpret_cast<const NdtPlanHead*>(plan);
    const uint32_t n0 = head.n_items[0], n2 = head.n_items[2], g = gridDim.x;
    uint32_t at = blockIdx.x;
    ndt_derivatives_walk<0, NNB>(sh, at, grids, pairs, evals, plan, 1u, partials);
    if (at < n0) at += (n0 - at + g - 1) / g * g;
    at -= n0;
    ndt_derivatives_walk<2, NNB>(sh, at, grids, pairs, evals, plan, 1u, partials);
    if (at < n2) at += (n2 - at + g - 1) / g * g;
    at -= n2;
    ndt_derivatives_walk<1, NNB>(sh, at, grids, pairs, evals, plan, 1u, partials);
    if (threadIdx.x < kNdtPartialStride) __threadfence();  // (the lanes that stored the records)
    __syncthreads();
    if (threadIdx.x == 0) s_last = atomicAdd(ticket, 1u) == gridDim.x - 1u ? 1u : 0u;
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    if (threadIdx.x == 0) atomicExch(ticket, 0u);  // for the next round's launch (behind this one on the stream)
    if (!evals[0].active) return;
    ndt_sum_records(pairs[0], evals[0].mode, head, partials, sh.red, results, 0u);
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence_system();
        __hip_atomic_store(&results[kNdtPartialStride], tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// PCL_NDT_HIP launches: the same plan, the f64 items
template <int MODE>
__global__ __launch_bounds__(256, 2) void ndt_derivatives_f64_kernel(const NdtGridDev* __restrict__ grids, const NdtPairDev* __restrict__ pairs, const NdtEvalDev* __restrict__ evals,
                                                                      const uint32_t* __restrict__ plan, uint32_t n_all_pairs, double* __restrict__ partials)
{
    const NdtPlanHead& head = *reinterpret_cast<const NdtPlanHead*>(plan);
    const uint32_t n_items = head.n_items[MODE];
    if (blockIdx.x >= n_items) return;
    const uint32_t ppt = head.ppt[MODE], n_busy = head.n_pairs[MODE];
    __shared__ NdtDerivShared<27> sh;
    for (uint32_t item = blockIdx.x; item < n_items; item += gridDim.x) {
        uint32_t pi, item_in_pair;
        ndt_plan_find(plan, n_all_pairs, MODE, n_busy, item, pi, item_in_pair);
        ndt_derivatives_f64_item<MODE>(sh, grids, pairs, evals, pi, item_in_pair, ppt, partials);
    }
}
template <int MODE>
__device__ __forceinline__ void ndt_derivatives_f64_walk(NdtDerivShared<27>& sh, uint32_t first, const NdtGridDev* __restrict__ grids, const NdtPairDev* __restrict__ pairs,
                                                         const NdtEvalDev* __restrict__ evals, const uint32_t* __restrict__ plan, uint32_t n_all_pairs, double* __restrict__ partials)
{
    const NdtPlanHead& head = *reinterpret_cast<const NdtPlanHead*>(plan);
    const uint32_t n_items = head.n_items[MODE], ppt = head.ppt[MODE], n_busy = head.n_pairs[MODE];
    for (uint32_t item = first; item < n_items; item += gridDim.x) {
        uint32_t pi, item_in_pair;
        ndt_plan_find(plan, n_all_pairs, MODE, n_busy, item, pi, item_in_pair);
        ndt_derivatives_f64_item<MODE>(sh, grids, pairs, evals, pi, item_in_pair, ppt, partials);
    }
}
__global__ __launch_bounds__(256, 2) void ndt_derivatives_f64_all_kernel(const NdtGridDev* __restrict__ grids, const NdtPairDev* __restrict__ pairs, const NdtEvalDev* __restrict__ evals,
                                                                          const uint32_t* __restrict__ plan, uint32_t n_all_pairs, double* __restrict__ partials)
{
    __shared__ NdtDerivShared<27> sh;
    const NdtPlanHead& head = *reinterpret_cast<const NdtPlanHead*>(plan);
    const uint32_t n0 = head.n_items[0], n2 = head.n_items[2], g = gridDim.x;
    uint32_t at = blockIdx.x;
    ndt_derivatives_f64_walk<0>(sh, at, grids, pairs, evals, plan, n_all_pairs, partials);
    if (at < n0) at += (n0 - at + g - 1) / g * g;
    at -= n0;
    ndt_derivatives_f64_walk<2>(sh, at, grids, pairs, evals, plan, n_all_pairs, partials);
    if (at < n2) at += (n2 - at + g - 1) / g * g;
    at -= n2;
    ndt_derivatives_f64_walk<1>(sh, at, grids, pairs, evals, plan, n_all_pairs, partials);
}

// ---- the round's plan -------------------------------------------------------------------------------------------------
// One workgroup scans the pending requests of all P pairs: per kernel variant the busy pairs (compacted, in pair order), the
// tiles-per-item of the launch and the exclusive prefix of the pairs' item counts.  Also tells the host how many pairs are
// still running (pinned memory; the host only uses it to stop enqueueing rounds).
__global__ __launch_bounds__(256) void ndt_plan_kernel(const NdtPairDev* __restrict__ pairs, const NdtEvalDev* __restrict__ evals, uint32_t P, uint32_t* __restrict__ plan,
                                                        uint32_t wg_target, uint32_t max_ppt, uint32_t forced_ppt, uint32_t round, NdtRoundInfo* __restrict__ host_info)
{
    __shared__ uint32_t s_scan[8];
    __shared__ uint32_t s_tiles[3], s_ppt[3];
    if (threadIdx.x < 3) s_tiles[threadIdx.x] = 0;
    __syncthreads();
    // pass 1: tiles of 256 points per variant
    uint32_t my_tiles[3] = {0, 0, 0};
    for (uint32_t i = threadIdx.x; i < P; i += 256) {
        const NdtEvalDev& ev = evals[i];
        if (ev.active) my_tiles[ev.mode] += (pairs[i].n_src + 255u) / 256u;
    }
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        const uint32_t t = wave_sum(my_tiles[m]);
        if (lane_id() == 0 && t) atomicAdd(&s_tiles[m], t);
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        uint32_t ppt = s_tiles[threadIdx.x] / wg_target;
        ppt = ppt < 1u ? 1u : (ppt > max_ppt ? max_ppt : ppt);
        s_ppt[threadIdx.x] = forced_ppt ? forced_ppt : ppt;
    }
    __syncthreads();
    // pass 2: compaction and item prefix per variant, 256 pairs at a time
    uint32_t run_pairs[3] = {0, 0, 0}, run_items[3] = {0, 0, 0}, n_active = 0;
    for (uint32_t c0 = 0; c0 < P; c0 += 256) {
        const uint32_t i = c0 + threadIdx.x;
        int      mode = -1;
        uint32_t tiles = 0;
        if (i < P) {
            const NdtEvalDev& ev = evals[i];
            if (ev.active) { mode = ev.mode; tiles = (pairs[i].n_src + 255u) / 256u; }
        }
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            const bool     mine = mode == m;
            const uint32_t items = mine ? (tiles + s_ppt[m] - 1) / s_ppt[m] : 0u;
            uint32_t tot_pairs, tot_items;
            const uint32_t k = block_exclusive_scan<256>(mine ? 1u : 0u, s_scan, &tot_pairs);
            const uint32_t st = block_exclusive_scan<256>(items, s_scan, &tot_items);
            if (mine) {
                plan[ndt_plan_pair_off(P, m) + run_pairs[m] + k] = i;
                plan[ndt_plan_start_off(P, m) + run_pairs[m] + k] = run_items[m] + st;
            }
            run_pairs[m] += tot_pairs;
            run_items[m] += tot_items;
        }
    }
    if (threadIdx.x == 0) {
        NdtPlanHead h;
        for (int m = 0; m < 3; ++m) {
            h.n_pairs[m] = run_pairs[m]; h.n_items[m] = run_items[m]; h.ppt[m] = s_ppt[m];
            plan[ndt_plan_start_off(P, m) + run_pairs[m]] = run_items[m];
            n_active += run_pairs[m];
        }
        h.n_active = n_active;
        h.round = round;
        for (int k = 0; k < 5; ++k) h.pad[k] = 0;
        *reinterpret_cast<NdtPlanHead*>(plan) = h;
        if (host_info) {
            NdtRoundInfo& o = host_info[round];
            o.n_active = n_active;
            for (int m = 0; m < 3; ++m) { o.n_pairs[m] = run_pairs[m]; o.n_items[m] = run_items[m]; }
            __threadfence_system();
            __hip_atomic_store(&o.tag, round + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// ---- reduce (+ controller step) -----------------------------------------------------------------------------------------
// One workgroup per pair with a pending request: fixed-order sum of the item partials of its evaluation (4 interleaved
// slices, then slice 0..3 in order: bitwise reproducible), then
//   CONTROL = true : the pair's optimiser state (HBM) is staged in LDS, lane 0 resumes the state machine of ndt_ctl.h with the
//                    44 sums and writes the next request — the batch advances without the host;
//   CONTROL = false: the sums go to `results` (pinned host memory) for the host-stepped controller.
template <bool CONTROL>
__global__ __launch_bounds__(256) void ndt_reduce_kernel(const NdtPairDev* __restrict__ pairs, NdtEvalDev* __restrict__ evals, const double* __restrict__ partials,
                                                          const uint32_t* __restrict__ plan, double* __restrict__ results, NdtCtlState* __restrict__ states, double tag,
                                                          uint32_t* __restrict__ ticket)
{
    const NdtPairDev pr = pairs[blockIdx.x];
    const NdtEvalDev& ev = evals[blockIdx.x];
    // host-stepped batches poll for `tag` (below): the workgroup that is done LAST writes it, so every workgroup takes a ticket — also the ones of pairs
    // without a request this round
    auto done_for_host = [&]() {
        if (CONTROL || tag == 0.0) return;
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence_system();  // this workgroup's record is in host memory before its ticket counts
            const bool last = gridDim.x == 1u || atomicAdd(ticket, 1u) == gridDim.x - 1u;
            if (last) {
                if (gridDim.x > 1u) atomicExch(ticket, 0u);  // for the next round's launch
                __threadfence_system();
                __hip_atomic_store(&results[(size_t)gridDim.x * kNdtPartialStride], tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    };
    if (!ev.active) { done_for_host(); return; }
#ifdef NDT_PHASE_CLOCK
    long long rc0 = 0, rc1 = 0, rc2 = 0, rc3 = 0, rc4 = 0, rc5 = 0, rc6 = 0, rsolves = 0;
    if (CONTROL && threadIdx.x == 0) rc0 = wall_clock64();
#endif
    const NdtPlanHead& head = *reinterpret_cast<const NdtPlanHead*>(plan);
    __shared__ double s[4][kNdtPartialStride];
    __shared__ double s_r[kNdtPartialStride];
    ndt_sum_records<CONTROL>(pr, ev.mode, head, partials, s, results, blockIdx.x, s_r);
    if (!CONTROL) {
        // a host-stepped alignment polls for its records instead of waiting for the stream: the slot behind the records gets `tag` once they are ALL visible
        // (one workgroup per pair; the last one to finish writes it)
        done_for_host();
        return;
    }
    // stage the state in LDS (coalesced), step it on one lane, write it and the next request back
    constexpr int kWords = sizeof(NdtCtlState) / 8;
    static_assert(sizeof(NdtCtlState) % 8 == 0, "state is copied as 8-byte words");
    __shared__ double s_state[kWords];
    __shared__ NdtEvalDev s_eval;
    __shared__ ctl::SvdWaveScratch s_svd;
    __shared__ double s_negg[6], s_delta[6];
#ifdef NDT_PHASE_CLOCK
    if (threadIdx.x == 0) rc1 = wall_clock64();
#endif
    const double* gs = reinterpret_cast<const double*>(states + blockIdx.x);
    for (int w = threadIdx.x; w < kWords; w += 256) s_state[w] = gs[w];
    __syncthreads();
#ifdef NDT_PHASE_CLOCK
    if (threadIdx.x == 0) rc2 = wall_clock64();
#endif
    if (wave_id() == 0) {
        // lane 0 runs the state machine; the 6x6 SVD solves of its Newton steps run on the whole wavefront
        NdtCtlState& st = *reinterpret_cast<NdtCtlState*>(s_state);
        int next = 0;
        if (threadIdx.x == 0) next = ctl::resume(st, s_r);
        next = __shfl(next, 0, kWave);
#ifdef NDT_PHASE_CLOCK
        if (threadIdx.x == 0) rc3 = wall_clock64();
#endif
        while (next == ctl::CTL_NEED_SOLVE) {
#ifdef NDT_PHASE_CLOCK
            ++rsolves;
#endif
            if (threadIdx.x < 6) s_negg[threadIdx.x] = -st.g[threadIdx.x];
            ctl::svd_wave_sync();
            // the LU fast path on lane 0 (ndt_ctl.h newton_solve6); a matrix it declines goes through the wavefront SVD
            int solved = 0;
            if (threadIdx.x == 0 && !st.svd_only) solved = ctl::lu_solve6(st.H, s_negg, s_delta) ? 1 : 0;
            solved = __shfl(solved, 0, kWave);
            if (!solved) ctl::svd_solve6_wave(st.H, s_negg, s_delta, s_svd);
            else         ctl::svd_wave_sync();
            if (threadIdx.x == 0) next = ctl::after_solve(st, s_delta);
            next = __shfl(next, 0, kWave);
        }
#ifdef NDT_PHASE_CLOCK
        if (threadIdx.x == 0) rc4 = wall_clock64();
#endif
        // the six double-precision sines / cosines of the next request's angle tables: one lane each instead of six in a row
        __shared__ double s_cs[6];
        if (threadIdx.x < 6 && !ctl::done(st)) s_cs[threadIdx.x] = ctl::angle_trig(st.req_p, threadIdx.x);
        ctl::svd_wave_sync();
        if (threadIdx.x == 0) {
            s_eval.active = 0;
            ctl::fill_eval_from(st, s_cs, s_eval);
        }
    }
#ifdef NDT_PHASE_CLOCK
    if (threadIdx.x == 0) rc5 = wall_clock64();
#endif
    __syncthreads();
    double* gd = reinterpret_cast<double*>(states + blockIdx.x);
    for (int w = threadIdx.x; w < kWords; w += 256) gd[w] = s_state[w];
    constexpr int kEvalWords = sizeof(NdtEvalDev) / 8;
    static_assert(sizeof(NdtEvalDev) % 8 == 0, "request record is copied as 8-byte words");
    const double* se = reinterpret_cast<const double*>(&s_eval);
    double*       ge = reinterpret_cast<double*>(evals + blockIdx.x);
    if (s_eval.active) { for (int w = threadIdx.x; w < kEvalWords; w += 256) ge[w] = se[w]; }
    else if (threadIdx.x == 0) evals[blockIdx.x].active = 0;
#ifdef NDT_PHASE_CLOCK
    if (threadIdx.x == 0) {
        rc6 = wall_clock64();
        atomicAdd(&g_rphase[0], (unsigned long long)(rc1 - rc0)); atomicAdd(&g_rphase[1], (unsigned long long)(rc2 - rc1)); atomicAdd(&g_rphase[2], (unsigned long long)(rc3 - rc2));
        atomicAdd(&g_rphase[3], (unsigned long long)(rc4 - rc3)); atomicAdd(&g_rphase[4], (unsigned long long)(rc5 - rc4)); atomicAdd(&g_rphase[5], (unsigned long long)(rc6 - rc5));
        atomicAdd(&g_rphase[6], (unsigned long long)rsolves); atomicAdd(&g_rphase[7], 1ull);
    }
#endif
}

// ---- diagnostic: the controller's scalar routines on the device (tests/test_gpu_control.py holds them against the host build
// of the same source): per case p[6], A[36], b[6] -> pose matrix (16 floats), angle tables (8*3 + 15*3 doubles), solve x[6]
__global__ __launch_bounds__(64) void ndt_ctl_math_kernel(const double* __restrict__ in, int n, float* __restrict__ M, double* __restrict__ tables, double* __restrict__ x)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const double* c = in + size_t(i) * 48;
    ctl::pose_to_matrix(c, M + size_t(i) * 16);
    double j[8][3], h[15][3];
    ctl::angle_tables(c, j, h);
    for (int a = 0; a < 8; ++a) for (int b = 0; b < 3; ++b) tables[size_t(i) * 69 + a * 3 + b] = j[a][b];
    for (int a = 0; a < 15; ++a) for (int b = 0; b < 3; ++b) tables[size_t(i) * 69 + 24 + a * 3 + b] = h[a][b];
    ctl::svd_solve6(c + 6, c + 42, x + size_t(i) * 6);
}
// the wavefront form of the solve, one case per workgroup of 64 lanes
__global__ __launch_bounds__(64) void ndt_ctl_svd_wave_kernel(const double* __restrict__ in, int n, double* __restrict__ x)
{
    __shared__ ctl::SvdWaveScratch s_svd;
    __shared__ double s_A[36], s_b[6], s_x[6];
    const double* c = in + size_t(blockIdx.x) * 48;
    if (threadIdx.x < 36) s_A[threadIdx.x] = c[6 + threadIdx.x];
    if (threadIdx.x < 6) s_b[threadIdx.x] = c[42 + threadIdx.x];
    ctl::svd_wave_sync();
    ctl::svd_solve6_wave(s_A, s_b, s_x, s_svd);
    if (threadIdx.x < 6) x[size_t(blockIdx.x) * 6 + threadIdx.x] = s_x[threadIdx.x];
}

// ---- diagnostic: glibc's exp restated (glibc_exp.h) on the device; tests hold it against the host build of the same header and against the host's libm
__global__ __launch_bounds__(256) void glibc_exp_kernel(const double* __restrict__ x, size_t n, double* __restrict__ out)
{
    const size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (i < n) out[i] = glibc_exp(x[i]);
}
int glibc_exp_device(mrgfe_ctx* ctx, const double* d_x, size_t n, double* d_out)
{
    if (n == 0) return MRGFE_OK;
    hipLaunchKernelGGL(glibc_exp_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, ctx->stream, d_x, n, d_out);
    MRGFE_HIP_CHECK(hipGetLastError());
    return MRGFE_OK;
}

// ---- diagnostic: the folded wave reduction against the plain one (tests/test_gpu_primitives.py) -----------------------------------
// in: cases x 64 lanes x N doubles; out_fold / out_plain: cases x N sums (wave_sum_fold hands every lane one total; wave_sum lane 0 all)
template <int N>
__global__ __launch_bounds__(64) void wave_fold_check_kernel(const double* __restrict__ in, double* __restrict__ out_fold, double* __restrict__ out_plain)
{
    double v[N];
    int    key[N];
    const double* mine = in + (size_t(blockIdx.x) * 64 + threadIdx.x) * N;
#pragma unroll
    for (int n = 0; n < N; ++n) { v[n] = mine[n]; key[n] = n; }
    double tv;
    int    tk;
    wave_sum_fold<N, 32>(v, key, tv, tk);
    out_fold[size_t(blockIdx.x) * N + tk] = tv;  // lanes that hold the same sum write the same double
#pragma unroll
    for (int n = 0; n < N; ++n) {
        const double r = wave_sum(v[n]);
        if (threadIdx.x == 0) out_plain[size_t(blockIdx.x) * N + n] = r;
    }
}
int wave_fold_check_device(mrgfe_ctx* ctx, int n_vals, const double* d_in, int cases, double* d_fold, double* d_plain)
{
    if (cases <= 0) return MRGFE_OK;
    if (n_vals == 44)      hipLaunchKernelGGL((wave_fold_check_kernel<44>), dim3(cases), dim3(64), 0, ctx->stream, d_in, d_fold, d_plain);
    else if (n_vals == 37) hipLaunchKernelGGL((wave_fold_check_kernel<37>), dim3(cases), dim3(64), 0, ctx->stream, d_in, d_fold, d_plain);
    else if (n_vals == 1)  hipLaunchKernelGGL((wave_fold_check_kernel<1>), dim3(cases), dim3(64), 0, ctx->stream, d_in, d_fold, d_plain);
    else { set_error("wave_fold_check: 44, 37 or 1 values per lane"); return MRGFE_ERR_INVALID; }
    MRGFE_HIP_CHECK(hipGetLastError());
    return MRGFE_OK;
}

int ndt_ctl_math_device(mrgfe_ctx* ctx, const double* d_in, int n, float* d_M, double* d_tables, double* d_x)
{
    if (n <= 0) return MRGFE_OK;
    hipLaunchKernelGGL(ndt_ctl_math_kernel, dim3((n + 63) / 64), dim3(64), 0, ctx->stream, d_in, n, d_M, d_tables, d_x);
    MRGFE_HIP_CHECK(hipGetLastError());
    return MRGFE_OK;
}
int ndt_ctl_svd_wave_device(mrgfe_ctx* ctx, const double* d_in, int n, double* d_x)
{
    if (n <= 0) return MRGFE_OK;
    hipLaunchKernelGGL(ndt_ctl_svd_wave_kernel, dim3(n), dim3(64), 0, ctx->stream, d_in, n, d_x);
    MRGFE_HIP_CHECK(hipGetLastError());
    return MRGFE_OK;
}

// final_transformation * source -> packed xyzi (the `output` cloud of pcl::Registration::align)
__global__ __launch_bounds__(256) void transform_cloud_kernel(const float4* __restrict__ src, float4* __restrict__ dst, uint32_t n, const float* __restrict__ T12)
{
    __shared__ float s_T[12];
    if (threadIdx.x < 12) s_T[threadIdx.x] = T12[threadIdx.x];
    __syncthreads();
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float4 p = src[i];
    float4 o;
    transform_point(s_T, p.x, p.y, p.z, o.x, o.y, o.z);
    o.w = p.w;
    dst[i] = o;
}

template <int MODE>
static void launch_mode(mrgfe_ctx* ctx, int nnb, uint32_t grid, const NdtGridDev* d_grids, const NdtPairDev* d_pairs, const NdtEvalDev* d_evals, const uint32_t* d_plan, uint32_t P,
                        double* d_partials)
{
    if (nnb == 7)       hipLaunchKernelGGL((ndt_derivatives_kernel<MODE, 7>), dim3(grid), dim3(256), 0, ctx->stream, d_grids, d_pairs, d_evals, d_plan, P, d_partials);
    else if (nnb == 1)  hipLaunchKernelGGL((ndt_derivatives_kernel<MODE, 1>), dim3(grid), dim3(256), 0, ctx->stream, d_grids, d_pairs, d_evals, d_plan, P, d_partials);
    else                hipLaunchKernelGGL((ndt_derivatives_kernel<MODE, 27>), dim3(grid), dim3(256), 0, ctx->stream, d_grids, d_pairs, d_evals, d_plan, P, d_partials);
}

int ndt_launch_derivatives_all(mrgfe_ctx* ctx, int search, uint32_t grid, const NdtGridDev* d_grids, const NdtPairDev* d_pairs, const NdtEvalDev* d_evals, const uint32_t* d_plan,
                               uint32_t P, double* d_partials, int formulation)
{
    if (grid == 0 || P == 0) return MRGFE_OK;
    if (formulation == 1)             hipLaunchKernelGGL(ndt_derivatives_f64_all_kernel, dim3(grid), dim3(256), 0, ctx->stream, d_grids, d_pairs, d_evals, d_plan, P, d_partials);
    else if (search == MRGFE_DIRECT7)      hipLaunchKernelGGL((ndt_derivatives_all_kernel<7>), dim3(grid), dim3(256), 0, ctx->stream, d_grids, d_pairs, d_evals, d_plan, P, d_partials);
    else if (search == MRGFE_DIRECT1) hipLaunchKernelGGL((ndt_derivatives_all_kernel<1>), dim3(grid), dim3(256), 0, ctx->stream, d_grids, d_pairs, d_evals, d_plan, P, d_partials);
    else                              hipLaunchKernelGGL((ndt_derivatives_all_kernel<27>), dim3(grid), dim3(256), 0, ctx->stream, d_grids, d_pairs, d_evals, d_plan, P, d_partials);
    MRGFE_HIP_CHECK(hipGetLastError());
    return MRGFE_OK;
}

void ndt_phase_dump()
{
#ifdef NDT_PHASE_CLOCK
    unsigned long long h[8];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_phase), sizeof(h)) != hipSuccess) return;
    const double t = 100.0 * (h[5] ? h[5] : 1);
    std::fprintf(stderr, "[mrgfe phase clocks] %llu tiles, us per tile: point loaded %.2f, probes back %.2f, terms staged %.2f, queue built %.2f | pair phase %.2f; %llu items: epilogue %.2f us per item\n",
                 h[5], h[0] / t, h[1] / t, h[2] / t, h[3] / t, h[4] / t, h[7], h[6] / 100.0 / (h[7] ? h[7] : 1));
    unsigned long long r[10];
    if (hipMemcpyFromSymbol(r, HIP_SYMBOL(g_rphase), sizeof(r)) != hipSuccess) return;
    const double n = 100.0 * (r[7] ? r[7] : 1);
    std::fprintf(stderr, "[mrgfe reduce clocks] %llu controller workgroups, us each: partial sums %.2f, state in %.2f, resume %.2f, solves %.2f (%.2f solves per workgroup, %.2f us per solve), "
                         "trig + request %.2f, state out %.2f\n", r[7], r[0] / n, r[1] / n, r[2] / n, r[3] / n, double(r[6]) / (r[7] ? r[7] : 1), r[6] ? r[3] / 100.0 / r[6] : 0.0, r[4] / n, r[5] / n);
#endif
}

// which pairs have finished, and where: one workgroup, records first, the request's tag last (system scope: the host polls it)
__global__ __launch_bounds__(256) void ndt_snapshot_kernel(const NdtCtlState* __restrict__ states, uint32_t P, uint32_t tag, NdtSnapshotHead* __restrict__ head, NdtSnapshotRec* __restrict__ recs)
{
    __shared__ uint32_t s_done[4];
    uint32_t mine = 0;
    for (uint32_t i = threadIdx.x; i < P; i += 256) {
        const NdtCtlState& st = states[i];
        const bool fin = ctl::done(st);
        recs[i].done = fin ? 1u : 0u;
        if (fin) {
            for (int k = 0; k < 12; ++k) recs[i].T12[k] = st.final_[k];
            ++mine;
        }
    }
    mine = wave_sum(mine);
    if (lane_id() == 0) s_done[wave_id()] = mine;
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        head->n_done = s_done[0] + s_done[1] + s_done[2] + s_done[3];
        __threadfence_system();
        __hip_atomic_store(&head->tag, tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

int ndt_launch_snapshot(mrgfe_ctx* ctx, const NdtCtlState* d_states, uint32_t P, uint32_t tag, NdtSnapshotHead* h_head, NdtSnapshotRec* h_recs)
{
    hipLaunchKernelGGL(ndt_snapshot_kernel, dim3(1), dim3(256), 0, ctx->stream, d_states, P, tag, h_head, h_recs);
    MRGFE_HIP_CHECK(hipGetLastError());
    return MRGFE_OK;
}

int ndt_launch_plan(mrgfe_ctx* ctx, const NdtPairDev* d_pairs, const NdtEvalDev* d_evals, uint32_t P, uint32_t* d_plan, uint32_t wg_target, uint32_t max_ppt, uint32_t forced_ppt,
                    uint32_t round, NdtRoundInfo* h_info)
{
    hipLaunchKernelGGL(ndt_plan_kernel, dim3(1), dim3(256), 0, ctx->stream, d_pairs, d_evals, P, d_plan, wg_target, max_ppt, forced_ppt, round, h_info);
    MRGFE_HIP_CHECK(hipGetLastError());
    return MRGFE_OK;
}

int ndt_launch_derivatives(mrgfe_ctx* ctx, int mode, int search, uint32_t grid, const NdtGridDev* d_grids, const NdtPairDev* d_pairs, const NdtEvalDev* d_evals,
                           const uint32_t* d_plan, uint32_t P, double* d_partials, int formulation)
{
    if (grid == 0 || P == 0) return MRGFE_OK;
    if (formulation == 1) {
        if (mode == 0)      hipLaunchKernelGGL((ndt_derivatives_f64_kernel<0>), dim3(grid), dim3(256), 0, ctx->stream, d_grids, d_pairs, d_evals, d_plan, P, d_partials);
        else if (mode == 1) hipLaunchKernelGGL((ndt_derivatives_f64_kernel<1>), dim3(grid), dim3(256), 0, ctx->stream, d_grids, d_pairs, d_evals, d_plan, P, d_partials);
        else                hipLaunchKernelGGL((ndt_derivatives_f64_kernel<2>), dim3(grid), dim3(256), 0, ctx->stream, d_grids, d_pairs, d_evals, d_plan, P, d_partials);
        MRGFE_HIP_CHECK(hipGetLastError());
        return MRGFE_OK;
    }
    const int nnb = (search == MRGFE_DIRECT7) ? 7 : (search == MRGFE_DIRECT1 ? 1 : 27);
    if (mode == 0)      launch_mode<0>(ctx, nnb, grid, d_grids, d_pairs, d_evals, d_plan, P, d_partials);
    else if (mode == 1) launch_mode<1>(ctx, nnb, grid, d_grids, d_pairs, d_evals, d_plan, P, d_partials);
    else                launch_mode<2>(ctx, nnb, grid, d_grids, d_pairs, d_evals, d_plan, P, d_partials);
    MRGFE_HIP_CHECK(hipGetLastError());
    return MRGFE_OK;
}

int ndt_launch_single_round(mrgfe_ctx* ctx, int search, uint32_t grid, const NdtGridDev* d_grids, const NdtPairDev* d_pairs, const NdtEvalDev* d_evals, const uint32_t* d_plan,
                            double* d_partials, uint32_t* d_ticket, double* h_results, double tag)
{
    if (grid == 0) return MRGFE_OK;
    if (search == MRGFE_DIRECT7)      hipLaunchKernelGGL((ndt_derivatives_single_kernel<7>), dim3(grid), dim3(256), 0, ctx->stream, d_grids, d_pairs, d_evals, d_plan, d_partials, d_ticket, h_results, tag);
    else if (search == MRGFE_DIRECT1) hipLaunchKernelGGL((ndt_derivatives_single_kernel<1>), dim3(grid), dim3(256), 0, ctx->stream, d_grids, d_pairs, d_evals, d_plan, d_partials, d_ticket, h_results, tag);
    else                              hipLaunchKernelGGL((ndt_derivatives_single_kernel<27>), dim3(grid), dim3(256), 0, ctx->stream, d_grids, d_pairs, d_evals, d_plan, d_partials, d_ticket, h_results, tag);
    MRGFE_HIP_CHECK(hipGetLastError());
    return MRGFE_OK;
}

int ndt_launch_reduce(mrgfe_ctx* ctx, uint32_t P, const NdtPairDev* d_pairs, NdtEvalDev* d_evals, const double* d_partials, const uint32_t* d_plan, double* d_results,
                      NdtCtlState* d_states, double tag, uint32_t* d_ticket)
{
    if (P == 0) return MRGFE_OK;
    if (d_states) hipLaunchKernelGGL((ndt_reduce_kernel<true>), dim3(P), dim3(256), 0, ctx->stream, d_pairs, d_evals, d_partials, d_plan, d_results, d_states, 0.0, static_cast<uint32_t*>(nullptr));
    else          hipLaunchKernelGGL((ndt_reduce_kernel<false>), dim3(P), dim3(256), 0, ctx->stream, d_pairs, d_evals, d_partials, d_plan, d_results, d_states,
                                     (P == 1 || d_ticket != nullptr) ? tag : 0.0, d_ticket);
    MRGFE_HIP_CHECK(hipGetLastError());
    return MRGFE_OK;
}

int launch_transform_cloud(mrgfe_ctx* ctx, const float4* d_src, float4* d_dst, uint32_t n, const float* d_T12)
{
    if (n == 0) return MRGFE_OK;
    hipLaunchKernelGGL(transform_cloud_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, d_src, d_dst, n, d_T12);
    MRGFE_HIP_CHECK(hipGetLastError());
    return MRGFE_OK;
}

}  // namespace mrgfe
