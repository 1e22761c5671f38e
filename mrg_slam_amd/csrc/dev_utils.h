// csrc/dev_utils.h — device helpers for gfx950 (wave64): wave/block reductions and scans.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mrgfe {

constexpr int kWave = 64;  // CDNA4 wavefront width (hard-coded: MI355X_MICROARCH.md "wave = 64 not 32")

// Pointers that come out of a descriptor in memory or LDS (NdtPairDev::src, NdtGridDev / NnGridDev members, job records) are generic to
// the compiler, which emits FLAT loads for them — issued to the LDS and the memory pipeline both, counted in lgkmcnt as well as vmcnt.
// These helpers name the address space (global), so the loads are global_load_*.
#define MRGFE_GLOBAL __attribute__((address_space(1)))
typedef float gvec4f __attribute__((ext_vector_type(4)));
template <class T>
__device__ __forceinline__ const MRGFE_GLOBAL T* as_global(const T* p) { return (const MRGFE_GLOBAL T*)p; }
__device__ __forceinline__ float4 load_point(const float4* p)
{
    const gvec4f v = *(const MRGFE_GLOBAL gvec4f*)p;
    return make_float4(v.x, v.y, v.z, v.w);
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & (kWave - 1); }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

// ---- wave-level reductions through cross-lane shuffles (DPP / ds_bpermute, no LDS traffic) -------------------
// the value lane `lane` holds, as a wave-uniform (scalar) value; `lane` must be the same in every lane.  Unlike __shfl with a uniform index
// this is a v_readlane (no LDS round trip) and the result lives in a scalar register.
__device__ __forceinline__ int      wave_read(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ uint32_t wave_read(uint32_t v, int lane) { return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), lane)); }
__device__ __forceinline__ float    wave_read(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }

template <class T>
__device__ __forceinline__ T wave_sum(T v)
{
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
    return v;  // valid in lane 0
}
// The same sums for N values per lane at once, folded: at offset OFF a lane keeps one value of a pair and hands the other to lane ^ OFF,
// so a step moves N / 2 values instead of N and the six steps together ~N instead of 6 N (the 44 sums of an NDT work item: 45
// shuffled doubles instead of 264).  Every value is still added in wave_sum's tree — (lane i) + (lane i + OFF) for OFF = 32 ... 1,
// and a + b == b + a bit for bit — so the totals are the same doubles.  Returns ONE total per lane: out_v is the total of the value
// that entered with key out_k (every value's total ends in at least one lane).
template <int N, int OFF>
__device__ __forceinline__ void wave_sum_fold(const double (&v)[N], const int (&key)[N], double& out_v, int& out_k)
{
    if constexpr (OFF == 0) {
        static_assert(N == 1, "more values than lanes");
        out_v = v[0];
        out_k = key[0];
    } else {
        constexpr int M = N / 2, R = N - M;  // M pairs (j, j + R); N odd: value M stays whole and takes the plain butterfly step
        double nv[R];
        int    nk[R];
        const bool up = (threadIdx.x & OFF) != 0;
#pragma unroll
        for (int j = 0; j < M; ++j) {
            const double keep = up ? v[j + R] : v[j], send = up ? v[j] : v[j + R];
            nv[j] = keep + __shfl_xor(send, OFF, kWave);
            nk[j] = up ? key[j + R] : key[j];
        }
        if constexpr (N % 2 == 1) {
            nv[M] = v[M] + __shfl_xor(v[M], OFF, kWave);
            nk[M] = key[M];
        }
        wave_sum_fold<R, OFF / 2>(nv, nk, out_v, out_k);
    }
}

__device__ __forceinline__ float wave_min(float v)
{
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) v = fminf(v, __shfl_down(v, off, kWave));
    return v;
}
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) v = fmaxf(v, __shfl_down(v, off, kWave));
    return v;
}

// inclusive scan across the 64 lanes of a wave
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v)
{
    const int lane = lane_id();
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        uint32_t t = __shfl_up(v, off, kWave);
        if (lane >= off) v += t;
    }
    return v;
}

// exclusive scan over a block of NT threads (NT multiple of 64, <= 1024). `lds` needs NT/64 + 1 words.
// Returns the exclusive prefix of `v`; *block_total receives the sum over the block (same value in every thread).
template <int NT>
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* lds, uint32_t* block_total)
{
    constexpr int NW = NT / kWave;
    const int lane = lane_id(), w = wave_id();
    uint32_t incl = wave_inclusive_scan(v);
    if (lane == kWave - 1) lds[w] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
#pragma unroll
        for (int i = 0; i < NW; ++i) { uint32_t t = lds[i]; lds[i] = run; run += t; }
        lds[NW] = run;
    }
    __syncthreads();
    uint32_t res = incl - v + lds[w];
    *block_total = lds[NW];
    __syncthreads();  // lds may be reused by the caller
    return res;
}

// lanes of the wave (among `active`) that hold the same 8-bit digit as this lane
__device__ __forceinline__ uint64_t wave_match_digit8_bits(uint32_t d, uint64_t active)
{
    uint64_t m = active;
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const bool     bit = (d >> b) & 1u;
        const uint64_t bal = __ballot(bit);
        m &= bit ? bal : ~bal;
    }
    return m;
}
// The same set, found by peeling: the digit of the first lane not yet matched is broadcast, one ballot names its lanes, and so on.  The keys of a
// wavefront are neighbours in a scan — a handful of distinct digits, often one — so this is a few ballots instead of eight (the bit-by-bit form
// is ~60 VALU instructions per key and was what bound the histogram and scatter kernels of the radix sort: 800 - 1100 per wavefront of eight
// keys).  After kPeel digits whatever is left takes the bit-by-bit form: a wavefront of unrelated keys pays a little more than before, not 64 ballots.
__device__ __forceinline__ uint64_t wave_match_digit8(uint32_t d, uint64_t active)
{
    constexpr int kPeel = 4;
    const bool    mine = (active >> lane_id()) & 1ull;
    uint64_t      left = active, m = 0ull;
#pragma unroll
    for (int it = 0; it < kPeel; ++it) {
        if (left == 0ull) break;  // uniform
        const int      leader = __ffsll(static_cast<unsigned long long>(left)) - 1;
        const uint32_t dv = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(d), leader));
        const uint64_t hit = __ballot(mine && d == dv);
        if (mine && d == dv) m = hit;
        left &= ~hit;
    }
    if (left != 0ull) {  // uniform
        const uint64_t full = wave_match_digit8_bits(d, active);
        if ((left >> lane_id()) & 1ull) m = full;
    }
    return m;
}

// eight consecutive words p[first .. first + 8) of an array of n (first a multiple of 4, p 16-byte aligned: the slices of cellsort.h): two 16-byte
// loads when all eight exist, else word by word with `fill` past the end
// h[d] += 1 for every `valid` lane (h: the workgroup's LDS histogram): the lanes that share one of the first few distinct digits of the wavefront
// add together (one atomic per digit: 64 atomics on one counter are served one after the other), whoever is left adds for itself — a count needs
// no ranks, so unrelated digits (the upper digits of a sort's later passes) cost a ballot or two and one atomic per lane instead of the full match
__device__ __forceinline__ void wave_hist_add(uint32_t* h, uint32_t d, bool valid)
{
    constexpr int kPeel = 3;
    uint64_t left = __ballot(valid);
    bool     mine = valid;
#pragma unroll
    for (int it = 0; it < kPeel; ++it) {
        if (left == 0ull) break;  // uniform
        const int      leader = __ffsll(static_cast<unsigned long long>(left)) - 1;
        const uint32_t dv = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(d), leader));
        const bool     same = mine && d == dv;
        const uint64_t hit = __ballot(same);
        if (lane_id() == leader) atomicAdd(&h[dv], static_cast<uint32_t>(__popcll(hit)));
        mine = mine && !same;
        left &= ~hit;
    }
    if (mine) atomicAdd(&h[d], 1u);
}

__device__ __forceinline__ void load8_u32(const uint32_t* __restrict__ p, uint32_t first, uint32_t n, uint32_t fill, uint32_t out[8])
{
    if (first + 8u <= n) {
        const uint4 a = *reinterpret_cast<const uint4*>(p + first), b = *reinterpret_cast<const uint4*>(p + first + 4);
        out[0] = a.x; out[1] = a.y; out[2] = a.z; out[3] = a.w; out[4] = b.x; out[5] = b.y; out[6] = b.z; out[7] = b.w;
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) out[k] = first + k < n ? p[first + k] : fill;
    }
}

__device__ __forceinline__ bool finite3(float x, float y, float z) { return isfinite(x) && isfinite(y) && isfinite(z); }

}  // namespace mrgfe
