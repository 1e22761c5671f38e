// csrc/cellsort.hip — bounding box, stable LSD radix sort, exclusive scan and run-head marking for gfx950.
//
// These are HBM-bound integer passes: every kernel streams its slice with coalesced 4-byte or
// 16-byte accesses, keeps the per-tile state (digit histogram, wave counters) in LDS and never reshapes the work
// into a GEMM.  Grids are (tiles, problems): a 120k-point cloud is 59 tiles, a 64-pair batch 3.8k workgroups.
#include "cellsort.h"
#include "dev_utils.h"

namespace mrgfe {

void SliceTable::build(const uint32_t* n, int nprob)
{
    h.resize(nprob);
    uint32_t off = 0, blk = 0;
    max_blks = 0;
    for (int p = 0; p < nprob; ++p) {
        Slice s;
        s.n = n[p];
        s.off = off;
        s.blk_off = blk;
        s.nblk = (n[p] + kTile - 1) / kTile;
        h[p] = s;
        off += (n[p] + 3u) & ~3u;
        blk += s.nblk;
        max_blks = s.nblk > max_blks ? s.nblk : max_blks;
    }
    total_elems = off;
    total_blks = blk;
}

// ------------------------------------------------------------------------------------------------------
// bounding boxes
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bbox_partial_kernel(const float4* const* __restrict__ clouds, const Slice* __restrict__ slices, BBox* __restrict__ partial)
{
    const Slice s = slices[blockIdx.y];
    if (blockIdx.x >= s.nblk) return;
    const uint32_t base = blockIdx.x * kTile;
    const float4* __restrict__ pts = clouds[blockIdx.y];
    float    mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    uint32_t cnt = 0;
#pragma unroll
    for (int k = 0; k < kTile / 256; ++k) {
        uint32_t i = base + k * 256 + threadIdx.x;
        if (i < s.n) {
            float4 p = load_point(pts + i);
            if (finite3(p.x, p.y, p.z)) {
                mn[0] = fminf(mn[0], p.x); mn[1] = fminf(mn[1], p.y); mn[2] = fminf(mn[2], p.z);
                mx[0] = fmaxf(mx[0], p.x); mx[1] = fmaxf(mx[1], p.y); mx[2] = fmaxf(mx[2], p.z);
                ++cnt;
            }
        }
    }
    __shared__ float    s_mn[4][3], s_mx[4][3];
    __shared__ uint32_t s_cnt[4];
#pragma unroll
    for (int a = 0; a < 3; ++a) { mn[a] = wave_min(mn[a]); mx[a] = wave_max(mx[a]); }
    cnt = wave_sum(cnt);
    if (lane_id() == 0) {
        for (int a = 0; a < 3; ++a) { s_mn[wave_id()][a] = mn[a]; s_mx[wave_id()][a] = mx[a]; }
        s_cnt[wave_id()] = cnt;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        BBox b;
        for (int a = 0; a < 3; ++a) {
            b.mn[a] = fminf(fminf(s_mn[0][a], s_mn[1][a]), fminf(s_mn[2][a], s_mn[3][a]));
            b.mx[a] = fmaxf(fmaxf(s_mx[0][a], s_mx[1][a]), fmaxf(s_mx[2][a], s_mx[3][a]));
        }
        b.n_finite = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        b.pad = 0;
        partial[s.blk_off + blockIdx.x] = b;
    }
}

__global__ __launch_bounds__(256) void bbox_final_kernel(const Slice* __restrict__ slices, const BBox* __restrict__ partial, BBox* __restrict__ out)
{
    const Slice s = slices[blockIdx.x];
    float    mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    uint32_t cnt = 0;
    for (uint32_t b = threadIdx.x; b < s.nblk; b += 256) {
        BBox q = partial[s.blk_off + b];
        for (int a = 0; a < 3; ++a) { mn[a] = fminf(mn[a], q.mn[a]); mx[a] = fmaxf(mx[a], q.mx[a]); }
        cnt += q.n_finite;
    }
    __shared__ float    s_mn[4][3], s_mx[4][3];
    __shared__ uint32_t s_cnt[4];
#pragma unroll
    for (int a = 0; a < 3; ++a) { mn[a] = wave_min(mn[a]); mx[a] = wave_max(mx[a]); }
    cnt = wave_sum(cnt);
    if (lane_id() == 0) {
        for (int a = 0; a < 3; ++a) { s_mn[wave_id()][a] = mn[a]; s_mx[wave_id()][a] = mx[a]; }
        s_cnt[wave_id()] = cnt;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        BBox b;
        for (int a = 0; a < 3; ++a) {
            b.mn[a] = fminf(fminf(s_mn[0][a], s_mn[1][a]), fminf(s_mn[2][a], s_mn[3][a]));
            b.mx[a] = fmaxf(fmaxf(s_mx[0][a], s_mx[1][a]), fmaxf(s_mx[2][a], s_mx[3][a]));
        }
        b.n_finite = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        b.pad = 0;
        out[blockIdx.x] = b;
    }
}

int bounding_box_partials(mrgfe_ctx* ctx, const float4* const* d_clouds, const Slice* d_slices, const SliceTable& t, BBox* d_partial)
{
    if (t.nprob() == 0 || t.max_blks == 0) return MRGFE_OK;
    hipLaunchKernelGGL(bbox_partial_kernel, dim3(t.max_blks, t.nprob()), dim3(256), 0, ctx->stream, d_clouds, d_slices, d_partial);
    MRGFE_HIP_CHECK(hipGetLastError());
    return MRGFE_OK;
}

int bounding_boxes(mrgfe_ctx* ctx, const float4* const* d_clouds, const Slice* d_slices, const SliceTable& t, BBox* d_partial, BBox* d_out)
{
    if (t.nprob() == 0) return MRGFE_OK;
    if (t.max_blks > 0) {
        dim3 grid(t.max_blks, t.nprob());
        hipLaunchKernelGGL(bbox_partial_kernel, grid, dim3(256), 0, ctx->stream, d_clouds, d_slices, d_partial);
    }
    hipLaunchKernelGGL(bbox_final_kernel, dim3(t.nprob()), dim3(256), 0, ctx->stream, d_slices, d_partial, d_out);
    MRGFE_HIP_CHECK(hipGetLastError());
    return MRGFE_OK;
}

// ------------------------------------------------------------------------------------------------------
// radix sort (8-bit digits, stable)
// ------------------------------------------------------------------------------------------------------
// pass 1: per-tile digit histogram, LDS atomics.  hist layout: [tile][256] so the scan kernel reads coalesced.
constexpr int kHistTiles = 2;  // tiles per workgroup of rs_hist_kernel: both tiles' loads go out before the first count (the kernel waited for one 8 KB tile per workgroup)
__global__ __launch_bounds__(256) void rs_hist_kernel(const uint32_t* __restrict__ keys, const Slice* __restrict__ slices, uint32_t* __restrict__ hist, int shift)
{
    const Slice s = slices[blockIdx.y];
    const uint32_t tile0 = blockIdx.x * kHistTiles;
    if (tile0 >= s.nblk) return;
    __shared__ uint32_t h[kHistTiles][256];
#pragma unroll
    for (int t = 0; t < kHistTiles; ++t) h[t][threadIdx.x] = 0;
    __syncthreads();
    // neighbours in a scan fall into the same voxel: a wavefront's keys share a handful of digits, and 64 LDS atomics on one
    // counter are served one after the other: wave_hist_add (dev_utils.h) lets the lanes that share a digit add together.
    // (a count does not mind the order: every lane takes eight CONSECUTIVE keys of each tile, two 16-byte loads instead of eight strided words)
    static_assert(kTile / 256 == 8, "eight keys per lane");
    uint32_t key[kHistTiles][8];
#pragma unroll
    for (int t = 0; t < kHistTiles; ++t) load8_u32(keys + s.off, (tile0 + t) * kTile + threadIdx.x * 8u, s.n, 0u, key[t]);  // (a tile past the end: n <= its first key, all fill)
#pragma unroll
    for (int t = 0; t < kHistTiles; ++t) {
        const uint32_t first = (tile0 + t) * kTile + threadIdx.x * 8u;
#pragma unroll
        for (int k = 0; k < kTile / 256; ++k) {
            const bool     valid = first + k < s.n;
            const uint32_t d = (key[t][k] >> shift) & 255u;
            wave_hist_add(h[t], d, valid);
        }
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < kHistTiles; ++t)
        if (tile0 + t < s.nblk) hist[(size_t)(s.blk_off + tile0 + t) * 256 + threadIdx.x] = h[t][threadIdx.x];
}

// pass 2: one workgroup per problem. Thread d turns column d of hist into an exclusive prefix over tiles, then the
// 256 digit totals are scanned to digit bases (stored in digit_base[problem][256]).
__global__ __launch_bounds__(256) void rs_scan_kernel(const Slice* __restrict__ slices, uint32_t* __restrict__ hist, uint32_t* __restrict__ digit_base)
{
    const Slice s = slices[blockIdx.x];
    __shared__ uint32_t lds[8];
    uint32_t run = 0;
    // sixteen tiles' counts in flight, then their prefixes out: read and written through the same pointer, a tile-by-tile loop
    // waits for every load in turn (64 tiles of a 130k-point cloud: 16 us, which a single registration's two sort passes paid twice)
    uint32_t* col = hist + (size_t)s.blk_off * 256 + threadIdx.x;
    uint32_t  b = 0;
    for (; b + 16 <= s.nblk; b += 16) {
        uint32_t v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = col[(size_t)(b + u) * 256];
#pragma unroll
        for (int u = 0; u < 16; ++u) { col[(size_t)(b + u) * 256] = run; run += v[u]; }
    }
    for (; b < s.nblk; ++b) {
        const uint32_t v = col[(size_t)b * 256];
        col[(size_t)b * 256] = run;
        run += v;
    }
    uint32_t total;
    uint32_t base = block_exclusive_scan<256>(run, lds, &total);
    digit_base[(size_t)blockIdx.x * 256 + threadIdx.x] = base;
}

// A problem of thousands of tiles (the map cloud: 6.5 M points, 3200 tiles) kept ONE workgroup busy for 110 us per pass walking its 3200 x 256
// counts.  Two launches instead: the digit counts of every chunk of 64 tiles, then one workgroup per chunk that adds up the chunks before its own
// (a few dozen rows) and scans its own 64 tiles; the workgroup of chunk 0 also leaves the digit bases.
constexpr uint32_t kScanChunk = 64;      // tiles per chunk
constexpr uint32_t kBigScanBlks = 512;   // a sort with a problem of more tiles than this takes the chunked scan
__global__ __launch_bounds__(256) void rs_chunk_sums_kernel(const Slice* __restrict__ slices, const uint32_t* __restrict__ hist, const uint32_t* __restrict__ chunk_off,
                                                             uint32_t* __restrict__ chunk_sums)
{
    const Slice s = slices[blockIdx.y];
    const uint32_t b0 = blockIdx.x * kScanChunk;
    if (b0 >= s.nblk) return;
    const uint32_t b1 = min(b0 + kScanChunk, s.nblk);
    const uint32_t* col = hist + (size_t)s.blk_off * 256 + threadIdx.x;
    uint32_t sum = 0, b = b0;
    for (; b + 16 <= b1; b += 16) {
        uint32_t v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = col[(size_t)(b + u) * 256];
#pragma unroll
        for (int u = 0; u < 16; ++u) sum += v[u];
    }
    for (; b < b1; ++b) sum += col[(size_t)b * 256];
    chunk_sums[(size_t)(chunk_off[blockIdx.y] + blockIdx.x) * 256 + threadIdx.x] = sum;
}
__global__ __launch_bounds__(256) void rs_scan_chunks_kernel(const Slice* __restrict__ slices, uint32_t* __restrict__ hist, const uint32_t* __restrict__ chunk_off,
                                                              const uint32_t* __restrict__ chunk_sums, uint32_t* __restrict__ digit_base)
{
    const Slice s = slices[blockIdx.y];
    const uint32_t nchunk = (s.nblk + kScanChunk - 1) / kScanChunk;
    if (blockIdx.x >= nchunk && blockIdx.x != 0) return;  // (chunk 0 of an empty problem still writes its digit bases: all zero)
    __shared__ uint32_t lds[8];
    const uint32_t* cs = chunk_sums + (size_t)chunk_off[blockIdx.y] * 256 + threadIdx.x;
    uint32_t before = 0, total = 0;
    for (uint32_t c = 0; c < nchunk; ++c) {
        const uint32_t v = cs[(size_t)c * 256];
        total += v;
        before += c < blockIdx.x ? v : 0u;
    }
    if (blockIdx.x == 0) {
        uint32_t all;
        digit_base[(size_t)blockIdx.y * 256 + threadIdx.x] = block_exclusive_scan<256>(total, lds, &all);
    }
    if (blockIdx.x >= nchunk) return;
    const uint32_t b0 = blockIdx.x * kScanChunk, b1 = min(b0 + kScanChunk, s.nblk);
    uint32_t* col = hist + (size_t)s.blk_off * 256 + threadIdx.x;
    uint32_t  run = before, b = b0;
    for (; b + 16 <= b1; b += 16) {
        uint32_t v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = col[(size_t)(b + u) * 256];
#pragma unroll
        for (int u = 0; u < 16; ++u) { col[(size_t)(b + u) * 256] = run; run += v[u]; }
    }
    for (; b < b1; ++b) {
        const uint32_t v = col[(size_t)b * 256];
        col[(size_t)b * 256] = run;
        run += v;
    }
}

constexpr uint32_t kOwnScanBlks = 512;  // tiles of a sort (all its problems) up to which the scatter kernel does the scan of the histograms as well
// pass 3: stable scatter. The tile is 8 rounds of 256 keys; the rank of a key among equal digits is (keys of earlier rounds and of
// earlier waves of its round: a scan over the 32 (round, wave) counts of the digit) + (lower lanes of its own wave, by ballot matching).
// kIota: the values are the element indices 0 .. n-1 of the problem (first pass of a sort of (key, index) pairs: nobody has to write or read them)
// kOwnScan (a sort of a few tiles, bound by its launches): `hist` holds the tiles' COUNTS as rs_hist_kernel left them and every workgroup adds up
// column d over the tiles before its own and over all of them itself — rs_scan_kernel's work, 64 KB of L2 reads per workgroup of a 130k-key
// sort instead of a launch per pass.
template <bool kIota, bool kOwnScan>
__global__ __launch_bounds__(256) void rs_scatter_kernel(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals, uint32_t* __restrict__ keys_out,
                                                          uint32_t* __restrict__ vals_out, const Slice* __restrict__ slices, const uint32_t* __restrict__ hist,
                                                          const uint32_t* __restrict__ digit_base, int shift)
{
    const Slice s = slices[blockIdx.y];
    if (blockIdx.x >= s.nblk) return;
    __shared__ uint32_t goff[256];             // global offset of this tile's first key of each digit
    __shared__ uint16_t cnt[kTile / 256][4][256];  // keys of each digit per (round, wavefront); after the scan: the tile's keys of the digit BEFORE that (round, wavefront)
    if (kOwnScan) {
        __shared__ uint32_t lds[8];
        const uint32_t* col = hist + (size_t)s.blk_off * 256 + threadIdx.x;
        uint32_t before = 0, total = 0, b = 0;
        for (; b + 16 <= s.nblk; b += 16) {
            uint32_t v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = col[(size_t)(b + u) * 256];
#pragma unroll
            for (int u = 0; u < 16; ++u) { total += v[u]; before += b + u < blockIdx.x ? v[u] : 0u; }
        }
        for (; b < s.nblk; ++b) {
            const uint32_t v = col[(size_t)b * 256];
            total += v;
            before += b < blockIdx.x ? v : 0u;
        }
        uint32_t all;
        goff[threadIdx.x] = block_exclusive_scan<256>(total, lds, &all) + before;
    } else {
        goff[threadIdx.x] = digit_base[(size_t)blockIdx.y * 256 + threadIdx.x] + hist[(size_t)(s.blk_off + blockIdx.x) * 256 + threadIdx.x];
    }
    {
        uint32_t* z = reinterpret_cast<uint32_t*>(&cnt[0][0][0]);
#pragma unroll
        for (int u = 0; u < (kTile / 256) * 4 * 256 / 2 / 256; ++u) z[u * 256 + threadIdx.x] = 0u;
    }
    __syncthreads();
    const uint32_t base = blockIdx.x * kTile;
    const int      lane = lane_id(), w = wave_id();
    // The tile is eight rounds of 256 keys (round k: keys base + 256 k + thread: the order of the input).  All keys and values first, then every round's
    // match; ONE barrier; thread d turns column d of the 32 (round, wavefront) counts into exclusive prefixes; one more barrier; every key knows
    // its place.  (Round 4 ran the rounds one after the other with three barriers each and a load at the top of each: 24 barriers and eight memory
    // round trips per tile, the wavefronts parked 80 % of their cycles.)
    uint32_t key8[kTile / 256], val8[kTile / 256], rank8[kTile / 256];
#pragma unroll
    for (int k = 0; k < kTile / 256; ++k) {
        const uint32_t i = base + k * 256 + threadIdx.x;
        key8[k] = i < s.n ? keys[s.off + i] : 0u;
    }
    if (!kIota) {
#pragma unroll
        for (int k = 0; k < kTile / 256; ++k) {
            const uint32_t i = base + k * 256 + threadIdx.x;
            val8[k] = i < s.n ? vals[s.off + i] : 0u;
        }
    }
#pragma unroll
    for (int k = 0; k < kTile / 256; ++k) {
        const bool     valid = base + k * 256 + threadIdx.x < s.n;
        const uint32_t d = (key8[k] >> shift) & 255u;
        const uint64_t m = wave_match_digit8(d, __ballot(valid));
        rank8[k] = __popcll(m & ((1ull << lane) - 1ull));
        if (valid && rank8[k] == 0) cnt[k][w][d] = static_cast<uint16_t>(__popcll(m));
    }
    __syncthreads();
    {
        uint32_t run = 0;
#pragma unroll
        for (int k = 0; k < kTile / 256; ++k)
#pragma unroll
            for (int ww = 0; ww < 4; ++ww) {
                const uint32_t c = cnt[k][ww][threadIdx.x];
                cnt[k][ww][threadIdx.x] = static_cast<uint16_t>(run);
                run += c;
            }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kTile / 256; ++k) {
        const uint32_t i = base + k * 256 + threadIdx.x;
        if (i < s.n) {
            const uint32_t d = (key8[k] >> shift) & 255u;
            const uint32_t pos = s.off + goff[d] + cnt[k][w][d] + rank8[k];
            keys_out[pos] = key8[k];
            vals_out[pos] = kIota ? i : val8[k];
        }
    }
}

int radix_sort_pairs(mrgfe_ctx* ctx, uint32_t* d_keys, uint32_t* d_vals, uint32_t* d_keys_tmp, uint32_t* d_vals_tmp, const Slice* d_slices, const SliceTable& t,
                     int key_bits, uint32_t* d_hist, uint32_t** out_keys, uint32_t** out_vals, bool iota_vals, bool first_hist_ready)
{
    *out_keys = d_keys;
    *out_vals = d_vals;
    if (t.nprob() == 0 || t.max_blks == 0) return MRGFE_OK;
    if (key_bits < 1) key_bits = 1;
    if (key_bits > 32) key_bits = 32;
    const int passes = (key_bits + 7) / 8;
    uint32_t* digit_base = d_hist + (size_t)t.total_blks * 256;
    uint32_t *ki = d_keys, *vi = d_vals, *ko = d_keys_tmp, *vo = d_vals_tmp;
    dim3 grid(t.max_blks, t.nprob());
    // a sort of a few hundred tiles lasts as long as its launches: the scatter kernel scans the tile histograms itself (see rs_scatter_kernel)
    static const uint32_t own_scan_blks = [] { const char* e = std::getenv("MRGFE_SORT_OWN_SCAN_BLKS"); return e ? static_cast<uint32_t>(std::atoi(e)) : kOwnScanBlks; }();
    const bool own_scan = t.total_blks <= own_scan_blks && t.max_blks <= 1024;
    // the chunked scan for problems of many tiles: chunk offsets per problem (host-known sizes) staged up once per sort
    const bool      big_scan = !own_scan && t.max_blks > kBigScanBlks;
    uint32_t        max_chunks = 0;
    const uint32_t* d_chunk_off = nullptr;
    uint32_t*       d_chunk_sums = nullptr;
    if (big_scan) {
        std::vector<uint32_t> off(t.nprob() + 1, 0);
        for (int p = 0; p < t.nprob(); ++p) {
            const uint32_t nc = (t.h[p].nblk + kScanChunk - 1) / kScanChunk;
            off[p + 1] = off[p] + nc;
            max_chunks = std::max(max_chunks, nc);
        }
        const size_t off_bytes = (sizeof(uint32_t) * off.size() + 255) & ~size_t(255);
        MRGFE_TRY(ctx->sort_chunks.ensure(off_bytes + sizeof(uint32_t) * 256 * std::max<uint32_t>(off.back(), 1)));
        MRGFE_TRY(ctx->stage_h2d(ctx->sort_chunks.p, off.data(), sizeof(uint32_t) * off.size(), ctx->stream));
        d_chunk_off = ctx->sort_chunks.as<uint32_t>();
        d_chunk_sums = reinterpret_cast<uint32_t*>(ctx->sort_chunks.as<char>() + off_bytes);
    }
    for (int p = 0; p < passes; ++p) {
        const int shift = 8 * p;
        if (!(p == 0 && first_hist_ready)) hipLaunchKernelGGL(rs_hist_kernel, dim3((t.max_blks + kHistTiles - 1) / kHistTiles, t.nprob()), dim3(256), 0, ctx->stream, ki, d_slices, d_hist, shift);
        if (own_scan) {
            if (p == 0 && iota_vals) hipLaunchKernelGGL((rs_scatter_kernel<true, true>), grid, dim3(256), 0, ctx->stream, ki, vi, ko, vo, d_slices, d_hist, digit_base, shift);
            else                     hipLaunchKernelGGL((rs_scatter_kernel<false, true>), grid, dim3(256), 0, ctx->stream, ki, vi, ko, vo, d_slices, d_hist, digit_base, shift);
        } else {
            if (big_scan) {
                hipLaunchKernelGGL(rs_chunk_sums_kernel, dim3(max_chunks, t.nprob()), dim3(256), 0, ctx->stream, d_slices, d_hist, d_chunk_off, d_chunk_sums);
                hipLaunchKernelGGL(rs_scan_chunks_kernel, dim3(max_chunks, t.nprob()), dim3(256), 0, ctx->stream, d_slices, d_hist, d_chunk_off, d_chunk_sums, digit_base);
            } else {
                hipLaunchKernelGGL(rs_scan_kernel, dim3(t.nprob()), dim3(256), 0, ctx->stream, d_slices, d_hist, digit_base);
            }
            if (p == 0 && iota_vals) hipLaunchKernelGGL((rs_scatter_kernel<true, false>), grid, dim3(256), 0, ctx->stream, ki, vi, ko, vo, d_slices, d_hist, digit_base, shift);
            else                     hipLaunchKernelGGL((rs_scatter_kernel<false, false>), grid, dim3(256), 0, ctx->stream, ki, vi, ko, vo, d_slices, d_hist, digit_base, shift);
        }
        uint32_t* tk = ki; ki = ko; ko = tk;
        uint32_t* tv = vi; vi = vo; vo = tv;
    }
    MRGFE_HIP_CHECK(hipGetLastError());
    *out_keys = ki;
    *out_vals = vi;
    return MRGFE_OK;
}

// ------------------------------------------------------------------------------------------------------
// exclusive scan
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void scan_tile_sum_kernel(const uint32_t* __restrict__ in, const Slice* __restrict__ slices, uint32_t* __restrict__ blk)
{
    const Slice s = slices[blockIdx.y];
    if (blockIdx.x >= s.nblk) return;
    const uint32_t base = blockIdx.x * kTile;
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < kTile / 256; ++k) {
        uint32_t i = base + k * 256 + threadIdx.x;
        if (i < s.n) acc += in[s.off + i];
    }
    __shared__ uint32_t sw[4];
    acc = wave_sum(acc);
    if (lane_id() == 0) sw[wave_id()] = acc;
    __syncthreads();
    if (threadIdx.x == 0) blk[s.blk_off + blockIdx.x] = sw[0] + sw[1] + sw[2] + sw[3];
}

__global__ __launch_bounds__(256) void scan_tiles_kernel(const Slice* __restrict__ slices, uint32_t* __restrict__ blk, uint32_t* __restrict__ totals)
{
    const Slice s = slices[blockIdx.x];
    __shared__ uint32_t lds[8];
    uint32_t carry = 0;
    for (uint32_t b0 = 0; b0 < s.nblk; b0 += 256) {
        const uint32_t b = b0 + threadIdx.x;
        uint32_t v = b < s.nblk ? blk[s.blk_off + b] : 0u;
        uint32_t total;
        uint32_t ex = block_exclusive_scan<256>(v, lds, &total);
        if (b < s.nblk) blk[s.blk_off + b] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = carry;
}

__global__ __launch_bounds__(256) void scan_apply_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, const Slice* __restrict__ slices,
                                                          const uint32_t* __restrict__ blk)
{
    const Slice s = slices[blockIdx.y];
    if (blockIdx.x >= s.nblk) return;
    __shared__ uint32_t lds[8];
    // blocked arrangement: thread t owns 8 consecutive elements (two 16-byte accesses when fully inside the slice)
    const uint32_t first = blockIdx.x * kTile + threadIdx.x * 8;
    uint32_t v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = (first + k < s.n) ? in[s.off + first + k] : 0u;
    uint32_t tsum = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) { uint32_t t = v[k]; v[k] = tsum; tsum += t; }
    uint32_t total;
    const uint32_t pre = block_exclusive_scan<256>(tsum, lds, &total) + blk[s.blk_off + blockIdx.x];
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if (first + k < s.n) out[s.off + first + k] = v[k] + pre;
}

int exclusive_scan(mrgfe_ctx* ctx, const uint32_t* d_in, uint32_t* d_out, const Slice* d_slices, const SliceTable& t, uint32_t* d_blk, uint32_t* d_totals)
{
    if (t.nprob() == 0) return MRGFE_OK;
    dim3 grid(t.max_blks ? t.max_blks : 1, t.nprob());
    if (t.max_blks) hipLaunchKernelGGL(scan_tile_sum_kernel, grid, dim3(256), 0, ctx->stream, d_in, d_slices, d_blk);
    hipLaunchKernelGGL(scan_tiles_kernel, dim3(t.nprob()), dim3(256), 0, ctx->stream, d_slices, d_blk, d_totals);
    if (t.max_blks) hipLaunchKernelGGL(scan_apply_kernel, grid, dim3(256), 0, ctx->stream, d_in, d_out, d_slices, d_blk);
    MRGFE_HIP_CHECK(hipGetLastError());
    return MRGFE_OK;
}

// ------------------------------------------------------------------------------------------------------
// run heads
// ------------------------------------------------------------------------------------------------------
// Exclusive scan of the run-head flags of a sorted key array (flag i = 1 iff i starts a run of equal keys among the first n_valid elements; the
// invalid keys sort behind them) without a flag array: both scan kernels read the keys and compare neighbours.  (Round 2 marked the heads
// in a kernel of their own and scanned the flags: one launch and 4 bytes per element more.)

__global__ __launch_bounds__(256) void scan_heads_tile_sum_kernel(const uint32_t* __restrict__ keys, const Slice* __restrict__ slices, const uint32_t* __restrict__ n_valid, uint32_t* __restrict__ blk)
{
    const Slice s = slices[blockIdx.y];
    if (blockIdx.x >= s.nblk) return;
    const uint32_t base = blockIdx.x * kTile, nv = n_valid[blockIdx.y];
    const uint32_t* __restrict__ k0 = keys + s.off;
    // eight consecutive keys per lane (two 16-byte loads) and the one before them
    const uint32_t first = base + threadIdx.x * 8u;
    uint32_t kk[9];
    kk[0] = (first > 0 && first - 1 < s.n) ? k0[first - 1] : 0u;
    load8_u32(k0, first, s.n, 0u, kk + 1);
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) acc += (first + k < nv && (first + k == 0 || kk[k] != kk[k + 1])) ? 1u : 0u;
    __shared__ uint32_t sw[4];
    acc = wave_sum(acc);
    if (lane_id() == 0) sw[wave_id()] = acc;
    __syncthreads();
    if (threadIdx.x == 0) blk[s.blk_off + blockIdx.x] = sw[0] + sw[1] + sw[2] + sw[3];
}

__global__ __launch_bounds__(256) void scan_heads_apply_kernel(const uint32_t* __restrict__ keys, uint32_t* __restrict__ out, const Slice* __restrict__ slices, const uint32_t* __restrict__ n_valid,
                                                                const uint32_t* __restrict__ blk)
{
    const Slice s = slices[blockIdx.y];
    if (blockIdx.x >= s.nblk) return;
    __shared__ uint32_t lds[8];
    const uint32_t nv = n_valid[blockIdx.y];
    const uint32_t* __restrict__ k0 = keys + s.off;
    const uint32_t first = blockIdx.x * kTile + threadIdx.x * 8;
    uint32_t kk[9];  // the thread's eight keys and the one before them
    kk[0] = (first > 0 && first - 1 < s.n) ? k0[first - 1] : 0u;
#pragma unroll
    for (int k = 0; k < 8; ++k) kk[k + 1] = (first + k < s.n) ? k0[first + k] : 0u;
    uint32_t v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = (first + k < nv && (first + k == 0 || kk[k] != kk[k + 1])) ? 1u : 0u;
    uint32_t tsum = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) { uint32_t t = v[k]; v[k] = tsum; tsum += t; }
    uint32_t total;
    const uint32_t pre = block_exclusive_scan<256>(tsum, lds, &total) + blk[s.blk_off + blockIdx.x];
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if (first + k < s.n) out[s.off + first + k] = v[k] + pre;
}

int run_head_tile_counts(mrgfe_ctx* ctx, const uint32_t* d_sorted_keys, const Slice* d_slices, const SliceTable& t, const uint32_t* d_n_valid, uint32_t* d_blk)
{
    if (t.nprob() == 0 || t.max_blks == 0) return MRGFE_OK;
    hipLaunchKernelGGL(scan_heads_tile_sum_kernel, dim3(t.max_blks, t.nprob()), dim3(256), 0, ctx->stream, d_sorted_keys, d_slices, d_n_valid, d_blk);
    MRGFE_HIP_CHECK(hipGetLastError());
    return MRGFE_OK;
}

int tile_sums(mrgfe_ctx* ctx, const uint32_t* d_in, const Slice* d_slices, const SliceTable& t, uint32_t* d_blk)
{
    if (t.nprob() == 0 || t.max_blks == 0) return MRGFE_OK;
    hipLaunchKernelGGL(scan_tile_sum_kernel, dim3(t.max_blks, t.nprob()), dim3(256), 0, ctx->stream, d_in, d_slices, d_blk);
    MRGFE_HIP_CHECK(hipGetLastError());
    return MRGFE_OK;
}

int exclusive_scan_run_heads(mrgfe_ctx* ctx, const uint32_t* d_sorted_keys, uint32_t* d_out, const Slice* d_slices, const SliceTable& t, const uint32_t* d_n_valid, uint32_t* d_blk,
                             uint32_t* d_totals)
{
    if (t.nprob() == 0) return MRGFE_OK;
    dim3 grid(t.max_blks ? t.max_blks : 1, t.nprob());
    if (t.max_blks) hipLaunchKernelGGL(scan_heads_tile_sum_kernel, grid, dim3(256), 0, ctx->stream, d_sorted_keys, d_slices, d_n_valid, d_blk);
    hipLaunchKernelGGL(scan_tiles_kernel, dim3(t.nprob()), dim3(256), 0, ctx->stream, d_slices, d_blk, d_totals);
    // d_out == nullptr: the caller wants the run count and the tiles' prefixes in d_blk only (ndt_launch_segments derives each head's ordinal from
    // them on the fly: no per-element ordinal array is written or read)
    if (t.max_blks && d_out) hipLaunchKernelGGL(scan_heads_apply_kernel, grid, dim3(256), 0, ctx->stream, d_sorted_keys, d_out, d_slices, d_n_valid, d_blk);
    MRGFE_HIP_CHECK(hipGetLastError());
    return MRGFE_OK;
}

}  // namespace mrgfe
