// csrc/bbox_device.h — device-side pieces of the bounding-box pass (cellsort.hip) for kernels that have the points in registers anyway:
// a workgroup of 256 threads merges its threads' boxes, or a list of partial boxes, into one.  min / max / integer count: the result does not
// depend on how the points were cut into partial boxes.
#pragma once
#include "cellsort.h"
#include "dev_utils.h"

namespace mrgfe {

struct BoxAcc {
    float    mn[3], mx[3];
    uint32_t cnt;
    __device__ __forceinline__ void init()
    {
        mn[0] = mn[1] = mn[2] = INFINITY;
        mx[0] = mx[1] = mx[2] = -INFINITY;
        cnt = 0;
    }
    __device__ __forceinline__ void add(const float4& p)  // pcl::getMinMax3D: finite points only
    {
        if (finite3(p.x, p.y, p.z)) {
            mn[0] = fminf(mn[0], p.x); mn[1] = fminf(mn[1], p.y); mn[2] = fminf(mn[2], p.z);
            mx[0] = fmaxf(mx[0], p.x); mx[1] = fmaxf(mx[1], p.y); mx[2] = fmaxf(mx[2], p.z);
            ++cnt;
        }
    }
    __device__ __forceinline__ void add(const BBox& q)
    {
        for (int a = 0; a < 3; ++a) { mn[a] = fminf(mn[a], q.mn[a]); mx[a] = fmaxf(mx[a], q.mx[a]); }
        cnt += q.n_finite;
    }
};

// all 256 threads of the workgroup call it; the merged box is returned to every thread
__device__ __forceinline__ BBox block_merge_box(BoxAcc a)
{
    __shared__ float    s_mn[4][3], s_mx[4][3];
    __shared__ uint32_t s_cnt[4];
    __shared__ BBox     s_out;
#pragma unroll
    for (int k = 0; k < 3; ++k) { a.mn[k] = wave_min(a.mn[k]); a.mx[k] = wave_max(a.mx[k]); }
    a.cnt = wave_sum(a.cnt);
    if (lane_id() == 0) {
        for (int k = 0; k < 3; ++k) { s_mn[wave_id()][k] = a.mn[k]; s_mx[wave_id()][k] = a.mx[k]; }
        s_cnt[wave_id()] = a.cnt;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        BBox b;
        for (int k = 0; k < 3; ++k) {
            b.mn[k] = fminf(fminf(s_mn[0][k], s_mn[1][k]), fminf(s_mn[2][k], s_mn[3][k]));
            b.mx[k] = fmaxf(fmaxf(s_mx[0][k], s_mx[1][k]), fmaxf(s_mx[2][k], s_mx[3][k]));
        }
        b.n_finite = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        b.pad = 0;
        s_out = b;
    }
    __syncthreads();
    const BBox r = s_out;
    __syncthreads();  // (the arrays may be reused by a second call)
    return r;
}

__device__ __forceinline__ BBox block_merge_partials(const BBox* __restrict__ partial, uint32_t n)
{
    BoxAcc a;
    a.init();
    for (uint32_t b = threadIdx.x; b < n; b += 256) a.add(partial[b]);
    return block_merge_box(a);
}

}  // namespace mrgfe
