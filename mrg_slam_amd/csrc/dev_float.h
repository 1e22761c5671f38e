// csrc/dev_float.h — float helpers whose operation ORDER is part of the contract with the CPU oracle: they are never
// contracted into FMAs (`#pragma clang fp contract(off)`), so the GPU rounds exactly where PCL / Eigen / FLANN do.
#pragma once
#include <hip/hip_runtime.h>

namespace mrgfe {

// (a0*b0 + a1*b1) + a2*b2
__device__ __forceinline__ float dot3f(float a0, float b0, float a1, float b1, float a2, float b2)
{
#pragma clang fp contract(off)
    const float p0 = a0 * b0, p1 = a1 * b1, p2 = a2 * b2;
    const float s = p0 + p1;
    return s + p2;
}

// NDT three-term products, accumulated left to right with fused multiply-adds: s = a0*b0; s = fma(a1,b1,s); s = fma(a2,b2,s).
// The CPU oracle runs the same sequence (oracle/ndt.cpp dot3f<true>): results are bit-identical.
__device__ __forceinline__ float fdot3f(float a0, float b0, float a1, float b1, float a2, float b2)
{
#pragma clang fp contract(off)
    const float s0 = a0 * b0;
    return __builtin_fmaf(a2, b2, __builtin_fmaf(a1, b1, s0));
}
__device__ __forceinline__ double fdot3d(double a0, double b0, double a1, double b1, double a2, double b2)
{
#pragma clang fp contract(off)
    const double s0 = a0 * b0;
    return __builtin_fma(a2, b2, __builtin_fma(a1, b1, s0));
}

// the same sum when the first product is a structural zero (0 * finite = +-0, and fma(a, b, +-0) = round(a * b)): equal
// to fdot3d(x, 0, a1, b1, a2, b2) for finite x except for the sign of an exact zero result
__device__ __forceinline__ double fdot3d_z(double a1, double b1, double a2, double b2)
{
#pragma clang fp contract(off)
    const double s1 = a1 * b1;
    return __builtin_fma(a2, b2, s1);
}

// pcl::transformPointCloud float path (pcl::detail::Transformer<float>::se3): x' = m0*x + (m1*y + (m2*z + m3)).
// T: row-major 3x4.
__device__ __forceinline__ void transform_point(const float* __restrict__ T, float x, float y, float z, float& ox, float& oy, float& oz)
{
#pragma clang fp contract(off)
    float a;
    a = T[2] * z;  float t0 = a + T[3];  a = T[1] * y; t0 = a + t0; a = T[0] * x; ox = a + t0;
    a = T[6] * z;  float t1 = a + T[7];  a = T[5] * y; t1 = a + t1; a = T[4] * x; oy = a + t1;
    a = T[10] * z; float t2 = a + T[11]; a = T[9] * y; t2 = a + t2; a = T[8] * x; oz = a + t2;
}

// FLANN L2_Simple<float>: ((dx*dx + dy*dy) + dz*dz)
__device__ __forceinline__ float sqdist3f(float ax, float ay, float az, float bx, float by, float bz)
{
#pragma clang fp contract(off)
    const float dx = ax - bx, dy = ay - by, dz = az - bz;
    const float xx = dx * dx, yy = dy * dy, zz = dz * dz;
    const float s = xx + yy;
    return s + zz;
}

}  // namespace mrgfe
