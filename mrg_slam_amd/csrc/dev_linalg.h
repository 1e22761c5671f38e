// csrc/dev_linalg.h — small dense f64 linear algebra for device (and host) code: symmetric 3x3 eigen-decomposition,
// 3x3 inverse. Used once per target voxel (pclomp::VoxelGridCovariance second pass) and per GICP covariance.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

namespace mrgfe {

// Cyclic Jacobi on a symmetric 3x3 (lower triangle read). w ascending, V columns = eigenvectors (row-major V[r*3+c]).
__host__ __device__ inline void dl_sym_eig3(const double A[9], double w[3], double V[9])
{
    double a00 = A[0], a11 = A[4], a22 = A[8], a01 = A[3], a02 = A[6], a12 = A[7];
    double v[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int sweep = 0; sweep < 32; ++sweep) {
        const double off = fabs(a01) + fabs(a02) + fabs(a12);
        const double dia = fabs(a00) + fabs(a11) + fabs(a22);
        if (off <= 1e-300 || off <= 2.2e-19 * dia) break;
        // rotation (0,1)
        if (a01 != 0.0) {
            double th = (a11 - a00) / (2.0 * a01);
            double t = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
            double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
            double n00 = a00 - t * a01, n11 = a11 + t * a01;
            double n02 = c * a02 - s * a12, n12 = s * a02 + c * a12;
            a00 = n00; a11 = n11; a01 = 0.0; a02 = n02; a12 = n12;
            for (int k = 0; k < 3; ++k) { double p = v[k * 3 + 0], q = v[k * 3 + 1]; v[k * 3 + 0] = c * p - s * q; v[k * 3 + 1] = s * p + c * q; }
        }
        // rotation (0,2)
        if (a02 != 0.0) {
            double th = (a22 - a00) / (2.0 * a02);
            double t = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
            double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
            double n00 = a00 - t * a02, n22 = a22 + t * a02;
            double n01 = c * a01 - s * a12, n12 = s * a01 + c * a12;
            a00 = n00; a22 = n22; a02 = 0.0; a01 = n01; a12 = n12;
            for (int k = 0; k < 3; ++k) { double p = v[k * 3 + 0], q = v[k * 3 + 2]; v[k * 3 + 0] = c * p - s * q; v[k * 3 + 2] = s * p + c * q; }
        }
        // rotation (1,2)
        if (a12 != 0.0) {
            double th = (a22 - a11) / (2.0 * a12);
            double t = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
            double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
            double n11 = a11 - t * a12, n22 = a22 + t * a12;
            double n01 = c * a01 - s * a02, n02 = s * a01 + c * a02;
            a11 = n11; a22 = n22; a12 = 0.0; a01 = n01; a02 = n02;
            for (int k = 0; k < 3; ++k) { double p = v[k * 3 + 1], q = v[k * 3 + 2]; v[k * 3 + 1] = c * p - s * q; v[k * 3 + 2] = s * p + c * q; }
        }
    }
    double d[3] = {a00, a11, a22};
    int i0 = 0, i1 = 1, i2 = 2;
    if (d[i0] > d[i1]) { int t = i0; i0 = i1; i1 = t; }
    if (d[i1] > d[i2]) { int t = i1; i1 = i2; i2 = t; }
    if (d[i0] > d[i1]) { int t = i0; i0 = i1; i1 = t; }
    const int idx[3] = {i0, i1, i2};
    for (int c = 0; c < 3; ++c) {
        w[c] = d[idx[c]];
        for (int r = 0; r < 3; ++r) V[r * 3 + c] = v[r * 3 + idx[c]];
    }
}

__host__ __device__ inline void dl_inv3(const double m[9], double out[9])
{
    const double c00 = m[4] * m[8] - m[5] * m[7];
    const double c10 = m[5] * m[6] - m[3] * m[8];
    const double c20 = m[3] * m[7] - m[4] * m[6];
    const double det = m[0] * c00 + m[1] * c10 + m[2] * c20;
    const double id = 1.0 / det;
    out[0] = c00 * id;
    out[1] = (m[2] * m[7] - m[1] * m[8]) * id;
    out[2] = (m[1] * m[5] - m[2] * m[4]) * id;
    out[3] = c10 * id;
    out[4] = (m[0] * m[8] - m[2] * m[6]) * id;
    out[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    out[6] = c20 * id;
    out[7] = (m[1] * m[6] - m[0] * m[7]) * id;
    out[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}

__host__ __device__ inline void dl_mul3(const double a[9], const double b[9], double out[9])
{
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) out[r * 3 + c] = a[r * 3 + 0] * b[0 * 3 + c] + a[r * 3 + 1] * b[1 * 3 + c] + a[r * 3 + 2] * b[2 * 3 + c];
}

}  // namespace mrgfe
