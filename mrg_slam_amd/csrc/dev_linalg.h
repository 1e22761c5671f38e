// csrc/dev_linalg.h — small dense f64 linear algebra for device (and host) code: symmetric 3x3 eigen-decomposition,
// 3x3 inverse. Used once per target voxel (pclomp::VoxelGridCovariance second pass) and per GICP covariance.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

namespace mrgfe {

// Cyclic Jacobi on a symmetric 3x3 (lower triangle read). w ascending, V columns = eigenvectors (row-major V[r*3+c]).
// The operation sequence (rotation formulas, two-sided updates, sweep test, stable ascending order) is fixed: with FMA
// contraction off every f64 operation rounds once, so a CPU running the same sequence produces the same bits.
__host__ __device__ inline void dl_sym_eig3(const double A[9], double w[3], double V[9])
{
    double a[3][3];
    a[0][0] = A[0]; a[1][1] = A[4]; a[2][2] = A[8];
    a[1][0] = a[0][1] = A[3];
    a[2][0] = a[0][2] = A[6];
    a[2][1] = a[1][2] = A[7];
    double v[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 64; ++sweep) {
        const double off = fabs(a[0][1]) + fabs(a[0][2]) + fabs(a[1][2]);
        const double diag = fabs(a[0][0]) + fabs(a[1][1]) + fabs(a[2][2]);
        if (off <= 1e-300 || off <= 2.220446049250313e-16 * 1e-3 * diag) break;
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int q = p + 1; q < 3; ++q) {
                if (a[p][q] == 0.0) continue;
                const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
                for (int k = 0; k < 3; ++k) {  // A <- A * J
                    const double akp = a[k][p], akq = a[k][q];
                    a[k][p] = c * akp - s * akq;
                    a[k][q] = s * akp + c * akq;
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) {  // A <- J^T * A
                    const double apk = a[p][k], aqk = a[q][k];
                    a[p][k] = c * apk - s * aqk;
                    a[q][k] = s * apk + c * aqk;
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const double vkp = v[k][p], vkq = v[k][q];
                    v[k][p] = c * vkp - s * vkq;
                    v[k][q] = s * vkp + c * vkq;
                }
            }
    }
    const double d[3] = {a[0][0], a[1][1], a[2][2]};
    int i0 = 0, i1 = 1, i2 = 2;  // stable ascending order of three values
    if (d[i1] < d[i0]) { const int t = i0; i0 = i1; i1 = t; }
    if (d[i2] < d[i1]) { const int t = i1; i1 = i2; i2 = t; }
    if (d[i1] < d[i0]) { const int t = i0; i0 = i1; i1 = t; }
    const int idx[3] = {i0, i1, i2};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        w[c] = d[idx[c]];
#pragma unroll
        for (int r = 0; r < 3; ++r) V[r * 3 + c] = v[r][idx[c]];
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
