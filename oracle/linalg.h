// oracle/linalg.h — TEST INFRASTRUCTURE (CPU oracle). Not part of the shipped product path.
//
// Tiny dependency-free linear algebra standing in for the Eigen 3.4 routines the upstream libraries call
// (Eigen is not installed here; PARITY UNPINNED — see quirks.h). Row-major storage: M[r*C + c].
//  * sym_eig3        ~ Eigen::SelfAdjointEigenSolver<Matrix3d>::compute (ascending eigenvalues)
//  * inv3            ~ Eigen 3x3 cofactor inverse (Matrix3d::inverse)
//  * JacobiSvd6      ~ Eigen::JacobiSVD<Matrix<double,6,6>>(FullU|FullV) + solve()
//  * euler_xyz_f     ~ Matrix3f::eulerAngles(0,1,2)
//  * pose_to_matrix_f~ (Translation3f * AngleAxisf(X) * AngleAxisf(Y) * AngleAxisf(Z)).matrix()
#pragma once
#include <cfloat>
#include <cmath>
#include <cstring>
#include <limits>
#include <algorithm>

namespace orc {

// ------------------------------------------------------------------------------------------------------
// 3x3 symmetric eigen-decomposition (cyclic Jacobi, double). evals ascending; evecs columns = eigenvectors
// (row-major 3x3: evecs[r*3+c] is component r of eigenvector c). Only the lower triangle of A is read, as
// SelfAdjointEigenSolver does.
inline void sym_eig3(const double A[9], double evals[3], double evecs[9])
{
    double a[3][3];
    a[0][0] = A[0]; a[1][1] = A[4]; a[2][2] = A[8];
    a[1][0] = a[0][1] = A[3];
    a[2][0] = a[0][2] = A[6];
    a[2][1] = a[1][2] = A[7];
    double v[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 64; ++sweep) {
        double off = std::fabs(a[0][1]) + std::fabs(a[0][2]) + std::fabs(a[1][2]);
        double diag = std::fabs(a[0][0]) + std::fabs(a[1][1]) + std::fabs(a[2][2]);
        if (off <= 1e-300 || off <= DBL_EPSILON * 1e-3 * diag) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (a[p][q] == 0.0) continue;
                double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
                double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; ++k) {  // A <- A * J
                    double akp = a[k][p], akq = a[k][q];
                    a[k][p] = c * akp - s * akq;
                    a[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < 3; ++k) {  // A <- J^T * A
                    double apk = a[p][k], aqk = a[q][k];
                    a[p][k] = c * apk - s * aqk;
                    a[q][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 3; ++k) {
                    double vkp = v[k][p], vkq = v[k][q];
                    v[k][p] = c * vkp - s * vkq;
                    v[k][q] = s * vkp + c * vkq;
                }
            }
    }
    int idx[3] = {0, 1, 2};
    double d[3] = {a[0][0], a[1][1], a[2][2]};
    std::sort(idx, idx + 3, [&](int i, int j) { return d[i] < d[j]; });
    for (int c = 0; c < 3; ++c) {
        evals[c] = d[idx[c]];
        for (int r = 0; r < 3; ++r) evecs[r * 3 + c] = v[r][idx[c]];
    }
}

// 3x3 general inverse via cofactors (adjugate / det), the structure Eigen uses for fixed 3x3.
inline void inv3(const double m[9], double out[9])
{
    double c00 = m[4] * m[8] - m[5] * m[7];
    double c10 = m[5] * m[6] - m[3] * m[8];  // cofactor of (1,0) wrt column 0 expansion pieces
    double c20 = m[3] * m[7] - m[4] * m[6];
    double det = m[0] * c00 + m[1] * c10 + m[2] * c20;
    double id = 1.0 / det;
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

inline void mul3(const double a[9], const double b[9], double out[9])
{
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) out[r * 3 + c] = a[r * 3 + 0] * b[0 * 3 + c] + a[r * 3 + 1] * b[1 * 3 + c] + a[r * 3 + 2] * b[2 * 3 + c];
}

// ------------------------------------------------------------------------------------------------------
// Two-sided Jacobi SVD of a real 6x6 (the algorithm of Eigen::JacobiSVD for square real matrices: no QR
// preconditioner, sweeps over (p,q) with p=1..5,q=0..p-1, real_2x2_jacobi_svd per pair, threshold
// 2*eps*maxDiagEntry), then solve() = V * diag(1/s_i, i<rank) * U^T * b with rank from
// s_i > max(s_0 * 6*eps, DBL_MIN).
struct JacobiSvd6 {
    double U[36], V[36], W[36], S[6];
    int rank_;

    static void make_jacobi(double x, double y, double z, double& c, double& s)
    {
        double deno = 2.0 * std::fabs(y);
        if (deno < std::numeric_limits<double>::min()) { c = 1; s = 0; return; }
        double tau = (x - z) / deno;
        double w = std::sqrt(tau * tau + 1.0);
        double t = (tau > 0) ? 1.0 / (tau + w) : 1.0 / (tau - w);
        double sign_t = t > 0 ? 1.0 : -1.0;
        double n = 1.0 / std::sqrt(t * t + 1.0);
        s = -sign_t * (y / std::fabs(y)) * std::fabs(t) * n;
        c = n;
    }
    // rows p,q of M <- J^T applied on the left with Eigen's convention applyOnTheLeft(p,q,j):
    //   row_p' = c*row_p + s*row_q ; row_q' = -s*row_p + c*row_q   (for j = (c,s), "adjoint" applied)
    static void rot_left(double* M, int p, int q, double c, double s)
    {
        for (int k = 0; k < 6; ++k) {
            double xp = M[p * 6 + k], xq = M[q * 6 + k];
            M[p * 6 + k] = c * xp + s * xq;
            M[q * 6 + k] = -s * xp + c * xq;
        }
    }
    // cols p,q of M <- applyOnTheRight(p,q,j): col_p' = c*col_p - s*col_q ; col_q' = s*col_p + c*col_q
    static void rot_right(double* M, int p, int q, double c, double s)
    {
        for (int k = 0; k < 6; ++k) {
            double xp = M[k * 6 + p], xq = M[k * 6 + q];
            M[k * 6 + p] = c * xp - s * xq;
            M[k * 6 + q] = s * xp + c * xq;
        }
    }

    void compute(const double A[36])
    {
        double scale = 0;
        for (int i = 0; i < 36; ++i) scale = std::max(scale, std::fabs(A[i]));
        for (int i = 0; i < 36; ++i) if (!std::isfinite(A[i])) scale = std::numeric_limits<double>::quiet_NaN();
        if (!std::isfinite(scale)) { rank_ = -1; return; }  // Eigen: m_info = InvalidInput
        if (scale == 0) scale = 1;
        for (int i = 0; i < 36; ++i) { W[i] = A[i] / scale; U[i] = V[i] = 0; }
        for (int i = 0; i < 6; ++i) U[i * 6 + i] = V[i * 6 + i] = 1;
        const double precision = 2.0 * DBL_EPSILON;
        const double consider_zero = std::numeric_limits<double>::min();
        double max_diag = 0;
        for (int i = 0; i < 6; ++i) max_diag = std::max(max_diag, std::fabs(W[i * 6 + i]));
        bool finished = false;
        int guard = 0;
        while (!finished && guard++ < 200) {
            finished = true;
            for (int p = 1; p < 6; ++p)
                for (int q = 0; q < p; ++q) {
                    double thr = std::max(consider_zero, precision * max_diag);
                    if (std::fabs(W[p * 6 + q]) > thr || std::fabs(W[q * 6 + p]) > thr) {
                        finished = false;
                        // real_2x2_jacobi_svd on [[W(p,p) W(p,q)],[W(q,p) W(q,q)]]
                        double m00 = W[p * 6 + p], m01 = W[p * 6 + q], m10 = W[q * 6 + p], m11 = W[q * 6 + q];
                        double t = m00 + m11, d = m10 - m01;
                        double c1, s1;
                        if (std::fabs(d) < std::numeric_limits<double>::min()) { s1 = 0; c1 = 1; }
                        else { double u = t / d; double tmp = std::sqrt(1.0 + u * u); s1 = 1.0 / tmp; c1 = u / tmp; }
                        // m.applyOnTheLeft(0,1,rot1)
                        double n00 = c1 * m00 + s1 * m10, n01 = c1 * m01 + s1 * m11;
                        double n11 = -s1 * m01 + c1 * m11;
                        double cr, sr;
                        make_jacobi(n00, n01, n11, cr, sr);
                        // j_left = rot1 * j_right^T  (composition of rotations (c,s)*(c',-s'))
                        double cl = c1 * cr + s1 * sr;  // real part
                        double sl = s1 * cr - c1 * sr;
                        rot_left(W, p, q, cl, sl);
                        // U.applyOnTheRight(p,q,j_left.transpose()) : transpose = (c,-s)
                        rot_right(U, p, q, cl, -sl);
                        rot_right(W, p, q, cr, sr);
                        rot_right(V, p, q, cr, sr);
                        max_diag = std::max(max_diag, std::max(std::fabs(W[p * 6 + p]), std::fabs(W[q * 6 + q])));
                    }
                }
        }
        for (int i = 0; i < 6; ++i) {
            double a = W[i * 6 + i];
            S[i] = std::fabs(a);
            if (a < 0) for (int k = 0; k < 6; ++k) U[k * 6 + i] = -U[k * 6 + i];
        }
        for (int i = 0; i < 6; ++i) {  // sort descending, swapping columns of U and V
            int best = i;
            for (int k = i + 1; k < 6; ++k) if (S[k] > S[best]) best = k;
            if (S[best] == 0) break;
            if (best != i) {
                std::swap(S[i], S[best]);
                for (int k = 0; k < 6; ++k) { std::swap(U[k * 6 + i], U[k * 6 + best]); std::swap(V[k * 6 + i], V[k * 6 + best]); }
            }
        }
        for (int i = 0; i < 6; ++i) S[i] *= scale;
        double thr = std::max(S[0] * 6.0 * DBL_EPSILON, std::numeric_limits<double>::min());
        rank_ = 0;
        for (int i = 0; i < 6; ++i) if (S[i] > thr) ++rank_;
    }
    void solve(const double b[6], double x[6]) const
    {
        double tmp[6];
        if (rank_ < 0) { for (int i = 0; i < 6; ++i) x[i] = std::numeric_limits<double>::quiet_NaN(); return; }
        for (int i = 0; i < 6; ++i) {
            double acc = 0;
            for (int k = 0; k < 6; ++k) acc += U[k * 6 + i] * b[k];
            tmp[i] = (i < rank_) ? acc / S[i] : 0.0;
        }
        for (int r = 0; r < 6; ++r) {
            double acc = 0;
            for (int i = 0; i < rank_; ++i) acc += V[r * 6 + i] * tmp[i];
            x[r] = acc;
        }
    }
};

// ------------------------------------------------------------------------------------------------------
// float 4x4 helpers, row-major M[r*4+c].
// Rotation of the Umeyama / Kabsch problem: R = U diag(1, 1, det(U) det(V)) V^T for sigma = U S V^T, through a one-sided
// (Hestenes) Jacobi SVD in f64.  A vanishing singular direction is completed by the cross product of the other two.
inline void umeyama_rotation(const double sigma[9], double R[9])
{
    double W[3][3], V[3][3];  // W = sigma V converges to U S
    double scale = 0;
    for (int i = 0; i < 9; ++i) scale = std::max(scale, std::fabs(sigma[i]));
    if (!(scale > 0) || !std::isfinite(scale)) { for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0) ? 1.0 : 0.0; return; }
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { W[r][c] = sigma[r * 3 + c] / scale; V[r][c] = r == c ? 1.0 : 0.0; }
    for (int sweep = 0; sweep < 60; ++sweep) {
        bool rotated = false;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double alpha = 0, beta = 0, gamma = 0;
                for (int k = 0; k < 3; ++k) { alpha += W[k][p] * W[k][p]; beta += W[k][q] * W[k][q]; gamma += W[k][p] * W[k][q]; }
                if (gamma == 0.0 || std::fabs(gamma) <= 1e-15 * std::sqrt(alpha * beta)) continue;
                rotated = true;
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / std::sqrt(1.0 + t * t), s = c * t;
                for (int k = 0; k < 3; ++k) {
                    const double wp = W[k][p], wq = W[k][q];
                    W[k][p] = c * wp - s * wq; W[k][q] = s * wp + c * wq;
                    const double vp = V[k][p], vq = V[k][q];
                    V[k][p] = c * vp - s * vq; V[k][q] = s * vp + c * vq;
                }
            }
        if (!rotated) break;
    }
    double sig[3], U[3][3];
    int    order[3] = {0, 1, 2};
    for (int j = 0; j < 3; ++j) sig[j] = std::sqrt(W[0][j] * W[0][j] + W[1][j] * W[1][j] + W[2][j] * W[2][j]);
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2 - a; ++b) if (sig[order[b]] < sig[order[b + 1]]) std::swap(order[b], order[b + 1]);  // descending
    const double tiny = 1e-12 * sig[order[0]];
    for (int jj = 0; jj < 3; ++jj) {
        const int j = order[jj];
        if (sig[j] > tiny) { for (int k = 0; k < 3; ++k) U[k][j] = W[k][j] / sig[j]; }
        else if (jj == 2) {  // rank 2: complete the basis
            const int a = order[0], b = order[1];
            U[0][j] = U[1][a] * U[2][b] - U[2][a] * U[1][b];
            U[1][j] = U[2][a] * U[0][b] - U[0][a] * U[2][b];
            U[2][j] = U[0][a] * U[1][b] - U[1][a] * U[0][b];
        } else { for (int k = 0; k < 3; ++k) U[k][j] = V[k][j]; }  // rank <= 1: no unique answer; stay finite
    }
    auto det3 = [](const double M[3][3]) { return M[0][0] * (M[1][1] * M[2][2] - M[1][2] * M[2][1]) - M[0][1] * (M[1][0] * M[2][2] - M[1][2] * M[2][0]) + M[0][2] * (M[1][0] * M[2][1] - M[1][1] * M[2][0]); };
    double S[3] = {1, 1, 1};
    if (det3(U) * det3(V) < 0) S[order[2]] = -1;  // the smallest singular direction flips
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            double s = 0;
            for (int j = 0; j < 3; ++j) s += U[r][j] * S[j] * V[c][j];
            R[r * 3 + c] = s;
        }
}

inline void mat4f_identity(float M[16]) { for (int i = 0; i < 16; ++i) M[i] = (i % 5 == 0) ? 1.0f : 0.0f; }
inline bool mat4f_is_identity(const float M[16])
{
    for (int i = 0; i < 16; ++i) if (M[i] != ((i % 5 == 0) ? 1.0f : 0.0f)) return false;
    return true;
}
inline void colmajor_to_rowmajor4(const float in[16], float out[16]) { for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) out[r * 4 + c] = in[c * 4 + r]; }
inline void rowmajor_to_colmajor4(const float in[16], float out[16]) { for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) out[c * 4 + r] = in[r * 4 + c]; }

// pcl::transformPointCloud float path (pcl::detail::Transformer<float>::se3, SSE form):
//   out = c0*x + (c1*y + (c2*z + c3)), no FMA.
inline void transform_point_f(const float M[16], float x, float y, float z, float& ox, float& oy, float& oz)
{
    float a;  // volatile stores keep gcc from reassociating/contracting
    a = M[2] * z;  float t0 = a + M[3];  a = M[1] * y;  t0 = a + t0;  a = M[0] * x;  ox = a + t0;
    a = M[6] * z;  float t1 = a + M[7];  a = M[5] * y;  t1 = a + t1;  a = M[4] * x;  oy = a + t1;
    a = M[10] * z; float t2 = a + M[11]; a = M[9] * y;  t2 = a + t2;  a = M[8] * x;  oz = a + t2;
}

// Matrix3f::eulerAngles(0,1,2) of the rotation block of a row-major 4x4 (Eigen 3.4 EulerAngles.h).
inline void euler_xyz_f(const float M[16], float res[3])
{
    auto m = [&](int r, int c) { return M[r * 4 + c]; };
    const float kPi = 3.14159265358979323846f;
    res[0] = std::atan2(m(1, 2), m(2, 2));
    float c2 = std::sqrt(m(0, 0) * m(0, 0) + m(0, 1) * m(0, 1));
    if (res[0] > 0.0f) {
        res[0] -= kPi;
        res[1] = std::atan2(-m(0, 2), -c2);
    } else {
        res[1] = std::atan2(-m(0, 2), c2);
    }
    float s1 = std::sin(res[0]);
    float c1 = std::cos(res[0]);
    res[2] = std::atan2(s1 * m(2, 0) - c1 * m(1, 0), c1 * m(1, 1) - s1 * m(2, 1));
    res[0] = -res[0]; res[1] = -res[1]; res[2] = -res[2];
}

// AngleAxisf(angle, unit axis k)::toRotationMatrix() (Eigen AngleAxis.h) for axis = e_k.
inline void angle_axis_unit_f(float angle, int axis, float R[9])
{
    float ax[3] = {0, 0, 0}; ax[axis] = 1.0f;
    float sn = std::sin(angle), c = std::cos(angle);
    float sin_axis[3] = {sn * ax[0], sn * ax[1], sn * ax[2]};
    float cos1_axis[3] = {(1.0f - c) * ax[0], (1.0f - c) * ax[1], (1.0f - c) * ax[2]};
    float tmp;
    tmp = cos1_axis[0] * ax[1]; R[0 * 3 + 1] = tmp - sin_axis[2]; R[1 * 3 + 0] = tmp + sin_axis[2];
    tmp = cos1_axis[0] * ax[2]; R[0 * 3 + 2] = tmp + sin_axis[1]; R[2 * 3 + 0] = tmp - sin_axis[1];
    tmp = cos1_axis[1] * ax[2]; R[1 * 3 + 2] = tmp - sin_axis[0]; R[2 * 3 + 1] = tmp + sin_axis[0];
    R[0] = cos1_axis[0] * ax[0] + c; R[4] = cos1_axis[1] * ax[1] + c; R[8] = cos1_axis[2] * ax[2] + c;
}
inline void mul3f(const float a[9], const float b[9], float out[9])
{
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            float p0 = a[r * 3 + 0] * b[0 * 3 + c], p1 = a[r * 3 + 1] * b[1 * 3 + c], p2 = a[r * 3 + 2] * b[2 * 3 + c];
            float s = p0 + p1; out[r * 3 + c] = s + p2;
        }
}
// (Translation3f(p0,p1,p2) * AngleAxisf(p3,X) * AngleAxisf(p4,Y) * AngleAxisf(p5,Z)).matrix(), row-major 4x4.
inline void pose_to_matrix_f(const double p[6], float M[16])
{
    float Rx[9], Ry[9], Rz[9], Rxy[9], R[9];
    angle_axis_unit_f(static_cast<float>(p[3]), 0, Rx);
    angle_axis_unit_f(static_cast<float>(p[4]), 1, Ry);
    angle_axis_unit_f(static_cast<float>(p[5]), 2, Rz);
    mul3f(Rx, Ry, Rxy);
    mul3f(Rxy, Rz, R);
    for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) M[r * 4 + c] = R[r * 3 + c]; M[r * 4 + 3] = static_cast<float>(p[r]); }
    M[12] = M[13] = M[14] = 0.0f; M[15] = 1.0f;
}

}  // namespace orc
