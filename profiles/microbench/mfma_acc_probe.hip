// Can the matrix pipe do the pair loop's f64 accumulate-adds?  v_mfma_f64_4x4x4f64 with B = identity is D[i][j] = C[i][j] + A[i][j] (+ three
// products with 0): a per-lane `acc += a` issued beside the VALU stream.  This probe finds (1) the lane pattern of B that makes it so and whether the
// result equals the f64 add bit for bit, (2) what EXEC-masked lanes do, (3) the time of a loop shaped like the derivative kernel's pair body —
// 300 f32 FMAs, 43 f32->f64 converts and 43 f64 accumulates — with the accumulates as v_add_f64 and as MFMAs, at 3 wavefronts per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_acc_probe mfma_acc_probe.hip && ./mfma_acc_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include <random>

__global__ void sem_kernel(const double* a, const double* b, const double* c, double* d, int masked)
{
    const int l = threadIdx.x;
    double r = c[l];
    if (!masked || l < 32) r = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], r, 0, 0, 0);
    d[l] = r;
}

constexpr int kAcc = 43;
template <bool MFMA>
__global__ __launch_bounds__(256, 3) void body_kernel(float* out, float seed, int iters, double bsel_in)
{
    float  x[8];
    for (int k = 0; k < 8; ++k) x[k] = seed + threadIdx.x * 1e-3f + k;
    double acc[kAcc];
    for (int k = 0; k < kAcc; ++k) acc[k] = 0.0;
    const int    l = threadIdx.x & 63;
    const double bsel = (((l >> 2) & 3) == ((l >> 4) & 3) || true) ? bsel_in : 0.0;  // (timing only: the value does not matter)
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < kAcc; ++k) {
            // ~7 f32 FMAs per accumulator: 300 per body
#pragma unroll
            for (int u = 0; u < 7; ++u) x[(k + u) & 7] = __builtin_fmaf(x[(k + u) & 7], 0.999f, x[(k + u + 1) & 7] * 1e-3f);
            const double t = static_cast<double>(x[k & 7]);
            if (MFMA) acc[k] = __builtin_amdgcn_mfma_f64_4x4x4f64(t, bsel, acc[k], 0, 0, 0);
            else      acc[k] += t;
        }
    }
    double s = 0;
    for (int k = 0; k < kAcc; ++k) s += acc[k];
    out[blockIdx.x * 256 + threadIdx.x] = static_cast<float>(s) + x[0];
}

int main()
{
    // ---- semantics
    std::mt19937_64 rng(5);
    std::uniform_real_distribution<double> U(-1e6, 1e6);
    std::vector<double> a(64), c(64), b(64), d(64);
    for (int l = 0; l < 64; ++l) { a[l] = U(rng) * 1e-7; c[l] = U(rng); }
    double *da, *db, *dc, *dd;
    hipMalloc(&da, 512); hipMalloc(&db, 512); hipMalloc(&dc, 512); hipMalloc(&dd, 512);
    hipMemcpy(da, a.data(), 512, hipMemcpyHostToDevice); hipMemcpy(dc, c.data(), 512, hipMemcpyHostToDevice);
    // hypotheses for (k, j) of lane l inside its block of 16: A lane = 4 k + i, B lane = 4 k + j ... try the four index pairings
    const char* names[4] = {"b = ((l>>2)&3) == (l&3)", "b = 1 for lanes 0,5,10,15 of a block (l%16 in {0,5,10,15})", "b = 1 everywhere", "b = (l&3) == ((l>>2)&3) transposed blocks"};
    for (int h = 0; h < 3; ++h) {
        for (int l = 0; l < 64; ++l) {
            const int m = l & 15;
            b[l] = h == 0 ? ((((l >> 2) & 3) == (l & 3)) ? 1.0 : 0.0) : h == 1 ? ((m == 0 || m == 5 || m == 10 || m == 15) ? 1.0 : 0.0) : 1.0;
        }
        hipMemcpy(db, b.data(), 512, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(sem_kernel, dim3(1), dim3(64), 0, 0, da, db, dc, dd, 0);
        hipMemcpy(d.data(), dd, 512, hipMemcpyDeviceToHost);
        int same = 0;
        for (int l = 0; l < 64; ++l) same += (d[l] == c[l] + a[l]);
        printf("hypothesis %d (%s): %d / 64 lanes have d == c + a bit for bit\n", h, names[h], same);
        if (h == 2) {  // b = 1 everywhere: d = c + sum_k a[i][k]: print which lanes' a make up lane 0..7's d
            for (int l = 0; l < 8; ++l) {
                // brute force: which 4-subset sum (in some order) matches
                bool found = false;
                for (int p = 0; p < 64 && !found; ++p)
                    for (int q = p + 1; q < 64 && !found; ++q)
                        for (int r = q + 1; r < 64 && !found; ++r)
                            for (int s2 = r + 1; s2 < 64 && !found; ++s2) {
                                const double exact = (double)((long double)c[l] + (long double)a[p] + (long double)a[q] + (long double)a[r] + (long double)a[s2]);
                                if (fabs(exact - d[l]) <= 1e-9 * fabs(d[l]) + 1e-12) { printf("  lane %d: d ~ c + a[%d] + a[%d] + a[%d] + a[%d]\n", l, p, q, r, s2); found = true; }
                            }
            }
        }
    }
    // identity by the pattern hypothesis 0, masked: lanes >= 32 skip the instruction in the source
    for (int l = 0; l < 64; ++l) b[l] = (((l >> 2) & 3) == (l & 3)) ? 1.0 : 0.0;
    hipMemcpy(db, b.data(), 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(sem_kernel, dim3(1), dim3(64), 0, 0, da, db, dc, dd, 1);
    hipMemcpy(d.data(), dd, 512, hipMemcpyDeviceToHost);
    int lo_ok = 0, hi_untouched = 0;
    for (int l = 0; l < 32; ++l) lo_ok += (d[l] == c[l] + a[l]);
    for (int l = 32; l < 64; ++l) hi_untouched += (d[l] == c[l]);
    printf("under `if (lane < 32)`: %d / 32 active lanes d == c + a, %d / 32 masked lanes untouched\n", lo_ok, hi_untouched);

    // ---- timing
    float* out;
    hipMalloc(&out, sizeof(float) * 256 * 256 * 3);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 400;
    for (int rep = 0; rep < 2; ++rep)
        for (int m = 0; m < 2; ++m) {
            hipEventRecord(e0);
            if (m == 0) hipLaunchKernelGGL((body_kernel<false>), dim3(256 * 3), dim3(256), 0, 0, out, 1.0f, iters, 1.0);
            else        hipLaunchKernelGGL((body_kernel<true>), dim3(256 * 3), dim3(256), 0, 0, out, 1.0f, iters, 1.0);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            printf("%s accumulates: %.3f ms for %d bodies per wavefront (3 wavefronts per SIMD) = %.0f ns per body\n", m ? "MFMA      " : "v_add_f64 ", ms, iters, 1e6 * ms / iters);
        }
    return 0;
}
