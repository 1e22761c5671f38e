// Issue rate of a few VALU instructions on gfx950, measured the blunt way: one workgroup of W wavefronts per CU-SIMD slot runs N
// back-to-back independent (or dependent) instructions of one kind; cycles per instruction and wavefront = wall cycles / N.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x
template <int OP>
__global__ void k(float* out, float seed, int iters, long long* cyc)
{
    float  a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7;
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) {  // v_cvt_f64_f32, 8 independent
            REP16(asm volatile("v_cvt_f64_f32 %0, %8\n v_cvt_f64_f32 %1, %9\n v_cvt_f64_f32 %2, %10\n v_cvt_f64_f32 %3, %11\n v_cvt_f64_f32 %4, %12\n v_cvt_f64_f32 %5, %13\n v_cvt_f64_f32 %6, %14\n v_cvt_f64_f32 %7, %15"
                               : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3), "=v"(d4), "=v"(d5), "=v"(d6), "=v"(d7) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7));)
        } else if (OP == 1) {  // v_add_f64, 8 independent chains
            REP16(asm volatile("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8"
                               : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(1.0));)
        } else if (OP == 2) {  // v_add_f64, ONE dependent chain
            REP16(asm volatile("v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1"
                               : "+v"(d0) : "v"(1.0));)
        } else if (OP == 3) {  // v_mul_f64 independent
            REP16(asm volatile("v_mul_f64 %0, %8, %9\n v_mul_f64 %1, %8, %9\n v_mul_f64 %2, %8, %9\n v_mul_f64 %3, %8, %9\n v_mul_f64 %4, %8, %9\n v_mul_f64 %5, %8, %9\n v_mul_f64 %6, %8, %9\n v_mul_f64 %7, %8, %9"
                               : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3), "=v"(d4), "=v"(d5), "=v"(d6), "=v"(d7), "+v"(d0), "+v"(d1) :);)
        } else if (OP == 4) {  // v_fma_f32 independent
            REP16(asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(1.0f));)
        } else if (OP == 5) {  // v_add_f32 ONE dependent chain
            REP16(asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1"
                               : "+v"(a0) : "v"(1.0f));)
        } else if (OP == 6) {  // v_cvt_f32_f64 independent
            REP16(asm volatile("v_cvt_f32_f64 %0, %8\n v_cvt_f32_f64 %1, %9\n v_cvt_f32_f64 %2, %10\n v_cvt_f32_f64 %3, %11\n v_cvt_f32_f64 %4, %12\n v_cvt_f32_f64 %5, %13\n v_cvt_f32_f64 %6, %14\n v_cvt_f32_f64 %7, %15"
                               : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(d0), "v"(d1), "v"(d2), "v"(d3), "v"(d4), "v"(d5), "v"(d6), "v"(d7));)
        } else if (OP == 7) {  // v_fma_f64 independent
            REP16(asm volatile("v_fma_f64 %0, %0, %8, %8\n v_fma_f64 %1, %1, %8, %8\n v_fma_f64 %2, %2, %8, %8\n v_fma_f64 %3, %3, %8, %8\n v_fma_f64 %4, %4, %8, %8\n v_fma_f64 %5, %5, %8, %8\n v_fma_f64 %6, %6, %8, %8\n v_fma_f64 %7, %7, %8, %8"
                               : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(1.0));)
        } else if (OP == 8) {  // v_pk_fma_f32 independent
            REP16(asm volatile("v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8"
                               : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(1.0));)
        }
    }
    const long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + float(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int OP>
void run(const char* name, int waves_per_simd)
{
    float*     out;
    long long* cyc;
    hipMalloc(&out, sizeof(float) * 256 * 64 * 4 * 16);
    hipMalloc(&cyc, 8);
    const int iters = 200, n = iters * 16 * 8;
    // one block of 256 * waves_per_simd / ... : blocks of 256 threads = 1 wave per SIMD of a CU; waves_per_simd blocks per CU
    const int   blocks = 256 * waves_per_simd;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 1.0f, iters, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 1.0f, iters, cyc);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    long long c;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    // wall: n instructions per wave, waves_per_simd waves per SIMD (if all blocks are resident at once)
    printf("%-28s waves/SIMD %d: %7.2f clock64 ticks per instruction (one wave), wall %.3f ms -> %.2f ns per instruction and SIMD\n", name, waves_per_simd, double(c) / n, ms,
           1e6 * ms / (double(n) * waves_per_simd));
    hipFree(out); hipFree(cyc);
}

int main()
{
    for (int w : {1, 4}) {
        run<4>("v_fma_f32 (8 indep)", w);
        run<8>("v_pk_fma_f32 (8 indep)", w);
        run<5>("v_add_f32 (dependent)", w);
        run<0>("v_cvt_f64_f32 (8 indep)", w);
        run<6>("v_cvt_f32_f64 (8 indep)", w);
        run<1>("v_add_f64 (8 indep)", w);
        run<2>("v_add_f64 (dependent)", w);
        run<3>("v_mul_f64 (8 indep)", w);
        run<7>("v_fma_f64 (8 indep)", w);
    }
    return 0;
}
