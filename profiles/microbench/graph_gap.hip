// Dependent-kernel gap on gfx950: a chain of N small kernels (each ~T us of work, 1024 workgroups) launched on a stream against the same chain as a
// captured hipGraph.   hipcc --offload-arch=gfx950 -O3 -o graph_gap graph_gap.hip && ./graph_gap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void work(float* p, int iters)
{
    float v = p[blockIdx.x * blockDim.x + threadIdx.x];
    for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
    p[blockIdx.x * blockDim.x + threadIdx.x] = v;
}
int main()
{
    float* d;
    CK(hipMalloc(&d, 1024 * 256 * 4));
    CK(hipMemset(d, 0, 1024 * 256 * 4));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int N = 300;
    for (int iters : {0, 2000, 20000}) {
        // stream
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0, st));
            for (int k = 0; k < N; ++k) hipLaunchKernelGGL(work, dim3(1024), dim3(256), 0, st, d, iters);
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
        }
        float ms_stream; CK(hipEventElapsedTime(&ms_stream, e0, e1));
        // one kernel alone
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL(work, dim3(1024), dim3(256), 0, st, d, iters);
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms_one; CK(hipEventElapsedTime(&ms_one, e0, e1));
        // graph
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int k = 0; k < N; ++k) hipLaunchKernelGGL(work, dim3(1024), dim3(256), 0, st, d, iters);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        float ms_graph = 0;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0, st));
            CK(hipGraphLaunch(ge, st));
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            CK(hipEventElapsedTime(&ms_graph, e0, e1));
        }
        printf("iters %6d: one kernel %.2f us | %d in a stream: %.2f us each | as a graph: %.2f us each\n", iters, 1e3 * ms_one, N, 1e3 * ms_stream / N, 1e3 * ms_graph / N);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
