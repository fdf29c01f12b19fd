// Can HIP events recorded INSIDE a captured graph be timed after the graph has run?
// (bench.py measures kernel durations with events on the library's stream over the timed region;
// a hipGraph for the step is only useful there if this works.)   hipcc --offload-arch=gfx950 -o mb_graph_events mb_graph_events.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void spin(double *p, int n) { double a = p[threadIdx.x]; for (int i = 0; i < n; i++) a = a * 1.0000001 + 1e-9; p[threadIdx.x] = a; }
int main() {
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    double *d; CK(hipMalloc(&d, 8 * 256)); CK(hipMemset(d, 0, 8 * 256));
    hipEvent_t e0, e1, e2; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    CK(hipEventRecord(e0, st));
    hipLaunchKernelGGL(spin, dim3(1), dim3(256), 0, st, d, 200000);
    CK(hipEventRecord(e1, st));
    hipLaunchKernelGGL(spin, dim3(1), dim3(256), 0, st, d, 400000);
    CK(hipEventRecord(e2, st));
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int it = 0; it < 3; it++) {
        CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        float a = -1, b = -1;
        hipError_t ra = hipEventElapsedTime(&a, e0, e1), rb = hipEventElapsedTime(&b, e1, e2);
        printf("launch %d: elapsed(e0,e1) = %.3f ms (%s), elapsed(e1,e2) = %.3f ms (%s)\n", it, a, hipGetErrorString(ra), b, hipGetErrorString(rb));
    }
    // plain stream for comparison
    CK(hipEventRecord(e0, st));
    hipLaunchKernelGGL(spin, dim3(1), dim3(256), 0, st, d, 200000);
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float a; CK(hipEventElapsedTime(&a, e0, e1)); printf("stream: %.3f ms\n", a);
    return 0;
}
