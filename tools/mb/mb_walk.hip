// microbenchmark of the 2-row recurrence step of k_render_hw (diagnostic, not product code)
// build: hipcc -O3 --offload-arch=gfx950 -DVARIANT=n -o mb_walk mb_walk.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#ifndef VARIANT
#define VARIANT 0
#endif
#ifndef LDS_BYTES
#define LDS_BYTES 20192
#endif
#define G 6
__global__ void __launch_bounds__(64) k_walk(const double *__restrict__ in, double *__restrict__ out, int trips, int segs) {
    __shared__ double acc[LDS_BYTES / 8];
    const int lane = threadIdx.x;
    for (int i = lane; i < LDS_BYTES / 8; i += 64) acc[i] = 0.0;
    __syncthreads();
    double g[G], r[G], q[G];
    double *col = acc + (lane & 31);
    for (int s = 0; s < segs; s++) {
#pragma unroll
        for (int i = 0; i < G; i++) {
            g[i] = in[(s * G + i) * 3 + 0] + lane * 1e-9;
            r[i] = in[(s * G + i) * 3 + 1];
            q[i] = in[(s * G + i) * 3 + 2];
        }
        int row = 0;
        for (int t = 0; t < trips; t++, row += 2) {
#pragma clang fp contract(off)
            double s0 = g[0], s1, g1[G], r1[G];
#pragma unroll
            for (int i = 1; i < G; i++) s0 += g[i];
#pragma unroll
            for (int i = 0; i < G; i++) { g1[i] = g[i] * r[i]; r1[i] = r[i] * q[i]; }
            s1 = g1[0];
#pragma unroll
            for (int i = 1; i < G; i++) s1 += g1[i];
#pragma unroll
            for (int i = 0; i < G; i++) { g[i] = g1[i] * r1[i]; r[i] = r1[i] * q[i]; }
#if VARIANT == 1
            col[0] += s0 + s1;           // plain LDS RMW on one address (no atomics)
#elif VARIANT == 2
            g[0] += (s0 + s1) * 1e-300;  // no LDS traffic at all
#else
            atomicAdd(&col[(row & 62) * 32], s0);
            atomicAdd(&col[((row & 62) + 1) * 32], s1);
#endif
        }
    }
    double v = 0;
#pragma unroll
    for (int i = 0; i < G; i++) v += g[i] + r[i];
    __syncthreads();
    out[blockIdx.x * 64 + lane] = v + acc[lane];
}
int main(int argc, char **argv) {
    int trips = 18, segs = 400, blocks = 10240;
    double *in, *out;
    hipMalloc(&in, sizeof(double) * segs * G * 3);
    hipMalloc(&out, sizeof(double) * blocks * 64);
    double *h = (double *)malloc(sizeof(double) * segs * G * 3);
    for (int i = 0; i < segs * G; i++) { h[3 * i] = 1.0 + i * 1e-6; h[3 * i + 1] = 0.999; h[3 * i + 2] = 0.9999; }
    hipMemcpy(in, h, sizeof(double) * segs * G * 3, hipMemcpyHostToDevice);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(a);
        hipLaunchKernelGGL(k_walk, dim3(blocks), dim3(64), 0, 0, in, out, trips, segs);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        double tr = (double)blocks * segs * trips;
        // per SIMD: blocks/(256 CUs*4 SIMDs) waves' worth of trips
        double cyc_per_trip_simd = ms * 1e-3 * 2.4e9 / (tr / 1024.0);
        printf("variant %d lds %d: %.3f ms, %.1f SIMD-cycles per trip at 2.4 GHz (35 VALU x 4 = 140 ideal)\n", VARIANT, LDS_BYTES, ms, cyc_per_trip_simd);
    }
    return 0;
}
