// fp64 VALU issue-rate microbenchmark (diagnostic): N independent chains of one instruction kind
#include <hip/hip_runtime.h>
#include <cstdio>
#ifndef VARIANT
#define VARIANT 0
#endif
#ifndef LDS_BYTES
#define LDS_BYTES 20192
#endif
#define N 16
__global__ void __launch_bounds__(64) k_rate(const double *__restrict__ in, double *__restrict__ out, int iters) {
    __shared__ double pad[LDS_BYTES / 8];
    const int lane = threadIdx.x;
    pad[lane] = in[lane];
    __syncthreads();
    double a[N], b[N];
#pragma unroll
    for (int i = 0; i < N; i++) { a[i] = in[i] + lane * 1e-9; b[i] = in[N + i]; }
    for (int t = 0; t < iters; t++) {
#pragma clang fp contract(off)
#pragma unroll
        for (int i = 0; i < N; i++) {
#if VARIANT == 0
            a[i] = a[i] * b[i];
#elif VARIANT == 1
            a[i] = a[i] + b[i];
#elif VARIANT == 2
            a[i] = __builtin_fma(a[i], b[i], b[i]);
#elif VARIANT == 3
            a[i] = a[i] * b[i]; b[i] = b[i] * a[(i + 1) % N];   // two dependent-free multiplies with mixed operands
#elif VARIANT == 4
            a[i] = __builtin_rint(a[i] * 1.0000001);            // v_mul_f64 + v_rndne_f64
#elif VARIANT == 5
            a[i] = (double)((int)(a[i]) + 1);                   // v_cvt_i32_f64 + v_add_u32 + v_cvt_f64_i32
#elif VARIANT == 6
            a[i] = __builtin_ldexp(a[i], (t & 1) ? 1 : -1);     // v_ldexp_f64
#elif VARIANT == 7
            a[i] = (a[i] + 6755399441055744.0) - 6755399441055743.0;   // the magic-number rounding: two v_add_f64
#elif VARIANT == 8
            a[i] = __builtin_amdgcn_frexp_mant(a[i]) + b[i];    // v_frexp_mant_f64 + v_add_f64
#endif
        }
    }
    double v = pad[lane];
#pragma unroll
    for (int i = 0; i < N; i++) v += a[i] + b[i];
    out[blockIdx.x * 64 + lane] = v;
}
int main() {
    int iters = 2000, blocks = 8192;
    double *in, *out;
    (void)hipMalloc(&in, sizeof(double) * 128);
    (void)hipMalloc(&out, sizeof(double) * blocks * 64);
    double h[128];
    for (int i = 0; i < 128; i++) h[i] = 1.0 + 1e-9 * i;
    (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(a);
        hipLaunchKernelGGL(k_rate, dim3(blocks), dim3(64), 0, 0, in, out, iters);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        double ninstr = (double)blocks * iters * N * ((VARIANT == 3 || VARIANT == 4 || VARIANT == 7 || VARIANT == 8) ? 2 : (VARIANT == 5 ? 3 : 1));
        printf("variant %d lds %d: %.3f ms, %.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", VARIANT, LDS_BYTES, ms,
               ms * 1e-3 * 2.4e9 / (ninstr / 1024.0));
    }
    return 0;
}
