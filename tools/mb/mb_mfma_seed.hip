// Can the seed exponents of k_render_hw's column recurrence run on the MFMA pipe beside the VALU?  (diagnostic)
//
// A pair of component groups (12 components) on a 32-column half-tile needs, per segment, for every (column, component)
//     e  = c0 + c1 X + c2 Y + c3 X^2 + c4 X Y + c5 Y^2          (the Gaussian's exponent at the segment's first row)
//     er = d0 + d1 X + d2 Y                                     (the exponent of the row-to-row ratio)
// i.e. a (32 x 8) . (8 x 24) contraction of per-column monomials with per-component coefficients, followed by two table
// exponentials each (VALU).  As MFMA: v_mfma_f64_16x16x4_f64, 2 row blocks x 2 column blocks x 2 k-steps = 8 instructions.
// Variants (one "step" = the seeds of one pair of groups on one half-tile, both halves of the wave working):
//   0  VALU only: the polynomials (10 fma per (column, component), 6 components per lane) + a stand-in for the two exp (18 fma)
//   1  the exp stand-in alone (what the VALU would keep)
//   2  the 8 MFMAs alone
//   3  MFMA polynomials issued beside the exp stand-in in the same wave (no data movement between them: the BEST case --
//      in the kernel the MFMA result (lane = component, 4 columns per lane) would still have to be transposed through LDS
//      into the walk's layout (lane = column, 6 components per lane): 24 x 32 doubles written and read per pair)
#include <hip/hip_runtime.h>
#include <cstdio>
#ifndef VARIANT
#define VARIANT 0
#endif
typedef double double4_t __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2)))
k_seed(const double *__restrict__ in, double *__restrict__ out, int iters) {
    __shared__ double pad[20192 / 8];            // the render kernel's LDS footprint: 8 waves per CU
    const int lane = threadIdx.x;
    pad[lane] = in[lane & 63];
    __syncthreads();
    double X = in[lane & 63] + lane, Y = in[(lane + 7) & 63];
    double c[6][6];
#pragma unroll
    for (int k = 0; k < 6; k++)
#pragma unroll
        for (int j = 0; j < 6; j++) c[k][j] = in[(k * 6 + j) & 63] * 1e-3;
    double acc = 0.0;
    double4_t d0 = {0, 0, 0, 0}, d1 = {0, 0, 0, 0}, d2 = {0, 0, 0, 0}, d3 = {0, 0, 0, 0};
    double a0 = X, b0 = Y;
    for (int t = 0; t < iters; t++) {
        const double XX = X * X, XY = X * Y, YY = Y * Y;
#if VARIANT == 0
#pragma unroll
        for (int k = 0; k < 6; k++) {
            double e = __builtin_fma(c[k][1], X, c[k][0]);
            e = __builtin_fma(c[k][2], Y, e);
            e = __builtin_fma(c[k][3], XX, e);
            e = __builtin_fma(c[k][4], XY, e);
            e = __builtin_fma(c[k][5], YY, e);
            double er = __builtin_fma(c[k][1], X, c[k][2]);
            er = __builtin_fma(c[k][3], Y, er);
            acc += e + er;
        }
#endif
#if VARIANT == 0 || VARIANT == 1 || VARIANT == 3
        // stand-in for two table exponentials per component: 18 dependent-free fma per component in 6 chains
#pragma unroll
        for (int k = 0; k < 6; k++) {
            double p = c[k][0], q = c[k][1];
#pragma unroll
            for (int j = 0; j < 9; j++) { p = __builtin_fma(p, XX, c[k][2]); q = __builtin_fma(q, YY, c[k][3]); }
            acc += p + q;
        }
#endif
#if VARIANT == 2 || VARIANT == 3
        d0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, d1, 0, 0, 0);
        d2 = __builtin_amdgcn_mfma_f64_16x16x4f64(b0, a0, d2, 0, 0, 0);
        d3 = __builtin_amdgcn_mfma_f64_16x16x4f64(b0, a0, d3, 0, 0, 0);
        d0 = __builtin_amdgcn_mfma_f64_16x16x4f64(XX, b0, d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f64_16x16x4f64(XY, b0, d1, 0, 0, 0);
        d2 = __builtin_amdgcn_mfma_f64_16x16x4f64(YY, a0, d2, 0, 0, 0);
        d3 = __builtin_amdgcn_mfma_f64_16x16x4f64(XX, a0, d3, 0, 0, 0);
#endif
        X += 1e-9; Y -= 1e-9;
    }
    acc += d0.x + d0.y + d0.z + d0.w + d1.x + d1.y + d2.z + d3.w + pad[lane];
    out[blockIdx.x * 64 + lane] = acc;
}

int main() {
    int iters = 2000, blocks = 8192;
    double *in, *out;
    (void)hipMalloc(&in, sizeof(double) * 64);
    (void)hipMalloc(&out, sizeof(double) * blocks * 64);
    double h[64];
    for (int i = 0; i < 64; i++) h[i] = 1.0 + 1e-3 * i;
    (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(a);
        hipLaunchKernelGGL(k_seed, dim3(blocks), dim3(64), 0, 0, in, out, iters);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        printf("variant %d: %.3f ms, %.0f cycles per step per SIMD at 2.4 GHz (2 waves per SIMD share it)\n", VARIANT, ms,
               ms * 1e-3 * 2.4e9 / ((double)blocks * iters / 1024.0));
    }
    return 0;
}
