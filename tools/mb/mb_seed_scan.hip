// Seeds of k_render_hw's column recurrence: two table exponentials per (component, column) against ONE seed per component
// and a product scan over the lanes.  (diagnostic; round-4 review item 6)
//
// A segment of a component group needs, for every (component k, column x) at the segment's first row y0,
//     g = A exp(E(x, y0)),   r = exp(-(qb dx + qc dy + qc / 2))          (k_render_hw.h, rec_group_hw)
// Today every lane evaluates both exponents for its column and takes two table exponentials (exp_tab64: 9 fp64 ops + rint,
// cvt, ldexp, an LDS read): ~38 VALU + 7 LDS reads per (component, column).
// Along a row the exponent is a quadratic in the column, so with t = x - x0
//     g(x0 + t) = g(x0) rho^t kappa^(t (t - 1) / 2),   rho = exp(-(qa dx0 + qb dy + qa / 2)),  kappa = exp(-qa)
//     r(x0 + t) = r(x0) sigma^t,                       sigma = exp(-qb)
// i.e. ONE lane per component evaluates five exponentials (g(x0), rho, kappa, r(x0), sigma), the other lanes get
//     kappa^t and sigma^t by five bit-selected multiplies each (the powers kappa^(2^i), sigma^(2^i) come with the five),
//     the running product of rho kappa^i over the lanes by a five-step scan (__shfl_up + multiply), then two multiplies.
// Variant 1 below is that scheme in its BEST case (x0 = the tile's first column: one direction, factors <= 1), the
// components' scalars written to LDS by lanes 0..5 and read back by all (a barrier pair per segment, as the table build has).
// Variant 0 is the kernel's code.  Both produce (g, r) for 6 components per lane; the harness checks that they agree.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#ifndef VARIANT
#define VARIANT 0
#endif
#define EXP_SCALE 92.332482616893656758   // 64 / ln 2
#define G 6

__device__ inline double exp_tab64(double t, const double *__restrict__ et) {
    const double c1 = 1.0830424696249145e-02, c2 = 5.864904955056169e-05, c3 = 2.1173137155464774e-07;
    const double c4 = 5.732851688640402e-10, c5 = 1.2417843701716923e-12;
    double n = rint(t);
    double f = t - n;
    int ni = (int)n;
    double p = fma(f, c5, c4);
    p = fma(p, f, c3);
    p = fma(p, f, c2);
    p = fma(p, f, c1);
    p = fma(p, f, 1.0);
    return ldexp(et[ni & 63] * p, ni >> 6);
}

struct Tab { double A[16], mx[16], my[16], qa[16], qb[16], qc[16]; };

__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2)))
k_seed(const double *__restrict__ in, double *__restrict__ out, int iters, int check) {
    __shared__ double pad[(20192 - sizeof(Tab) - 64 * 8 - 16 * 8 * 16) / 8];   // the render kernel's LDS footprint: 8 waves per CU
    __shared__ Tab T;
    __shared__ double et[64];
    __shared__ double sc[16][16];            // variant 1: per component g0, rho, r0, kappa^(2^i) x5, sigma^(2^i) x5
    const int lane = threadIdx.x;
    const int half = lane >> 5, col = lane & 31;
    et[lane] = exp2((double)lane * (1.0 / 64.0));
    pad[lane] = in[lane];
    if (lane < 16) {
        // components of a galaxy-like source left of the tile (the scan's best case): widths 1.5 ... 9 pixels
        const double s2x = 2.0 + 5.0 * lane, s2y = 3.0 + 4.0 * lane, cxy = 0.3 * sqrt(s2x * s2y);
        const double det = s2x * s2y - cxy * cxy;
        T.qa[lane] = s2y / det * EXP_SCALE; T.qb[lane] = -cxy / det * EXP_SCALE; T.qc[lane] = s2x / det * EXP_SCALE;
        T.A[lane] = 1.0 / (2.0 * M_PI * sqrt(det));
        T.mx[lane] = -3.0 - 0.25 * lane + in[0] * 1e-9; T.my[lane] = 20.0 + 0.5 * lane;
    }
    __syncthreads();
    const double x = (double)col;
    const int k0 = half * G;
    double acc = 0.0, worst = 0.0;
    for (int it = 0; it < iters; it++) {
        const double y0 = (double)(it & 31);
        double g[G], r[G];
#if VARIANT == 0 || defined(CHECK_BOTH)
#pragma unroll
        for (int i = 0; i < G; i++) {
            const int k = k0 + i;
            double dx = x - T.mx[k], dy = y0 - T.my[k];
            double qb = T.qb[k], qc = T.qc[k];
            double hx = qb * dx + qc * dy;
            double e = -0.5 * (T.qa[k] * dx * dx + (qb * dx + hx) * dy);
            double er = fmin(fmax(-(hx + 0.5 * qc), -680.0 * EXP_SCALE), 680.0 * EXP_SCALE);
            g[i] = T.A[k] * exp_tab64(e, et);
            r[i] = exp_tab64(er, et);
        }
#endif
#if VARIANT == 1
        double g2[G], r2[G];
        __syncthreads();
        if (lane < 2 * G) {                 // one lane per component: five exponentials, the powers by squaring
            const int k = lane;
            const double dx0 = 0.0 - T.mx[k], dy = y0 - T.my[k];
            const double qa = T.qa[k], qb = T.qb[k], qc = T.qc[k];
            const double hx = qb * dx0 + qc * dy;
            sc[k][0] = T.A[k] * exp_tab64(-0.5 * (qa * dx0 * dx0 + (qb * dx0 + hx) * dy), et);
            sc[k][1] = exp_tab64(-(qa * dx0 + qb * dy + 0.5 * qa), et);
            sc[k][2] = exp_tab64(-(hx + 0.5 * qc), et);
            double kp = exp_tab64(-qa, et), sg = exp_tab64(-qb, et);
#pragma unroll
            for (int j = 0; j < 5; j++) { sc[k][3 + j] = kp; sc[k][8 + j] = sg; kp *= kp; sg *= sg; }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < G; i++) {
            const int k = k0 + i;
            double kt = 1.0, st = 1.0;      // kappa^t, sigma^t, t = col
#pragma unroll
            for (int j = 0; j < 5; j++) {
                kt *= ((col >> j) & 1) ? sc[k][3 + j] : 1.0;
                st *= ((col >> j) & 1) ? sc[k][8 + j] : 1.0;
            }
            double f = sc[k][1] * kt;        // the factor from column t to t + 1
            // exclusive product scan over the half-wave's 32 lanes: P(t) = prod_{i < t} f(i)
            double p = f;
#pragma unroll
            for (int o = 1; o < 32; o <<= 1) {
                const double up = __shfl_up(p, o, 32);
                if (col >= o) p *= up;
            }
            const double excl = __shfl_up(p, 1, 32);
            g2[i] = sc[k][0] * ((col > 0) ? excl : 1.0);
            r2[i] = sc[k][2] * st;
        }
#ifdef CHECK_BOTH
        if (check)
#pragma unroll
            for (int i = 0; i < G; i++) {
                if (g[i] > 1e-280) worst = fmax(worst, fabs(g2[i] / g[i] - 1.0));
                worst = fmax(worst, fabs(r2[i] / r[i] - 1.0));
            }
#endif
#pragma unroll
        for (int i = 0; i < G; i++) { g[i] = g2[i]; r[i] = r2[i]; }
#endif
#pragma unroll
        for (int i = 0; i < G; i++) acc += g[i] * r[i];
    }
    out[blockIdx.x * 64 + lane] = check ? worst : acc + pad[lane];
}

int main() {
    int iters = 2000, blocks = 8192;
    double *in, *out;
    (void)hipMalloc(&in, sizeof(double) * 4096);
    (void)hipMalloc(&out, sizeof(double) * blocks * 64);
    double h[4096];
    for (int i = 0; i < 4096; i++) h[i] = 1.0 + 1e-3 * i;
    (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(a);
        hipLaunchKernelGGL(k_seed, dim3(blocks), dim3(64), 0, 0, in, out, iters, 0);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        printf("variant %d: %.3f ms, %.0f cycles per segment seed (6 components per lane) per SIMD at 2.4 GHz (2 waves per SIMD share it)\n", VARIANT, ms,
               ms * 1e-3 * 2.4e9 / ((double)blocks * iters / 1024.0));
    }
#ifdef CHECK_BOTH
    hipLaunchKernelGGL(k_seed, dim3(1), dim3(64), 0, 0, in, out, 64, 1);
    double w[64];
    (void)hipMemcpy(w, out, sizeof(w), hipMemcpyDeviceToHost);
    double m = 0;
    for (int i = 0; i < 64; i++) m = fmax(m, w[i]);
    printf("scan seeds against table seeds: worst relative difference %.3e\n", m);
#endif
    return 0;
}
