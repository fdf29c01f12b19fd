// k_estep.h -- E-step sufficient statistics without materialising the (S+1, H, W) layers
//
// CelestePy's EM (celeste_em.py:38-88) builds gen_src_prob_layers (celeste.py:222-234), an
// (S+1) x H x W tensor of responsibilities eps/lambda and F_s/lambda, only to reduce it:
//   X~[n][s]  = sum_pixels nelec * F_s / lambda      photons source s is responsible for (:85)
//   noise[n]  = sum_pixels nelec * eps / lambda      photons the sky is responsible for  (:62)
//   mass[n][s]= sum_pixels unit stamp of s           fraction of s's light inside image n (:89)
// At 10 000 sources x 2048^2 that tensor is 335 GB per band; the three reductions need only the
// resident model image.  One 256-thread block per (source, band) re-evaluates the source's
// stamp on its own box (direct evaluator, exact) against lambda and nelec; a strided kernel
// reduces the sky term.  Both reductions run in a fixed order.
//
// k_estep_src_hw is the same reduction on the column recurrence (hw_source.h): one wave per
// (source, band), the box covered by 32 x 64 chunks, each rendered as the unit stamp into an LDS
// tile with the drop rule relative to the source itself (both sums are linear in the stamp, so
// their relative error stays below K e^-T), then reduced against nelec / lambda.
#pragma once
#include "hw_source.h"

__global__ void __launch_bounds__(256)
k_estep_src(const BandDev *__restrict__ bands, int B, int H, int W, int64_t S, const SrcRec *__restrict__ recs,
            const double *__restrict__ nelec, const double *__restrict__ lambda,
            double *__restrict__ xt /* S*B */, double *__restrict__ mass /* S*B */) {
    __shared__ CompTab T;
    __shared__ double red[256], red2[256];
    const int tid = threadIdx.x;
    const int64_t job = blockIdx.x;
    const int b = (int)(job % B);
    const int64_t s = job / B;
    const SrcRec *rp = recs + (int64_t)b * S + s;
    const int type = rp->type;
    if (type < 0) {
        if (tid == 0) { xt[job] = 0.0; mass[job] = 0.0; }
        return;
    }
    const BandDev *bd = bands + b;
    const int K = (type == 0) ? K_PSF : K_GAL;
    if (tid < K) {
        Comp c = make_comp(tid, type, rp->px, rp->py, 1.0, rp->w00, rp->w01, rp->w11, rp->theta, bd);
        T.A[tid] = c.A; T.mx[tid] = c.mx; T.my[tid] = c.my;
        T.qa[tid] = c.qa; T.qb[tid] = c.qb; T.qc[tid] = c.qc;
    }
    __syncthreads();
    const int x0 = rp->x0, y0 = rp->y0, nx = rp->x1 - rp->x0, ny = rp->y1 - rp->y0;
    const double counts = rp->scale;
    const int64_t plane = (int64_t)b * H * W;
    double a = 0.0, m = 0.0;
    const int n = nx * ny;
    for (int i = tid; i < n; i += 256) {
        int yy = i / nx, xx = i - yy * nx;
        double u = eval_direct(T, 0, K, (double)(x0 + xx), (double)(y0 + yy), 1.0);
        int64_t idx = plane + (int64_t)(y0 + yy) * W + (x0 + xx);
        a += nelec[idx] * (counts * u) / lambda[idx];
        m += u;
    }
    red[tid] = a; red2[tid] = m;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) { red[tid] += red[tid + o]; red2[tid] += red2[tid + o]; }
        __syncthreads();
    }
    if (tid == 0) { xt[job] = red[0]; mass[job] = red2[0]; }
}

__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2)))
k_estep_src_hw(const BandDev *__restrict__ bands, int B, int H, int W, int64_t S, const SrcRec *__restrict__ recs,
               const double *__restrict__ nelec, const double *__restrict__ lambda, double Tdrop,
               double *__restrict__ xt /* S*B */, double *__restrict__ mass /* S*B */) {
    __shared__ double acc[HW_TH * HW_TW];
    __shared__ CompTab T;
    __shared__ double et[64];
    const int lane = threadIdx.x;
    const int half = lane >> 5, col = lane & 31;
    const int64_t job = blockIdx.x;
    const int b = (int)(job % B);
    const int64_t s = job / B;
    RecU rec = rec_unpack(rec_fetch(recs + (int64_t)b * S, (int)s, lane));
    if (rec.type < 0) {
        if (lane == 0) { xt[job] = 0.0; mass[job] = 0.0; }
        return;
    }
    const BandDev *bd = bands + b;
    const double counts = rec.scale;
    rec.scale = 1.0;
    et[lane] = exp2((double)lane * (1.0 / 64.0));
    const LaneConst lc = lane_consts(lane, bd);
    const int dropmode = (Tdrop > 0.0) ? HW_DROP_SELF : HW_DROP_NONE;
    const int64_t plane = (int64_t)b * H * W;
    double a = 0.0, m = 0.0;
    for (int Y0 = rec.y0; Y0 < rec.y1; Y0 += HW_TH) {
        const int rb = min(HW_TH, rec.y1 - Y0);
        for (int X0 = rec.x0; X0 < rec.x1; X0 += HW_TW) {
            const int xi = X0 + col;
            const bool on = xi < rec.x1;
#pragma unroll
            for (int r = 0; r < HW_TH / 2; r++) acc[r * 64 + lane] = 0.0;
            bool direct;
            const int Kk = hw_build(T, lc, rec, lane, dropmode, Tdrop, 0.0, Y0, X0, min(rec.x1, X0 + HW_TW) - 1, 0, rb, direct);
            hw_walk(T, et, Kk, (double)xi, Y0, 0, rb, on, direct, acc, lane);
            __syncthreads();
            const int64_t base = plane + (int64_t)Y0 * W + min(xi, rec.x1 - 1);
            for (int r0 = 0; r0 < HW_TH / 2 && 2 * r0 < rb; r0 += 8) {
                double ne[8], la[8];
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    const int64_t idx = base + (int64_t)min(2 * (r0 + r) + half, rb - 1) * W;
                    ne[r] = nelec[idx];
                    la[r] = lambda[idx];
                }
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    if (on && 2 * (r0 + r) + half < rb) {
                        const double u = acc[(r0 + r) * 64 + lane];
                        a += ne[r] * (counts * u) / la[r];
                        m += u;
                    }
                }
            }
            __syncthreads();
        }
    }
    a = wave_sum(a);
    m = wave_sum(m);
    if (lane == 0) { xt[job] = a; mass[job] = m; }
}

// sky responsibility: partial[b][blk] = sum over the block's pixel chunk of nelec * eps / lambda
__global__ void __launch_bounds__(256)
k_estep_noise(const BandDev *__restrict__ bands, int64_t npix, int nblk, const double *__restrict__ nelec,
              const double *__restrict__ lambda, double *__restrict__ partial) {
    __shared__ double red[256];
    const int b = blockIdx.x / nblk, blk = blockIdx.x - b * nblk;
    const double eps = bands[b].eps;
    const int64_t chunk = (npix + nblk - 1) / nblk;
    const int64_t lo = chunk * blk, hi = (lo + chunk < npix) ? lo + chunk : npix;
    const double *ne = nelec + (int64_t)b * npix, *la = lambda + (int64_t)b * npix;
    double a = 0.0;
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) a += ne[i] * eps / la[i];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
