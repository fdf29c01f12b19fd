// k_patch_ll.h -- per-source conditional Poisson log-likelihoods on fixed patches
//
// The inner call of the per-source samplers: Source.log_likelihood (CelestePy/sources.py:134-183)
// and Source.log_likelihood_isolated (:188-237), which slice sampling / HMC evaluate 10-50 times
// per source per sweep with one of (u, fluxes, shape) varied (sources.py:308-319).  A batch of
// P proposals is scored in one launch: one 256-thread block per (proposal, band) renders the
// proposal's unit stamp on the band's FIXED patch limits with the direct evaluator (exact, no
// component dropping) and reduces, in a fixed order,
//   mode 0:  sum_{m>0} log(m) * z  -  counts * sum(psf weights),   m = counts * stamp
//   mode 1:  sum log(m + eps) * z  -  sum (m + eps)
// z = the patch data (photons attributed to the source, or nelec for the isolated form).
#pragma once
#include "k_render.h"

__global__ void __launch_bounds__(256)
k_patch_ll(const BandDev *__restrict__ bands, int B, int64_t P, const SrcRec *__restrict__ recs,
           const int *__restrict__ owner /* P: which patch set a proposal is scored on, or nullptr = 0 */,
           const int4 *__restrict__ pbox /* NB*B: x0, x1, y0, y1 */, const int64_t *__restrict__ offsets /* NB*B+1 */,
           const double *__restrict__ data, const double *__restrict__ nelec /* used when data == nullptr */,
           int H, int W, int mode, double *__restrict__ out /* P*B */) {
    __shared__ CompTab T;
    __shared__ double red[256], red2[256];
    const int tid = threadIdx.x;
    const int64_t job = blockIdx.x;
    const int b = (int)(job % B);
    const int64_t p = job / B;
    const BandDev *bd = bands + b;
    const SrcRec *rp = recs + (int64_t)b * P + p;
    const int64_t ob = (int64_t)(owner ? owner[p] : 0) * B + b;
    const int4 bx = pbox[ob];
    const int nx = bx.y - bx.x, ny = bx.w - bx.z;
    double wsum = bd->w[0] + bd->w[1] + bd->w[2];
    int type = rp->type;
    const double counts = rp->scale;
    if (nx <= 0 || ny <= 0) {           // no sample image in this band
        if (tid == 0) out[job] = 0.0;
        return;
    }
    // record types: 0/1 star/galaxy, -1/-2 star/galaxy whose own box is empty, -3 overlap-test miss
    if (type == -3 && mode == 0) {      // psf_ns is None (:160-163)
        if (tid == 0) out[job] = -counts * wsum;
        return;
    }
    if (type < 0) type = (type == -2) ? 1 : 0;           // imposed limits: the kind still renders
    const int K = (type == 0) ? K_PSF : K_GAL;
    if (tid < K) {
        Comp c = make_comp(tid, type, rp->px, rp->py, 1.0, rp->w00, rp->w01, rp->w11, rp->theta, bd);
        T.A[tid] = c.A; T.mx[tid] = c.mx; T.my[tid] = c.my;
        T.qa[tid] = c.qa; T.qb[tid] = c.qb; T.qc[tid] = c.qc;
    }
    __syncthreads();
    const double eps = bd->eps;
    // patch data: a packed buffer (photons attributed to the source), or -- when none is given --
    // the observed image itself on the box (the isolated form reads nelec, sources.py:204)
    const double *z = data ? data + offsets[ob] : nelec + (int64_t)b * H * W + (int64_t)bx.z * W + bx.x;
    const int zpitch = data ? nx : W;
    double a = 0.0, m = 0.0;
    const int n = nx * ny;
    for (int i = tid; i < n; i += 256) {
        int yy = i / nx, xx = i - yy * nx;
        double v = counts * eval_direct(T, 0, K, (double)(bx.x + xx), (double)(bx.z + yy), 1.0);
        const double zi = z[(int64_t)yy * zpitch + xx];
        if (mode == 0) {
            if (v > 0.0) a += log(v) * zi;
        } else {
            v += eps;
            a += log(v) * zi;
            m += v;
        }
    }
    red[tid] = a; red2[tid] = m;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) { red[tid] += red[tid + o]; red2[tid] += red2[tid + o]; }
        __syncthreads();
    }
    if (tid == 0) out[job] = (mode == 0) ? red[0] - counts * wsum : red[0] - red2[0];
}
