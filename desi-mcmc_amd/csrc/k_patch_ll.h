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
//   mode 2:  sum_{m>0} log(m) * z  -  sum m        (galaxy_source_like, celeste_galaxy_conditionals.py:15-42,
//            on given limits; a pixel the model does not reach contributes nothing)
//   mode 4:  sum_{z>=0, m+bg>0} log(m + bg) * z - (m + bg)   on a given BACKGROUND: the patch data is two planes,
//            z then bg (everything else in the field rendered on the box); z < 0 marks a masked pixel.  The
//            image_like closure of the star <-> galaxy move (sources.py:277-291).
// z = the patch data (photons attributed to the source, or nelec for the isolated form).
//
// Two kernels with the same contract:
//   k_patch_ll     reference form: one 256-thread block per (proposal, band), direct evaluator,
//                  every component on every pixel (CEL_OPT_KERNEL = 0).
//   k_patch_ll_hw  default: one wave per (proposal, band); the patch is covered by 32 x 64 chunks,
//                  each rendered into an LDS tile by the column recurrence (hw_source.h) with the
//                  drop rule relative to the source itself (mode 0) or to the sky (mode 1), then
//                  reduced against the patch data with the table log.
#pragma once
#include "hw_source.h"

// TZ: the patch data's element type -- double for patches a caller hands over, int for the device-resident photon
// split (photon counts are integers: 4 bytes per pixel instead of 8, 1.6 GB instead of 3.2 GB at config 3)
template <typename TZ>
__global__ void __launch_bounds__(256)
k_patch_ll(const BandDev *__restrict__ bands, int B, int64_t P, const SrcRec *__restrict__ recs,
           const int *__restrict__ owner /* P: which patch set a proposal is scored on, or nullptr = 0 */,
           const int4 *__restrict__ pbox /* NB*B: x0, x1, y0, y1 */, const int64_t *__restrict__ offsets /* NB*B+1 */,
           const TZ *__restrict__ data, const double *__restrict__ nelec /* used when data == nullptr */,
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
    const TZ *zd = data ? data + offsets[ob] : nullptr;
    const double *zn = nelec + (int64_t)b * H * W + (int64_t)bx.z * W + bx.x;
    const int zpitch = data ? nx : W;
    double a = 0.0, m = 0.0;
    const int n = nx * ny;
    for (int i = tid; i < n; i += 256) {
        int yy = i / nx, xx = i - yy * nx;
        double v = counts * eval_direct(T, 0, K, (double)(bx.x + xx), (double)(bx.z + yy), 1.0);
        const double zi = zd ? (double)zd[(int64_t)yy * zpitch + xx] : zn[(int64_t)yy * zpitch + xx];
        if (mode == 0) {
            if (v > 0.0) a += log(v) * zi;
        } else if (mode == 4) {
            v += (double)zd[(int64_t)n + (int64_t)yy * zpitch + xx];         // the background plane follows the data plane
            if (v > 0.0 && zi >= 0.0) { a += log(v) * zi; m += v; }
        } else if (mode == 2) {
            if (v > 0.0) { a += log(v) * zi; m += v; }
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

// The smallest rectangle of a patch that holds all of its nonzero data.  The conditional form
// (mode 0) sums log(m) * z, to which a pixel with z = 0 contributes exactly nothing, so the model
// need only be evaluated inside this rectangle -- for a faint source a small fraction of its box
// (the photons sit in the core, the box reaches out to the 1e-5 contour).  One wave per patch.
template <typename TZ>
__global__ void __launch_bounds__(64)
k_patch_nzbox(const int4 *__restrict__ pbox, const int64_t *__restrict__ offsets, const TZ *__restrict__ data,
              int4 *__restrict__ nz /* x0, x1, y0, y1 (absolute), all 0 when the patch holds no photon */) {
    const int64_t i = blockIdx.x;
    const int4 bx = pbox[i];
    const int nx = bx.y - bx.x, ny = bx.w - bx.z;
    const int lane = threadIdx.x;
    int xlo = INT_MAX, xhi = -1, ylo = INT_MAX, yhi = -1;
    if (nx > 0 && ny > 0) {
        const TZ *z = data + offsets[i];
        const int64_t n = (int64_t)nx * ny;
        int yy = 0, xx = lane;
        while (xx >= nx) { xx -= nx; yy++; }
        for (int64_t k = lane; k < n; k += 64) {
            if (z[k] != (TZ)0) {
                xlo = min(xlo, xx); xhi = max(xhi, xx);
                ylo = min(ylo, yy); yhi = max(yhi, yy);
            }
            xx += 64;
            while (xx >= nx) { xx -= nx; yy++; }
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        xlo = min(xlo, __shfl_xor(xlo, o)); xhi = max(xhi, __shfl_xor(xhi, o));
        ylo = min(ylo, __shfl_xor(ylo, o)); yhi = max(yhi, __shfl_xor(yhi, o));
    }
    if (lane == 0) nz[i] = (xhi >= 0) ? make_int4(bx.x + xlo, bx.x + xhi + 1, bx.z + ylo, bx.z + yhi + 1) : make_int4(0, 0, 0, 0);
}

// One trip of the evaluation at the photons: lane l takes the list entries i0 + l, i0 + 64 + l, ... (P of them),
// sums the K components at each (exponent = c0 + c1 X + c2 Y + c3 X^2 + c4 X Y + c5 Y^2, X, Y relative to the
// source; cq holds c0..c5, the amplitude and a pad per component) and returns its share of sum z log(counts * stamp).
template <int P>
__device__ __forceinline__ double nz_trip(const NzEntry *__restrict__ L, int n, int i0, int lane, int K,
                                          const double *__restrict__ cq, const double *__restrict__ ltq,
                                          const double *__restrict__ et, double px, double py, double counts) {
    double X[P], Y[P], XX[P], XY[P], YY[P], v[P], z[P];
#pragma unroll
    for (int p = 0; p < P; p++) {
        const int i = i0 + 64 * p + lane;
        const NzEntry en = L[min(i, n - 1)];
        X[p] = (double)(en.xy & 0xffff) - px;
        Y[p] = (double)((unsigned)en.xy >> 16) - py;
        XX[p] = X[p] * X[p]; XY[p] = X[p] * Y[p]; YY[p] = Y[p] * Y[p];
        z[p] = (i < n) ? (double)en.z : 0.0;
        v[p] = 0.0;
    }
    for (int k = 0; k < K; k++) {
        const double *c = cq + 8 * k;
        const double c0 = c[0], c1 = c[1], c2 = c[2], c3 = c[3], c4 = c[4], c5 = c[5], A = c[6];
#pragma unroll
        for (int p = 0; p < P; p++) {
            double e = fma(c1, X[p], c0);
            e = fma(c2, Y[p], e);
            e = fma(c3, XX[p], e);
            e = fma(c4, XY[p], e);
            e = fma(c5, YY[p], e);
            v[p] = fma(A, exp_tab64(fmax(e, -1.0e5), et), v[p]);
        }
    }
    double a = 0.0;
#pragma unroll
    for (int p = 0; p < P; p++) {
        const double m = counts * v[p];
        if (z[p] != 0.0 && m > 0.0) a += log_tab(m, ltq) * z[p];
    }
    return a;
}

#define PLL_PARTS 4
template <int MODE, typename TZ = double>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2)))
k_patch_ll_hw(const BandDev *__restrict__ bands, int B, int64_t P, const SrcRec *__restrict__ recs,
              const int *__restrict__ owner, const int4 *__restrict__ pbox, const int64_t *__restrict__ offsets,
              const TZ *__restrict__ data, const double *__restrict__ nelec, int H, int W,
              const int4 *__restrict__ nzbox /* NB*B from k_patch_nzbox, or nullptr: evaluate the whole patch */,
              double Tdrop, double *__restrict__ out /* P*B*nsplit */,
              const int *__restrict__ job_order = nullptr /* P*B: launch order of the (proposal, band) jobs, heaviest first */,
              int nsplit = 1 /* 1 or PLL_PARTS blocks per job: block (job, part) takes the job's chunks c with
                                c % PLL_PARTS == part and writes out[job * PLL_PARTS + part]; part 0 carries the terms that
                                are not sums over pixels.  A round of few, long one-wave jobs (the late rounds of the slice
                                sampler, a caller with a handful of proposals) otherwise lasts as long as its longest job
                                while most of the GPU idles.  MODE 0 sums its chunks in PLL_PARTS classes either way and
                                whoever adds the parts adds them in order, so a value does not depend on nsplit */,
              const int *__restrict__ job_count = nullptr /* with job_order: only its first *job_count entries are jobs */,
              const int *__restrict__ nzmode = nullptr /* MODE 0, resident patches: per patch 1 = evaluate at its photons (k_nz_layout) */,
              const int64_t *__restrict__ nzoff = nullptr, const NzEntry *__restrict__ nzlist = nullptr /* the photon lists */) {
    __shared__ double acc[HW_TH * HW_TW];
    __shared__ CompTab T;
    __shared__ double et[64];
    // the log table lives in the component table's LDS between a chunk's walk and the next chunk's
    // build (20 184 B per wave: 8 waves per CU; with a table of its own 21 208 B: 7)
    double *lt = reinterpret_cast<double *>(&T);
    const int lane = threadIdx.x;
    const int half = lane >> 5, col = lane & 31;
    const int part = (nsplit > 1) ? (int)(blockIdx.x % (unsigned)PLL_PARTS) : 0;
    const int64_t jslot = (nsplit > 1) ? (int64_t)(blockIdx.x / (unsigned)PLL_PARTS) : (int64_t)blockIdx.x;
    if (job_count && jslot >= *job_count) return;      // wave-uniform: behind the end of a compacted job list
    const int64_t job = job_order ? job_order[jslot] : jslot;
    double *const outp = out + job * (nsplit > 1 ? PLL_PARTS : 1) + part;
    const int b = (int)(job % B);
    const int64_t p = job / B;
    const BandDev *bd = bands + b;
    if (owner && owner[p] < 0) {        // a retired proposal slot (the device-resident slice sampler's finished chains)
        if (lane == 0) *outp = 0.0;
        return;
    }
    const int64_t ob = (int64_t)(owner ? owner[p] : 0) * B + b;
    RecU rec = rec_unpack(rec_fetch(recs + (int64_t)b * P, (int)p, lane));
    // MODE 3 (mass of the unit stamp on the source's OWN box, sources.py:338-339): the box is the record's
    const int4 bx = (MODE == 3) ? make_int4(rec.x0, rec.x1, rec.y0, rec.y1) : pbox[ob];
    const int nx = bx.y - bx.x, ny = bx.w - bx.z;
    const int4 ev = (MODE == 0 && nzbox) ? nzbox[ob] : bx;     // the rectangle that has to be evaluated
    const double counts = rec.scale;
    const double wsum = bd->w[0] + bd->w[1] + bd->w[2];
    if (nx <= 0 || ny <= 0 || (MODE == 3 && rec.type < 0)) {   // no sample image in this band / no stamp
        if (lane == 0) *outp = 0.0;
        return;
    }
    if (rec.type == -3 && MODE == 0) {  // psf_ns is None (sources.py:160-163)
        if (lane == 0) *outp = (part == 0) ? -counts * wsum : 0.0;
        return;
    }
    if (rec.type < 0) rec.type = (rec.type == -2) ? 1 : 0;    // imposed limits: the kind still renders
    rec.scale = 1.0;                                          // the tile holds the unit stamp
    et[lane] = exp2((double)lane * (1.0 / 64.0));
    const double lt_ic = c_log_ic[lane], lt_lc = c_log_lc[lane];      // this lane's two entries of the log table
    const LaneConst lc = lane_consts(lane, bd);
    const double eps = bd->eps;
    // the job's components do not depend on the chunk: one per lane, kept for every chunk's table
    Comp cj;
    if (lane < ((rec.type == 0) ? K_PSF : K_GAL)) cj = make_comp_lc(lc, rec);
    if (MODE == 0) {
        // A proposal so far from the patch that every component's exponent stays below -750 on the
        // whole rectangle evaluates to exactly 0 there (exp underflows below -745.2): every pixel
        // is masked out (model_patch > 0, sources.py:172) and only the mass term is left.  The
        // first shrink steps of a slice sampler started from a wide interval are of this kind.
        bool alive = false;
        if (lane < ((rec.type == 0) ? K_PSF : K_GAL)) {
            const Comp &c = cj;
            const double qmin = quad_min_rect(c.qa, c.qb, c.qc, (double)ev.x - c.mx, (double)(ev.y - 1) - c.mx,
                                              (double)ev.z - c.my, (double)(ev.w - 1) - c.my);
            alive = !(0.5 * qmin > 750.0);
        }
        if (__ballot(alive) == 0ull) {
            if (lane == 0) *outp = (part == 0) ? -counts * wsum : 0.0;
            return;
        }
    }
    if (MODE == 0 && nzmode && nzmode[ob]) {
        // ---- at the photons: sum over the patch's photon list of z log(counts * stamp(x, y)) ---------------------------
        // Every lane takes a photon-holding pixel and sums all K components there by direct exponentials (table exp):
        // e_k(x, y) = c0 + c1 X + c2 Y + c3 X^2 + c4 X Y + c5 Y^2 in coordinates X, Y relative to the SOURCE (so that no
        // term is large where the value matters), 5 fma + the table exp + 1 fma per component.  Nothing is dropped.  A job
        // is never dealt to several blocks here (it is short): part 0 does it, the other parts return 0.
        if (part != 0) {
            if (lane == 0) *outp = 0.0;
            return;
        }
        double *ltq = acc;                       // the log table (128 doubles) and the components (8 doubles each) live in the tile's LDS
        double *cq = acc + 128;
        const int K = (rec.type == 0) ? K_PSF : K_GAL;
        ltq[lane] = lt_ic;
        ltq[64 + lane] = lt_lc;
        if (lane < K) {
            const Comp &c = cj;
            const double ux = c.mx - rec.px, uy = c.my - rec.py;         // the component's centre seen from the source
            const double qa = c.qa * EXP_SCALE, qb = c.qb * EXP_SCALE, qc = c.qc * EXP_SCALE;
            // -1/2 (qa (X-ux)^2 + 2 qb (X-ux)(Y-uy) + qc (Y-uy)^2)
            cq[8 * lane + 0] = -0.5 * (qa * ux * ux + 2.0 * qb * ux * uy + qc * uy * uy);
            cq[8 * lane + 1] = qa * ux + qb * uy;
            cq[8 * lane + 2] = qb * ux + qc * uy;
            cq[8 * lane + 3] = -0.5 * qa;
            cq[8 * lane + 4] = -qb;
            cq[8 * lane + 5] = -0.5 * qc;
            cq[8 * lane + 6] = c.A;
            cq[8 * lane + 7] = 0.0;
        }
        __syncthreads();
        const NzEntry *L = nzlist + nzoff[ob];
        const int n = (int)(nzoff[ob + 1] - nzoff[ob]);
        double a = 0.0;
        // A component's eight constants come from LDS (the same address for every lane: the read still returns 512 B
        // per wave-instruction), so a lane takes up to FOUR photons per trip and the constants are read once for all
        // of them: with one photon per lane the kernel was bound by those reads, not by the arithmetic.
        int i0 = 0;
        for (; n - i0 > 128; i0 += 256) a += nz_trip<4>(L, n, i0, lane, K, cq, ltq, et, rec.px, rec.py, counts);
        if (n - i0 > 64) { a += nz_trip<2>(L, n, i0, lane, K, cq, ltq, et, rec.px, rec.py, counts); i0 += 128; }
        if (n - i0 > 0) a += nz_trip<1>(L, n, i0, lane, K, cq, ltq, et, rec.px, rec.py, counts);
        a = wave_sum(a);
        if (lane == 0) *outp = a - counts * wsum;
        return;
    }
    // mode 1 drops against the sky seen from the unit stamp: counts * g < eps e^-T
    int dropmode = HW_DROP_NONE;
    double log_floor = 0.0;
    if (Tdrop > 0.0) {
        if (MODE == 0 || MODE == 2 || MODE == 3 || MODE == 4) dropmode = HW_DROP_SELF;
        else if (eps > 0.0 && counts > 0.0) { dropmode = HW_DROP_SKY; log_floor = (double)__logf((float)(eps / counts)); }
    }
    // the patch data: the caller's / the resident split's packed buffer (TZ), or -- none given -- the observed image on the box
    const TZ *zd = (MODE == 3 || !data) ? nullptr : data + offsets[ob];
    const double *zn = (MODE == 3 || data) ? nullptr : nelec + (int64_t)b * H * W + (int64_t)bx.z * W + bx.x;
    const int64_t zpitch = data ? nx : W;
    double a = 0.0, m = 0.0;
    double apart[PLL_PARTS] = {0.0, 0.0, 0.0, 0.0};     // MODE 0: the chunk classes' sums (statically indexed below)
    int chunk = 0;
    for (int Y0 = ev.z; Y0 < ev.w; Y0 += HW_TH) {
        const int rb = min(HW_TH, ev.w - Y0);
        for (int X0 = ev.x; X0 < ev.y; X0 += HW_TW, chunk++) {
            if (nsplit > 1 && chunk % PLL_PARTS != part) continue;
            const int xi = X0 + col;
            const bool on = xi < ev.y;
#pragma unroll
            for (int r = 0; r < HW_TH / 2; r++) acc[r * 64 + lane] = 0.0;
            bool direct;
            const int Kk = hw_build(T, lc, rec, lane, dropmode, Tdrop, log_floor, Y0, X0, min(ev.y, X0 + HW_TW) - 1, 0, rb, direct, &cj);
            hw_walk(T, et, Kk, (double)xi, Y0, 0, rb, on, direct, acc, lane);
            __syncthreads();
            if (MODE != 3) {
                lt[lane] = lt_ic;
                lt[64 + lane] = lt_lc;
                __syncthreads();
            }
            if (MODE == 3) {
#pragma unroll
                for (int r = 0; r < HW_TH / 2; r++)
                    if (on && 2 * r + half < rb) m += acc[r * 64 + lane];
                __syncthreads();
                continue;
            }
            // the chunk's patch data, 16 rows of loads in flight at a time (addresses clamped into
            // the chunk instead of predicated), issued only once the walk's registers are free
            const int64_t zo = (int64_t)(Y0 - bx.z) * zpitch + (min(xi, ev.y - 1) - bx.x);
            for (int r0 = 0; r0 < HW_TH / 2 && 2 * r0 < rb; r0 += 8) {
                double zz[8], bg[8];
                if (zd) {
                    TZ raw[8];
#pragma unroll
                    for (int r = 0; r < 8; r++) raw[r] = zd[zo + (int64_t)min(2 * (r0 + r) + half, rb - 1) * zpitch];
#pragma unroll
                    for (int r = 0; r < 8; r++) zz[r] = (double)raw[r];
                } else {
#pragma unroll
                    for (int r = 0; r < 8; r++) zz[r] = zn[zo + (int64_t)min(2 * (r0 + r) + half, rb - 1) * zpitch];
                }
                if (MODE == 4) {        // the background plane follows the data plane (nx * ny values further on)
#pragma unroll
                    for (int r = 0; r < 8; r++) bg[r] = (double)zd[zo + (int64_t)nx * ny + (int64_t)min(2 * (r0 + r) + half, rb - 1) * zpitch];
                }
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    if (on && 2 * (r0 + r) + half < rb) {
                        double v = counts * acc[(r0 + r) * 64 + lane];
                        if (MODE == 0) {
#ifndef PLL_NO_ZSKIP
                            // a pixel without photons adds 0 * log(v) = 0: skipped (a step of 64 such pixels skips the log)
                            if (v > 0.0 && zz[r] != 0.0) a += log_tab(v, lt) * zz[r];
#else
                            if (v > 0.0) a += log_tab(v, lt) * zz[r];
#endif
                        } else if (MODE == 2) {
                            if (v > 0.0) { a += log_tab(v, lt) * zz[r]; m += v; }
                        } else if (MODE == 4) {
                            v += bg[r];
                            if (v > 0.0 && zz[r] >= 0.0) { a += log_tab(v, lt) * zz[r]; m += v; }
                        } else {
                            v += eps;
                            a += log_tab(v, lt) * zz[r];
                            m += v;
                        }
                    }
                }
            }
            if (MODE == 0) {        // wave-uniform class: four predicated adds, no dynamically indexed registers
#pragma unroll
                for (int k = 0; k < PLL_PARTS; k++)
                    if ((chunk % PLL_PARTS) == k) apart[k] += a;
                a = 0.0;
            }
            __syncthreads();
        }
    }
    if (MODE == 0) {
        // ((A0 - counts * wsum) + A1) + A2) + A3: one block forms it itself, PLL_PARTS blocks leave the additions to the reader
        double tot = 0.0;
#pragma unroll
        for (int k = 0; k < PLL_PARTS; k++) {
            if (nsplit > 1 && k != part) continue;
            if (k > 0 && k >= chunk) continue;      // a class without any chunk is 0.0: adding it changes nothing (`chunk` = the job's chunk count here)
            double ak = wave_sum(apart[k]);
            if (k == 0) ak -= counts * wsum;
            tot = (nsplit > 1 || k == 0) ? ak : tot + ak;
        }
        if (lane == 0) *outp = tot;
        return;
    }
    a = wave_sum(a);
    m = wave_sum(m);
    if (lane == 0) *outp = (MODE == 3) ? m : a - m;
}

// work estimate of every (chain, band) job of the device slice sampler: components x pixels of the
// photon rectangle its likelihood evaluates; k_order turns it into a heaviest-first launch order
// (a round's 50 000 one-wave jobs differ by two orders of magnitude: in index order the launch
// ends on a few late galaxies)
__global__ void __launch_bounds__(256)
k_job_work(const int *__restrict__ type, const int4 *__restrict__ nzbox, int64_t S, int B, int *__restrict__ work) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S * B) return;
    const int4 q = nzbox[i];
    const long long area = (q.y > q.x && q.w > q.z) ? (long long)(q.y - q.x) * (q.w - q.z) : 0;
    const long long w = area * ((type[i / B] == 0) ? K_PSF : K_GAL) + 64;
    work[i] = (int)min(w, (long long)0x3fffffff);
}
