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
//   mode 4:  sum_{z unmasked, m+bg>0} log(m + bg) * z - (m + bg)   on a given BACKGROUND: the patch data is two planes,
//            z then bg (everything else in the field rendered on the box); a NaN z marks a masked pixel (a negative count is data).  The
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
#include "k_slice_state.h"
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
            if (v > 0.0 && zi == zi) { a += log(v) * zi; m += v; }       // a NaN count marks a masked pixel
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
            v[p] = fma(A, exp_tab256_p3(e, et), v[p]);     // e is finite; far tails flush to 0 inside; et: the 256-entry table
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
// MODE 3 (the stamp-mass kernel) at three waves per SIMD keeps 168 VGPRs and 24 B of scratch (5 spilled registers, re-read
// once per chunk); -DPLL_MASS_TWO_WAVES gives it 256 registers, no scratch and two waves: the A/B of tools/ab_scratch.sh
#ifdef PLL_MASS_TWO_WAVES
#define PLL_MASS_WAVES 2
#else
#define PLL_MASS_WAVES 3
#endif
template <int MODE, typename TZ = double>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(MODE == 3 ? PLL_MASS_WAVES : 2, MODE == 3 ? PLL_MASS_WAVES : 2)))
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
              const int *__restrict__ nzmode = nullptr /* MODE 0, resident patches: per patch 1 = scored at its photons by k_patch_ll_nz, not here */,
              int encoded = 0 /* job_order holds (job << 3 | part << 1 | split) per BLOCK: a list may deal some jobs (the long ones) to
                                 PLL_PARTS blocks and leave the others whole -- what the slice sampler launches in every round */,
              int ostride = 0 /* doubles per job in `out` (0: PLL_PARTS when nsplit > 1, else 1).  A job that is not dealt writes its
                                 value to slot 0 and zeros behind it, so that whoever adds a job's slots in order gets the same bits
                                 whether the job was dealt or not */) {
    __shared__ double acc[HW_TH * HW_TW];
    __shared__ CompTab T;
    __shared__ double et[64];
    // the log table lives in the component table's LDS between a chunk's walk and the next chunk's
    // build (20 184 B per wave: 8 waves per CU; with a table of its own 21 208 B: 7)
    double *lt = reinterpret_cast<double *>(&T);
    const int lane = threadIdx.x;
    const int half = lane >> 5, col = lane & 31;
    int part = (nsplit > 1) ? (int)(blockIdx.x % (unsigned)PLL_PARTS) : 0;
    const int64_t jslot = (nsplit > 1 && !encoded) ? (int64_t)(blockIdx.x / (unsigned)PLL_PARTS) : (int64_t)blockIdx.x;
    if (job_count && jslot >= *job_count) return;      // wave-uniform: behind the end of a compacted job list
    int64_t job = job_order ? job_order[jslot] : jslot;
    bool split = nsplit > 1;
    if (encoded) { split = (job & 1) != 0; part = (int)((job >> 1) & 3); job >>= 3; }
    const int ostr = ostride ? ostride : (nsplit > 1 ? PLL_PARTS : 1);
    double *const outp = out + job * ostr + (split ? part : 0);
    // a job that is not dealt owns all of its ostr slots: value in the first, zeros behind
    const int nfill = split ? 1 : ostr;
#define PLL_PUT(v) do { if (lane < nfill) outp[lane] = (lane == 0) ? (v) : 0.0; } while (0)
    const int b = (int)(job % B);
    const int64_t p = job / B;
    const BandDev *bd = bands + b;
    if (owner && owner[p] < 0) {        // a retired proposal slot (the device-resident slice sampler's finished chains)
        PLL_PUT(0.0);
        return;
    }
    const int64_t ob = (int64_t)(owner ? owner[p] : 0) * B + b;
    if (MODE == 0 && nzmode && nzmode[ob]) return;      // this patch is scored at its photons, by k_patch_ll_nz
    RecU rec = rec_unpack(rec_fetch(recs + (int64_t)b * P, (int)p, lane));
    // MODE 3 (mass of the unit stamp on the source's OWN box, sources.py:338-339): the box is the record's
    const int4 bx = (MODE == 3) ? make_int4(rec.x0, rec.x1, rec.y0, rec.y1) : pbox[ob];
    const int nx = bx.y - bx.x, ny = bx.w - bx.z;
    const int4 ev = (MODE == 0 && nzbox) ? nzbox[ob] : bx;     // the rectangle that has to be evaluated
    const double counts = rec.scale;
    const double wsum = bd->w[0] + bd->w[1] + bd->w[2];
    if (nx <= 0 || ny <= 0 || (MODE == 3 && rec.type < 0)) {   // no sample image in this band / no stamp
        PLL_PUT(0.0);
        return;
    }
    if (rec.type == -3 && MODE == 0) {  // psf_ns is None (sources.py:160-163)
        PLL_PUT((part == 0 || !split) ? -counts * wsum : 0.0);
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
            PLL_PUT((part == 0 || !split) ? -counts * wsum : 0.0);
            return;
        }
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
            if (split && chunk % PLL_PARTS != part) continue;
            const int xi = X0 + col;
            const bool on = xi < ev.y;
            if (MODE != 3) {
#pragma unroll
                for (int r = 0; r < HW_TH / 2; r++) acc[r * 64 + lane] = 0.0;
            }
            bool direct;
            const int Kk = hw_build(T, lc, rec, lane, dropmode, Tdrop, log_floor, Y0, X0, min(ev.y, X0 + HW_TW) - 1, 0, rb, direct, &cj);
            if (MODE == 3) {
                // the stamp's mass wants no tile: every lane sums what it evaluates (its columns' rows of its groups), in a fixed order
                hw_walk<true>(T, et, Kk, (double)xi, Y0, 0, rb, on, direct, nullptr, lane, &m);
                continue;
            }
            hw_walk(T, et, Kk, (double)xi, Y0, 0, rb, on, direct, acc, lane);
            __syncthreads();
            lt[lane] = lt_ic;
            lt[64 + lane] = lt_lc;
            __syncthreads();
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
                            if (v > 0.0 && zz[r] == zz[r]) { a += log_tab(v, lt) * zz[r]; m += v; }    // NaN: masked
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
            if (split && k != part) continue;
            if (k > 0 && k >= chunk) continue;      // a class without any chunk is 0.0: adding it changes nothing (`chunk` = the job's chunk count here)
            double ak = wave_sum(apart[k]);
            if (k == 0) ak -= counts * wsum;
            tot = (split || k == 0) ? ak : tot + ak;
        }
        tot = __shfl(tot, 0);
        PLL_PUT(tot);
        return;
    }
    a = wave_sum(a);
    m = wave_sum(m);
    if (lane == 0) *outp = (MODE == 3) ? m : a - m;
#undef PLL_PUT
}

// A galaxy's 42 components are 3 x 14: covariance v_j W + P_k about the same centre for the 14 profile variances v_j of PSF
// component k.  In the basis that diagonalises the pair (W, P_k) -- M_k with M^T W M = I, M^T P_k M = diag(l1, l2) -- every
// one of the 14 quadratic forms is a1^2 / (v_j + l1) + a2^2 / (v_j + l2) with a = M_k^T (x - centre_k): the rotated
// coordinates are formed once per (PSF component, photon) (6 instructions) and a component's exponent is ONE multiply and
// ONE fma (both terms <= 0: nothing cancels) instead of the five fma of the general quadratic -- 15 VALU per (component,
// photon) instead of 19.  gk: per PSF component p0 p1 p2 q0 q1 q2 (a1 = p0 + p1 X + p2 Y, a2 likewise; 8 doubles apart);
// gc: per component s1 s2 A (s = -1/2 * 64/ln2 / (v_j + l); 4 doubles apart), PSF-major like the lanes.
template <int P>
__device__ __forceinline__ double nz_trip_gal(const NzEntry *__restrict__ L, int n, int i0, int lane,
                                              const double *__restrict__ gk, const double *__restrict__ gc,
                                              const double *__restrict__ ltq, const double *__restrict__ et,
                                              double px, double py, double counts) {
    double X[P], Y[P], v[P], z[P];
#pragma unroll
    for (int p = 0; p < P; p++) {
        const int i = i0 + 64 * p + lane;
        const NzEntry en = L[min(i, n - 1)];
        X[p] = (double)(en.xy & 0xffff) - px;
        Y[p] = (double)((unsigned)en.xy >> 16) - py;
        z[p] = (i < n) ? (double)en.z : 0.0;
        v[p] = 0.0;
    }
    // (rolled loops with their table reads inside: unrolled, the compiler hoists all 144 table values out of the trip loop
    // into registers it does not have -- 1 KB of scratch per lane)
#pragma nounroll
    for (int k = 0; k < K_PSF; k++) {
        asm volatile("" ::: "memory");
        const double *g = gk + 8 * k;
        const double p0 = g[0], p1 = g[1], p2 = g[2], q0 = g[3], q1 = g[4], q2 = g[5];
        double r1[P], r2[P];
#pragma unroll
        for (int p = 0; p < P; p++) {
            const double a1 = fma(p1, X[p], fma(p2, Y[p], p0));
            const double a2 = fma(q1, X[p], fma(q2, Y[p], q0));
            r1[p] = a1 * a1;
            r2[p] = a2 * a2;
        }
#pragma nounroll
        for (int j = 0; j < K_PROF; j++) {
            asm volatile("" ::: "memory");
            const double *c = gc + 4 * (k * K_PROF + j);
            const double s1 = c[0], s2 = c[1], A = c[2];
#pragma unroll
            for (int p = 0; p < P; p++) {
                const double e = fma(s1, r1[p], s2 * r2[p]);
                v[p] = fma(A, exp_tab256_p3(e, et), v[p]);
            }
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

// ---- mode 0 at the photons ---------------------------------------------------------------------------------
// sum over a patch's photon list (k_nz_compact) of z log(counts * stamp(x, y)) - counts * sum(psf weights): the same
// value k_patch_ll_hw<0> forms over the photon rectangle, where only pixels with z > 0 contribute.  Every lane takes
// photon-holding pixels and sums all K components there by direct exponentials (table exp), nothing is dropped:
//     exponent = c0 + c1 X + c2 Y + c3 X^2 + c4 X Y + c5 Y^2,  X, Y relative to the SOURCE (no term is large where the
//     value matters): 5 fma + the table exp + 1 fma per component and photon.
// A kernel of its own because it needs a third of the dense kernel's registers and a fifth of its LDS: four waves per
// SIMD instead of two hide the job's chain of dependent loads (job -> owner -> record -> list).
#define NZ_TRIP 256          // photons per trip of the main loop (four per lane)
#define NZ_SPLIT_PHOTONS 2048   // a list longer than this is worth dealing to PLL_PARTS blocks (k_job_work, k_slice_live_jobs)
// FUSE = true: the instantiation of the fused slice rounds (CEL_OPT_SLICE_FUSE, an opt-in experiment: k_slice_state.h).  It is
// a kernel of its own because the code behind `if (fzp)` cost the plain kernel 1.3 % (10.96 -> 11.10 ms per sweep, same box, same
// chains) although the branch was never taken.
template <bool FUSE>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4)))
k_patch_ll_nz(const BandDev *__restrict__ bands, int B, int64_t P, const SrcRec *__restrict__ recs,
              const int *__restrict__ owner, const int4 *__restrict__ pbox, const int4 *__restrict__ nzbox,
              const int *__restrict__ nzmode, const int64_t *__restrict__ nzoff, const NzEntry *__restrict__ nzlist,
              double *__restrict__ out /* PLL_PARTS doubles per job */,
              const int *__restrict__ job_order /* per BLOCK: job << 3 | part << 1 | split, or nullptr: block = job, whole */,
              const int *__restrict__ job_count,
              const SliceFuse *__restrict__ fzp /* FUSE: the block that finishes a chain's last job of the round steps the chain
                              (k_slice_state.h).  A pointer, not the 200-byte struct: as a kernel argument its fields sat in scalar
                              registers through the photon loop (40 SGPR spills, the kernel 11 % slower) */) {
    __shared__ double et[256];                 // 2^(j/256): the photon kernel's exponentials take a cubic on it (exp_tab256_p3)
    __shared__ double ltq[128];
    __shared__ double cq[8 * K_GAL];
    __shared__ double gq[32 + 4 * K_GAL];      // a galaxy's table in the rotated form (nz_trip_gal): 3 x 8 + 42 x 4 doubles
    const int lane = threadIdx.x;
    const int64_t jslot = blockIdx.x;
    if (job_count && jslot >= *job_count) return;
    int64_t job = job_order ? job_order[jslot] : jslot;
    bool split = false;
    int part = 0;
    if (job_order) { split = (job & 1) != 0; part = (int)((job >> 1) & 3); job >>= 3; }
    // The trips of a list fall into PLL_PARTS classes (trip t -> class t mod PLL_PARTS) that are ALWAYS summed apart: a
    // whole job writes the four class sums to its four slots, a dealt job's block its class to its slot, and whoever
    // adds the slots in order (k_slice_step, the host loop of cel_patch_loglik_multi) gets the same bits either way.
    double *const outp = out + job * PLL_PARTS;
    const int b = (int)(job % B);
    const int64_t p = job / B;
    if (owner && owner[p] < 0) return;                  // a retired chain: nobody reads its slots
    const int64_t ob = (int64_t)(owner ? owner[p] : 0) * B + b;
    if (!nzmode[ob]) return;                            // scored densely, by k_patch_ll_hw<0>
    const BandDev *bd = bands + b;
    RecU rec = rec_unpack(rec_fetch(recs + (int64_t)b * P, (int)p, lane));
    const int4 bx = pbox[ob], ev = nzbox[ob];
    const double counts = rec.scale;
    const double wsum = bd->w[0] + bd->w[1] + bd->w[2];
    double mass_only = 0.0;                             // the value when no pixel contributes
    bool done = false;
    if (bx.y - bx.x <= 0 || bx.w - bx.z <= 0) done = true;                              // no sample image in this band
    else if (rec.type == -3) { mass_only = -counts * wsum; done = true; }               // psf_ns is None (sources.py:160-163)
    const int K = (rec.type == 0 || rec.type == -1) ? K_PSF : K_GAL;
    bool use_gal = false;
    if (!done) {
        if (rec.type < 0) rec.type = (rec.type == -2) ? 1 : 0;
        rec.scale = 1.0;
        {   // 2^(j/256), j = 4 lane + k: the 64-entry value times 2^(k/256)
            const double e64 = exp2((double)lane * (1.0 / 64.0));
            et[4 * lane + 0] = e64;
            et[4 * lane + 1] = e64 * 1.0027112750502025;      // 2^(1/256)
            et[4 * lane + 2] = e64 * 1.0054299011128027;      // 2^(2/256)
            et[4 * lane + 3] = e64 * 1.0081558981184175;      // 2^(3/256)
        }
        ltq[lane] = c_log_ic[lane];
        ltq[64 + lane] = c_log_lc[lane];
        const LaneConst lc = lane_consts(lane, bd);
        bool alive = false, gal_ok = true;
        if (lane < K) {
            const Comp c = make_comp_lc(lc, rec);
            // exactly 0 on the whole photon rectangle (exp underflows below -745.2): only the mass term is left, as in the dense kernel
            const double qmin = quad_min_rect(c.qa, c.qb, c.qc, (double)ev.x - c.mx, (double)(ev.y - 1) - c.mx,
                                              (double)ev.z - c.my, (double)(ev.w - 1) - c.my);
            alive = !(0.5 * qmin > 750.0);
            const double ux = c.mx - rec.px, uy = c.my - rec.py;         // the component's centre seen from the source
            const double qa = c.qa * EXP_SCALE256, qb = c.qb * EXP_SCALE256, qc = c.qc * EXP_SCALE256;
            cq[8 * lane + 0] = -0.5 * (qa * ux * ux + 2.0 * qb * ux * uy + qc * uy * uy);
            cq[8 * lane + 1] = qa * ux + qb * uy;
            cq[8 * lane + 2] = qb * ux + qc * uy;
            cq[8 * lane + 3] = -0.5 * qa;
            cq[8 * lane + 4] = -qb;
            cq[8 * lane + 5] = -0.5 * qc;
            cq[8 * lane + 6] = c.A;
            cq[8 * lane + 7] = 0.0;
            if (K == K_GAL) {
                // the pair (W, P_k) diagonalised: W = L L^T, C = L^-1 P_k L^-T, one Jacobi rotation of C (nz_trip_gal)
                // (round 6: reciprocals and reciprocal roots by fp32 seed + two Newton steps -- rounding -- where the operand is a
                // variance-like number; the one quotient whose denominator can be anything, tau, stays an IEEE division.  Four
                // square roots and six divisions were a sixth of a job's ~600 set-up instructions.)
                const double ia = rsqrt64(rec.w00), l21 = rec.w01 * ia, d22 = rec.w11 - l21 * l21;
                const double ic = rsqrt64(d22), ib = -l21 * ia * ic;
                const double C11 = ia * ia * lc.g_cxx, C12 = ia * (ib * lc.g_cxx + ic * lc.g_cxy);
                const double C22 = ib * ib * lc.g_cxx + 2.0 * ib * ic * lc.g_cxy + ic * ic * lc.g_cyy;
                double cs = 1.0, sn = 0.0, lam1 = C11, lam2 = C22;
                if (C12 != 0.0) {
                    const double tau = (C22 - C11) / (2.0 * C12);
                    const double t = copysign(1.0, tau) / (fabs(tau) + sqrt(1.0 + tau * tau));      // (tau may be anything: IEEE)
                    cs = rsqrt64(1.0 + t * t); sn = t * cs;                                        // |t| <= 1
                    lam1 = C11 - t * C12; lam2 = C22 + t * C12;
                }
                const double d1 = lc.g_var + lam1, d2 = lc.g_var + lam2;
                const double gs1 = -0.5 * EXP_SCALE256 * rcp64(d1), gs2 = -0.5 * EXP_SCALE256 * rcp64(d2);
                const double al1 = cs * ia - sn * ib, be1 = -sn * ic, al2 = sn * ia + cs * ib, be2 = cs * ic;
                const double g0 = -(al1 * ux + be1 * uy), g3 = -(al2 * ux + be2 * uy);
                gq[32 + 4 * lane + 0] = gs1; gq[32 + 4 * lane + 1] = gs2; gq[32 + 4 * lane + 2] = c.A;
                if (lane % K_PROF == 0) {
                    double *g = gq + 8 * (lane / K_PROF);
                    g[0] = g0; g[1] = al1; g[2] = be1; g[3] = g3; g[4] = al2; g[5] = be2;
                }
                gal_ok = (rec.w00 > 0.0) && (d22 > 0.0) && (d1 > 0.0) && (d2 > 0.0) && (gs1 == gs1) && (gs2 == gs2) &&
                         (g0 == g0) && (g3 == g3) && (fabs(g0) < 1e300) && (fabs(g3) < 1e300);
            }
        }
        if (__ballot(alive) == 0ull) { mass_only = -counts * wsum; done = true; }
        // a galaxy whose decomposition is sound on every lane takes the rotated form: the table is rewritten in its layout
        use_gal = (K == K_GAL) && (__ballot(lane < K && !gal_ok) == 0ull);
    }
    if (done) {
        if (split) { if (lane == 0) sl_put(outp + part, (part == 0) ? mass_only : 0.0, FUSE); }
        else if (lane < PLL_PARTS) sl_put(outp + lane, (lane == 0) ? mass_only : 0.0, FUSE);
        if (FUSE) sl_fused_step(*fzp, p, lane);
        return;
    }
    __syncthreads();
    const NzEntry *L = nzlist + nzoff[ob];
    const int n = (int)(nzoff[ob + 1] - nzoff[ob]);
    double ac[PLL_PARTS] = {0.0, 0.0, 0.0, 0.0};
    // A component's constants come from LDS (one address for every lane: the read still returns 512 B per
    // wave-instruction), so a lane takes up to FOUR photons per trip and the constants are read once for all of
    // them: with one photon per lane the reads, not the arithmetic, bound the kernel.
    int t = 0;
    for (int i0 = 0; i0 < n; i0 += NZ_TRIP, t++) {
        const int cls = t & (PLL_PARTS - 1);
        if (split && cls != part) continue;
        const int left = n - i0;
        double a;
        if (use_gal) {
            if (left > 128) a = nz_trip_gal<4>(L, n, i0, lane, gq, gq + 32, ltq, et, rec.px, rec.py, counts);
            else if (left > 64) a = nz_trip_gal<2>(L, n, i0, lane, gq, gq + 32, ltq, et, rec.px, rec.py, counts);
            else a = nz_trip_gal<1>(L, n, i0, lane, gq, gq + 32, ltq, et, rec.px, rec.py, counts);
        } else if (left > 128) a = nz_trip<4>(L, n, i0, lane, K, cq, ltq, et, rec.px, rec.py, counts);
        else if (left > 64) a = nz_trip<2>(L, n, i0, lane, K, cq, ltq, et, rec.px, rec.py, counts);
        else a = nz_trip<1>(L, n, i0, lane, K, cq, ltq, et, rec.px, rec.py, counts);
#pragma unroll
        for (int k = 0; k < PLL_PARTS; k++)
            if (cls == k) ac[k] += a;               // wave-uniform class: predicated adds, no dynamically indexed registers
    }
    double mine = 0.0;                                  // lane k < PLL_PARTS ends up with class k's sum
#pragma unroll
    for (int k = 0; k < PLL_PARTS; k++) {
        if (split && k != part) continue;
        double s = wave_sum(ac[k]);
        if (k == 0) s -= counts * wsum;
        s = __shfl(s, 0);
        if (lane == k) mine = s;
    }
    if (split) { if (lane == part) sl_put(outp + part, mine, FUSE); }
    else if (lane < PLL_PARTS) sl_put(outp + lane, mine, FUSE);
    if (FUSE) sl_fused_step(*fzp, p, lane);
}

// work estimate of every (chain, band) job of the device slice sampler: components x pixels of the
// photon rectangle its likelihood evaluates; k_order turns it into a heaviest-first launch order
// (a round's 50 000 one-wave jobs differ by two orders of magnitude: in index order the launch
// ends on a few late galaxies)
// Entries are per (job, part): index i = job * PLL_PARTS + part.  A long job (more than PLL_SPLIT_CHUNKS chunks /
// NZ_SPLIT_PHOTONS photons) is dealt to PLL_PARTS blocks -- all its parts are members, each with a quarter of the work
// -- a short one stays whole (part 0 the member).  work = estimate << 1 | dealt; -1: not a block of this list.
#define PLL_SPLIT_CHUNKS 6
__global__ void __launch_bounds__(256)
k_job_work(const int *__restrict__ type, const int4 *__restrict__ nzbox, int64_t S, int B, int *__restrict__ work /* S*B*PLL_PARTS */,
           const int *__restrict__ nzmode = nullptr, const int *__restrict__ nnz = nullptr,
           int *__restrict__ work_nz = nullptr /* with nzmode: the blocks of the jobs scored at their photons */,
           int *__restrict__ need = nullptr /* per chain, zeroed by the caller: the blocks the two lists hold for it (SliceFuse) */) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S * B * PLL_PARTS) return;
    const int64_t job = i / PLL_PARTS;
    const int part = (int)(i % PLL_PARTS);
    const int4 q = nzbox[job];
    const long long area = (q.y > q.x && q.w > q.z) ? (long long)(q.y - q.x) * (q.w - q.z) : 0;
    const int K = (type[job / B] == 0) ? K_PSF : K_GAL;
    const bool sparse = nzmode && nzmode[job];
    const long long chunks = area > 0 ? (long long)((q.y - q.x + HW_TW - 1) / HW_TW) * ((q.w - q.z + HW_TH - 1) / HW_TH) : 0;
    const long long wd = area * K + 64;
    const bool deal_d = chunks > PLL_SPLIT_CHUNKS;
    int w = -1;
    if (!sparse && (deal_d || part == 0)) w = (int)(min(deal_d ? wd / PLL_PARTS : wd, (long long)0x1fffffff) << 1) | (deal_d ? 1 : 0);
    work[i] = w;
    if (work_nz) {
        int wn = -1;
        if (sparse) {
            const long long cnt = nnz[job];
            const bool deal_n = cnt > NZ_SPLIT_PHOTONS;
            const long long ws = cnt * K * 3 + 64;
            if (deal_n || part == 0) wn = (int)(min(deal_n ? ws / PLL_PARTS : ws, (long long)0x1fffffff) << 1) | (deal_n ? 1 : 0);
        }
        work_nz[i] = wn;
        if (need && (w >= 0 || wn >= 0)) atomicAdd(&need[job / B], 1);
    }
}

// k_order (k_prep_bin.h) for the (job, part) entries of k_job_work: the members (work >= 0) come first, heaviest first,
// each written as the kernels' block descriptor job << 3 | part << 1 | dealt (= 2 i + the work's low bit); *nmember
// receives their number.  One block.
__global__ void __launch_bounds__(1024)
k_order_members(const int *__restrict__ work, int T, int *__restrict__ order, int *__restrict__ nmember) {
    __shared__ int hist[257];
    __shared__ int red[1024];
    const int tid = threadIdx.x;
    int mx = 0;
    for (int i = tid; i < T; i += 1024) mx = max(mx, work[i] >> 1);
    red[tid] = mx;
    if (tid < 257) hist[tid] = 0;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if (tid < o) red[tid] = max(red[tid], red[tid + o]);
        __syncthreads();
    }
    const long long wmax = red[0] > 0 ? red[0] : 1;
    for (int i = tid; i < T; i += 1024) {
        const int w = work[i];
        atomicAdd(&hist[w < 0 ? 256 : 255 - (int)(((long long)(w >> 1) * 255) / wmax)], 1);
    }
    __syncthreads();
    if (tid == 0) {   // exclusive scan, bucket 0 = heaviest, bucket 256 = the non-members
        int run = 0;
        for (int k = 0; k < 257; k++) { int c = hist[k]; hist[k] = run; run += c; }
        *nmember = hist[256];
    }
    __syncthreads();
    for (int i = tid; i < T; i += 1024) {
        const int w = work[i];
        order[atomicAdd(&hist[w < 0 ? 256 : 255 - (int)(((long long)(w >> 1) * 255) / wmax)], 1)] = 2 * i + (w < 0 ? 0 : (w & 1));
    }
}

// The same in two steps, 4 x faster (three quarters of the (job, part) entries are not members): the members first
// compacted by the whole GPU (one atomic per wave), then ordered by one block.
__global__ void __launch_bounds__(256)
k_list_members(const int *__restrict__ work, int T, int *__restrict__ work_c, int *__restrict__ desc_c, int *__restrict__ nmember) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int w = (i < T) ? work[i] : -1;
    const int at = wave_reserve(nmember, w >= 0 ? 1 : 0);
    if (w >= 0) { work_c[at] = w >> 1; desc_c[at] = 2 * i + (w & 1); }
}

__global__ void __launch_bounds__(1024)
k_order_compact(const int *__restrict__ work_c, const int *__restrict__ desc_c, const int *__restrict__ n_ptr, int *__restrict__ order) {
    __shared__ int hist[256];
    __shared__ int red[1024];
    const int tid = threadIdx.x;
    const int T = *n_ptr;
    int mx = 0;
    for (int i = tid; i < T; i += 1024) mx = max(mx, work_c[i]);
    red[tid] = mx;
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if (tid < o) red[tid] = max(red[tid], red[tid + o]);
        __syncthreads();
    }
    const long long wmax = red[0] > 0 ? red[0] : 1;
    for (int i = tid; i < T; i += 1024) atomicAdd(&hist[255 - (int)(((long long)work_c[i] * 255) / wmax)], 1);
    __syncthreads();
    if (tid == 0) {   // exclusive scan, bucket 0 = heaviest
        int run = 0;
        for (int k = 0; k < 256; k++) { int c = hist[k]; hist[k] = run; run += c; }
    }
    __syncthreads();
    for (int i = tid; i < T; i += 1024) order[atomicAdd(&hist[255 - (int)(((long long)work_c[i] * 255) / wmax)], 1)] = desc_c[i];
}
