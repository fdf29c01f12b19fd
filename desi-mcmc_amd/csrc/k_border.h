// k_border.h -- the photon split's totals image from a model image that is already on the device
//
// The split (celeste_sample_sources.pyx:61-156) needs every pixel's total rate under the reference's membership rule: a
// source takes part at a pixel only STRICTLY inside its box on the low side (x > x0 and y > y0, :50-51).  Round 3 rendered
// that image from scratch before every split (k_render_hw with CEL_RENDER_STRICT: 1.3 ms of a 25 ms sweep at configs[4]).
// But a Gibbs chain that traces its log-likelihood has just rendered the SAME catalogue with the same sky levels on full
// boxes, and the two images differ only on the first row and the first column of every source's box (86 pixels of a star's
// 43 x 43): with lambda, the records and the tile lists of that render still on the device,
//     totals = lambda - sum over the tile's sources of their stamp on those pixels.
// One wave per 32 x 64 render tile: the tile's sources are taken in list order, a source's border pixels on the tile are
// evaluated directly (every component, table exponential) and collected in an LDS tile with ds_add (one wave: a fixed
// order), then the tile of lambda is streamed through.  What is subtracted carries the direct evaluator's rounding and the
// components the render dropped below eps e^-T on the tile, so the totals agree with a strict render to ~1e-10 of a pixel's
// rate on border pixels (exactly elsewhere): a perturbation of the split's probabilities far below anything a sampler can
// see -- but enough to change a draw now and then, so the two ways are not photon-for-photon interchangeable
// (CEL_OPT_SPLIT_REUSE = 0 renders from scratch).
#pragma once
#include "k_render_hw.h"

__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4)))
k_strict_totals(RenderArgs a /* lambda: the full-box model image (in); lists / recs / tile_* of that render */, double *__restrict__ rate,
                unsigned long long *__restrict__ massfx /* per (source, band): += the unit stamp on these border pixels, 2^-60 units (k_split.h), or nullptr */) {
    __shared__ double acc[HW_TH * HW_TW];
    __shared__ double tA[K_GAL], tmx[K_GAL], tmy[K_GAL], tqa[K_GAL], tqb[K_GAL], tqc[K_GAL];
    __shared__ double et[64];
    const int lane = threadIdx.x;
    const int tile = blockIdx.x;
    const int per_band = a.ntx * a.nty;
    const int b = tile / per_band;
    const int t = tile - b * per_band;
    const int ty = t / a.ntx, tx = t - ty * a.ntx;
    const int X0 = tx * HW_TW, Y0 = ty * HW_TH;
    const BandDev *bd = a.bands + b;
    const int cnt = a.tile_cnt[tile];
    const int64_t off = a.tile_off[tile];
    const int nent = (int)min((int64_t)cnt, a.capacity > off ? a.capacity - off : (int64_t)0);
    const int half = lane >> 5, col = lane & 31;
    bool any = false;
    if (nent > 0) {
        et[lane] = exp2((double)lane * (1.0 / 64.0));
#pragma unroll
        for (int r = 0; r < HW_TH * HW_TW / 64; r++) acc[r * 64 + lane] = 0.0;
        const LaneConst lc = lane_consts(lane, bd);
        const SrcRec *recs = a.recs + (int64_t)b * a.S;
        const double log_eps = (bd->eps > 0.0) ? (double)__logf((float)bd->eps) : -700.0;
        __syncthreads();
        for (int e = 0; e < nent; e++) {
            const int s = a.lists[off + e];
            const RecU rec = rec_unpack(rec_fetch(recs, s, lane));
            if (rec.type < 0) continue;
            // does the first row / the first column of this box cross the tile?
            const bool row_here = rec.y0 >= Y0 && rec.y0 < Y0 + HW_TH && rec.x1 > X0 && rec.x0 < X0 + HW_TW;
            const bool col_here = rec.x0 >= X0 && rec.x0 < X0 + HW_TW && rec.y1 > Y0 && rec.y0 + 1 < Y0 + HW_TH;
            if (!row_here && !col_here) continue;
            const int K = (rec.type == 0) ? K_PSF : K_GAL;
            __syncthreads();            // the previous source's table has been read
            // The border lies where the source has faded to 1e-5 / 1e-3 of its mass: most of a galaxy's 42 components are
            // nothing there.  A component is kept when it can exceed eps e^-40 somewhere on the piece of the box's first row or
            // column that lies on this tile: what is left out is below 42 e^-40 = 2e-16 of the sky level per pixel.  Kept
            // components are compacted (ballot + prefix count).
            bool keep = false;
            Comp c;
            if (lane < K) {
                c = make_comp_lc(lc, rec);
                // ... on THIS tile's piece of the row / of the column: the form's minimum over the segment (a quadratic in one
                // variable: the free minimiser clamped to the segment)
                const double dy = (double)rec.y0 - c.my, dx = (double)rec.x0 - c.mx;
                double qrow = INFINITY, qcol = INFINITY;
                if (row_here) {
                    const double lo = (double)max(rec.x0, X0) - c.mx, hi = (double)(min(rec.x1, X0 + HW_TW) - 1) - c.mx;
                    const double t = fmin(fmax(-c.qb * dy / c.qa, lo), hi);
                    qrow = c.qa * t * t + (2.0 * c.qb * t + c.qc * dy) * dy;
                }
                if (col_here) {
                    const double lo = (double)max(rec.y0 + 1, Y0) - c.my, hi = (double)(min(rec.y1, Y0 + HW_TH) - 1) - c.my;
                    const double t = fmin(fmax(-c.qb * dx / c.qc, lo), hi);
                    qcol = c.qa * dx * dx + (2.0 * c.qb * dx + c.qc * t) * t;
                }
                const double lim = 2.0 * (40.0 + (double)__logf((float)fmax(fabs(c.A), 1e-300)) - log_eps);
                keep = (fmin(qrow, qcol) <= lim * (1.0 + 1e-6) + 1e-6) || !(lim == lim);
            }
            const unsigned long long km = __ballot(keep);
            const int Kk = __popcll(km);
            if (keep) {
                const int p = __popcll(km & ((1ull << lane) - 1ull));
                tA[p] = c.A; tmx[p] = c.mx; tmy[p] = c.my;
                tqa[p] = c.qa * EXP_SCALE; tqb[p] = c.qb * EXP_SCALE; tqc[p] = c.qc * EXP_SCALE;
            }
            __syncthreads();
            if (Kk == 0) continue;
            any = true;
            double bsum = 0.0;
            if (row_here) {             // row y0: the tile's 32 columns, the components dealt to the two half-waves
                const double x = (double)(X0 + col), y = (double)rec.y0;
                const bool on = (X0 + col >= rec.x0) && (X0 + col < rec.x1);
                double v = 0.0;
                for (int k = half; k < Kk; k += 2) {
                    const double dx = x - tmx[k], dy = y - tmy[k];
                    const double q = tqa[k] * dx * dx + (2.0 * tqb[k] * dx + tqc[k] * dy) * dy;
                    v = fma(tA[k], exp_tab64(-0.5 * q, et), v);
                }
                if (on) { lds_add(&acc[(rec.y0 - Y0) * HW_TW + col], v); bsum += v; }
            }
            if (col_here) {             // column x0 below the first row: the tile's 64 rows, one per lane
                const int yi = Y0 + lane;
                const double x = (double)rec.x0, y = (double)yi;
                const bool on = (yi > rec.y0) && (yi < rec.y1);
                double v = 0.0;
                for (int k = 0; k < Kk; k++) {
                    const double dx = x - tmx[k], dy = y - tmy[k];
                    const double q = tqa[k] * dx * dx + (2.0 * tqb[k] * dx + tqc[k] * dy) * dy;
                    v = fma(tA[k], exp_tab64(-0.5 * q, et), v);
                }
                if (on) { lds_add(&acc[lane * HW_TW + (rec.x0 - X0)], v); bsum += v; }
            }
            if (massfx) {
                bsum = wave_sum_lane63(bsum);
                if (lane == 63 && bsum > 0.0 && rec.scale > 0.0)
                    atomicAdd(massfx + ((int64_t)s * a.B + b), (unsigned long long)__double2ull_rn(bsum / rec.scale * MASS_FX));
            }
        }
        __syncthreads();
    }
    // stream the tile: two 256-B row segments per wave-instruction
    const int xi = X0 + col;
    const int64_t base = (int64_t)b * a.H * a.W + (int64_t)(Y0 + half) * a.W + xi;
    if (X0 + HW_TW <= a.W && Y0 + HW_TH <= a.H) {
        double lam[HW_TH / 2];
#pragma unroll
        for (int r = 0; r < HW_TH / 2; r++) lam[r] = a.lambda[base + (int64_t)(2 * r) * a.W];
#pragma unroll
        for (int r = 0; r < HW_TH / 2; r++) rate[base + (int64_t)(2 * r) * a.W] = any ? lam[r] - acc[(2 * r + half) * HW_TW + col] : lam[r];
    } else if (xi < a.W) {
        for (int r = 0; r < HW_TH / 2; r++)
            if (Y0 + 2 * r + half < a.H) {
                const double lam = a.lambda[base + (int64_t)(2 * r) * a.W];
                rate[base + (int64_t)(2 * r) * a.W] = any ? lam - acc[(2 * r + half) * HW_TW + col] : lam;
            }
    }
}
