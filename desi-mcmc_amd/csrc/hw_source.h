// hw_source.h -- ONE source onto a 32-column x 64-row fp64 LDS tile
//
// The building block of the per-source kernels (conditional log-likelihoods, photon split,
// stamps): the same lane mapping, component table, drop test and column recurrence as the
// field kernel k_render_hw (k_render_hw.h), factored so that a kernel can render a single
// source into a scratch tile, consume the tile, and go on to the next source.
//
// Two ways to decide that a component cannot matter on a rectangle:
//   HW_DROP_SKY   against the band's sky level: A e^E < eps e^-T everywhere on the rectangle
//                 (what the field kernel does; the tile is added to a lambda >= eps).
//   HW_DROP_SELF  against the source itself: below e^-T times a lower bound of the source's own
//                 value on the rectangle (the largest over components of the component's minimum
//                 there).  Needs no sky, so it serves log(m) of the conditional likelihood
//                 (sources.py:134-183), where only the relative accuracy of m matters.
// Either way the relative error of what the tile is used for stays below K e^-T (T = 32: 6e-13).
// When a kept component's threshold lies beyond the exponent range the recurrence's seeds are
// safe on (-E > 300), the source is evaluated directly on that rectangle instead.
#pragma once
#include "k_render_hw.h"

#define HW_DROP_NONE 0
#define HW_DROP_SKY 1
#define HW_DROP_SELF 2

// largest value of the convex form a x^2 + 2 b x y + c y^2 on a rectangle: at one of the corners
__device__ inline double quad_max_rect(double a, double b, double c, double x1, double x2, double y1, double y2) {
    double m = a * x1 * x1 + (2.0 * b * x1 + c * y1) * y1;
    m = fmax(m, a * x2 * x2 + (2.0 * b * x2 + c * y1) * y1);
    m = fmax(m, a * x1 * x1 + (2.0 * b * x1 + c * y2) * y2);
    m = fmax(m, a * x2 * x2 + (2.0 * b * x2 + c * y2) * y2);
    return m * 1.00001;
}

__device__ inline float wave_max_f(float v) {
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// Build the compacted component table of `rec` (its scale already what the tile should hold) for
// the rectangle columns [xa, xb] x rows Y0 + [ra, rb).  Returns the number of kept components;
// `direct` comes back true when the rectangle must be evaluated without the recurrence.
// All 64 lanes call this; it contains the two barriers that fence the table.
__device__ __forceinline__ int hw_build(CompTab &T, const LaneConst &lc, const RecU &rec, int lane, int dropmode,
                               double Tdrop, double log_floor /* HW_DROP_SKY: log(eps) */, int Y0, int xa,
                               int xb, int ra, int rb, bool &direct,
                               const Comp *pre = nullptr /* this lane's component, when the caller has it already */,
                               const double *__restrict__ et = nullptr /* the 2^(j/64) table in LDS: exp(-qc) by table */) {
    const int K = (rec.type == 0) ? K_PSF : K_GAL;
    const double xad = (double)xa, xbd = (double)xb;
    const double yad = (double)(Y0 + ra), ybd = (double)(Y0 + rb - 1);
    Comp c;
    float logA = -INFINITY;
    if (lane < K) {
        c = pre ? *pre : make_comp_lc(lc, rec);
        logA = __logf((float)fabs(c.A));
    }
    if (dropmode == HW_DROP_SELF) {
        float mine = -INFINITY;
        if (lane < K)
            mine = logA - 0.5f * (float)quad_max_rect(c.qa, c.qb, c.qc, xad - c.mx, xbd - c.mx, yad - c.my, ybd - c.my);
        log_floor = (double)wave_max_f(mine);
    }
    bool keep = false, far = false;
    int Lk = 0, rlo = ra, rhi = rb;
    if (lane < K) {
        // A e^E >= floor e^-T  <=>  -E <= T + log A - log floor =: Tk
        double Tk = Tdrop + (double)logA - log_floor;
        if (dropmode == HW_DROP_NONE) {
            // nothing is dropped, so the component matters wherever it is evaluated: the seeds must
            // be safe down to its smallest value on the rectangle
            Tk = 0.5 * quad_max_rect(c.qa, c.qb, c.qc, xad - c.mx, xbd - c.mx, yad - c.my, ybd - c.my);
            keep = true;
            far = !(Tk <= 300.0);
        } else if (Tk == Tk && Tk < 1e30) {
            double qmin = quad_min_rect(c.qa, c.qb, c.qc, xad - c.mx, xbd - c.mx, yad - c.my, ybd - c.my);
            keep = (0.5 * qmin <= Tk);
            // rows on which the component can matter on THIS rectangle's columns
            float ylo, yhi;
            quad_rows_on_columns(c.qa, c.qb, c.qc, 2.0 * fmax(Tk, 0.0), xad - c.mx, xbd - c.mx, ylo, yhi);
            const float cy = (float)(c.my - (double)Y0);
            rlo = max(ra, (int)floorf(fmaxf(cy + ylo - 0.02f, -1.0f)));
            rhi = min(rb, (int)ceilf(fminf(cy + yhi + 0.02f, 4096.0f)) + 1);
            keep = keep && (rhi > rlo);
            far = keep && (Tk > 300.0);
        } else {
            keep = (logA > -INFINITY);      // no usable floor (a zero source, or no sky): keep, evaluate directly
            far = keep;
        }
        Lk = seg_len(c.qc, fmin(fmax(Tk, 1.0), 300.0));
    }
    const unsigned long long km = __ballot(keep);
    const int Kk = __popcll(km);
    direct = (__ballot(far) != 0ull);
    __syncthreads();   // the previous table's reads are done
    if (lane < 4) { T.gL[lane] = 4096; T.gr0[lane] = HW_TH; T.gr1[lane] = 0; }
    const int slot = slot_by_rows(keep, rlo, rhi);
    if (keep) {
        const int p = slot;
        T.A[p] = c.A; T.mx[p] = c.mx; T.my[p] = c.my;
        T.qa[p] = c.qa * EXP_SCALE; T.qb[p] = c.qb * EXP_SCALE; T.qc[p] = c.qc * EXP_SCALE;
        T.eq[p] = et ? exp_tab64(-c.qc * EXP_SCALE, et) : exp(-c.qc);
        const int gi = p / (2 * REC_G);     // per pair of groups: shortest segment, union of the row ranges
        atomicMin(&T.gL[gi], Lk);
        atomicMin(&T.gr0[gi], rlo);
        atomicMax(&T.gr1[gi], rhi);
    }
    if (lane < HW_PAD) {   // zero components behind the table (amplitude 0, ratio 1)
        int p = Kk + lane;
        T.A[p] = 0.0; T.mx[p] = 0.0; T.my[p] = 0.0;
        T.qa[p] = 0.0; T.qb[p] = 0.0; T.qc[p] = 0.0;
        T.eq[p] = 1.0;
    }
    __syncthreads();
    return Kk;
}

// Add the table's Kk components into the tile: lanes 0..31 / 32..63 are the same 32 columns
// (x = this lane's column, `on` = the column belongs to the rectangle) working on two groups of
// components; rows [ra, rb) bound the direct path, the recurrence walks each group's own rows.
// (forced inline: as a called function it cost a scratch frame per chunk -- 12 GB of scratch writes
// per 160 000-proposal launch of k_patch_ll_hw)
// SUM: no tile -- every lane adds what it evaluates to `msum` (the sum over the 64 lanes is the source's total on the
// rectangle's `on` columns and rows [ra, rb)); acc is not touched.
template <bool SUM = false>
__device__ __forceinline__ void hw_walk(const CompTab &T, const double *__restrict__ et, int Kk, double x, int Y0,
                               int ra, int rb, bool on, bool direct, double *__restrict__ acc, int lane,
                               double *__restrict__ msum = nullptr) {
    const int half = lane >> 5, col = lane & 31;
    if (direct) {
        const int kh = (Kk + 1) / 2;
        const int k0 = half ? kh : 0, k1 = half ? Kk : kh;
        for (int row = ra; row < rb; row++) {
            double v = eval_direct(T, k0, k1, x, (double)(Y0 + row), 1.0 / EXP_SCALE);
            if (on) { if (SUM) *msum += v; else lds_add(&acc[row * HW_TW + col], v); }
        }
        return;
    }
    for (int p0 = 0; p0 < Kk; p0 += 2 * REC_G) {
        const int R = min(2 * REC_G, Kk - p0);
        const int gA = (R + 1) / 2;
        const int gi = p0 / (2 * REC_G);
        const int L = __builtin_amdgcn_readfirstlane(T.gL[gi]);
        const int ga = __builtin_amdgcn_readfirstlane(T.gr0[gi]);
        const int gb = __builtin_amdgcn_readfirstlane(T.gr1[gi]);
        const int k0 = half ? p0 + gA : p0;
        if (L < 4) {   // pathologically sharp component: evaluate this pair of groups directly
            const int k1 = half ? p0 + R : p0 + gA;
            for (int row = ga; row < gb; row++) {
                double v = eval_direct(T, k0, k1, x, (double)(Y0 + row), 1.0 / EXP_SCALE);
                if (on) { if (SUM) *msum += v; else lds_add(&acc[row * HW_TW + col], v); }
            }
            continue;
        }
        double *colp = SUM ? nullptr : acc + col;
        switch (gA) {
        case 6: rec_group_hw<6, HW_TW, SUM>(T, et, k0, x, Y0, ga, gb, L, on, colp, msum); break;
        case 5: rec_group_hw<5, HW_TW, SUM>(T, et, k0, x, Y0, ga, gb, L, on, colp, msum); break;
        case 4: rec_group_hw<4, HW_TW, SUM>(T, et, k0, x, Y0, ga, gb, L, on, colp, msum); break;
        case 3: rec_group_hw<3, HW_TW, SUM>(T, et, k0, x, Y0, ga, gb, L, on, colp, msum); break;
        case 2: rec_group_hw<2, HW_TW, SUM>(T, et, k0, x, Y0, ga, gb, L, on, colp, msum); break;
        default: rec_group_hw<1, HW_TW, SUM>(T, et, k0, x, Y0, ga, gb, L, on, colp, msum); break;
        }
    }
}
