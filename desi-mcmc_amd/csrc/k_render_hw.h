// k_render_hw.h -- the render kernel on 32-column x 64-row tiles ("half-wave" layout)
//
// Same algorithm as k_render (k_render.h) with a different mapping of the 64 lanes:
//   lanes  0..31  : the tile's 32 pixel columns, working on one group of components
//   lanes 32..63  : the SAME 32 columns, working on the next group of components
// Both halves add into one 32 x 64 fp64 LDS accumulator tile (16 KB, as before) with ds_add_f64.
// Why: the recurrence's cost per (component, column) is a seed (two exp, ~54 instructions) plus
// ~3.3 instructions per row.  With 64 x 32 tiles a column is walked for ~22 rows per source on
// average, so seeds were 37 % of the instruction stream.  Here a column is walked for up to 64
// rows (~36 on average): half as many seeds per evaluated pixel, and a 32-wide tile wastes fewer
// lanes on the columns outside a source's box (84 % of lanes useful against 70 %).
// Component parameters are read from LDS per half (two broadcast addresses per read).
#pragma once
#include "k_render.h"

// timing-only ablation bits of CEL_OPT_DEBUG ride in bits 8.. of the render flags; they exist only in a
// -DCEL_ABLATE build (tools/ablate_render.py builds its own library), never in the shipped one
#ifdef CEL_ABLATE
#define CEL_ABLATE_BITS(flags) ((flags) >> 8)
#else
#define CEL_ABLATE_BITS(flags) 0
#endif

#define HW_TW 32
#define HW_TH 64
// The blocks of these kernels are ONE wave, and the LDS operations of one wave execute in the order they were issued: a table
// written by some lanes is what the other lanes read afterwards without anyone waiting.  Between a source's table writes and
// reads the compiler only has to keep the order (no instruction; __syncthreads() there made the wave wait for every LDS operation
// in flight -- the previous pair's last ds_add_f64 among them -- twice per (source, tile) entry).
#ifdef HW_SYNC_BARRIER
#define HW_WAVE_SYNC() __syncthreads()
#else
#define HW_WAVE_SYNC()                                              \
    do {                                                            \
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      \
        __builtin_amdgcn_wave_barrier();                            \
    } while (0)
#endif
#define HW_PAD 12   // zero components behind the compacted table: a half's group may read past the end
#ifndef HW_PART_ENTRIES
#define HW_PART_ENTRIES 12   // k_render_hw<, PARTS>: list entries per working part of a tile
#endif

// largest value of the convex form a x^2 + 2 b x y + c y^2 on a rectangle: at one of the corners
__host__ __device__ inline double quad_max_rect_hw(double a, double b, double c, double x1, double x2, double y1, double y2) {
    double m = a * x1 * x1 + (2.0 * b * x1 + c * y1) * y1;
    m = fmax(m, a * x2 * x2 + (2.0 * b * x2 + c * y1) * y1);
    m = fmax(m, a * x1 * x1 + (2.0 * b * x1 + c * y2) * y2);
    m = fmax(m, a * x2 * x2 + (2.0 * b * x2 + c * y2) * y2);
    return m * 1.00001;
}

// SUM: nothing is written to a tile; the rows' sums are added to *msum (this lane's share of the source's total on the
// rectangle: the stamp-mass kernel needs no accumulator tile at all)
template <int G, int STRIDE = HW_TW, bool SUM = false>   // STRIDE = doubles between consecutive rows of the accumulator tile
__device__ inline void rec_group_hw(const CompTab &T, const double *__restrict__ et, int k0, double x,
                                    int Y0, int ra, int rb, int L, bool on, double *__restrict__ acc_col,
                                    double *__restrict__ msum = nullptr) {
    double g[G], r[G], q[G];
    const double aon = on ? 1.0 : 0.0;
    for (int sa = ra; sa < rb; sa += L) {
        const int sb = min(sa + L, rb);
        const double y0 = (double)(Y0 + sa);
#pragma unroll
        for (int i = 0; i < G; i++) {
            const int k = k0 + i;                               // differs between the two halves
            double dx = x - T.mx[k], dy = y0 - T.my[k];
            double qb = T.qb[k], qc = T.qc[k];
            double hx = qb * dx + qc * dy;
            double e = -0.5 * (T.qa[k] * dx * dx + (qb * dx + hx) * dy);
            double er = fmin(fmax(-(hx + 0.5 * qc), -REC_EMAX * EXP_SCALE), REC_EMAX * EXP_SCALE);
            g[i] = (T.A[k] * aon) * exp_tab64(e, et);
            r[i] = exp_tab64(er, et);
            q[i] = T.eq[k];
        }
        int row = sa;
        for (; row + 1 < sb; row += 2) {
            // no FMA contraction here: fusing g*r into the row sum would not save the product
            // (g*r is needed by itself for the next row) and costs one extra multiply per component pair
#pragma clang fp contract(off)
            double s0 = g[0], s1, g1[G], r1[G];
#pragma unroll
            for (int i = 1; i < G; i++) s0 += g[i];
#pragma unroll
            for (int i = 0; i < G; i++) {
                g1[i] = g[i] * r[i];
                r1[i] = r[i] * q[i];
            }
            s1 = g1[0];
#pragma unroll
            for (int i = 1; i < G; i++) s1 += g1[i];
#pragma unroll
            for (int i = 0; i < G; i++) {
                g[i] = g1[i] * r1[i];
                r[i] = r1[i] * q[i];
            }
            if (SUM) { *msum += s0; *msum += s1; }
            else {
                lds_add(&acc_col[row * STRIDE], s0);
                lds_add(&acc_col[(row + 1) * STRIDE], s1);
            }
        }
        if (row < sb) {
            double s0 = g[0];
#pragma unroll
            for (int i = 1; i < G; i++) s0 += g[i];
            if (SUM) *msum += s0;
            else lds_add(&acc_col[row * STRIDE], s0);
        }
    }
}

// ---- the nested walk (round 6) ------------------------------------------------------------------------------------------------
// A pair of groups walked the UNION of its 12 components' row ranges with all 12 -- 24 % more component-rows than the components
// need one by one (tools/row_waste.py).  The slots are ordered by decreasing row count, so the pair's later slots need a
// shorter stretch of rows than its first ones: the pair's first 2 NB slots (NB per half) are walked over the whole union
// [ga, gb) as before, the remaining ones (NS per half) only over THEIR union [sa, sb), seeded at its first row -- three
// phases of one recurrence, the middle one with NB + NS components per lane, the outer ones with NB.  A component is still
// added on every row of its own range: what is left out lies below the drop level by the range's construction.  One segment
// only (the caller checks gb - ga <= L).  Slots of a half: big p0 + half * NB + i, small p0 + 2 NB + half * NS + i.
template <int N, int NT>
__device__ __forceinline__ void rec_walk_n(double (&g)[NT], double (&r)[NT], const double (&q)[NT], int row, int rend, bool advance,
                                           double *__restrict__ acc_col) {
    for (; row + 1 < rend; row += 2) {
#pragma clang fp contract(off)
        double s0 = g[0], s1, g1[N], r1[N];
#pragma unroll
        for (int i = 1; i < N; i++) s0 += g[i];
#pragma unroll
        for (int i = 0; i < N; i++) {
            g1[i] = g[i] * r[i];
            r1[i] = r[i] * q[i];
        }
        s1 = g1[0];
#pragma unroll
        for (int i = 1; i < N; i++) s1 += g1[i];
#pragma unroll
        for (int i = 0; i < N; i++) {
            g[i] = g1[i] * r1[i];
            r[i] = r1[i] * q[i];
        }
        lds_add(&acc_col[row * HW_TW], s0);
        lds_add(&acc_col[(row + 1) * HW_TW], s1);
    }
    if (row < rend) {
#pragma clang fp contract(off)
        double s0 = g[0];
#pragma unroll
        for (int i = 1; i < N; i++) s0 += g[i];
        lds_add(&acc_col[row * HW_TW], s0);
        if (advance) {                       // the next phase goes on from the row behind this one
#pragma unroll
            for (int i = 0; i < N; i++) {
                g[i] = g[i] * r[i];
                r[i] = r[i] * q[i];
            }
        }
    }
}

template <int NB, int NS>
__device__ inline void rec_group_nested(const CompTab &T, const double *__restrict__ et, int kb, int ks, double x, int Y0,
                                        int ga, int gb, int sa, int sb, bool on, double *__restrict__ acc_col) {
    constexpr int NT = NB + NS;
    double g[NT], r[NT], q[NT];
    const double aon = on ? 1.0 : 0.0;
    auto seed = [&](int i, int k, double y0) {
        double dx = x - T.mx[k], dy = y0 - T.my[k];
        double qb = T.qb[k], qc = T.qc[k];
        double hx = qb * dx + qc * dy;
        double e = -0.5 * (T.qa[k] * dx * dx + (qb * dx + hx) * dy);
        double er = fmin(fmax(-(hx + 0.5 * qc), -REC_EMAX * EXP_SCALE), REC_EMAX * EXP_SCALE);
        g[i] = (T.A[k] * aon) * exp_tab64(e, et);
        r[i] = exp_tab64(er, et);
        q[i] = T.eq[k];
    };
#pragma unroll
    for (int i = 0; i < NB; i++) seed(i, kb + i, (double)(Y0 + ga));
    rec_walk_n<NB, NT>(g, r, q, ga, sa, true, acc_col);
#pragma unroll
    for (int i = 0; i < NS; i++) seed(NB + i, ks + i, (double)(Y0 + sa));
    rec_walk_n<NT, NT>(g, r, q, sa, sb, true, acc_col);
    rec_walk_n<NB, NT>(g, r, q, sb, gb, false, acc_col);
}

// components per half in the pair's first (larger) set when a half holds gA: 3 of 5 or 6, 2 of 4; fewer than 4: no second set
__host__ __device__ inline int nested_big(int gA) { return gA >= 5 ? 3 : (gA == 4 ? 2 : gA); }

// Table slot of a kept component: by decreasing number of tile rows it can matter on, in eight
// classes of eight rows (inside a class: component order).  The 12 components of a pair of groups walk
// the UNION of their row ranges, so similar ranges belong together: 2.55e8 -> 2.2e8 walked
// component-rows at config 3, k_render_hw -10 %.  (An exact rank by counting over the kept lanes
// groups no better and costs 3 VALU + a scalar loop trip per component: 1.36 against 1.32 ms; 16
// classes of four rows 1.34.)  Depends on the data only; all 64 lanes call it.
template <int SHIFT = 3>   // 8 << SHIFT = rows of the tile
__device__ __forceinline__ int slot_by_rows(bool keep, int rlo, int rhi) {
    const int cls = keep ? 7 - min((rhi - rlo - 1) >> SHIFT, 7) : 8;
    int slot = 0, base = 0;
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const unsigned long long bq = __ballot(cls == q);
        const int within = __builtin_amdgcn_mbcnt_hi((unsigned)(bq >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bq, 0));
        if (cls == q) slot = base + within;
        base += __popcll(bq);
    }
    return slot;
}

// ---- the stars of a tile, batched ---------------------------------------------------------------
// A star's three components are the band's PSF components shifted to the star: inverse covariance,
// normaliser, row-to-row ratio are BAND constants (computed once per tile by lanes 0..2); a star
// brings only its position, counts and box.  So the stars of a tile (they come first in its list,
// k_bin2.h) need no per-source component table, no division, no drop test and no barrier pair per
// source: up to 64 of them are staged in LDS per batch (40 bytes each, one lane per star) and walked as
// (star, column) tasks, 64 per step (star_pass below), each all three components in one group over the
// star's rows on this tile.  One segment per star: legal when no component's exponent can exceed 600
// anywhere on a star's box (checked per tile from the band's bounding radius; otherwise the stars take the
// general path below).
// Nothing is dropped here, so a star's pixels carry all three components (the general path skips
// components below eps * e^-T on the tile): both agree with the reference to the test tolerances.
struct StarTab {           // lives in the component table's LDS while the stars are processed
    double px[64], py[64], scale[64];
    int4 box[64];          // x0, x1, y0, y1
    int cum[64];           // first (star, column) task of the sorted batch's star j
    double qa[K_PSF], qb[K_PSF], qc[K_PSF], eq[K_PSF], A0[K_PSF], mux[K_PSF], muy[K_PSF];
};
static_assert(sizeof(StarTab) <= sizeof(CompTab), "the star table must fit the component table's storage");
#define STAR_EMAX 600.0


// band constants of the three PSF components into the star table (lanes 0..2) and the check that a
// star may take the one-segment path; contains one barrier, all 64 lanes call it
__device__ __forceinline__ bool star_setup(StarTab &ST, const BandDev *__restrict__ bd, const double *__restrict__ et, int lane) {
    bool bad = false;
    if (lane < K_PSF) {
        const double cxx = bd->cxx[lane], cxy = bd->cxy[lane], cyy = bd->cyy[lane];
        const double inv = 1.0 / (cxx * cyy - cxy * cxy);
        const double qa = cyy * inv, qb = -cxy * inv, qc = cxx * inv;
        ST.qa[lane] = qa * EXP_SCALE; ST.qb[lane] = qb * EXP_SCALE; ST.qc[lane] = qc * EXP_SCALE;
        ST.eq[lane] = exp_tab64(-qc * EXP_SCALE, et);
        ST.A0[lane] = bd->w[lane] * (0.5 / PI_D) * sqrt(inv);
        ST.mux[lane] = bd->mux[lane]; ST.muy[lane] = bd->muy[lane];
        // largest exponent of this component anywhere on a star's box (half-width R + 2 about the star)
        const double rb_ = bd->R + 2.0;
        const double emax = 0.5 * quad_max_rect_hw(qa, qb, qc, -rb_ - bd->mux[lane], rb_ - bd->mux[lane],
                                                   -rb_ - bd->muy[lane], rb_ - bd->muy[lane]);
        bad = !(emax <= STAR_EMAX);
    }
    const bool ok = (__ballot(bad) == 0ull);
    __syncthreads();
    return ok;
}

// the first nstar entries of the tile's list (its stars) into the accumulator tile.  DIAG: the tile-timing
// counters and (CEL_ABLATE builds) the timing-only ablation switches; the production instantiation has neither.
//
// Column tasks (round 3).  A star's 43-pixel box covers 32, or only a few, of a 32-column tile's columns: with one
// star per half-wave 59 % of the lanes carried a column of their star.  Here the unit of work is one COLUMN of one
// star: the batch's stars are sorted by the number of rows they have on the tile (descending), their column counts
// are prefix-summed, and each step hands 64 consecutive (star, column) tasks to the 64 lanes -- a lane finds its
// star by bisection of the prefix sums (6 LDS reads), seeds the three components at ITS star's first row and walks
// ITS star's rows, row i of the step being row ra_lane + i of the tile.  Neighbouring tasks belong to the same or the
// next star of the sorted order, so the lanes of a step have nearly equal row counts, and every lane carries a
// column (all but the batch's last step).  The order in which a pixel receives its terms depends on the data only.
// star_stage puts one batch (<= 64 stars of the tile's list) into the table, sorted; star_walk adds the batch's
// columns [Xa, Xa + CW) into an accumulator of CW doubles per row (the whole tile: Xa = X0, CW = HW_TW); a caller
// that walks the same batch again (another part of the columns) puts a barrier between the walks.
__device__ __forceinline__ void star_stage(const RenderArgs &a, StarTab &ST, const SrcRec *__restrict__ recs, int64_t off, int base, int nb,
                                           int lane, int X0, int Y0, int strict, int lfirst = 0, int lstride = 1 /* the batch's star j is
                                           entry lfirst + lstride * (base + j) of the tile's list: one PART of a tile's stars (k_render_hw<, PARTS>) */) {
    __syncthreads();                   // the previous batch has been read
    // One lane per star loads it; the batch is then SORTED by the number of rows the star has on this tile
    // (descending; ties by list position).  The rank of a star is a count over the batch (<= 64 LDS
    // broadcasts) and depends only on the data.  A star without a row or a column here sorts last.
    double2 pp = make_double2(0.0, 0.0);
    double sc = 0.0;
    int4 bx4 = make_int4(0, 0, 0, 0);
    int nrows = -1;
    if (lane < nb) {
        const SrcRec *rp = recs + a.lists[off + lfirst + (int64_t)lstride * (base + lane)];
        pp = *reinterpret_cast<const double2 *>(&rp->px);
        sc = rp->scale;
        bx4 = *reinterpret_cast<const int4 *>(&rp->x0);
        nrows = max(min(bx4.w, Y0 + HW_TH) - max(bx4.z + strict, Y0), 0);
        const int ncols = max(min(bx4.y, X0 + HW_TW) - max(bx4.x + strict, X0), 0);
        if (ncols == 0) nrows = 0;
    }
    int *srows = reinterpret_cast<int *>(ST.scale);      // scratch until the sorted table is written
    srows[lane] = nrows;
    __syncthreads();
    int rank = 0;
    for (int j = 0; j < nb; j++) {
        const int rj = srows[j];
        rank += (rj > nrows || (rj == nrows && j < lane)) ? 1 : 0;
    }
    __syncthreads();
    if (lane < nb) {
        ST.px[rank] = pp.x; ST.py[rank] = pp.y; ST.scale[rank] = sc;
        ST.box[rank] = bx4;
    }
    __syncthreads();
}

// `own` (optional, 64 * CW bytes of LDS): the star of every task, written once per batch by the stars' lanes (CW byte stores
// each) -- a step then finds its lane's star with ONE LDS read instead of a six-step bisection of the prefix sums (six
// dependent reads, a quarter of a short step's cycles); meant for narrow parts (k_small_stars: CW = 8)
template <bool DIAG, int CW>
__device__ __forceinline__ void star_walk(const RenderArgs &a, StarTab &ST, const double *__restrict__ et, double *__restrict__ acc,
                                          int nb, int lane, int Xa, int Y0, int strict, unsigned &dbg_halfrows,
                                          unsigned char *__restrict__ own = nullptr) {
    double cqa[K_PSF], cqb[K_PSF], cqc[K_PSF], ceq[K_PSF], cA0[K_PSF], cmx[K_PSF], cmy[K_PSF];
#pragma unroll
    for (int k = 0; k < K_PSF; k++) {
        cqa[k] = ST.qa[k]; cqb[k] = ST.qb[k]; cqc[k] = ST.qc[k]; ceq[k] = ST.eq[k];
        cA0[k] = ST.A0[k]; cmx[k] = ST.mux[k]; cmy[k] = ST.muy[k];
    }
    const int dbg = DIAG ? CEL_ABLATE_BITS(a.flags) : 0;
    // exclusive prefix sum of the column counts in sorted order (lane j = sorted star j): cum[j] = first task of star j
    int w = 0;
    if (lane < nb) {
        const int4 q = ST.box[lane];
        const int nr = max(min(q.w, Y0 + HW_TH) - max(q.z + strict, Y0), 0);
        w = (nr > 0) ? max(min(q.y, Xa + CW) - max(q.x + strict, Xa), 0) : 0;
    }
    int incl = w;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(incl, o);
        if (lane >= o) incl += up;
    }
    const int total = __builtin_amdgcn_readlane(incl, 63);
    ST.cum[lane] = (lane < nb) ? incl - w : 0x3fffffff;
    if (own) {
#pragma unroll
        for (int cidx = 0; cidx < CW; cidx++)
            if (cidx < w) own[incl - w + cidx] = (unsigned char)lane;
    }
    __syncthreads();
    for (int t0 = 0; t0 < total && !(dbg & 2); t0 += 64) {
        const int t = t0 + lane;
        const bool valid = t < total;
        // the star this task belongs to: the last j with cum[j] <= t (stars without a column share their
        // successor's cum and are passed over; entries behind the batch hold a sentinel)
        int j = 0;
        if (own) {
            j = own[min(t, total - 1)];
        } else {
#pragma unroll
            for (int step = 32; step > 0; step >>= 1)
                if (ST.cum[min(j + step, 63)] <= t && j + step < 64) j += step;
        }
        const double px = ST.px[j], py = ST.py[j];
        const int4 bx = ST.box[j];
        const int bx0 = max(bx.x + strict, Xa), by0 = bx.z + strict;
        const int xi = bx0 + (t - ST.cum[j]);
        const int ra = max(by0, Y0) - Y0;
        const int n = (valid && !(dbg & 1)) ? max(min(bx.w, Y0 + HW_TH) - Y0 - ra, 0) : 0;
        const double amp = valid ? ST.scale[j] : 0.0;
        // sorted by rows: the step's first task has the most -- of the stars WITH a column here; one without
        // passes its place to its successor, whose rows are no more
        const int nmax = __builtin_amdgcn_readlane(n, 0);
        if (DIAG && a.timing) dbg_halfrows += 2u * (unsigned)nmax * K_PSF;   // in half-tile (32-lane) widths, as the general path counts
        const double x = (double)xi;
        double g[K_PSF], r[K_PSF];
        const double y0 = (double)(Y0 + ra);
#pragma unroll
        for (int k = 0; k < K_PSF; k++) {
            const double dx = x - (px + cmx[k]), dy = y0 - (py + cmy[k]);
            const double hx = cqb[k] * dx + cqc[k] * dy;
            const double e = -0.5 * (cqa[k] * dx * dx + (cqb[k] * dx + hx) * dy);
            const double er = fmin(fmax(-(hx + 0.5 * cqc[k]), -REC_EMAX * EXP_SCALE), REC_EMAX * EXP_SCALE);
            g[k] = (cA0[k] * amp) * exp_tab64(e, et);
            r[k] = exp_tab64(er, et);
        }
        double *rowp = acc + ra * CW + (valid ? xi - Xa : 0);
        int i = 0;
        for (; i + 3 < nmax; i += 4, rowp += 4 * CW) {       // four rows per trip: one address update, one bound test
#pragma clang fp contract(off)
            double g1[K_PSF], r1[K_PSF];
            const double s0 = (g[0] + g[1]) + g[2];
#pragma unroll
            for (int k = 0; k < K_PSF; k++) { g1[k] = g[k] * r[k]; r1[k] = r[k] * ceq[k]; }
            const double s1 = (g1[0] + g1[1]) + g1[2];
#pragma unroll
            for (int k = 0; k < K_PSF; k++) { g[k] = g1[k] * r1[k]; r[k] = r1[k] * ceq[k]; }
            const double s2 = (g[0] + g[1]) + g[2];
#pragma unroll
            for (int k = 0; k < K_PSF; k++) { g1[k] = g[k] * r[k]; r1[k] = r[k] * ceq[k]; }
            const double s3 = (g1[0] + g1[1]) + g1[2];
#pragma unroll
            for (int k = 0; k < K_PSF; k++) { g[k] = g1[k] * r1[k]; r[k] = r1[k] * ceq[k]; }
            if (i + 3 < n) {            // the whole trip lies inside this lane's rows (most lanes, most trips)
                lds_add(&rowp[0], s0);
                lds_add(&rowp[CW], s1);
                lds_add(&rowp[2 * CW], s2);
                lds_add(&rowp[3 * CW], s3);
            } else {
                if (i < n) lds_add(&rowp[0], s0);
                if (i + 1 < n) lds_add(&rowp[CW], s1);
                if (i + 2 < n) lds_add(&rowp[2 * CW], s2);
            }
        }
        for (; i < nmax; i++, rowp += CW) {
#pragma clang fp contract(off)
            if (i < n) lds_add(&rowp[0], (g[0] + g[1]) + g[2]);
#pragma unroll
            for (int k = 0; k < K_PSF; k++) { g[k] = g[k] * r[k]; r[k] = r[k] * ceq[k]; }
        }
    }
}

template <bool DIAG>
__device__ __forceinline__ void star_pass(const RenderArgs &a, StarTab &ST, const double *__restrict__ et, double *__restrict__ acc,
                                          const SrcRec *__restrict__ recs, int64_t off, int nstar, int lane, int X0, int Y0,
                                          int strict, unsigned &dbg_halfrows, unsigned &dbg_pairs, int lfirst = 0, int lstride = 1) {
    for (int base = 0; base < nstar; base += 64) {
        const int nb = min(64, nstar - base);
        star_stage(a, ST, recs, off, base, nb, lane, X0, Y0, strict, lfirst, lstride);
        if (DIAG && a.timing) { dbg_pairs += (unsigned)nb; }
        star_walk<DIAG, HW_TW>(a, ST, et, acc, nb, lane, X0, Y0, strict, dbg_halfrows);
    }
}


// epilogue of a tile: lambda = eps + acc written once, the Poisson term reduced to one partial.
// One wave-instruction covers two 256-B row segments (rows 2r and 2r+1).  `lt` is the component /
// star table's LDS, dead by now: it takes the log table (128 doubles).  All 32 nelec loads of a lane
// are issued before the first use (16 KB in flight per wave); PRE (nelec already in registers) is
// kept for experiments.
template <bool PRE, bool DIAG = false>
__device__ __forceinline__ void hw_epilogue(const RenderArgs &a, const double *__restrict__ acc, double *__restrict__ lt,
                                            const BandDev *__restrict__ bd, int tile, int b, int xi, int Y0, int lane,
                                            const double *ne_pre) {
    const int half = lane >> 5;
    __syncthreads();
    lt[lane] = c_log_ic[lane];
    lt[64 + lane] = c_log_lc[lane];
    __syncthreads();
    const double eps = bd->eps;
    const bool store = !(a.flags & CEL_RENDER_NO_STORE);
    const bool ll = (a.flags & CEL_RENDER_LOGLIK) != 0;
    double part = 0.0;
    const int64_t plane = (int64_t)b * a.H * a.W;
    const int dbg = DIAG ? CEL_ABLATE_BITS(a.flags) : 0;
    // A tile wholly inside the frame takes a form without per-row bounds: with loads and stores under conditions the
    // compiler cannot count what is outstanding where their paths meet and waits for EVERYTHING (s_waitcnt vmcnt(0))
    // at every use of a loaded value and before every store -- once per row for the previous row's store to be
    // acknowledged.  Here it counts (vmcnt(31) ... vmcnt(0)): the stores leave without anyone waiting for them.
    const bool inside = (xi - (lane & 31) + HW_TW <= a.W) && (Y0 + HW_TH <= a.H);
    if (inside && ll && !PRE && !dbg) {
        const int64_t base = plane + (int64_t)(Y0 + half) * a.W + xi;
        double ne[HW_TH / 2];
#pragma unroll
        for (int r = 0; r < HW_TH / 2; r++) ne[r] = a.nelec[base + (int64_t)(2 * r) * a.W];
        if (store) {
#pragma unroll
            for (int r = 0; r < HW_TH / 2; r++) {
                const double lam = eps + acc[r * 64 + lane];
                a.lambda[base + (int64_t)(2 * r) * a.W] = lam;
                part += ne[r] * log_tab(lam, lt) - lam;
            }
        } else {
#pragma unroll
            for (int r = 0; r < HW_TH / 2; r++) {
                const double lam = eps + acc[r * 64 + lane];
                part += ne[r] * log_tab(lam, lt) - lam;
            }
        }
    } else if (inside && !ll && !dbg) {      // model images only (gen_model_image): stores, none of them under a condition
        if (store) {
            const int64_t base = plane + (int64_t)(Y0 + half) * a.W + xi;
#pragma unroll
            for (int r = 0; r < HW_TH / 2; r++) a.lambda[base + (int64_t)(2 * r) * a.W] = eps + acc[r * 64 + lane];
        }
    } else if (xi < a.W && !(dbg & 16)) {
        const int64_t base = plane + (int64_t)(Y0 + half) * a.W + xi;
        double ne[HW_TH / 2];
#pragma unroll
        for (int r = 0; r < HW_TH / 2; r++)
            ne[r] = PRE ? ne_pre[r] : ((ll && Y0 + 2 * r + half < a.H) ? a.nelec[base + (int64_t)(2 * r) * a.W] : 0.0);
#pragma unroll
        for (int r = 0; r < HW_TH / 2; r++) {
            if (Y0 + 2 * r + half < a.H) {
                double lam = eps + acc[r * 64 + lane];
                if (store) a.lambda[base + (int64_t)(2 * r) * a.W] = lam;
                if (ll) part += (dbg & 4) ? ne[r] - lam : ne[r] * log_tab(lam, lt) - lam;
            }
        }
    }
    if (ll) {
        part = wave_sum(part);
        if (lane == 0) a.partials[tile] = part;
    }
}

// an empty tile: lambda = eps everywhere -- pure streaming (nelec in, eps out), one log per wave instead of
// one per pixel, no LDS.  Same arithmetic per pixel as the general epilogue (ne * log(lam) - lam, rows in the
// same order).  Shared by k_render_hw and k_render_stars.
__device__ __forceinline__ void hw_empty_tile(const RenderArgs &a, const BandDev *__restrict__ bd, int tile, int b, int X0, int Y0,
                                              int lane, unsigned long long t_start) {
    const int half = lane >> 5, xi = X0 + (lane & 31);
    const double eps = bd->eps;
    const bool store = !(a.flags & CEL_RENDER_NO_STORE);
    const bool ll = (a.flags & CEL_RENDER_LOGLIK) != 0;
    const int64_t base = (int64_t)b * a.H * a.W + (int64_t)(Y0 + half) * a.W + xi;
    double part = 0.0;
    if ((a.W & 31) == 0 && Y0 + HW_TH <= a.H) {
        // whole tile inside an aligned frame: 16 B per lane (4 rows of 16 lane-pairs per
        // instruction), all 16 loads in flight before the first use
        const int cp = lane & 15, rq = lane >> 4;
        const int64_t b2 = (int64_t)b * a.H * a.W + (int64_t)(Y0 + rq) * a.W + X0 + 2 * cp;
        const double leps = ll ? log(eps) : 0.0;
        double2 ne2[HW_TH / 4];
        if (ll) {
#pragma unroll
            for (int r = 0; r < HW_TH / 4; r++)
                ne2[r] = *reinterpret_cast<const double2 *>(a.nelec + b2 + (int64_t)(4 * r) * a.W);
        }
#pragma unroll
        for (int r = 0; r < HW_TH / 4; r++) {
            if (store) *reinterpret_cast<double2 *>(a.lambda + b2 + (int64_t)(4 * r) * a.W) = make_double2(eps, eps);
            if (ll) part += (ne2[r].x * leps - eps) + (ne2[r].y * leps - eps);
        }
    } else if (xi < a.W) {
        double ne[HW_TH / 2];
#pragma unroll
        for (int r = 0; r < HW_TH / 2; r++)
            ne[r] = (ll && Y0 + 2 * r + half < a.H) ? a.nelec[base + (int64_t)(2 * r) * a.W] : 0.0;
        const double leps = ll ? log(eps) : 0.0;
#pragma unroll
        for (int r = 0; r < HW_TH / 2; r++) {
            if (Y0 + 2 * r + half < a.H) {
                if (store) a.lambda[base + (int64_t)(2 * r) * a.W] = eps;
                if (ll) part += ne[r] * leps - eps;
            }
        }
    }
    if (ll) {
        part = wave_sum(part);
        if (lane == 0) a.partials[tile] = part;
    }
    if (a.cost && lane == 0) a.cost[tile] = (int)min(wall_clock64() - t_start, 0x3fffffffull) + 1;
}

static_assert(sizeof(CompTab) >= 128 * sizeof(double), "log table must fit the component table");
// DIAG = false is the production kernel.  DIAG = true adds the per-tile work counters / time stamps of
// CEL_OPT_TILE_TIMING and, in a -DCEL_ABLATE build only, the timing-only ablation switches of CEL_OPT_DEBUG:
// the host launches it only when one of them is asked for, so the hottest loop of the library carries no
// diagnostic branch, scalar register or counter (round 2: 8 flag tests inside per-source / per-batch code).
//
// PARTS > 1 (round 5): a FRAME OF FEW TILES -- one rank's strip of an 8-way cut is 1 280 tiles for 2 048 wave slots, a real
// 51 x 51 field is five -- finishes when its heaviest tile does, a single wave walking ~40 sources (0.35-0.43 ms for a
// strip whose share of the work is 0.14 ms).  Here PARTS one-wave blocks share a tile: part p takes the stars and the
// galaxies p, p + PARTS, ... of the tile's list (by list position: the dealing depends on the data only), accumulates
// them in its own LDS tile and stores that as a 16 KB slab; the block that draws the last ticket of the tile's counter
// adds the slabs IN PART ORDER -- whoever it is -- and runs the epilogue.  The hand-off is the in-launch split-K reduction
// of the CDNA guide (plain slab stores, vmcnt drain, agent-scope release, relaxed agent fetch_add; the last arriver: agent
// acquire, plain loads): correct wherever a tile's parts run.  A tile's parts sit on ONE XCD (block id -> XCD is round
// robin: ids that agree mod 8 share an L2), where the slabs are read at twice the cross-XCD rate.  Values agree with the
// one-wave form to rounding (a pixel's terms are added part by part), and are the same bits in every run.
template <bool DIAG, int PARTS = 1>
__global__ void __launch_bounds__(64)
k_render_hw(RenderArgs a) {
    __shared__ double acc[HW_TH * HW_TW];
    __shared__ CompTab T;
    __shared__ double et[64];
    const int lane = threadIdx.x;
    const int half = lane >> 5, col = lane & 31;
    unsigned long long *const timing = DIAG ? a.timing : nullptr;
    const int dbg = DIAG ? CEL_ABLATE_BITS(a.flags) : 0;
    const unsigned long long t_start = (timing || a.cost) ? wall_clock64() : 0ull;
    int tpos = blockIdx.x, part = 0;
    if (PARTS > 1) {                   // ids b, b + 8, ..., b + 8 (PARTS - 1) -- one XCD -- are the parts of one tile
        const int q = blockIdx.x >> 3;
        tpos = (q / PARTS) * 8 + (blockIdx.x & 7);
        part = q % PARTS;
        if (tpos >= a.B * a.ntx * a.nty) return;
    }
    const int tile = a.order ? a.order[tpos] : tpos;
    if (PARTS == 1 && !DIAG && a.dirty && !a.dirty[tile]) return;      // an incremental render: this tile did not change
    const int per_band = a.ntx * a.nty;
    const int b = tile / per_band;
    const int t = tile - b * per_band;
    const int ty = t / a.ntx, tx = t - ty * a.ntx;
    const int X0 = tx * HW_TW, Y0 = ty * HW_TH;
    const int xi = X0 + col;
    const double x = (double)xi;
    const BandDev *bd = a.bands + b;

    const int cnt = a.tile_cnt[tile];
    if (dbg & 8) return;             // ablation: launch + header load only
    if (cnt == 0 && !timing) {
        if (part == 0) hw_empty_tile(a, bd, tile, b, X0, Y0, lane, t_start);
        return;
    }
    // how many of the tile's PARTS blocks work: one per HW_PART_ENTRIES list entries (a tile of a few sources is not worth a
    // slab hand-off; the heaviest tiles, which decide when a launch of few tiles ends, get all of them).  A function of the
    // tile's list length only: every block of the tile computes the same number, the others leave at once.
    const int peff = (PARTS > 1) ? min(PARTS, max(1, (cnt + HW_PART_ENTRIES - 1) / HW_PART_ENTRIES)) : 1;
    if (PARTS > 1 && part >= peff) return;

    et[lane] = exp2((double)lane * (1.0 / 64.0));
#pragma unroll
    for (int r = 0; r < HW_TH / 2; r++) acc[r * 64 + lane] = 0.0;

    const int64_t off = a.tile_off[tile];
    const SrcRec *recs = a.recs + (int64_t)b * a.S;
    const double Tdrop = a.tail_T;
    const double eps_sky = bd->eps;
    const bool dropping = (a.variant != 0) && (Tdrop > 0.0) && (eps_sky > 0.0);
    const float log_eps = dropping ? __logf((float)eps_sky) : 0.0f;
    const int strict = (a.flags >> 2) & 1;   // photon-split totals: boxes open on the low side (internal flag)

    unsigned dbg_pairrows = 0, dbg_comprows = 0, dbg_pairs = 0;   // only counted under CEL_OPT_TILE_TIMING
    float dbg_area = 0.f;
    unsigned dbg_halfrows = 0;                                   // star path: component-rows (each walked on 32 lanes, like the general path's)
    const int nent_all = (int)min((int64_t)cnt, a.capacity > off ? a.capacity - off : (int64_t)0);
    int nstar = min(a.tile_nstar ? a.tile_nstar[tile] : 0, nent_all);
    if (dbg & 32) nstar = 0;         // ablation: no star pass at all (and no general pass: see below)
    if (a.variant == 0) nstar = 0;             // the direct evaluator takes every source through the general path
    if (nstar > 0) {
        StarTab &ST = *reinterpret_cast<StarTab *>(&T);
        __syncthreads();                       // et[] is written
        if (!star_setup(ST, bd, et, lane)) nstar = 0;   // a very sharp PSF component: general path (segments, direct fallback)
    }
    const int nstar_all = nstar;               // the rest of the list begins behind ALL of the tile's stars
    if (PARTS > 1) nstar = (nstar > part) ? (nstar - part + peff - 1) / peff : 0;         // this part's stars: entries part, part + peff, ...
    // (Requesting a star-only tile's nelec BEFORE the star pass, so that the loads land under the
    // arithmetic, was tried here -- before and after the first batch's record loads: a wave's loads
    // return in order -- and in a persistent, software-pipelined star kernel: slower in every form,
    // 0.179 / 0.180 / 0.205 against 0.171 ms on the dense star field -- DESIGN.md 5.)
    if (nstar > 0)
        star_pass<DIAG>(a, *reinterpret_cast<StarTab *>(&T), et, acc, recs, off, nstar, lane, X0, Y0, strict, dbg_halfrows, dbg_pairs,
                        PARTS > 1 ? part : 0, peff);

    const LaneConst lc = lane_consts(lane, bd);
    // the rest of the tile's list (everything when there was no star pass), 64 indices per coalesced
    // load; the next source's record is in flight while the current one is evaluated
    const int64_t off2 = off + nstar_all + (PARTS > 1 ? part : 0);
    int nent = (dbg & 32) ? 0 : nent_all - nstar_all;
    if (PARTS > 1) nent = (nent > part) ? (nent - part + peff - 1) / peff : 0;            // entries part, part + peff, ... of the rest
    int idx64 = (lane < nent) ? a.lists[off2 + (int64_t)peff * lane] : 0;
    int recw_next = (nent > 0) ? rec_fetch(recs, __builtin_amdgcn_readlane(idx64, 0), lane) : 0;

    for (int e = 0; e < nent; e++) {
        const int recw = recw_next;
        if (e + 1 < nent) {
            if (((e + 1) & 63) == 0) idx64 = (e + 1 + lane < nent) ? a.lists[off2 + (int64_t)peff * (e + 1 + lane)] : 0;
            recw_next = rec_fetch(recs, __builtin_amdgcn_readlane(idx64, (e + 1) & 63), lane);
        }
        const RecU rec = rec_unpack(recw);
        const RecU *rp = &rec;
        const int type = rec.type;
        const int K = (type == 0) ? K_PSF : K_GAL;
        const int bx0 = rp->x0 + strict, bx1 = rp->x1, by0 = rp->y0 + strict, by1 = rp->y1;
        const int ra = max(by0, Y0) - Y0, rb = min(by1, Y0 + HW_TH) - Y0;
        const bool on = (xi >= bx0) && (xi < bx1);
        const double xa = (double)max(bx0, X0), xb = (double)(min(bx1, X0 + HW_TW) - 1);
        const double ya = (double)(Y0 + ra), yb = (double)(Y0 + rb - 1);

        bool keep = false;
        Comp c;
        int Lk = 0, rlo = ra, rhi = rb;
        if (lane < K) {
            c = make_comp_lc(lc, rec);
            double Tk = dropping ? Tdrop + (double)(__logf((float)fabs(c.A)) - log_eps) : 100.0;   // T + log(A / eps), no fp64 division
            if (dropping) {
                double qmin = quad_min_rect(c.qa, c.qb, c.qc, xa - c.mx, xb - c.mx, ya - c.my, yb - c.my);
                keep = (0.5 * qmin <= Tk);
                // rows on which the component can matter on THIS tile's columns
                float ylo, yhi;
                quad_rows_on_columns(c.qa, c.qb, c.qc, 2.0 * fmax(Tk, 0.0), xa - c.mx, xb - c.mx, ylo, yhi);
                const float cy = (float)(c.my - (double)Y0);
                // (the integer rows inside [cy + ylo, cy + yhi], a fiftieth of a row of margin for the fp32 ends: rounds 1-5 took
                // floor and ceil + 1, a row more at either end of every component that ends inside the tile)
                rlo = max(ra, (int)ceilf(cy + ylo - 0.02f));
                rhi = min(rb, (int)floorf(cy + yhi + 0.02f) + 1);
                keep = keep && (rhi > rlo);
            } else {
                keep = true;
            }
            Lk = seg_len(c.qc, fmin(fmax(Tk, 1.0), 300.0));
        }
        const unsigned long long km = __ballot(keep);
        const int Kk = __popcll(km);
        if (timing) {   // diagnostic: rows / column-clipped area the kept components need one by one
            int ir = keep ? rhi - rlo : 0;
            float cw = 0.f;
            if (keep) {
                double Tk2 = Tdrop + (double)__logf((float)(fabs(c.A) / eps_sky));
                float hx = sqrt_f32(2.0f * (float)fmax(Tk2, 0.0) / (float)c.ixx) + 1.0f;
                float lo = fmaxf((float)xa, (float)c.mx - hx), hi = fminf((float)xb + 1.f, (float)c.mx + hx);
                cw = fmaxf(hi - lo, 0.f);
            }
            float ar = (float)ir * cw;
            for (int o = 32; o; o >>= 1) { ir += __shfl_xor(ir, o); ar += __shfl_xor(ar, o); }
            dbg_pairrows += (unsigned)ir; dbg_area += ar;
        }
        HW_WAVE_SYNC();    // previous source's table reads are done
        if (lane < 4) { T.gL[lane] = 4096; T.gr0[lane] = HW_TH; T.gr1[lane] = 0; T.gs0[lane] = HW_TH; T.gs1[lane] = 0; }
        const int slot = slot_by_rows(keep, rlo, rhi);
        if (keep) {
            const int p = slot;
            T.A[p] = c.A; T.mx[p] = c.mx; T.my[p] = c.my;
            T.qa[p] = c.qa * EXP_SCALE; T.qb[p] = c.qb * EXP_SCALE; T.qc[p] = c.qc * EXP_SCALE;
            T.eq[p] = exp_tab64(-c.qc * EXP_SCALE, et);
            // LDS operations of one wave execute in order: the min/max below land after the resets above
            const int gi = p / (2 * REC_G);
            atomicMin(&T.gL[gi], Lk);
            atomicMin(&T.gr0[gi], rlo);
            atomicMax(&T.gr1[gi], rhi);
            // the pair's second set: the slots behind its first 2 NB (rec_group_nested)
            const int Rp = min(2 * REC_G, Kk - gi * (2 * REC_G));
            if (p - gi * (2 * REC_G) >= 2 * nested_big((Rp + 1) / 2)) {
                atomicMin(&T.gs0[gi], rlo);
                atomicMax(&T.gs1[gi], rhi);
            }
        }
        if (lane < HW_PAD) {   // zero components behind the table (amplitude 0, ratio 1)
            int p = Kk + lane;
            T.A[p] = 0.0; T.mx[p] = 0.0; T.my[p] = 0.0;
            T.qa[p] = 0.0; T.qb[p] = 0.0; T.qc[p] = 0.0;
            T.eq[p] = 1.0;
        }
        HW_WAVE_SYNC();
        if (a.variant == 0) {
            // direct evaluator: the halves split the kept components
            const int kh = (Kk + 1) / 2;
            const int k0 = half ? kh : 0, k1 = half ? Kk : kh;
            for (int row = ra; row < rb; row++) {
                double v = eval_direct(T, k0, k1, x, (double)(Y0 + row), 1.0 / EXP_SCALE);
                if (on) lds_add(&acc[row * HW_TW + col], v);
            }
            continue;
        }
        // pairs of groups: the lower half takes gA components, the upper half the next gB
        for (int p0 = 0; p0 < Kk; p0 += 2 * REC_G) {
            const int R = min(2 * REC_G, Kk - p0);
            const int gA = (R + 1) / 2;
            const int gi = p0 / (2 * REC_G);
            const int L = __builtin_amdgcn_readfirstlane(T.gL[gi]);
            const int ga = __builtin_amdgcn_readfirstlane(T.gr0[gi]);
            const int gb = __builtin_amdgcn_readfirstlane(T.gr1[gi]);
            const int k0 = half ? p0 + gA : p0;
            // the nested walk: a pair of at least 7 components in one segment
            const int nb = nested_big(gA);
            const bool nested = (gA >= 4) && (gb - ga <= L);
            int sa = 0, sb = 0;
            if (nested) {
                sa = __builtin_amdgcn_readfirstlane(T.gs0[gi]);
                sb = __builtin_amdgcn_readfirstlane(T.gs1[gi]);
                sa = min(max(sa, ga), gb);
                sb = max(min(sb, gb), sa);
            }
            if (timing) {
                dbg_comprows += nested ? (unsigned)(gb - ga) * (unsigned)(2 * nb) + (unsigned)(sb - sa) * (unsigned)(R - 2 * nb)
                                       : (unsigned)(gb - ga) * (unsigned)R;
                dbg_pairs += 1;
            }
            if (nested) {
                double *colp = acc + col;
                const int kb = p0 + half * nb, ks = p0 + 2 * nb + half * (gA - nb);
                switch (gA) {
                case 6: rec_group_nested<3, 3>(T, et, kb, ks, x, Y0, ga, gb, sa, sb, on, colp); break;
                case 5: rec_group_nested<3, 2>(T, et, kb, ks, x, Y0, ga, gb, sa, sb, on, colp); break;
                default: rec_group_nested<2, 2>(T, et, kb, ks, x, Y0, ga, gb, sa, sb, on, colp); break;
                }
                continue;
            }
            if (L < 4) {
                // pathologically sharp component: evaluate this pair of groups directly
                const int k1 = half ? p0 + R : p0 + gA;
                for (int row = ga; row < gb; row++) {
                    double v = eval_direct(T, k0, k1, x, (double)(Y0 + row), 1.0 / EXP_SCALE);
                    if (on) lds_add(&acc[row * HW_TW + col], v);
                }
                continue;
            }
            double *colp = acc + col;
            switch (gA) {
            case 6: rec_group_hw<6>(T, et, k0, x, Y0, ga, gb, L, on, colp); break;
            case 5: rec_group_hw<5>(T, et, k0, x, Y0, ga, gb, L, on, colp); break;
            case 4: rec_group_hw<4>(T, et, k0, x, Y0, ga, gb, L, on, colp); break;
            case 3: rec_group_hw<3>(T, et, k0, x, Y0, ga, gb, L, on, colp); break;
            case 2: rec_group_hw<2>(T, et, k0, x, Y0, ga, gb, L, on, colp); break;
            default: rec_group_hw<1>(T, et, k0, x, Y0, ga, gb, L, on, colp); break;
            }
        }
    }

    if (PARTS > 1 && peff > 1) {
        // the tile's parts meet: this part's accumulator goes out as a slab; the last of the tile's parts to arrive adds all of
        // them up in part order
        __syncthreads();
        double *slab = a.slabs + ((size_t)tile * PARTS + part) * (HW_TH * HW_TW);
#pragma unroll
        for (int r = 0; r < HW_TH / 2; r++) slab[r * 64 + lane] = acc[r * 64 + lane];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int ticket = 0;
        if (lane == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            ticket = __hip_atomic_fetch_add(&a.part_cnt[tile], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        ticket = __builtin_amdgcn_readfirstlane(ticket);
        if (ticket != peff - 1) return;
        if (lane == 0) {
            __hip_atomic_store(&a.part_cnt[tile], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // for the next launch (zeroed when allocated)
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        const double *s0 = a.slabs + (size_t)tile * PARTS * (HW_TH * HW_TW);
#pragma unroll 4
        for (int r = 0; r < HW_TH / 2; r++) {
            double v = s0[r * 64 + lane];
            for (int p = 1; p < peff; p++) v += s0[(size_t)p * (HW_TH * HW_TW) + r * 64 + lane];
            acc[r * 64 + lane] = v;
        }
    }
    hw_epilogue<false, DIAG>(a, acc, reinterpret_cast<double *>(&T), bd, tile, b, xi, Y0, lane, nullptr);
    if (a.cost && lane == 0) a.cost[tile] = (int)min((wall_clock64() - t_start) * (unsigned long long)peff, 0x3fffffffull) + 1;
    if (timing && lane == 0) {
        timing[3 * (size_t)tile + 0] = t_start;
        timing[3 * (size_t)tile + 1] = wall_clock64();
        // work counters of this tile (diagnostic): sources | pairs of groups << 12 | kept component-rows << 32
        timing[3 * (size_t)tile + 2] = (unsigned long long)(unsigned)cnt | ((unsigned long long)dbg_pairs << 12) |
                                               ((unsigned long long)(dbg_comprows + dbg_halfrows) << 32);
        if ((a.flags >> 8) & 128)   // diagnostic (tools/row_waste.py): what the kept components need one by one
            timing[3 * (size_t)tile + 2] = (unsigned long long)dbg_pairrows | ((unsigned long long)(unsigned)(dbg_area / 32.f) << 32);
    }
}
