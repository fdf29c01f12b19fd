// k_small_stars.h -- a small star field's whole evaluation in ONE launch (BASELINE configs[1]: 1 000 stars x 5 bands x 512^2)
//
// Why.  On a frame of a few hundred render tiles the step of the general path is four launches -- k_prep, k_bin_direct,
// the render kernel, k_reduce -- each far too short to fill the chip (640 one-wave tiles on 2 048 wave slots) and each
// paying a dependent-launch boundary: at configs[1] the render kernel ran 37 us at 7 % of the HBM roof and was still
// less than half of the 75 us step.  For a catalogue of at most SMALL_MAX_S stars on at most STAR_TILES_MIN tiles this kernel
// does all four jobs:
//   * a block is one PART of a tile: SMALL_NP column parts of 32 / SMALL_NP columns x 64 rows, every part a block of its own
//     (k_render_stars walks a tile's parts one behind the other; the unit of work -- one column of one star -- belongs to
//     exactly one part, so nothing is seeded twice and four times as many waves are in flight);
//   * no lists: the block tests the band's stars against its rectangle itself -- equa2pixel, the overlap test and the int()
//     box are k_prep's own expressions (prep_pixel / prep_star_box / prep_window) -- and keeps the hits' indices in LDS, in
//     ascending order; a designated block per 64 sources also writes their records, boxes and status words, so everything
//     that reads k_prep's outputs after a render (cel_field_stats, cel_stamp_boxes, ...) finds them;
//   * no reduction launch: a block's Poisson partial is one double; the B x blocks-per-band partials ride back to the host in
//     the step's one copy (20 KB at configs[1]) and the host adds each band's in index order (Kahan), so the per-band
//     log-likelihoods are reproducible bit for bit.  (A last-block-done sum inside the kernel -- write-through partial,
//     agent-scope counter add, the last arriver reads the band's partials with sc1 loads -- was built and measured first:
//     correct, and 20 us of a 60 us kernel: 512 blocks per band finish together and queue on one counter word.)
// Measured at configs[1] by timing-only ablations (tools/ab_small.sh): launch + tables 5 us, walk 5 us, epilogue 2 us; the
// scan was 28 us while every block ran k_prep's full arithmetic on every star, hence the cheap position filter below.
#pragma once
#include "k_render_stars.h"

#define SMALL_NP 4            // column parts per tile = blocks per tile
#define SMALL_CW (HW_TW / SMALL_NP)
#define SMALL_MAX_S 4096      // = BIN_DIRECT_MAX_S: the catalogue sizes the one-wave scans are meant for
#define SMALL_CAP 256         // candidate stars of one part (8 x 64 pixels grown by the star radius) the kernel can hold; more: the host takes the general path
#ifndef SMALL_ABL
#define SMALL_ABL 0           // timing-only ablations (tools/ab_small.sh builds them): 1 no scan, 2 no walk, 4 no epilogue
#endif

struct SmallArgs {
    const double *radec, *counts;            // the catalogue as cel_sources holds it: radec[S][2], counts[S][B]
    SrcRec *recs; int4 *boxes; int *kind; int *status;       // k_prep's outputs
    double *partials;                         // out: one Poisson partial per block (band-major); the host sums them
    unsigned long long *flag;                 // set to `stamp` when a part holds more than SMALL_CAP stars
    unsigned long long stamp;
    const double *consts;                     // per band SMALL_CONSTS doubles: what every block used to compute for itself (k_small_consts)
    unsigned long long *stamps;               // diagnostic (CEL_SMALL_STAMPS): 6 wall-clock stamps (100 MHz) + XCC/CU id per block, or nullptr
    int full_H, win_y0;
};

// Per-band constants of the star pass, formed ONCE per image set by k_small_consts with the device's own arithmetic (so a
// block that loads them holds the bits it would have computed): the star table's 21 doubles (star_setup), cos(phi_1),
// the 2^(j/64) table and the two log tables.  2 560 blocks each spent ~3 us of dependent fp64 library code (exp2, cos, a
// division, a square root) on them before touching a star.
#define SMALL_CONSTS (21 + 1 + 64 + 128)
__global__ void __launch_bounds__(64) k_small_consts(const BandDev *__restrict__ bands, double *__restrict__ out) {
    __shared__ StarTab ST;
    __shared__ double et[64];
    const int lane = threadIdx.x, b = blockIdx.x;
    et[lane] = exp2((double)lane * (1.0 / 64.0));
    __syncthreads();
    star_setup(ST, bands + b, et, lane);
    double *o = out + (int64_t)b * SMALL_CONSTS;
    if (lane < 21) o[lane] = (&ST.qa[0])[lane];          // qa, qb, qc, eq, A0, mux, muy: contiguous at the table's end
    if (lane == 0) o[21] = cos(bands[b].phi[1] / 180.0 * PI_D);
    o[22 + lane] = et[lane];
    o[86 + lane] = c_log_ic[lane];
    o[150 + lane] = c_log_lc[lane];
}
static_assert(offsetof(StarTab, muy) - offsetof(StarTab, qa) == 18 * sizeof(double), "the star table's constants must be contiguous");


// k_prep's record of one star (position, counts, box on the window, type / status)
__device__ __forceinline__ void small_prep(const RenderArgs &a, const SmallArgs &x, const BandDev &bd, double cphi, int b, int64_t s,
                                           SrcRec &r) {
    memset(&r, 0, sizeof(r));
    double px, py;
    prep_pixel(bd, x.radec[2 * s], x.radec[2 * s + 1], cphi, px, py);
    r.px = px; r.py = py;
    r.scale = x.counts[s * a.B + b];
    r.type = 0;
    prep_star_box(bd, px, py, x.full_H, a.W, r);
    prep_window(r, py, x.win_y0, a.H);
}

// one batch of the part's hit list into the star table, sorted by the rows a star has on this part (star_stage's rule:
// descending, ties by list position; a star without a row or a column here sorts last).  A hit's position comes
// from the scan's LDS tables, its box is k_prep's expressions on that position; its counts are the one global load.
__device__ __forceinline__ void small_stage(const RenderArgs &a, const SmallArgs &x, StarTab &ST, const unsigned short *__restrict__ hits,
                                            const double *__restrict__ cpx, const double *__restrict__ cpy,
                                            int base, int nb, int lane, int b, const BandDev &bd, int Xa, int Y0) {
    __syncthreads();                   // the previous batch has been read
    double2 pp = make_double2(0.0, 0.0);
    double sc = 0.0;
    int4 bx4 = make_int4(0, 0, 0, 0);
    int nrows = -1;
    if (lane < nb) {
        sc = x.counts[(int64_t)hits[base + lane] * a.B + b];
        SrcRec r;
        r.x0 = r.x1 = r.y0 = r.y1 = 0;
        r.type = 0;
        const double px = cpx[base + lane], py = cpy[base + lane];
        prep_star_box(bd, px, py, x.full_H, a.W, r);
        prep_window(r, py, x.win_y0, a.H);
        pp = make_double2(px, r.py);
        bx4 = make_int4(r.x0, r.x1, r.y0, r.y1);
        nrows = max(min(bx4.w, Y0 + HW_TH) - max(bx4.z, Y0), 0);
        const int ncols = max(min(bx4.y, Xa + SMALL_CW) - max(bx4.x, Xa), 0);
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

__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3)))
k_small_stars(RenderArgs a, SmallArgs x) {
    __shared__ double acc[HW_TH * SMALL_CW];
    __shared__ StarTab ST;
    __shared__ double et[64];
    __shared__ double lt[128];
    __shared__ unsigned short hits[SMALL_CAP];
    __shared__ double cpx[SMALL_CAP], cpy[SMALL_CAP];      // the candidates' pixel positions (full-frame rows)
    __shared__ unsigned char own[64 * SMALL_CW];           // the star of every (star, column) task of a batch (star_walk)
    const int lane = threadIdx.x;
    const int per_band = a.ntx * a.nty;
    const int nblk_band = per_band * SMALL_NP;
    const int b = blockIdx.x / nblk_band;
    const int q = blockIdx.x - b * nblk_band;           // this block among its band's
    const int t = q / SMALL_NP, p = q - t * SMALL_NP;
    const int ty = t / a.ntx, tx = t - ty * a.ntx;
    const int X0 = tx * HW_TW, Y0 = ty * HW_TH;
    const int Xa = X0 + p * SMALL_CW;
    const BandDev *bd = a.bands + b;
    const BandDev &bdr = *bd;
    const int S = (int)a.S;
    unsigned long long tstamp[6];
    if (x.stamps) tstamp[0] = wall_clock64();

    // ---- nearly everything this block reads from global memory is requested NOW -- the band's constants, the first 1024
    // stars' positions, the part's observed pixels -- in one round trip
    const double *sc = x.consts + (int64_t)b * SMALL_CONSTS;
    const double v_st = sc[min(lane, 20)], v_et = sc[22 + lane], v_l0 = sc[86 + lane], v_l1 = sc[150 + lane];
    const double cphi = sc[21];
    constexpr int U = 16;                       // stars per lane in flight
    double ra[U], de[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
        const int s = min(64 * u + lane, S - 1);
        const double2 rd = *reinterpret_cast<const double2 *>(x.radec + 2 * (int64_t)s);
        ra[u] = rd.x; de[u] = rd.y;
    }
    const bool in_frame = (X0 + HW_TW <= a.W) && (Y0 + HW_TH <= a.H);
    const bool inside = in_frame && (a.flags & CEL_RENDER_LOGLIK);
    const bool store = !(a.flags & CEL_RENDER_NO_STORE);
    double ne[HW_TH * SMALL_CW / 64];
    if (inside && !(SMALL_ABL & 4)) stars_nelec<SMALL_CW, true>(a, b, Xa, Y0, lane, ne);

#pragma unroll
    for (int r = 0; r < HW_TH * SMALL_CW / 64; r++) acc[r * 64 + lane] = 0.0;
    et[lane] = v_et;
    lt[lane] = v_l0;
    lt[64 + lane] = v_l1;
    if (lane < 21) (&ST.qa[0])[lane] = v_st;    // star_setup's values (the host checked the one-segment condition for every band)
    const double eps = bd->eps;
    __syncthreads();

    if (x.stamps) tstamp[1] = wall_clock64();
    // ---- the band's stars against this part's rectangle.  Pass 1, every star: the pixel position only (six flops) against
    // the rectangle grown by the star radius + 3 -- the int() box reaches less than R + 2 from the position, so no star
    // whose box meets the rectangle is lost; the candidates' indices and positions go to LDS in ascending order.  Pass 2,
    // the candidates (a few dozen): k_prep's exact box against the rectangle.
    int nh = 0;
    {
        const double grow = bdr.R + 3.0;
        const double xlo = (double)Xa - grow, xhi = (double)(Xa + SMALL_CW) + grow;
        const double ylo = (double)(Y0 + x.win_y0) - grow, yhi = (double)(Y0 + x.win_y0 + HW_TH) + grow;
        int nc = 0;
        for (int s0 = 0; s0 < S && !(SMALL_ABL & 1); s0 += 64 * U) {
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int s = s0 + 64 * u + lane;
                double px, py;
                prep_pixel(bdr, ra[u], de[u], cphi, px, py);
                const bool cand = (s < S) && (px > xlo) && (px < xhi) && (py > ylo) && (py < yhi);
                const unsigned long long m = __ballot(cand);
                if (cand) {
                    const int at = nc + __popcll(m & ((1ull << lane) - 1ull));
                    if (at < SMALL_CAP) { hits[at] = (unsigned short)s; cpx[at] = px; cpy[at] = py; }
                }
                nc += __popcll(m);
            }
            if (s0 + 64 * U < S) {              // a catalogue of more than 1024 stars: the next 1024
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const int s = min(s0 + 64 * U + 64 * u + lane, S - 1);
                    const double2 rd = *reinterpret_cast<const double2 *>(x.radec + 2 * (int64_t)s);
                    ra[u] = rd.x; de[u] = rd.y;
                }
            }
        }
        if (nc > SMALL_CAP) {                   // the host renders this call again on the general path
            if (lane == 0) *x.flag = x.stamp;
            nc = SMALL_CAP;
        }
        __syncthreads();
        for (int c0 = 0; c0 < nc; c0 += 64) {   // exact test (no loads); the tables are compacted in place (writes trail reads)
            const int i = c0 + lane, ic = min(i, nc - 1);
            const double px = cpx[ic], py = cpy[ic];
            const unsigned short sh = hits[ic];
            SrcRec r;
            r.x0 = r.x1 = r.y0 = r.y1 = 0;
            r.type = 0;
            prep_star_box(bdr, px, py, x.full_H, a.W, r);
            prep_window(r, py, x.win_y0, a.H);
            const bool hit = (i < nc) && r.type >= 0 && box_hits(make_int4(r.x0, r.x1, r.y0, r.y1), Xa, Xa + SMALL_CW, Y0, Y0 + HW_TH);
            const unsigned long long m = __ballot(hit);
            __syncthreads();
            if (hit) {
                const int at = nh + __popcll(m & ((1ull << lane) - 1ull));
                hits[at] = sh; cpx[at] = px; cpy[at] = py;
            }
            nh += __popcll(m);
            __syncthreads();
        }
    }

    // ---- the part's stars into its accumulator
    if (x.stamps) tstamp[2] = wall_clock64();
    unsigned d0 = 0;
    for (int base = 0; base < nh; base += 64) {
        const int nb = min(64, nh - base);
        small_stage(a, x, ST, hits, cpx, cpy, base, nb, lane, b, bdr, Xa, Y0);
        if (!(SMALL_ABL & 2)) star_walk<false, SMALL_CW>(a, ST, et, acc, nb, lane, Xa, Y0, 0, d0, own);
    }
    __syncthreads();

    // ---- epilogue: lambda = eps + acc written once, the Poisson terms of the part
    if (x.stamps) tstamp[3] = wall_clock64();
    double part = 0.0;
    if (SMALL_ABL & 4) {
        part = acc[lane];
    } else if (inside) {
        part = store ? stars_epilogue<SMALL_CW, true, true>(a, acc, lt, eps, b, Xa, Y0, lane, ne)
                     : stars_epilogue<SMALL_CW, true, false>(a, acc, lt, eps, b, Xa, Y0, lane, ne);
    } else if (in_frame) {                      // model images only: stores, none of them under a condition
        if (store) {
            constexpr int RPI = 64 / SMALL_CW;
            const int64_t base = (int64_t)b * a.H * a.W + (int64_t)(Y0 + lane / SMALL_CW) * a.W + Xa + lane % SMALL_CW;
#pragma unroll
            for (int r = 0; r < HW_TH / RPI; r++) a.lambda[base + (int64_t)(RPI * r) * a.W] = eps + acc[r * 64 + lane];
        }
    } else {
        stars_nelec<SMALL_CW, false>(a, b, Xa, Y0, lane, ne);
        part = store ? stars_epilogue<SMALL_CW, false, true>(a, acc, lt, eps, b, Xa, Y0, lane, ne)
                     : stars_epilogue<SMALL_CW, false, false>(a, acc, lt, eps, b, Xa, Y0, lane, ne);
    }
    if (a.flags & CEL_RENDER_LOGLIK) {          // the part's Poisson partial; the host adds a band's partials in index order
        part = wave_sum(part);
        if (lane == 0) x.partials[(int64_t)b * nblk_band + q] = part;
    }

    if (x.stamps) tstamp[4] = wall_clock64();
    // ---- k_prep's outputs: block q writes the records of sources [64 k, 64 k + 64), k = q, q + nblk_band, ...
    for (int64_t s = (int64_t)q * 64 + lane; s - lane < a.S; s += (int64_t)nblk_band * 64) {
        if (s < a.S) {
            SrcRec r;
            small_prep(a, x, bdr, cphi, b, s, r);
            prep_store(r, (int64_t)b * a.S + s, x.recs, x.boxes, x.kind, x.status);
        }
    }
    if (x.stamps && lane == 0) {
        tstamp[5] = wall_clock64();
        unsigned long long *o = x.stamps + (int64_t)blockIdx.x * 8;
        for (int k = 0; k < 6; k++) o[k] = tstamp[k];
        o[6] = (unsigned long long)nh;
        o[7] = (unsigned long long)__builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20) /* XCC_ID */;
    }
}
