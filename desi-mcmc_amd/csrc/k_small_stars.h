// k_small_stars.h -- a small star field's whole evaluation in ONE launch (BASELINE configs[1]: 1 000 stars x 5 bands x 512^2)
//
// Why.  On a frame of a few hundred render tiles the step of the general path is four launches -- k_prep, k_bin_direct,
// the render kernel, k_reduce -- each far too short to fill the chip (640 one-wave tiles on 2 048 wave slots) and each
// paying a dependent-launch boundary: at configs[1] the render kernel ran 37 us at 7 % of the HBM roof and was still
// less than half of the 75 us step.  For a catalogue of at most SMALL_MAX_S stars on at most STAR_TILES_MIN tiles this kernel
// does all four jobs:
//   * a block is one tile, its four waves each OWN eight of the tile's 32 columns (k_render_stars walks a tile's column parts
//     one behind the other in one wave; the unit of work -- one column of one star -- belongs to exactly one part, so
//     nothing is seeded twice, four times as many waves are in flight, and a pixel receives its terms from one wave in a
//     fixed order).  The waves share the scan (four stars per thread and pass instead of sixteen) and the staging of the
//     tile's hits; each walks and finishes its own columns without waiting for the others.  (Round 4 first ran the four
//     parts as four one-wave blocks, each scanning the whole catalogue for itself: 26.9 us against 25.3.  Narrower waves --
//     eight per tile, or 16-column blocks of four -- fill a walk step's 64 lanes worse, 56 % against 73 % of the lane-rows
//     at four against eight columns, and measured 29-33 us.)
//   * no lists: the block tests the band's stars against its rectangle itself -- equa2pixel, the overlap test and the int()
//     box are k_prep's own expressions (prep_pixel / prep_star_box / prep_window) -- and keeps the hits' indices in LDS, in
//     ascending order; a designated block per 256 sources also writes their records, boxes and status words, so everything
//     that reads k_prep's outputs after a render (cel_field_stats, cel_stamp_boxes, ...) finds them;
//   * no reduction launch: a block's Poisson partial is one double; the B x blocks-per-band partials ride back to the host in
//     the kernel's own stores to mapped host memory (5 KB at configs[1]) and the host adds each band's in index order (Kahan), so the per-band
//     log-likelihoods are reproducible bit for bit.  (A last-block-done sum inside the kernel -- write-through partial,
//     agent-scope counter add, the last arriver reads the band's partials with sc1 loads -- was built and measured first:
//     correct, and 20 us of a 60 us kernel: 512 blocks per band finish together and queue on one counter word.)
// Measured at configs[1] by timing-only ablations (tools/ab_small.sh): launch + tables 5 us, walk 5 us, epilogue 2 us; the
// scan was 28 us while every block ran k_prep's full arithmetic on every star, hence the cheap position filter below.
#pragma once
#include "k_render_stars.h"

#ifndef SMALL_NB
#define SMALL_NB 1            // blocks per tile (column parts of the tile)
#endif
#define SMALL_BW (HW_TW / SMALL_NB)     // columns per block
#ifndef SMALL_NWV
#define SMALL_NWV 4           // waves per block
#endif
#define SMALL_CW (SMALL_BW / SMALL_NWV) // columns per wave
static_assert(SMALL_NB * SMALL_NWV * SMALL_CW == HW_TW && 16 % SMALL_NWV == 0 && SMALL_CW <= 8, "k_small_stars: block shape");
#define SMALL_MAX_S 4096      // = BIN_DIRECT_MAX_S: the catalogue sizes the one-wave scans are meant for
#define SMALL_CAP 256         // candidate stars of one block (its columns x 64 pixels grown by the star radius) the kernel can hold; more: the host takes the general path
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
    unsigned long long *stamps;               // diagnostic (a -DSMALL_STAMPS build + CEL_SMALL_STAMPS): 6 wall-clock stamps (100 MHz) + XCC id per block, or nullptr
    int full_H, win_y0;
    int perm_mul, perm_add;                   // block -> tile shuffle (see the kernel)
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

// LDS traffic of ONE wave is in order (a wave's DS instructions execute in issue order), so a table one lane writes and
// another lane of the same wave reads needs no workgroup barrier -- only the compiler kept from moving the accesses
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// one batch of the block's hit list into the star table, sorted by the rows a star has on this tile (star_stage's rule:
// descending, ties by list position; a star without a row or a column here sorts last).  A hit's position and counts come
// from the scan's LDS tables, its box is k_prep's expressions on that position.  The block's first wave does the work (a
// batch is at most 64 stars); every wave takes the barriers.
__device__ __forceinline__ void small_stage(const RenderArgs &a, const SmallArgs &x, StarTab &ST, const double *__restrict__ cpx,
                                            const double *__restrict__ cpy, const double *__restrict__ ccn,
                                            int base, int nb, int tid, const BandDev &bd, int Xb, int Y0) {
    __syncthreads();                   // the previous batch has been read
    double2 pp = make_double2(0.0, 0.0);
    double sc = 0.0;
    int4 bx4 = make_int4(0, 0, 0, 0);
    int nrows = -1;
    if (tid < nb) {
        sc = ccn[base + tid];
        SrcRec r;
        r.x0 = r.x1 = r.y0 = r.y1 = 0;
        r.type = 0;
        const double px = cpx[base + tid], py = cpy[base + tid];
        prep_star_box(bd, px, py, x.full_H, a.W, r);
        prep_window(r, py, x.win_y0, a.H);
        pp = make_double2(px, r.py);
        bx4 = make_int4(r.x0, r.x1, r.y0, r.y1);
        nrows = max(min(bx4.w, Y0 + HW_TH) - max(bx4.z, Y0), 0);
        const int ncols = max(min(bx4.y, Xb + SMALL_BW) - max(bx4.x, Xb), 0);
        if (ncols == 0) nrows = 0;
    }
    int *srows = reinterpret_cast<int *>(ST.scale);      // scratch until the sorted table is written
    if (tid < 64) srows[tid] = nrows;
    __syncthreads();
    int rank = 0;
    if (tid < nb)
        for (int j = 0; j < nb; j++) {
            const int rj = srows[j];
            rank += (rj > nrows || (rj == nrows && j < tid)) ? 1 : 0;
        }
    __syncthreads();
    if (tid < nb) {
        ST.px[rank] = pp.x; ST.py[rank] = pp.y; ST.scale[rank] = sc;
        ST.box[rank] = bx4;
    }
    __syncthreads();
}

// star_walk (k_render_hw.h) for one WAVE of a multi-wave block: the wave adds the staged batch's columns [Xa, Xa + CW) into
// ITS accumulator (CW doubles per row), with its own task tables (cum: first task of sorted star j, own: the star of every
// task) and no workgroup barrier.  The PSF constants are read from the star table where a task is seeded (one LDS address
// for all lanes) instead of living in 42 VGPRs across the walk: five waves share a SIMD.
template <int CW>
__device__ __forceinline__ void small_walk(const StarTab &ST, const double *__restrict__ et, double *__restrict__ acc,
                                           int *__restrict__ cum, unsigned char *__restrict__ own, int nb, int lane, int Xa, int Y0) {
    int w = 0;
    if (lane < nb) {
        const int4 q = ST.box[lane];
        const int nr = max(min(q.w, Y0 + HW_TH) - max(q.z, Y0), 0);
        w = (nr > 0) ? max(min(q.y, Xa + CW) - max(q.x, Xa), 0) : 0;
    }
    int incl = w;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(incl, o);
        if (lane >= o) incl += up;
    }
    const int total = __builtin_amdgcn_readlane(incl, 63);
    if (total == 0) return;
    cum[lane] = incl - w;
#pragma unroll
    for (int cidx = 0; cidx < CW; cidx++)
        if (cidx < w) own[incl - w + cidx] = (unsigned char)lane;
    wave_lds_fence();
    const double ceq0 = ST.eq[0], ceq1 = ST.eq[1], ceq2 = ST.eq[2];
    for (int t0 = 0; t0 < total; t0 += 64) {
        const int t = t0 + lane;
        const bool valid = t < total;
        const int j = own[min(t, total - 1)];
        const double px = ST.px[j], py = ST.py[j];
        const int4 bx = ST.box[j];
        const int bx0 = max(bx.x, Xa);
        const int xi = bx0 + (t - cum[j]);
        const int ra = max(bx.z, Y0) - Y0;
        const int n = valid ? max(min(bx.w, Y0 + HW_TH) - Y0 - ra, 0) : 0;
        const double amp = valid ? ST.scale[j] : 0.0;
        const int nmax = __builtin_amdgcn_readlane(n, 0);      // sorted by rows: the step's first task has the most
        const double xx = (double)xi;
        const double y0 = (double)(Y0 + ra);
        double g[K_PSF], r[K_PSF];
#pragma unroll
        for (int k = 0; k < K_PSF; k++) {
            const double qb = ST.qb[k], qc = ST.qc[k];
            const double dx = xx - (px + ST.mux[k]), dy = y0 - (py + ST.muy[k]);
            const double hx = qb * dx + qc * dy;
            const double e = -0.5 * (ST.qa[k] * dx * dx + (qb * dx + hx) * dy);
            const double er = fmin(fmax(-(hx + 0.5 * qc), -REC_EMAX * EXP_SCALE), REC_EMAX * EXP_SCALE);
            g[k] = (ST.A0[k] * amp) * exp_tab64(e, et);
            r[k] = exp_tab64(er, et);
        }
        double *rowp = acc + ra * CW + (valid ? xi - Xa : 0);
        int i = 0;
        for (; i + 3 < nmax; i += 4, rowp += 4 * CW) {       // four rows per trip, as star_walk
#pragma clang fp contract(off)
            double g1[K_PSF], r1[K_PSF];
            const double s0 = (g[0] + g[1]) + g[2];
            g1[0] = g[0] * r[0]; r1[0] = r[0] * ceq0; g1[1] = g[1] * r[1]; r1[1] = r[1] * ceq1; g1[2] = g[2] * r[2]; r1[2] = r[2] * ceq2;
            const double s1 = (g1[0] + g1[1]) + g1[2];
            g[0] = g1[0] * r1[0]; r[0] = r1[0] * ceq0; g[1] = g1[1] * r1[1]; r[1] = r1[1] * ceq1; g[2] = g1[2] * r1[2]; r[2] = r1[2] * ceq2;
            const double s2 = (g[0] + g[1]) + g[2];
            g1[0] = g[0] * r[0]; r1[0] = r[0] * ceq0; g1[1] = g[1] * r[1]; r1[1] = r[1] * ceq1; g1[2] = g[2] * r[2]; r1[2] = r[2] * ceq2;
            const double s3 = (g1[0] + g1[1]) + g1[2];
            g[0] = g1[0] * r1[0]; r[0] = r1[0] * ceq0; g[1] = g1[1] * r1[1]; r[1] = r1[1] * ceq1; g[2] = g1[2] * r1[2]; r[2] = r1[2] * ceq2;
            if (i + 3 < n) {
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
            g[0] = g[0] * r[0]; r[0] = r[0] * ceq0; g[1] = g[1] * r[1]; r[1] = r[1] * ceq1; g[2] = g[2] * r[2]; r[2] = r[2] * ceq2;
        }
    }
    wave_lds_fence();
}

// ordered compaction over the block: the position of this thread's flagged element among all flagged elements of the block's
// SMALL_NWV waves (waves in order, lanes in order).  wcnt: SMALL_NWV ints of LDS.  Two barriers; returns the block's count.
__device__ __forceinline__ int block_rank(bool flag, int wave, int lane, int *__restrict__ wcnt, int &pos) {
    const unsigned long long m = __ballot(flag);
    if (lane == 0) wcnt[wave] = __popcll(m);
    __syncthreads();
    int before = 0, all = 0;
#pragma unroll
    for (int v = 0; v < SMALL_NWV; v++) {
        const int c = wcnt[v];
        before += (v < wave) ? c : 0;
        all += c;
    }
    pos = before + __popcll(m & ((1ull << lane) - 1ull));
    __syncthreads();
    return all;
}

#ifndef SMALL_WAVES
#define SMALL_WAVES 3
#endif
__global__ void __launch_bounds__(64 * SMALL_NWV) __attribute__((amdgpu_waves_per_eu(SMALL_WAVES, SMALL_WAVES)))
k_small_stars(RenderArgs a, SmallArgs x) {
    __shared__ double acc[SMALL_NWV][HW_TH * SMALL_CW];
    __shared__ StarTab ST;
    __shared__ double et[64];
    __shared__ double lt[128];
    __shared__ unsigned short hits[SMALL_CAP];
    __shared__ double cpx[SMALL_CAP], cpy[SMALL_CAP], ccn[SMALL_CAP];   // the candidates' pixel positions (full-frame rows) and counts
    __shared__ int cum[SMALL_NWV][64];
    __shared__ unsigned char own[SMALL_NWV][64 * SMALL_CW];
    __shared__ int wcnt[16];
    __shared__ double wpart[SMALL_NWV];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per_band = a.ntx * a.nty;
    const int nblk_band = per_band * SMALL_NB;
    const int b = blockIdx.x / nblk_band;
    const int q0 = blockIdx.x - b * nblk_band;
    // Which tile a block takes is shuffled per band (a multiplier coprime to the band's block count, an offset per band):
    // blocks i, i + 256, i + 512 of a launch tend to share a CU, and the same tile of every band holds the same stars --
    // unshuffled, a crowded tile's blocks sat on one CU in three bands at once and the launch ended 1.8 us later (25.2 against
    // 23.4 us; tools/small_placement.py shows who shares a CU).
    const int q = (int)(((long long)q0 * x.perm_mul + (long long)b * x.perm_add) % nblk_band);     // this block's tile among its band's
    const int t = q / SMALL_NB, p = q - t * SMALL_NB;
    const int ty = t / a.ntx, tx = t - ty * a.ntx;
    const int Y0 = ty * HW_TH;
    const int Xb = tx * HW_TW + p * SMALL_BW;           // the block's 16 columns
    const int Xa = Xb + wave * SMALL_CW;                // this wave's four
    const BandDev *bd = a.bands + b;
    const BandDev &bdr = *bd;
    const int S = (int)a.S;
#ifdef SMALL_STAMPS
    unsigned long long tstamp[6];
#define SMALL_STAMP(i) do { if (x.stamps) tstamp[i] = wall_clock64(); } while (0)
#else
#define SMALL_STAMP(i) do { } while (0)
#endif
    SMALL_STAMP(0);

    // ---- nearly everything this block reads from global memory is requested NOW -- the band's constants, the first 1024
    // stars' positions, the block's observed pixels -- in one round trip
    const double *sc = x.consts + (int64_t)b * SMALL_CONSTS;
    constexpr int NT = 64 * SMALL_NWV, NCL = (SMALL_CONSTS + NT - 1) / NT;
    double v_c[NCL];
#pragma unroll
    for (int k = 0; k < NCL; k++) v_c[k] = sc[min(k * NT + tid, SMALL_CONSTS - 1)];
    const double cphi = sc[21];
    constexpr int U = 16 / SMALL_NWV;           // stars per thread in flight: 1024 per pass
    double ra[U], de[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
        const int s = min(64 * SMALL_NWV * u + tid, S - 1);
        const double2 rd = *reinterpret_cast<const double2 *>(x.radec + 2 * (int64_t)s);
        ra[u] = rd.x; de[u] = rd.y;
    }
    const bool in_frame = (Xb + SMALL_BW <= a.W) && (Y0 + HW_TH <= a.H);
    const bool inside = in_frame && (a.flags & CEL_RENDER_LOGLIK);
    const bool store = !(a.flags & CEL_RENDER_NO_STORE);
    double ne[HW_TH * SMALL_CW / 64];
    if (inside && !(SMALL_ABL & 4)) stars_nelec<SMALL_CW, true>(a, b, Xa, Y0, lane, ne);

#pragma unroll
    for (int r = 0; r < HW_TH * SMALL_CW / 64; r++) acc[wave][r * 64 + lane] = 0.0;
#pragma unroll
    for (int k = 0; k < NCL; k++) {
        const int i = k * NT + tid;
        if (i < 21) (&ST.qa[0])[i] = v_c[k];    // star_setup's values (the host checked the one-segment condition for every band)
        else if (i >= 22 && i < 86) et[i - 22] = v_c[k];
        else if (i >= 86 && i < SMALL_CONSTS) lt[i - 86] = v_c[k];
    }
    const double eps = bd->eps;

    SMALL_STAMP(1);
    // ---- the band's stars against this block's rectangle.  Pass 1, every star: the pixel position only (six flops) against
    // the rectangle grown by the star radius + 3 -- the int() box reaches less than R + 2 from the position, so no star
    // whose box meets the rectangle is lost; the candidates' indices and positions go to LDS in ascending order.  Pass 2,
    // the candidates (a few dozen): k_prep's exact box against the rectangle.
    int nh = 0;
    {
        const double grow = bdr.R + 3.0;
        const double xlo = (double)Xb - grow, xhi = (double)(Xb + SMALL_BW) + grow;
        const double ylo = (double)(Y0 + x.win_y0) - grow, yhi = (double)(Y0 + x.win_y0 + HW_TH) + grow;
        int nc = 0;
        for (int s0 = 0; s0 < S && !(SMALL_ABL & 1); s0 += 64 * SMALL_NWV * U) {
            // the flags of all U rounds first, then ONE ordered compaction over (round, wave, lane) = ascending star index
            double px[U], py[U];
            unsigned long long m[U];
            bool cand[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int s = s0 + 64 * SMALL_NWV * u + tid;
                prep_pixel(bdr, ra[u], de[u], cphi, px[u], py[u]);
                cand[u] = (s < S) && (px[u] > xlo) && (px[u] < xhi) && (py[u] > ylo) && (py[u] < yhi);
                m[u] = __ballot(cand[u]);
                if (lane == 0) wcnt[u * SMALL_NWV + wave] = __popcll(m[u]);
            }
            __syncthreads();
            int before[U], all = 0;
#pragma unroll
            for (int u = 0; u < U; u++) before[u] = 0;
#pragma unroll
            for (int e = 0; e < U * SMALL_NWV; e++) {
                const int c = wcnt[e];
#pragma unroll
                for (int u = 0; u < U; u++) before[u] += (e < u * SMALL_NWV + wave) ? c : 0;
                all += c;
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int at = nc + before[u] + __popcll(m[u] & ((1ull << lane) - 1ull));
                if (cand[u] && at < SMALL_CAP) { hits[at] = (unsigned short)(s0 + 64 * SMALL_NWV * u + tid); cpx[at] = px[u]; cpy[at] = py[u]; }
            }
            nc += all;
            __syncthreads();                    // wcnt is written again
            if (s0 + 64 * SMALL_NWV * U < S) {  // a catalogue of more than 1024 stars: the next 1024
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const int s = min(s0 + 64 * SMALL_NWV * (U + u) + tid, S - 1);
                    const double2 rd = *reinterpret_cast<const double2 *>(x.radec + 2 * (int64_t)s);
                    ra[u] = rd.x; de[u] = rd.y;
                }
            }
        }
        if (nc > SMALL_CAP) {                   // the host renders this call again on the general path
            if (tid == 0) *x.flag = x.stamp;
            nc = SMALL_CAP;
        }
        __syncthreads();
        // exact test, one candidate per thread and trip; its counts are requested first and arrive under the box arithmetic;
        // the tables are compacted in place (a trip's writes stay inside what its threads have already read)
        for (int c0 = 0; c0 < nc; c0 += 64 * SMALL_NWV) {
            const int i = c0 + tid, ic = min(i, nc - 1);
            const unsigned short sh = hits[ic];
            const double cn = x.counts[(int64_t)sh * a.B + b];
            const double px = cpx[ic], py = cpy[ic];
            SrcRec r;
            r.x0 = r.x1 = r.y0 = r.y1 = 0;
            r.type = 0;
            prep_star_box(bdr, px, py, x.full_H, a.W, r);
            prep_window(r, py, x.win_y0, a.H);
            const bool hit = (i < nc) && r.type >= 0 && box_hits(make_int4(r.x0, r.x1, r.y0, r.y1), Xb, Xb + SMALL_BW, Y0, Y0 + HW_TH);
            int at;
            const int all = block_rank(hit, wave, lane, wcnt, at);
            if (hit) { hits[nh + at] = sh; cpx[nh + at] = px; cpy[nh + at] = py; ccn[nh + at] = cn; }
            nh += all;
        }
    }

    // ---- the block's stars into its waves' accumulators
    SMALL_STAMP(2);
#ifdef SMALL_STAMPS
    unsigned long long stage_ticks = 0;
#endif
    for (int base = 0; base < nh; base += 64) {
        const int nb = min(64, nh - base);
#ifdef SMALL_STAMPS
        const unsigned long long ts0 = wall_clock64();
#endif
        small_stage(a, x, ST, cpx, cpy, ccn, base, nb, tid, bdr, Xb, Y0);
#ifdef SMALL_STAMPS
        stage_ticks += wall_clock64() - ts0;
#endif
        if (!(SMALL_ABL & 2)) small_walk<SMALL_CW>(ST, et, acc[wave], cum[wave], own[wave], nb, lane, Xa, Y0);
    }

    // ---- epilogue, every wave its own columns: lambda = eps + acc written once, the Poisson terms
    SMALL_STAMP(3);
    double part = 0.0;
    const double *wacc = acc[wave];
    if (SMALL_ABL & 4) {
        part = wacc[lane];
    } else if (inside) {
        part = store ? stars_epilogue<SMALL_CW, true, true>(a, wacc, lt, eps, b, Xa, Y0, lane, ne)
                     : stars_epilogue<SMALL_CW, true, false>(a, wacc, lt, eps, b, Xa, Y0, lane, ne);
    } else if (in_frame) {                      // model images only: stores, none of them under a condition
        if (store) {
            constexpr int RPI = 64 / SMALL_CW;
            const int64_t base = (int64_t)b * a.H * a.W + (int64_t)(Y0 + lane / SMALL_CW) * a.W + Xa + lane % SMALL_CW;
#pragma unroll
            for (int r = 0; r < HW_TH / RPI; r++) a.lambda[base + (int64_t)(RPI * r) * a.W] = eps + wacc[r * 64 + lane];
        }
    } else {
        stars_nelec<SMALL_CW, false>(a, b, Xa, Y0, lane, ne);
        part = store ? stars_epilogue<SMALL_CW, false, true>(a, wacc, lt, eps, b, Xa, Y0, lane, ne)
                     : stars_epilogue<SMALL_CW, false, false>(a, wacc, lt, eps, b, Xa, Y0, lane, ne);
    }
    if (a.flags & CEL_RENDER_LOGLIK) {          // the block's Poisson partial: its waves' in order; the host adds a band's partials in index order
        part = wave_sum(part);
        if (lane == 0) wpart[wave] = part;
        __syncthreads();
        if (tid == 0) {
            double s = wpart[0];
#pragma unroll
            for (int v = 1; v < SMALL_NWV; v++) s += wpart[v];
            x.partials[(int64_t)b * nblk_band + q] = s;
        }
    }

    SMALL_STAMP(4);
    // ---- k_prep's outputs: block q writes the records of sources [256 k, 256 k + 256), k = q, q + nblk_band, ...
    for (int64_t s = (int64_t)q * NT + tid; s - tid < a.S; s += (int64_t)nblk_band * NT) {
        if (s < a.S) {
            SrcRec r;
            small_prep(a, x, bdr, cphi, b, s, r);
            prep_store(r, (int64_t)b * a.S + s, x.recs, x.boxes, x.kind, x.status);
        }
    }
#ifdef SMALL_STAMPS
    if (x.stamps && tid == 0) {
        tstamp[5] = wall_clock64();
        unsigned long long *o = x.stamps + (int64_t)blockIdx.x * 8;
        for (int k = 0; k < 6; k++) o[k] = tstamp[k];
        o[6] = (unsigned long long)nh | (stage_ticks << 32);
        o[7] = (unsigned long long)__builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20) /* XCC_ID */ |
               ((unsigned long long)__builtin_amdgcn_s_getreg(((32 - 1) << 11) | (0 << 6) | 4) /* HW_ID */ << 8);
    }
#endif
}
