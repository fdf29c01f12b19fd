// k_split.h -- the per-pixel multinomial split of observed photons among overlapping sources
//
// Gibbs step of CelestePy's sampler: sample_source_counts / sample_multinomial
// (CelestePy/celeste_sample_sources.pyx:61-156), driven by
// sample_source_photons_single_image_cython (celeste_mcmc.py:98-150) and Field.resample_photons
// (models.py:123-160).  For every pixel the nelec observed photons are split among the sources
// whose patch contains the pixel and the sky, z ~ Multinomial(nelec; F_1, ..., F_k, eps), by
// the conditional-binomial method in source order with the sky last (:139-155).
//
// Reference semantics kept on purpose:
//   * a source takes part at a pixel only if the pixel is STRICTLY inside its box in x0 and y0
//     (`x > x0 and x < x1 and y > y0 and y < y1`, :50-51): the first row and column of every
//     sample patch stay 0;
//   * photons of a pixel nobody covers all go to the noise sum (:91-93).
// The reference draws binomials from randomkit's MT19937 stream (deps/randomkit); a GPU cannot
// reproduce that stream, so parity is statistical: the draws here come from a
// counter-based Philox4x32-10 generator keyed by (seed; band, pixel, source), so a result
// depends only on the seed and the inputs -- not on tiling, launch order or GPU count.
// Binomial variates: inversion for n*min(p,1-p) <= 30, BTPE (Kachitvichyanukul & Schmeiser 1988)
// above, the same split randomkit makes (distributions.c:422-455).
//
// One wave per render tile (the tile lists of k_bin_*).  Pass A accumulates every pixel's total
// rate in LDS, pass B walks the sources again in order and draws.  Stamps use the direct
// evaluator: exact, every component.
#pragma once
// timing-only ablations of the split (1 = no draws, 2 = every draw 1, 4 = no stamp walk) exist only in a
// -DCEL_ABLATE build (tools/ablate_render.py); the shipped kernel has no such switch
#ifdef CEL_ABLATE
#define SPLIT_ABLATE(a) ((a).debug)
#else
#define SPLIT_ABLATE(a) 0
#endif
#include "hw_source.h"

#ifndef PHILOX_ROUNDS
#define PHILOX_ROUNDS 10      // Philox4x32-10 (Salmon et al. 2011); lower values only for timing experiments
#endif
struct Philox {
    unsigned k0, k1, c0, c1, c2, c3;
    unsigned out[4];
    int have;
};

__device__ inline void philox_block(Philox &g) {
    unsigned c0 = g.c0, c1 = g.c1, c2 = g.c2, c3 = g.c3, k0 = g.k0, k1 = g.k1;
#pragma unroll
    for (int r = 0; r < PHILOX_ROUNDS; r++) {
        unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
        unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
        unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1;
        unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    g.out[0] = c0; g.out[1] = c1; g.out[2] = c2; g.out[3] = c3;
    g.have = 4;
    g.c0 += 1;      // next block of this stream
}

__device__ inline Philox philox_init(unsigned long long seed, unsigned long long pixel, unsigned stream) {
    Philox g;
    g.k0 = (unsigned)seed; g.k1 = (unsigned)(seed >> 32);
    g.c0 = 0; g.c1 = stream; g.c2 = (unsigned)pixel; g.c3 = (unsigned)(pixel >> 32);
    g.have = 0;
    return g;
}

// uniform double in [0, 1) with 53 random bits
__device__ inline double philox_double(Philox &g) {
    if (g.have < 2) philox_block(g);
    // words (3, 2) first, then (1, 0); selected, not indexed: a dynamically indexed register array
    // would live in scratch memory
    const bool first = (g.have == 4);
    unsigned a = (first ? g.out[3] : g.out[1]) >> 5, b = (first ? g.out[2] : g.out[0]) >> 6;
    g.have -= 2;
    return ((double)a * 67108864.0 + (double)b) * (1.0 / 9007199254740992.0);
}

// Stirling-series tail used by BTPE's final acceptance test (btpe_st_r below).  The leading constant is randomkit's 13680
// (deps/randomkit/distributions.c:362-365, the sampler the reference's split calls); Kachitvichyanukul & Schmeiser print
// 13860 = 166320/12.  Kept as the reference's dependency has it: the term is a 1e-3 relative change of a 1/(12 x)
// correction inside an accept/reject bound, far below anything the pmf tests resolve.

// 1/x to ~4e-15 relative for x of fp32 range: v_rcp_f32 (1 ulp) + one Newton step in fp64 (5 instructions with the two
// conversions, against ~15 for an IEEE fp64 division).  Used where a probability or a pmf ratio is formed.
__device__ inline double fast_rcp(double x) {
    double y = (double)rcp_f32((float)x);
    return fma(y, fma(-x, y, 1.0), y);
}

// Binomial(n, r) for r <= 1/2, n r <= 30: sequential inversion (BINV).
// P(X = 0) = (1 - r)^n >= 1 - n r, so a uniform at or below 1 - n r returns 0 before anything
// transcendental is evaluated (most draws of a split: a pixel in a source's tail).  Otherwise
// (1 - r)^n comes from the table log / table exp of k_render.h (et: 64 doubles, lt: 128 doubles in
// LDS): the exponent n log(1 - r) >= -42 carries an absolute error of ~n * 1e-16.
__device__ inline long long binom_inversion(long long n, double r, double U /* the first uniform */, Philox &g,
                                            const double *__restrict__ et, const double *__restrict__ lt) {
    const double q = 1.0 - r;
    const double nd = (double)n;
    const double np = nd * r;
    if (U <= 1.0 - np) return 0;
    const double qn = exp_tab64(nd * log_tab(q, lt) * EXP_SCALE, et);
    if (U <= qn) return 0;
    // p(X) = p(X-1) (n - X + 1)/X r/q: one division for r/q, 1/X from an fp32 reciprocal and
    // one Newton step (4e-15; the recurrence needs no more)
    const double s = r / q;
    const double bound = fmin(nd, np + 10.0 * sqrt(np * q + 1.0));
    double X = 0.0, px = qn;
    while (U > px) {
        X += 1.0;
        if (X > bound) {            // numerical tail: start over
            X = 0.0; px = qn; U = philox_double(g);
        } else {
            U -= px;
            px *= (nd + 1.0 - X) * s * fast_rcp(X);
        }
    }
    return (long long)X;
}

// Binomial(n, r) for r <= 1/2, n r > 30: BTPE (triangle / parallelogram / exponential tails).
// Round 6: no IEEE fp64 division and no library log in it -- quotients by fast_rcp (4e-15 relative: fp32 reciprocal + a Newton
// step, 4 instructions against ~15), logarithms by the table log of k_render.h (the sampler's LDS table, to the last bits); the
// square root that sizes the triangle stays exact (the envelope's constants assume that p1).  These touch the algorithm at accept/reject
// boundaries only, 1e-14 wide; the pmf tests (chi^2 against the exact pmf, moments) are unchanged.  A trip of 64 queued draws
// holding one BTPE lane costs every lane this code: ~1 500 instructions before, ~600 now (tools/ablate_split.py).
__device__ inline double btpe_st_r(double x, double rx /* ~1/x */) {
    const double r2 = rx * rx;
    return (13680.0 - (462.0 - (132.0 - (99.0 - 140.0 * r2) * r2) * r2) * r2) * rx * (1.0 / 166320.0);
}
__device__ inline long long binom_btpe(long long n, double r, Philox &g, const double *__restrict__ lt) {
    const double q = 1.0 - r, nd = (double)n;
    const double nrq = nd * r * q;
    const double fm = nd * r + r;
    const long long m = (long long)floor(fm);
    const double md = (double)m;
    const double p1 = floor(2.195 * sqrt(nrq) - 4.6 * q) + 0.5;
    const double xm = md + 0.5, xl = xm - p1, xr = xm + p1;
    const double c = 0.134 + 20.5 * fast_rcp(15.3 + md);
    double a = (fm - xl) * fast_rcp(fm - xl * r);
    const double laml = a * (1.0 + a * 0.5);
    a = (xr - fm) * fast_rcp(xr * q);
    const double lamr = a * (1.0 + a * 0.5);
    const double rc = fast_rcp(c), rp1 = fast_rcp(p1), rlaml = fast_rcp(laml), rlamr = fast_rcp(lamr), rnrq = fast_rcp(nrq);
    const double p2 = p1 * (1.0 + 2.0 * c), p3 = p2 + c * rlaml, p4 = p3 + c * rlamr;
    const double s = r * fast_rcp(q), aa = s * (nd + 1.0);
    long long y;
    for (int guard = 0; guard < 100000; guard++) {
        double u = philox_double(g) * p4, v = philox_double(g);
        if (u <= p1) {                                   // triangular centre: accept at once
            y = (long long)floor(xm - p1 * v + u);
            return y;
        }
        if (u <= p2) {                                   // parallelograms
            double x = xl + (u - p1) * rc;
            v = v * c + 1.0 - fabs(md - x + 0.5) * rp1;
            if (v > 1.0) continue;
            y = (long long)floor(x);
        } else if (u <= p3) {                            // left exponential tail
            y = (long long)floor(xl + log_tab(v, lt) * rlaml);
            if (y < 0) continue;
            v = v * (u - p2) * laml;
        } else {                                         // right exponential tail
            y = (long long)floor(xr - log_tab(v, lt) * rlamr);
            if (y > n) continue;
            v = v * (u - p3) * lamr;
        }
        const double yd = (double)y;
        const double k = fabs(yd - md);
        if (k <= 20.0 || k >= nrq * 0.5 - 1.0) {
            // evaluate f(y)/f(m) by the recurrence: the product of (aa / i - s) over the integers between m and y, numerator
            // and denominator kept apart so that the loop holds no reciprocal
            double num = 1.0, den = 1.0;
            if (m < y) { for (long long i = m + 1; i <= y; i++) { const double di = (double)i; num *= (aa - s * di); den *= di; } if (v * den > num) continue; }
            else if (m > y) { for (long long i = y + 1; i <= m; i++) { const double di = (double)i; num *= (aa - s * di); den *= di; } if (v * num > den) continue; }
            else if (v > 1.0) continue;
            return y;
        }
        // squeezes, then the Stirling bound
        const double knrq = k * rnrq;
        const double rho = knrq * ((k * (k * (1.0 / 3.0) + 0.625) + 0.16666666666666666) * rnrq + 0.5);
        const double t = -0.5 * k * knrq;
        const double A = log_tab(v, lt);
        if (A < t - rho) return y;
        if (A > t + rho) continue;
        const double x1 = yd + 1.0, f1 = md + 1.0, z = nd + 1.0 - md, w = nd - yd + 1.0;
        const double rx1 = fast_rcp(x1), rf1 = fast_rcp(f1), rz = fast_rcp(z), rw = fast_rcp(w);
        const double bound = xm * log_tab(f1 * rx1, lt) + (nd - md + 0.5) * log_tab(z * rw, lt) +
                             (yd - md) * log_tab(w * s * rx1, lt) +
                             btpe_st_r(f1, rf1) + btpe_st_r(z, rz) + btpe_st_r(x1, rx1) + btpe_st_r(w, rw);
        if (A > bound) continue;
        return y;
    }
    return m;   // unreachable in practice (acceptance > 0.8 per trial); keeps the loop bounded
}

#ifndef BINV_MAX_NP
#define BINV_MAX_NP 30.0
#endif
// `a`: the inversion's first uniform is drawn from [a, 1) instead of [0, 1) -- the split's callers have already decided, on
// the top 32 bits of that uniform, that it is not below a (split_first_word below); 0 everywhere else
__device__ inline long long binomial_draw(long long n, double p, double a, Philox &g, const double *__restrict__ et,
                                          const double *__restrict__ lt) {
    if (n <= 0 || !(p > 0.0)) return 0;
    if (p >= 1.0) return n;
    const bool flip = p > 0.5;
    const double r = flip ? 1.0 - p : p;
    long long y;
    if (r * (double)n <= BINV_MAX_NP) {
        const double u = philox_double(g);
        y = binom_inversion(n, r, a + u * (1.0 - a), g, et, lt);
    } else {
        y = binom_btpe(n, r, g, lt);
    }
    return flip ? n - y : y;
}
__device__ inline long long binomial_draw(long long n, double p, Philox &g, const double *__restrict__ et,
                                          const double *__restrict__ lt) {
    return binomial_draw(n, p, 0.0, g, et, lt);
}

// ---- the split's first decision on a shared 16-bit word -----------------------------------------------------------------
// Most of a split's binomials are 0 and are decided by the sampler's first test, U <= 1 - n p (a pixel in a source's tail):
// that test needs no 53-bit uniform of its own.  Let V be the top 16 bits of U: if V + 1 <= floor((1 - n p) 2^16) =: tf the
// test holds whatever the other bits are.  So ONE Philox block (128 bits), keyed by (seed; band, column, the pixel's row with
// bits 1, 2 and 3 cleared; source), serves the EIGHT pixels y, y + 2, ..., y + 14 of a column -- the eight rows a lane of
// k_photon_split_hw takes in consecutive steps -- half-word (y >> 1) & 7 each (round 4: 32-bit words, four rows per block;
// round 6: half the blocks).  A pixel that does not pass (V >= tf) goes to the sampler proper with its own stream (seed; band,
// pixel; source), whose first uniform is drawn from [tf 2^-16, 1): the law of U given V >= tf.  Together: P(first test
// holds) = tf 2^-16 + (1 - n p - tf 2^-16) = 1 - n p, exactly the sampler's; the coarser word only sends 2^-16 more of the
// pixels to the sampler (6 000 of 4e8 at config 3).
#define SPLIT_GROUP_BIT (1ull << 63)      // keys of the shared blocks: never a pixel's own stream (pixel indices stay below 2^63)
#define SPLIT_GROUP_MASK ((int64_t)14)    // the row bits a group shares
#define SPLIT_WORD_SCALE 65536.0
__device__ inline double split_tf(long long n, double pr) {         // floor((1 - n p) 2^16): V < tf passes; <= 0: nobody does
    return floor((1.0 - (double)n * pr) * SPLIT_WORD_SCALE);
}
__device__ inline unsigned philox_word(const Philox &g, int w /* 0..7 */) {
    // (shifts, not a selection among the four words: the compiler turns a chain of selects on an index into a table in scratch)
    const unsigned long long lo = ((unsigned long long)g.out[1] << 32) | g.out[0], hi = ((unsigned long long)g.out[3] << 32) | g.out[2];
    return (unsigned)(((w & 4) ? hi : lo) >> ((w & 3) * 16)) & 0xffffu;
}
// the whole draw for one (pixel, source): what k_photon_split_hw does in two passes
__device__ inline long long split_draw(long long n, double pr, unsigned long long seed, unsigned long long group_key, int word,
                                       unsigned long long pixel_key, unsigned s, const double *__restrict__ et,
                                       const double *__restrict__ lt) {
    if (n <= 0 || !(pr > 0.0)) return 0;
    double a = 0.0;
    if (pr <= 0.5) {
        const double tf = split_tf(n, pr);
        Philox h = philox_init(seed, group_key | SPLIT_GROUP_BIT, s);
        philox_block(h);
        if ((double)philox_word(h, word) < tf) return 0;
        a = fmax(tf, 0.0) * (1.0 / SPLIT_WORD_SCALE);
    }
    Philox g = philox_init(seed, pixel_key, s);
    return binomial_draw(n, pr, a, g, et, lt);
}

// ---- device-resident sample patches ------------------------------------------------------------
// The split's output is consumed on the device by the per-source conditional likelihoods
// (k_patch_ll), so the patches need never cross PCIe (3.2 GB at config 3).  k_samp_layout turns
// the (band, source) records into source-major patch boxes and offsets (an exclusive scan of the
// box areas) in ONE block; k_patch_sums reduces every patch to its photon count (what the flux
// Gibbs step conditions on, sources.py:327-345).
// inclusive prefix sum over the 1024 threads of a block (+ the block's total, to every thread): a shuffle scan inside
// every wave, the 16 wave totals through LDS -- two barriers; the Hillis-Steele scan over LDS it replaces took twenty
__device__ __forceinline__ long long block_scan_1024(long long v, long long *__restrict__ wtot /* 16 */, long long &total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const long long up = __shfl_up(v, o);
        if (lane >= o) v += up;
    }
    if (lane == 63) wtot[w] = v;
    __syncthreads();
    long long before = 0, all = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const long long t = wtot[j];
        before += (j < w) ? t : 0;
        all += t;
    }
    __syncthreads();                  // wtot is rewritten by the next call
    total = all;
    return v + before;
}

// Three launches instead of one block walking the table: every block of 1024 scans its own entries (k_*_layout: local
// exclusive prefix + the block's total), one block scans the totals (k_scan_totals), every block adds its offset
// (k_scan_apply).  The single block took 0.15 + 0.11 ms per sweep for the two tables, all of it load latency in sequence.
__global__ void __launch_bounds__(1024)
k_samp_layout(const SrcRec *__restrict__ recs, int64_t S, int B, int4 *__restrict__ sbox /* S*B: x0,x1,y0,y1 */,
              int64_t *__restrict__ soff /* S*B + 1: the block-local exclusive prefix (k_scan_apply finishes it) */,
              long long *__restrict__ btot /* per block: its total */) {
    __shared__ long long wtot[16];
    const int64_t n = S * B;
    const int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x;       // i = s*B + b
    int4 bx = make_int4(0, 0, 0, 0);
    long long area = 0;
    if (i < n) {
        const int64_t s = i / B;
        const int b = (int)(i - s * B);
        const SrcRec &r = recs[(int64_t)b * S + s];
        if (r.type >= 0) { bx = make_int4(r.x0, r.x1, r.y0, r.y1); area = (long long)(r.x1 - r.x0) * (r.y1 - r.y0); }
    }
    long long total;
    const long long incl = block_scan_1024(area, wtot, total);
    if (i < n) { sbox[i] = bx; soff[i] = incl - area; }
    if (threadIdx.x == 0) btot[blockIdx.x] = total;
}

// exclusive scan of up to 1024 * 1024 block totals in place (one block; chunks of 1024 with a carry), grand total -> *out_total
__global__ void __launch_bounds__(1024)
k_scan_totals(long long *__restrict__ btot, int nb, int64_t *__restrict__ out_total) {
    __shared__ long long wtot[16];
    long long carry = 0;
    for (int base = 0; base < nb; base += 1024) {
        const int j = base + threadIdx.x;
        const long long v = (j < nb) ? btot[j] : 0;
        long long total;
        const long long incl = block_scan_1024(v, wtot, total);
        if (j < nb) btot[j] = carry + incl - v;
        carry += total;
    }
    if (threadIdx.x == 0) *out_total = carry;
}

__global__ void __launch_bounds__(1024)
k_scan_apply(int64_t *__restrict__ off, int64_t n, const long long *__restrict__ btot) {
    const int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    if (i < n) off[i] += btot[blockIdx.x];
}

template <typename TS>
__global__ void __launch_bounds__(256)
k_patch_sums(const int64_t *__restrict__ soff, const TS *__restrict__ samp, double *__restrict__ sums) {
    __shared__ double red[256];
    const int64_t i = blockIdx.x;
    const int64_t lo = soff[i], hi = soff[i + 1];
    double a = 0.0;
    for (int64_t k = lo + threadIdx.x; k < hi; k += 256) a += (double)samp[k];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) sums[i] = red[0];
}

// Before a resident split on the recurrence kernel: that kernel writes every pixel strictly inside
// a box exactly once, so only the first row and column of each patch need zeroing (not the whole
// 3.2 GB buffer), and the photon rectangles it will reduce with atomics need their identity.
#define SAMP_PREP_PER_WAVE 4      // patches per wave (one wave per patch was 50 000 blocks of a few stores: launch-bound)
template <typename TS>
__global__ void __launch_bounds__(256)
k_samp_prepare(const int4 *__restrict__ sbox, const int64_t *__restrict__ soff, TS *__restrict__ samp,
               int4 *__restrict__ nz, int64_t n) {
    const int lane = threadIdx.x & 63;
    const int64_t first = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * SAMP_PREP_PER_WAVE;
    for (int64_t i = first; i < first + SAMP_PREP_PER_WAVE && i < n; i++) {
        const int4 bx = sbox[i];
        const int nx = bx.y - bx.x, ny = bx.w - bx.z;
        if (lane == 0) nz[i] = make_int4(INT_MAX, 0, INT_MAX, 0);      // x0, x1, y0, y1: empty
        if (nx <= 0 || ny <= 0) continue;
        TS *p = samp + soff[i];
        for (int x = lane; x < nx; x += 64) p[x] = (TS)0;
        for (int y = lane; y < ny; y += 64) p[(int64_t)y * nx] = (TS)0;
    }
}

// ---- photon lists: the pixels of a sample patch that hold a photon ------------------------------------
// The conditional likelihood of a source (k_patch_ll_hw, mode 0) is sum z log m over its patch: only pixels with
// z > 0 contribute, and at config 3 they are a fifth of the rectangle that holds them (median fill: galaxies 0.21,
// stars 0.29; tools/nz_fill.py).  So every patch also gets a compact list (x | y << 16, z) of its photons, in
// row-major order (one wave per patch: the order, and with it every sum taken over the list, depends on the data
// only), and the likelihood kernel evaluates a patch either at its photons (direct exponentials, every lane a
// photon) or densely by the column recurrence, whichever the layout pass estimated cheaper:
//     at the photons   ceil(nnz / 64) steps x K components x ~17 instructions
//     densely          chunks of 32 x 64 pixels x (set-up + K x (seeds + rows)): ~1 900 (star) / ~6 800 (galaxy) each
__global__ void __launch_bounds__(1024)
k_nz_layout(const int *__restrict__ nnz, const int4 *__restrict__ nzbox, const int *__restrict__ type /* per source */,
            int64_t S, int B, int force /* 0 = estimate, 1 = every patch at its photons, 2 = every patch densely */,
            double bias /* force 0: at the photons unless that is estimated more than `bias` times the dense cost */,
            int64_t *__restrict__ loff /* S*B + 1: list offsets -- block-local here, finished by k_scan_totals / k_scan_apply */,
            int *__restrict__ mode /* S*B: 1 = evaluate at the photons */, long long *__restrict__ btot) {
    __shared__ long long wtot[16];
    const int64_t n = S * B;
    const int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    long long cnt = 0;
    int md = 0;
    if (i < n) {
        cnt = nnz[i];
        const int4 q = nzbox[i];
        const int K = (type[i / B] == 0) ? K_PSF : K_GAL;
        const long long chunks = (cnt > 0) ? (long long)((q.y - q.x + HW_TW - 1) / HW_TW) * ((q.w - q.z + HW_TH - 1) / HW_TH) : 0;
        const long long sparse_cost = ((cnt + 63) / 64) * K * 17 + 150;
        const long long dense_cost = chunks * ((K == K_PSF) ? 1900 : 6800);
        md = (force == 1) ? 1 : (force == 2) ? 0 : ((double)sparse_cost < bias * (double)dense_cost ? 1 : 0);
    }
    long long total;
    const long long incl = block_scan_1024(cnt, wtot, total);
    if (i < n) { mode[i] = md; loff[i] = incl - cnt; }
    if (threadIdx.x == 0) btot[blockIdx.x] = total;
}

__global__ void __launch_bounds__(64)
k_nz_compact(const int4 *__restrict__ sbox, const int64_t *__restrict__ soff, const int *__restrict__ samp,
             const int4 *__restrict__ nzbox, const int64_t *__restrict__ loff, NzEntry *__restrict__ list) {
    const int64_t i = blockIdx.x;
    const int4 bx = sbox[i], q = nzbox[i];
    if (!(q.y > q.x && q.w > q.z)) return;
    const int lane = threadIdx.x;
    const int nx = bx.y - bx.x;
    const int *p = samp + soff[i];
    NzEntry *out = list + loff[i];
    const int64_t cap = loff[i + 1] - loff[i];
    int64_t pos = 0;
    // Steps in row-major order, NZC_FLIGHT at a time: their loads are issued together (unconditional, clamped addresses) instead
    // of one memory round trip per step, and taken in order afterwards.  A step is one 64-column chunk of one row -- or, for a
    // rectangle of at most 32 (16) columns, two (four) whole rows: lane = sub-row * cw + column, so the lanes of a step are in
    // row-major order too and the list's order (with it every sum taken over the list) is the same as with one row per step.
    // Round 6: a photon rectangle is ~25 columns wide on average; with one row per step 60 % of a step's lanes had no pixel and
    // a wave kept 1 KB in flight: 0.43 ms for 1.04 GB = 2.4 TB/s.
#define NZC_FLIGHT 6
    const int w = q.y - q.x, h = q.w - q.z;
    const int rp = (w <= 16) ? 4 : (w <= 32) ? 2 : 1;          // rows per step
    const int cw = 64 / rp;                                    // columns per sub-row
    const int nch = (w + cw - 1) / cw;                         // column chunks per row (1 unless rp == 1)
    const int nsteps = ((h + rp - 1) / rp) * nch;
    const int sub = lane / cw, cl = lane - sub * cw;
    for (int u0 = 0; u0 < nsteps; u0 += NZC_FLIGHT) {
        int z[NZC_FLIGHT], xs[NZC_FLIGHT], ys[NZC_FLIGHT];
#pragma unroll
        for (int j = 0; j < NZC_FLIGHT; j++) {
            const int u = min(u0 + j, nsteps - 1);
            const int r = u / nch, ch = u - r * nch;
            ys[j] = q.z + r * rp + sub;
            xs[j] = q.x + cw * ch + cl;
            z[j] = p[(int64_t)(min(ys[j], q.w - 1) - bx.z) * nx - bx.x + min(xs[j], q.y - 1)];
        }
#pragma unroll
        for (int j = 0; j < NZC_FLIGHT; j++) {
            const int zz = (u0 + j < nsteps && xs[j] < q.y && ys[j] < q.w) ? z[j] : 0;
            const unsigned long long m = __ballot(zz != 0);
            if (zz != 0) {
                const int64_t k = pos + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
                if (k < cap) out[k] = NzEntry{xs[j] | (ys[j] << 16), zz};
            }
            pos += __popcll(m);
        }
    }
}

// diagnostic: N independent Binomial(n, p) draws (stream i), for the sampler's own tests
__global__ void __launch_bounds__(256)
k_binomial_draws(long long n, double p, unsigned long long seed, int64_t N, long long *__restrict__ out) {
    __shared__ double et[64], lt[128];
    if (threadIdx.x < 64) {
        et[threadIdx.x] = exp2((double)threadIdx.x * (1.0 / 64.0));
        lt[threadIdx.x] = c_log_ic[threadIdx.x];
        lt[64 + threadIdx.x] = c_log_lc[threadIdx.x];
    }
    __syncthreads();
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    Philox g = philox_init(seed, (unsigned long long)i, 0u);
    out[i] = binomial_draw(n, p, g, et, lt);
}

// smallest / largest value of the observed image (and whether it holds a NaN), per block: the host finishes the
// reduction.  Decides the photon split's instantiation (TL above).
__global__ void __launch_bounds__(256)
k_nelec_range(const double *__restrict__ x, int64_t n, double *__restrict__ out /* 3 per block: min, max, nan count */) {
    __shared__ double slo[256], shi[256], snan[256];
    double lo = INFINITY, hi = -INFINITY, bad = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double v = x[i];
        if (v == v) { lo = fmin(lo, v); hi = fmax(hi, v); } else bad += 1.0;
    }
    slo[threadIdx.x] = lo; shi[threadIdx.x] = hi; snan[threadIdx.x] = bad;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            slo[threadIdx.x] = fmin(slo[threadIdx.x], slo[threadIdx.x + o]);
            shi[threadIdx.x] = fmax(shi[threadIdx.x], shi[threadIdx.x + o]);
            snan[threadIdx.x] += snan[threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[3 * blockIdx.x] = slo[0]; out[3 * blockIdx.x + 1] = shi[0]; out[3 * blockIdx.x + 2] = snan[0]; }
}

// int(num_photons_xy) of the reference (celeste_sample_sources.pyx:105), without the undefined
// behaviour of an out-of-range cast
__device__ inline int photons_int(double nelec) {
    return (int)fmin(fmax(nelec, -2147483648.0), 2147483647.0);
}

struct SplitArgs {
    const BandDev *bands;
    const SrcRec *recs;
    const int *lists;
    const int *tile_cnt;
    const int64_t *tile_off;
    const double *nelec;
    const int64_t *offsets;     // [S*B + 1] packed position of the sample patch of (source s, band b) at s*B + b
    void *samp;                 // packed sample patches (the kernels' TS: double for a caller's buffer, int for the
                                // device-resident split), zero-initialised by the caller
    double *partials;           // per-tile noise sums
    int64_t S, capacity;
    int B, H, W, ntx, nty, TW, TH;
    int win_y0, full_H;         // the images hold rows [win_y0, win_y0 + H) of a full_H-row frame: the random
                                // streams are keyed on FULL-FRAME pixel indices, so a pixel draws the same
                                // numbers whichever row strip (GPU) holds it
    unsigned long long seed;
    const double *rate_img;     // k_photon_split_hw: every pixel's total rate (strict boxes), rendered beforehand
    double tail_T;              // k_photon_split_hw: drop threshold of the per-source tiles
    int4 *nz;                   // k_photon_split_hw: per patch, the rectangle holding its photons (min/max by atomics), or nullptr
    const int *order;           // k_photon_split_hw: tile launch order (heaviest first, from the totals render), or nullptr
    int noise_y0, noise_y1;     // rows [noise_y0, noise_y1) of the window whose sky photons the noise sums count (a rank of a
                                // strip-partitioned chain splits a halo beyond its strip and counts its strip only)
    int debug;                  // CEL_OPT_DEBUG bits (timing-only ablations; results are wrong when set)
    int strict;                 // 1: a source takes part strictly inside its box on the low side (the reference, :50-51); 0: on its
                                // whole box (CEL_OPT_SPLIT_FULL_BOX: the split of the model the renderer draws from)
    int *nnz;                   // k_photon_split_hw: pixels that received a photon, per (source, band), zeroed by the caller, or nullptr
    double *sums;               // k_photon_split_hw: photons per (source, band), index s*B + b, zeroed by the caller, or nullptr.
                                // Integer-valued doubles: the atomic sums are exact, so their order does not matter
    unsigned long long *massfx; // k_photon_split_hw: per (source, band), the unit stamp summed over the pixels the split walks
                                // (strictly inside the box), in units of 2^-60 (MASS_FX): integer atomics, so the order of the
                                // (source, half-tile) partials does not matter either; zeroed by the caller, or nullptr
};

template <typename TS>
__global__ void __launch_bounds__(64)
k_photon_split(SplitArgs a) {
    __shared__ double rate[2048];     // remaining total rate of the pixel (sources not yet drawn + sky)
    __shared__ int left[2048];        // photons of the pixel not yet attributed
    __shared__ CompTab T;
    __shared__ double et[64], lt[128];
    const int lane = threadIdx.x;
    const int tile = blockIdx.x;
    et[lane] = exp2((double)lane * (1.0 / 64.0));
    lt[lane] = c_log_ic[lane];
    lt[64 + lane] = c_log_lc[lane];
    const int per_band = a.ntx * a.nty;
    const int b = tile / per_band;
    const int t = tile - b * per_band;
    const int ty = t / a.ntx, tx = t - ty * a.ntx;
    const int X0 = tx * a.TW, Y0 = ty * a.TH;
    const int col = lane % a.TW, rsub = lane / a.TW, rstep = 64 / a.TW;
    const int niter = a.TH / rstep;              // TW * TH = 2048 in both layouts
    const int xi = X0 + col;
    const BandDev *bd = a.bands + b;
    const double eps = bd->eps;
    const int64_t plane = (int64_t)b * a.H * a.W;
    const int cnt = a.tile_cnt[tile];
    const int64_t off = a.tile_off[tile];
    const SrcRec *recs = a.recs + (int64_t)b * a.S;
    unsigned covered = 0u;            // bit i: pixel i of this lane lies strictly inside some source's box

    for (int i = 0; i < niter; i++) {
        const int y = Y0 + i * rstep + rsub;
        const bool in = (xi < a.W) && (y < a.H);
        rate[i * 64 + lane] = eps;
        left[i * 64 + lane] = in ? photons_int(a.nelec[plane + (int64_t)y * a.W + xi]) : 0;
    }

    for (int pass = 0; pass < 2; pass++) {
        for (int e = 0; e < cnt; e++) {
            const int64_t at = off + e;
            if (at >= a.capacity) break;
            const int s = __builtin_amdgcn_readfirstlane(a.lists[at]);
            const SrcRec *rp = recs + s;
            const int type = rp->type;
            const int K = (type == 0) ? K_PSF : K_GAL;
            const int bx0 = rp->x0, bx1 = rp->x1, by0 = rp->y0, by1 = rp->y1;
            const double counts = rp->scale;
            __syncthreads();
            if (lane < K) {
                Comp c = make_comp(lane, type, rp->px, rp->py, 1.0, rp->w00, rp->w01, rp->w11, rp->theta, bd);
                T.A[lane] = c.A; T.mx[lane] = c.mx; T.my[lane] = c.my;
                T.qa[lane] = c.qa; T.qb[lane] = c.qb; T.qc[lane] = c.qc;
            }
            __syncthreads();
            const int nx = bx1 - bx0;
            TS *patch = static_cast<TS *>(a.samp) + a.offsets[(int64_t)s * a.B + b];    // source-major, like cel_patch_loglik_multi
            const bool colin = (xi >= bx0 + a.strict) && (xi < bx1);     // strict on the low side (:50)
            for (int i = 0; i < niter; i++) {
                const int y = Y0 + i * rstep + rsub;
                if (!(colin && y >= by0 + a.strict && y < by1)) continue;
                const double F = counts * eval_direct(T, 0, K, (double)xi, (double)y, 1.0);
                const int li = i * 64 + lane;
                if (pass == 0) {
                    rate[li] += F;
                    covered |= 1u << i;
                } else {
                    const int n = left[li];
                    double tot = rate[li];
                    long long z = 0;
                    if (n > 0) {
                        const int64_t yf = (int64_t)a.win_y0 + y;            // full-frame row
                        const int64_t kb = (int64_t)b * a.full_H * a.W;
                        z = split_draw((long long)n, F * fast_rcp(tot), a.seed, (unsigned long long)(kb + (yf & ~SPLIT_GROUP_MASK) * a.W + xi),
                                       (int)((yf >> 1) & 7), (unsigned long long)(kb + yf * a.W + xi), (unsigned)s, et, lt);   // curr_prob / sum_probs (:147)
                    }
                    left[li] = n - (int)z;
                    rate[li] = tot - F;                                   // sum_probs -= curr_prob (:152)
                    patch[(int64_t)(y - by0) * nx + (xi - bx0)] = (TS)z;
                }
            }
        }
    }
    // what is left belongs to the sky (:153); a pixel nobody covers adds its nelec as it is, not
    // truncated to an integer (:91-93)
    double noise = 0.0;
    for (int i = 0; i < niter; i++) {
        const int y = Y0 + i * rstep + rsub;
        if (y < a.noise_y0 || y >= a.noise_y1) continue;
        if ((covered >> i) & 1u) noise += (double)left[i * 64 + lane];
        else if (xi < a.W && y < a.H) noise += a.nelec[plane + (int64_t)y * a.W + xi];
    }
    noise = wave_sum(noise);
    if (lane == 0) a.partials[tile] = noise;
}

// ---- the same split on the column recurrence ------------------------------------------------------
// k_photon_split above evaluates every (source, pixel) twice with the direct evaluator (once for
// the pixel totals, once for the draws): ~85 ms at config 3, 45 x the render step.  Here
//   * the totals are an image rendered beforehand by the field kernel with the reference's strict
//     boxes (CEL_RENDER_STRICT: a source takes part only where x > x0 and y > y0), and
//   * each source is rendered ONCE, by the recurrence, into a scratch LDS tile (hw_source.h), from
//     which the lanes draw their pixels' binomials.
// One wave per 32 x 32 half of a render tile (three 32 x 32 LDS planes: the source's tile, the
// remaining rate, the photons left), sources in list order = source order, as the reference.
// A pixel's draw uses the Philox stream (seed; band, pixel, source) as above; the probabilities
// agree with the direct kernel's to ~1e-13, so the two kernels make the same draws except where a
// uniform falls within that distance of a decision boundary.
#define SP_TH 32
// TL: the type of the photons-left plane.  unsigned short (valid while every pixel of the image holds 0 ... 65 535 photons: the
// host knows the image's range, k_nelec_range) takes the block from 25 312 to 23 264 B of LDS, 7 waves per CU instead of 6.
template <typename TS, typename TL>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2)))
k_photon_split_hw(SplitArgs a) {
    __shared__ double one[SP_TH * HW_TW];
    __shared__ double rate[SP_TH * HW_TW];
    __shared__ TL left[SP_TH * HW_TW];
    __shared__ CompTab T;
#ifdef SPLIT_LDS_PAD        // occupancy experiments only
    __shared__ double lds_pad[SPLIT_LDS_PAD / 8];
    if (a.S < 0) lds_pad[threadIdx.x] = 1.0;
#endif
    // pixels of the current source whose draw needs the sampler proper: queued in the component table's
    // LDS, which is dead between a source's walk and the next source's table (a barrier either side)
    // ... and behind the queue the sampler's log table (written per source once its table is dead: a block of its own
    // kilobyte would cost the seventh wave -- LDS is handed out in 512-B granules)
    static_assert(sizeof(CompTab) >= sizeof(unsigned short) * SP_TH * HW_TW + 128 * sizeof(double),
                  "the draw queue and the log table live in the component table");
    unsigned short *queue = reinterpret_cast<unsigned short *>(&T);
    double *lt = reinterpret_cast<double *>(reinterpret_cast<char *>(&T) + sizeof(unsigned short) * SP_TH * HW_TW);
    __shared__ double et[64];
    // the band's three PSF components as a STAR needs them (round 6): inverse covariance in the table exponential's units,
    // amplitude per unit count, centre offsets, and exp(-4 qc): the ratio of a stride-two column recurrence's ratios
    __shared__ double sc[8 * K_PSF];
    const int lane = threadIdx.x;
    const int half = lane >> 5, col = lane & 31;
    const int sub = blockIdx.x & 1;
    const int tile = a.order ? a.order[blockIdx.x >> 1] : (int)(blockIdx.x >> 1);
    const int per_band = a.ntx * a.nty;
    const int b = tile / per_band;
    const int t = tile - b * per_band;
    const int ty = t / a.ntx, tx = t - ty * a.ntx;
    const int X0 = tx * HW_TW, Y0 = ty * HW_TH + sub * SP_TH;
    const int xi = X0 + col;
    const double x = (double)xi;
    const BandDev *bd = a.bands + b;
    const double eps = bd->eps;
    const int64_t plane = (int64_t)b * a.H * a.W;
    const int64_t kband = (int64_t)b * a.full_H * a.W;                            // full-frame pixel index of the band's (0, 0)
    const int64_t key0 = kband + (int64_t)a.win_y0 * a.W;                         // ... of this window's (x=0, y=0)
    unsigned covered = 0u;            // bit r: this lane's pixel of row pair r lies strictly inside some source's box
    et[lane] = exp2((double)lane * (1.0 / 64.0));
    // Every load unconditional (a pixel outside the frame reads the band's first pixel and drops the value): with a
    // load under a condition the compiler waits for each one before it issues the next (s_waitcnt vmcnt(0) where the
    // paths meet) -- 16 memory round trips one behind the other at the start of every block.
    {
        double rt[SP_TH / 2], ne[SP_TH / 2];
#pragma unroll
        for (int r = 0; r < SP_TH / 2; r++) {
            const int y = Y0 + 2 * r + half;
            const bool in = (xi < a.W) && (y < a.H);
            const int64_t idx = in ? plane + (int64_t)y * a.W + xi : plane;
            rt[r] = a.rate_img[idx];
            ne[r] = a.nelec[idx];
        }
#pragma unroll
        for (int r = 0; r < SP_TH / 2; r++) {
            const int y = Y0 + 2 * r + half;
            const bool in = (xi < a.W) && (y < a.H);
            one[r * 64 + lane] = 0.0;
            rate[r * 64 + lane] = in ? rt[r] : eps;
            left[r * 64 + lane] = (TL)(in ? photons_int(ne[r]) : 0);
        }
    }
    const int cnt = a.tile_cnt[tile];
    const int64_t off = a.tile_off[tile];
    const SrcRec *recs = a.recs + (int64_t)b * a.S;
    const int dropmode = (a.tail_T > 0.0 && eps > 0.0) ? HW_DROP_SKY : HW_DROP_NONE;
    const double log_sky = (eps > 0.0) ? (double)__logf((float)eps) : 0.0;
    // The band's lane constants (28 VGPRs) are read again per source instead of living in registers across the walk: 253
    // VGPRs and NO scratch (round 4: 256 + 20 B, round 5: + 28 B with them held).  Measured on the benchmark field, three
    // repeats each way (tools/ab_scratch.sh, profiles/r05_ab_scratch.txt): 6.50-6.52 ms held, 6.52-6.53 ms re-read --
    // equal; -DSPLIT_LC_IN_REGISTERS builds the other form.
#ifdef SPLIT_LC_IN_REGISTERS
    const LaneConst lc = lane_consts(lane, bd);
#endif
#ifdef CEL_ABLATE
    // work counters of the diagnostic build (CEL_OPT_DEBUG bits 8..): returned in place of the noise sums, summed over the tiles
    //   8 queued draws | 16 sampler trips of 64 | 32 (source, half-tile) pairs walked | 64 draws by BTPE | 128 queued draws that
    //   left a photon | 256 first-pass steps (row pairs) | 512 pairs with a non-empty queue
    double dbg_count = 0.0;
#endif
    // ---- a star takes no component table (round 6) ----------------------------------------------------------------------
    // Half of a field's sources are stars, and a star's three components are the band's PSF shifted to the star: nothing to
    // invert, nothing to drop, no slots to sort.  Yet every (star, half-tile) pair went through hw_build -- ~600 instructions
    // and two barriers -- before its walk.  Here a lane seeds the three components at its own column and first row and walks ITS
    // pixels (row = 2 r + half: every other row, so the recurrence runs with stride two: g(y + 2) = g(y) R(y), R(y + 2) =
    // R(y) e^(-4 qc)) inside the first-test loop itself: the value arrives in a register, the scratch tile is written only for
    // the pixels that go to the sampler.  Legal when no exponent on a star's box leaves the range one segment is safe on (the
    // check of the field kernel's star pass, star_setup); a sharper PSF takes the general path.  Nothing is dropped, so a star's
    // pixel carries all three components (the general path skips those below eps e^-T): 1e-14 apart.
    bool star_bad = false;
    if (lane < K_PSF) {
        const double cxx = bd->cxx[lane], cxy = bd->cxy[lane], cyy = bd->cyy[lane];
        double inv, rsq;
        rcp_rsqrt(cxx * cyy - cxy * cxy, inv, rsq);
        const double qa = cyy * inv, qb = -cxy * inv, qc = cxx * inv;
        sc[8 * lane + 0] = qa * EXP_SCALE; sc[8 * lane + 1] = qb * EXP_SCALE; sc[8 * lane + 2] = qc * EXP_SCALE;
        sc[8 * lane + 3] = bd->w[lane] * (0.5 / PI_D) * rsq;
        sc[8 * lane + 4] = bd->mux[lane]; sc[8 * lane + 5] = bd->muy[lane];
        sc[8 * lane + 6] = exp_tab64(-4.0 * qc * EXP_SCALE, et);
        sc[8 * lane + 7] = 0.0;
        const double rb_ = bd->R + 2.0;
        star_bad = !(0.5 * quad_max_rect_hw(qa, qb, qc, -rb_ - bd->mux[lane], rb_ - bd->mux[lane], -rb_ - bd->muy[lane], rb_ - bd->muy[lane]) <= STAR_EMAX);
    }
    const bool star_ok = (__ballot(star_bad) == 0ull) && !(SPLIT_ABLATE(a) & (4 | 1024));
    lt[lane] = c_log_ic[lane];              // (a star leaves the table's storage alone: the sampler's log table must be there from the start)
    lt[64 + lane] = c_log_lc[lane];
    __syncthreads();
    const int nent = (Y0 < a.H) ? (int)min((int64_t)cnt, a.capacity > off ? a.capacity - off : (int64_t)0) : 0;
    int idx64 = (lane < nent) ? a.lists[off + lane] : 0;
    int s_next = __builtin_amdgcn_readlane(idx64, 0);
    int recw_next = (nent > 0) ? rec_fetch(recs, s_next, lane) : 0;
    int64_t poff_next = (nent > 0) ? a.offsets[(int64_t)s_next * a.B + b] : 0;

    for (int e = 0; e < nent; e++) {
        const int recw = recw_next;
        const int s = s_next;
        const int64_t poff = poff_next;
        if (e + 1 < nent) {
            if (((e + 1) & 63) == 0) idx64 = (e + 1 + lane < nent) ? a.lists[off + e + 1 + lane] : 0;
            s_next = __builtin_amdgcn_readlane(idx64, (e + 1) & 63);
            recw_next = rec_fetch(recs, s_next, lane);
            poff_next = a.offsets[(int64_t)s_next * a.B + b];
        }
        const RecU rec = rec_unpack(recw);
        // strictly inside the box on the low side (celeste_sample_sources.pyx:50-51)
        const int sx0 = rec.x0 + a.strict, sy0 = rec.y0 + a.strict;
        const int ra = max(sy0, Y0) - Y0, rb = min(rec.y1, Y0 + SP_TH) - Y0;
        const int xa = max(sx0, X0), xb = min(rec.x1, X0 + HW_TW) - 1;
        if (ra >= rb || xa > xb) continue;          // touches the tile's other half only (wave-uniform)
        const bool on = (xi >= xa) && (xi <= xb);
        const bool star_fast = star_ok && rec.type == 0;
        // Two passes over the source's pixels on this half-tile.  Most draws are decided by ONE
        // uniform (U <= 1 - n p gives 0: a pixel in the source's tail); the few that are not would
        // each hold their whole wave in the sampler's loops.  Pass 1 settles the easy pixels and
        // queues the others (a ballot + prefix count per step); pass 2 runs the sampler on the queue,
        // 64 pixels per trip.  A pixel's draw takes the same numbers of its Philox stream either way.
        int zlo = INT_MAX, zhi = -1, xlo = INT_MAX, xhi = -1;   // where this lane's draws left photons
        double zsum = 0.0, fsum = 0.0;
        int zcnt = 0;
        const int nx = rec.x1 - rec.x0;
        TS *patch0 = static_cast<TS *>(a.samp) + poff + (int64_t)(Y0 - rec.y0) * nx - rec.x0;   // + row * nx + x
        int nq = 0;
        // pass 1, in two instantiations (a generic lambda: the star form's six recurrence registers must not be live through the
        // general form's walk -- as plain locals of this loop they cost the general path six spilled registers and 0.2 ms)
        auto pass1 = [&](auto star_tag) {
            constexpr bool STAR = decltype(star_tag)::value;
            double sg0 = 0.0, sg1 = 0.0, sg2 = 0.0, sr0 = 1.0, sr1 = 1.0, sr2 = 1.0, sq0 = 1.0, sq1 = 1.0, sq2 = 1.0;
            if (STAR) {
                sq0 = sc[6]; sq1 = sc[14]; sq2 = sc[22];      // e^(-4 qc): registers of the star form only
                // seeds at this lane's column and first row of the loop below (row 2 (ra >> 1) + half: possibly one above the box)
                const double y0 = (double)(Y0 + 2 * (ra >> 1) + half);
#define SPLIT_STAR_SEED(K, G, R)                                                                           \
                {                                                                                          \
                    const double *cs_ = sc + 8 * (K);                                                      \
                    const double dx = x - (rec.px + cs_[4]), dy = y0 - (rec.py + cs_[5]);                  \
                    const double hx = cs_[1] * dx + cs_[2] * dy;                                           \
                    G = (cs_[3] * rec.scale) * exp_tab64(-0.5 * (cs_[0] * dx * dx + (cs_[1] * dx + hx) * dy), et); \
                    R = exp_tab64(-2.0 * (hx + cs_[2]), et);                                               \
                }
                SPLIT_STAR_SEED(0, sg0, sr0)
                SPLIT_STAR_SEED(1, sg1, sr1)
                SPLIT_STAR_SEED(2, sg2, sr2)
#undef SPLIT_STAR_SEED
            }
            int64_t grp_have = -1;              // the row group (full-frame row, bits 1, 2 and 3 cleared) whose block this lane holds
            Philox hg;
            hg.out[0] = hg.out[1] = hg.out[2] = hg.out[3] = 0u;
            for (int r = ra >> 1; 2 * r < rb; r++) {
                const int row = 2 * r + half;
                const int li = r * 64 + lane;
                // the shared block of this lane's eight consecutive rows: one per eight steps, every lane at the same step
                // (windows start on even rows), whether or not it has a pixel here
                const int64_t yf = (int64_t)a.win_y0 + Y0 + row;
                const int64_t grp = yf & ~SPLIT_GROUP_MASK;
                if (grp != grp_have && !(SPLIT_ABLATE(a) & 1)) {
                    hg = philox_init(a.seed, (unsigned long long)(kband + grp * a.W + (xi < a.W ? xi : 0)) | SPLIT_GROUP_BIT, (unsigned)s);
                    philox_block(hg);
                    grp_have = grp;
                }
                const unsigned vword = philox_word(hg, (int)((yf >> 1) & 7));
                bool slow = false;
                double Fstar = 0.0;
                if (STAR) {             // this lane's pixel of the step, then one stride-two step of the three recurrences
#pragma clang fp contract(off)
                    Fstar = (sg0 + sg1) + sg2;
                    sg0 *= sr0; sg1 *= sr1; sg2 *= sr2;
                    sr0 *= sq0; sr1 *= sq1; sr2 *= sq2;
                }
                if (on && row >= ra && row < rb) {
                    const double F = STAR ? Fstar : one[li];
                    const int n = (int)left[li];
                    const double tot = rate[li];
                    fsum += F;
                    covered |= 1u << r;
                    if (n > 0 && !(SPLIT_ABLATE(a) & 1)) {
                        const double pr = F * fast_rcp(tot);                  // curr_prob / sum_probs (:147)
                        if (pr > 0.0) slow = !(pr <= 0.5 && (double)vword < split_tf((long long)n, pr));     // the first test, on the shared word
                    }
                    if (!slow) {
                        if (!STAR) one[li] = 0.0;         // the scratch tile is clean again for the next source
                        rate[li] = tot - F;               // sum_probs -= curr_prob (:152)
                        patch0[(int64_t)row * nx + xi] = (TS)0;
                    } else if (STAR) one[li] = F;         // the sampler's pass reads the value there (and clears it)
                }
                const unsigned long long sm = __ballot(slow);
                if (slow) queue[nq + __builtin_amdgcn_mbcnt_hi((unsigned)(sm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)sm, 0))] = (unsigned short)li;
                nq += __popcll(sm);
            }
        };
        if (star_fast) {
            pass1(std::true_type{});
        } else {
            bool direct;
#ifndef SPLIT_LC_IN_REGISTERS
            asm volatile("" ::: "memory");      // (keeps the compiler from hoisting the loads out of the loop again)
            const LaneConst lc = lane_consts(lane, bd);
#endif
            const int Kk = hw_build(T, lc, rec, lane, dropmode, a.tail_T, log_sky, Y0, xa, xb, ra, rb, direct, nullptr, et);
            if (!(SPLIT_ABLATE(a) & 4)) hw_walk(T, et, Kk, x, Y0, ra, rb, on, direct, one, lane);
            __syncthreads();
            lt[lane] = c_log_ic[lane];          // the component table is dead until the next source: queue + log table
            lt[64 + lane] = c_log_lc[lane];
            pass1(std::false_type{});
        }
        __syncthreads();
#ifdef CEL_ABLATE
        if (lane == 0) {
            if (a.debug & 8) dbg_count += (double)nq;
            if (a.debug & 16) dbg_count += (double)((nq + 63) >> 6);
            if (a.debug & 32) dbg_count += 1.0;
            if (a.debug & 256) dbg_count += (double)(((rb + 1) >> 1) - (ra >> 1));
            if ((a.debug & 512) && nq > 0) dbg_count += 1.0;
        }
#endif
        for (int q0 = 0; q0 < nq; q0 += 64) {
            if (q0 + lane < nq) {
                const int li = queue[q0 + lane];
                const int row = 2 * (li >> 6) + ((li >> 5) & 1), xq = X0 + (li & 31);
                const double F = one[li];
                one[li] = 0.0;
                const int n = (int)left[li];
                const double tot = rate[li];
                Philox g = philox_init(a.seed, (unsigned long long)(key0 + (int64_t)(Y0 + row) * a.W + xq), (unsigned)s);
                const double pr = F * fast_rcp(tot);
                const double afirst = (pr <= 0.5) ? fmax(split_tf((long long)n, pr), 0.0) * (1.0 / SPLIT_WORD_SCALE) : 0.0;   // pass 1 saw V >= tf
                const long long z = (SPLIT_ABLATE(a) & 2) ? 1 : binomial_draw((long long)n, pr, afirst, g, et, lt);
                left[li] = (TL)(n - (int)z);
                rate[li] = tot - F;
                patch0[(int64_t)row * nx + xq] = (TS)z;
#ifdef CEL_ABLATE
                if ((a.debug & 64) && fmin(pr, 1.0 - pr) * (double)n > BINV_MAX_NP) dbg_count += 1.0;
                if ((a.debug & 128) && z > 0) dbg_count += 1.0;
#endif
                if (z > 0) {
                    zlo = min(zlo, row); zhi = max(zhi, row); xlo = min(xlo, xq); xhi = max(xhi, xq);
                    zsum += (double)z;
                    zcnt += 1;
                }
            }
        }
        if (a.massfx) {                  // this half-tile's share of the source's stamp mass (cel_stamp_mass's short cut)
            fsum = wave_sum_lane63(fsum);
            if (lane == 63 && fsum > 0.0 && rec.scale > 0.0)
                atomicAdd(a.massfx + ((int64_t)s * a.B + b), (unsigned long long)__double2ull_rn(fsum * fast_rcp(rec.scale) * MASS_FX));   // 1 / scale to 4e-15
        }
        if (a.nz && __ballot(zhi >= 0)) {
            // the patch's photon rectangle (what the conditional likelihoods will evaluate): one
            // wave reduction per (source, half-tile), four atomics by one lane.  (Tracking it in
            // scalar registers from a ballot per step needs a wave-uniform step loop and measured
            // slower: 10.7 against 10.4 ms; a separate pass over the 3.2 GB of patches costs 1.1 ms.)
            for (int o = 32; o > 0; o >>= 1) {
                zlo = min(zlo, __shfl_xor(zlo, o)); zhi = max(zhi, __shfl_xor(zhi, o));
                xlo = min(xlo, __shfl_xor(xlo, o)); xhi = max(xhi, __shfl_xor(xhi, o));
            }
            if (a.sums) zsum = wave_sum(zsum);
            if (a.nnz) for (int o = 32; o > 0; o >>= 1) zcnt += __shfl_down(zcnt, o);
            if (lane == 0) {
                int *q = reinterpret_cast<int *>(a.nz + ((int64_t)s * a.B + b));
                atomicMin(q + 0, xlo); atomicMax(q + 1, xhi + 1);
                atomicMin(q + 2, Y0 + zlo); atomicMax(q + 3, Y0 + zhi + 1);
                if (a.sums) atomicAdd(a.sums + ((int64_t)s * a.B + b), zsum);
                if (a.nnz) atomicAdd(a.nnz + ((int64_t)s * a.B + b), zcnt);
            }
        }
        __syncthreads();
    }
    // what is left belongs to the sky (:153); a pixel nobody covers adds its nelec as it is, not
    // truncated to an integer (:91-93)
    double noise = 0.0;
    double raw[SP_TH / 2];            // unconditional loads again (the tile is in L2): all 16 in flight at once
#pragma unroll
    for (int r = 0; r < SP_TH / 2; r++) {
        const int y = Y0 + 2 * r + half;
        const bool in = (xi < a.W) && (y < a.H);
        raw[r] = a.nelec[in ? plane + (int64_t)y * a.W + xi : plane];
    }
#pragma unroll
    for (int r = 0; r < SP_TH / 2; r++) {
        const int y = Y0 + 2 * r + half;
        const bool counted = (y >= a.noise_y0) && (y < a.noise_y1);
        const bool in = (xi < a.W) && (y < a.H);
        const double v = ((covered >> r) & 1u) ? (double)left[r * 64 + lane] : (in ? raw[r] : 0.0);
        noise += counted ? v : 0.0;
    }
#ifdef CEL_ABLATE
    if (a.debug & ~7) noise = dbg_count;
#endif
    noise = wave_sum(noise);
    if (lane == 0) a.partials[2 * tile + sub] = noise;
}
