// celeste_hip.hip -- MI355X (gfx950 / CDNA4) implementation of CelestePy's model-image
// rendering + Poisson log-likelihood path behind the C ABI of include/celeste_hip.h.
//
// Path (reference file:line, relative to the HIPS/DESI-MCMC root):
//   equa2pixel / cd_at_pixel            CelestePy/fits_image.py:166-216
//   calc_bounding_radius                CelestePy/util/bound/bounding_box.py:9-31
//   galaxy transform + MoG (x) MoG      CelestePy/celeste_galaxy_conditionals.py:90-125,185-214
//                                       CelestePy/util/dists/mog.py:75-100
//   mixture evaluation on pixel grids   CelestePy/util/dists/mog.py:5-21,
//                                       CelestePy/util/like/gmm_like_fast.pyx:130-176
//   gen_model_image / celeste_likelihood CelestePy/celeste.py:203-252
//
// Design (DESIGN.md has the long form):
//   k_prep    one thread per (band, source): pixel position, galaxy shape matrix, bounding
//             radius, clipped box -> a 128-byte record + a 16-byte box.
//   k_bin     one wave per (band, 64 x TH image tile): scans the band's boxes 64 at a time
//             (ballot + prefix popcount) and writes the tile's source list in source order:
//             deterministic, no sort, no atomics on the data path.
//   k_render  one wave per tile, lane = pixel column (coalesced 512-B rows).  Gathers every
//             source of the tile's list into an LDS accumulator tile, then writes
//             lambda = eps + acc ONCE and fuses the Poisson term nelec*log(lambda) - lambda
//             with a wavefront shuffle reduction.  Component tables (K = 3 star, 42 galaxy)
//             are built lane-parallel in LDS from the 128-byte record.  Two evaluators:
//               direct    : exp() per Gaussian-pixel
//               recurrence: along a pixel column a Gaussian obeys g(y+1) = g(y) r(y),
//                           r(y+1) = r(y) q with q = exp(-c): 2 mul + 1 add per
//                           Gaussian-pixel after a per-segment seed; segments are bounded so
//                           that no significant lane ever underflows.
//   k_reduce  fixed-order sum of the per-tile partials -> ll per band (bitwise reproducible).
//   k_stamps  per-source stamps into a packed buffer (same column evaluators, no accumulator).
//   k_gmm     generic N-point evaluator (gmm_like_2d).
// All arithmetic is fp64 on the vector ALU; MFMA is not used (no contraction in this path).

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/celeste_hip.h"

#define K_PSF 3
#define K_EXP 6
#define K_PROF 14
#define K_GAL 42
#define TILE_W 64
#define MAX_BANDS 16

// ------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(e_ == hipErrorOutOfMemory ? CEL_ERR_NOMEM : CEL_ERR_HIP, "%s: %s (%s:%d)", \
                        #expr, hipGetErrorString(e_), __FILE__, __LINE__);                    \
    } while (0)

// ------------------------------------------------------------------------------------------
// device-side data
// ------------------------------------------------------------------------------------------
struct BandDev {        // per band, SoA-friendly PSF so that lane k can index component k
    double eps;
    double w[K_PSF], mux[K_PSF], muy[K_PSF], cxx[K_PSF], cxy[K_PSF], cyy[K_PSF];
    double rho[2], phi[2], ups[4], ups_inv[4];
    double R;
};

struct alignas(16) SrcRec {   // one per (band, source); 128 bytes
    double px, py;            // pixel position (x = column, y = row)
    double scale;             // expected photons of this source in this band
    double w00, w01, w11;     // galaxy: Tinv Tinv^T (cov_j = var_j * W + psf_cov_k)
    double theta;             // exp-profile fraction
    int x0, x1, y0, y1;       // clipped box [x0,x1) x [y0,y1); empty when x1<=x0 or y1<=y0
    int type;                 // 0 star, 1 galaxy, -1 no contribution
    int pad[3];
    double rsv[5];
};
static_assert(sizeof(SrcRec) == 128, "SrcRec must be 128 bytes");

// exp/dev profile mixtures (Hogg & Lang; CelestePy/mixture_profiles.py:9-19), amplitudes
// normalised on the host exactly as the reference does (:13,:19) and uploaded once.
__constant__ double c_prof_amp[K_PROF];
__constant__ double c_prof_var[K_PROF];

static const double H_EXP_AMP[6] = {2.34853813e-03, 3.07995260e-02, 2.23364214e-01,
                                    1.17949102e+00, 4.33873750e+00, 5.99820770e+00};
static const double H_EXP_VAR[6] = {1.20078965e-03, 8.84526493e-03, 3.91463084e-02,
                                    1.39976817e-01, 4.60962500e-01, 1.50159566e+00};
static const double H_DEV_AMP[8] = {4.26347652e-02, 2.40127183e-01, 6.85907632e-01, 1.51937350e+00,
                                    2.83627243e+00, 4.46467501e+00, 5.72440830e+00, 5.60989349e+00};
static const double H_DEV_VAR[8] = {2.23759216e-04, 1.00220099e-03, 4.18731126e-03, 1.69432589e-02,
                                    6.84850479e-02, 2.87207080e-01, 1.33320254e+00, 8.40215071e+00};

#define PI_D 3.14159265358979323846

// ------------------------------------------------------------------------------------------
// k_prep: (band, source) -> record + box
// ------------------------------------------------------------------------------------------
__device__ inline void dev_pixel2equa(const BandDev &b, double x, double y, double cphi, double &ra,
                                      double &dec) {
    double d0 = x - b.rho[0], d1 = y - b.rho[1];
    double i0 = b.ups[0] * d0 + b.ups[1] * d1;
    double i1 = b.ups[2] * d0 + b.ups[3] * d1;
    ra = i0 / cphi + b.phi[0];
    dec = i1 + b.phi[1];
}

__device__ inline double clampd(double v, double lo, double hi) { return fmin(fmax(v, lo), hi); }

// calc_bounding_radius for one component (bounding_box.py:13-27)
__device__ inline double comp_radius(double cxx, double cxy, double cyy, double rsq_inv, double dist) {
    double s1 = sqrt(cxx), s2 = sqrt(cyy);
    double rho = cxy / (s1 * s2);
    double A11 = s1, A21 = rho * s2, A22 = s2 * sqrt(1.0 - rho * rho);
    double An = rsq_inv * (1.0 / (A11 * A11) + (A21 * A21) / (A22 * A22));
    double Bn = rsq_inv * (-2.0 * A21 / (A11 * (A22 * A22)));
    double Cn = rsq_inv * 1.0 / (A22 * A22);
    double maj = 1.0 / sqrt(0.5 * (An + Cn - sqrt(Bn * Bn + (An - Cn) * (An - Cn))));
    return maj + dist;
}

__global__ void __launch_bounds__(256)
k_prep(const BandDev *__restrict__ bands, int B, int H, int W, int win_y0, int win_h, int64_t S,
       const int *__restrict__ type, const double *__restrict__ radec,
       const double *__restrict__ counts, const double *__restrict__ shape, double rsq_gal,
       SrcRec *__restrict__ recs, int4 *__restrict__ boxes) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S * B) return;
    int b = (int)(i / S);
    int64_t s = i - (int64_t)b * S;
    const BandDev &bd = bands[b];
    SrcRec r;
    memset(&r, 0, sizeof(r));
    int t = type[s];
    double ra = radec[2 * s], dec = radec[2 * s + 1];
    // equa2pixel (fits_image.py:166-174)
    double cphi = cos(bd.phi[1] / 180.0 * PI_D);
    double s0 = (ra - bd.phi[0]) * cphi, s1 = dec - bd.phi[1];
    double px = (bd.ups_inv[0] * s0 + bd.ups_inv[1] * s1) + bd.rho[0];
    double py = (bd.ups_inv[2] * s0 + bd.ups_inv[3] * s1) + bd.rho[1];
    r.px = px; r.py = py;
    r.scale = counts[s * B + b];
    r.type = t;
    const double BIG = 1073741824.0;
    if (t == 0) {
        // celeste.py:130-140: overlap test (with the reference's axis mix-up, Q1) + int() box
        bool miss = (px < -50 || px > 2.0 * H || py < -50 || px > 2.0 * W);
        if (miss || !(px == px) || !(py == py)) {
            r.type = -1;
        } else {
            double bound = bd.R;
            int lx = (int)clampd(px - bound, -BIG, BIG), hx = (int)clampd(px + bound + 1, -BIG, BIG);
            int ly = (int)clampd(py - bound, -BIG, BIG), hy = (int)clampd(py + bound + 1, -BIG, BIG);
            r.x0 = max(0, lx); r.x1 = min(hx, W);
            r.y0 = max(0, ly); r.y1 = min(hy, H);
        }
    } else if (t == 1) {
        double theta = shape[4 * s], sig = shape[4 * s + 1], phi_s = shape[4 * s + 2], rho_s = shape[4 * s + 3];
        // cd_at_pixel (fits_image.py:196-216): 10-px finite difference of pixel2equa
        double ra0, dec0, rax, decx, ray, decy;
        dev_pixel2equa(bd, px, py, cphi, ra0, dec0);
        dev_pixel2equa(bd, px + 10.0, py, cphi, rax, decx);
        dev_pixel2equa(bd, px, py + 10.0, cphi, ray, decy);
        double cosd = cos(dec0 * (PI_D / 180.0));
        double cd0 = (rax - ra0) / 10.0 * cosd, cd1 = (ray - ra0) / 10.0 * cosd;
        double cd2 = (decx - dec0) / 10.0, cd3 = (decy - dec0) / 10.0;
        // gen_galaxy_transformation (celeste_galaxy_conditionals.py:90-125); phi in degrees (Q7)
        double phi = (90.0 - phi_s) * PI_D / 180.0;
        double re_deg = fmax(1.0 / 30, sig) / 3600.0;
        double cp = cos(phi), sp = sin(phi);
        double g0 = re_deg * cp, g1 = re_deg * (sp * rho_s), g2 = re_deg * (-sp), g3 = re_deg * (cp * rho_s);
        double gd = g0 * g3 - g1 * g2;
        double gi0 = g3 / gd, gi1 = -g1 / gd, gi2 = -g2 / gd, gi3 = g0 / gd;
        double t0 = gi0 * cd0 + gi1 * cd2, t1 = gi0 * cd1 + gi1 * cd3;
        double t2 = gi2 * cd0 + gi3 * cd2, t3 = gi2 * cd1 + gi3 * cd3;
        double td = t0 * t3 - t1 * t2;
        double ti0 = t3 / td, ti1 = -t1 / td, ti2 = -t2 / td, ti3 = t0 / td;   // Tinv
        double w00 = ti0 * ti0 + ti1 * ti1, w01 = ti0 * ti2 + ti1 * ti3, w11 = ti2 * ti2 + ti3 * ti3;
        r.w00 = w00; r.w01 = w01; r.w11 = w11; r.theta = theta;
        // calc_bounding_radius over the 42 convolved components, error 1e-5, centre (px, py)
        double rsq_inv = 1.0 / rsq_gal;
        double bound = -INFINITY;
        for (int k = 0; k < K_PSF; k++) {
            double mx = (px + bd.mux[k]) - px, my = (py + bd.muy[k]) - py;
            double dist = sqrt(mx * mx + my * my);
            for (int j = 0; j < K_PROF; j++) {
                double v = c_prof_var[j];
                double rr = comp_radius(v * w00 + bd.cxx[k], v * w01 + bd.cxy[k], v * w11 + bd.cyy[k],
                                        rsq_inv, dist);
                bound = fmax(bound, rr);
            }
        }
        if (!(bound == bound) || !(px == px) || !(py == py)) {
            r.type = -1;
        } else {
            // celeste_galaxy_conditionals.py:208-211: floor/ceil box (Q6)
            r.x0 = (int)clampd(fmax(0.0, floor(px - bound)), -BIG, BIG);
            r.x1 = (int)clampd(fmin((double)W, ceil(px + bound)), -BIG, BIG);
            r.y0 = (int)clampd(fmax(0.0, floor(py - bound)), -BIG, BIG);
            r.y1 = (int)clampd(fmin((double)H, ceil(py + bound)), -BIG, BIG);
        }
    } else {
        r.type = -1;
    }
    // row window [win_y0, win_y0 + win_h) of the H-row frame (strip partition across GPUs):
    // boxes are formed against the FULL frame exactly as above, then cut to the window and
    // re-based, so a strip renders the same pixels the whole frame would.
    r.y0 = max(r.y0, win_y0) - win_y0;
    r.y1 = min(r.y1, win_y0 + win_h) - win_y0;
    r.py = py - (double)win_y0;
    if (r.type < 0 || r.x1 <= r.x0 || r.y1 <= r.y0) {
        r.x0 = r.x1 = r.y0 = r.y1 = 0;
        if (r.type >= 0) r.type = -1 - r.type;   // remember the kind, mark "no contribution"
    }
    recs[i] = r;
    boxes[i] = make_int4(r.x0, r.x1, r.y0, r.y1);
}

// work counters of one render: sum of box areas and K-weighted areas (on demand, not timed)
__global__ void k_stats(const SrcRec *__restrict__ recs, int64_t n, double *out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double a = 0.0, g = 0.0;
    if (i < n) {
        const SrcRec &r = recs[i];
        if (r.type >= 0) {
            a = (double)(r.x1 - r.x0) * (double)(r.y1 - r.y0);
            g = a * (r.type == 0 ? K_PSF : K_GAL);
        }
    }
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_down(a, o); g += __shfl_down(g, o); }
    if ((threadIdx.x & 63) == 0 && a != 0.0) { atomicAdd(out, a); atomicAdd(out + 1, g); }
}

// ------------------------------------------------------------------------------------------
// k_bin: per-tile source lists, in source order
// ------------------------------------------------------------------------------------------
// pass 0 (lists == nullptr): count; the last lane-0 of each tile reserves its segment with one
// atomicAdd on `cursor` (segment ORDER in the buffer is arbitrary, list CONTENT is not).
// pass 1: fill.  One wave per tile.
__global__ void __launch_bounds__(64)
k_bin(const int4 *__restrict__ boxes, int64_t S, int ntx, int nty, int TH, int pass,
      int *__restrict__ tile_cnt, int64_t *__restrict__ tile_off, unsigned long long *cursor,
      int *__restrict__ lists, int64_t capacity, int *overflow) {
    int tile = blockIdx.x;
    int lane = threadIdx.x;
    int per_band = ntx * nty;
    int b = tile / per_band;
    int t = tile - b * per_band;
    int ty = t / ntx, tx = t - ty * ntx;
    int X0 = tx * TILE_W, X1 = X0 + TILE_W, Y0 = ty * TH, Y1 = Y0 + TH;
    const int4 *bx = boxes + (int64_t)b * S;
    int64_t base = 0;
    if (pass == 1) base = tile_off[tile];
    int count = 0;
    for (int64_t s0 = 0; s0 < S; s0 += 64) {
        int64_t s = s0 + lane;
        bool hit = false;
        if (s < S) {
            int4 q = bx[s];
            hit = (q.x < X1) && (q.y > X0) && (q.z < Y1) && (q.w > Y0) && (q.y > q.x) && (q.w > q.z);
        }
        unsigned long long m = __ballot(hit);
        if (pass == 1 && hit) {
            int pos = __popcll(m & ((1ull << lane) - 1ull));
            int64_t at = base + count + pos;
            if (at < capacity) lists[at] = (int)s; else *overflow = 1;
        }
        count += __popcll(m);
    }
    if (pass == 0 && lane == 0) {
        tile_cnt[tile] = count;
        tile_off[tile] = (int64_t)atomicAdd(cursor, (unsigned long long)count);
    }
}

// ------------------------------------------------------------------------------------------
// component tables in LDS
// ------------------------------------------------------------------------------------------
// Per component, 8 doubles, SoA over k (reads in the evaluators are wave-uniform broadcasts):
//   A   = scale * weight / (2 pi sqrt(det))     mx, my = mean
//   qa, qb, qc = inverse covariance [[qa, qb], [qb, qc]]
//   ixx = qa - qb^2/qc = 1/Sigma_xx,  iyy = qc - qb^2/qa = 1/Sigma_yy   (marginal bounds)
struct CompTab {
    double A[K_GAL + 6], mx[K_GAL + 6], my[K_GAL + 6], qa[K_GAL + 6], qb[K_GAL + 6], qc[K_GAL + 6],
        ixx[K_GAL + 6], iyy[K_GAL + 6], eq[K_GAL + 6];   // eq = exp(-qc): the row-to-row ratio of r
};

__device__ inline void build_comp(CompTab &T, int k, int type, double px, double py, double scale,
                                  double w00, double w01, double w11, double theta,
                                  const BandDev *__restrict__ bd) {
    int kk = (type == 0) ? k : (k % K_PSF);
    int j = k / K_PSF;
    double cxx = bd->cxx[kk], cxy = bd->cxy[kk], cyy = bd->cyy[kk], wt = bd->w[kk];
    if (type == 1) {
        double var = c_prof_var[j];
        double amp = (j < K_EXP) ? theta * c_prof_amp[j] : (1.0 - theta) * c_prof_amp[j];
        cxx += var * w00; cxy += var * w01; cyy += var * w11;
        wt *= amp;
    }
    double det = cxx * cyy - cxy * cxy;
    double inv = 1.0 / det;
    double qa = cyy * inv, qb = -cxy * inv, qc = cxx * inv;
    T.A[k] = scale * wt / (2.0 * PI_D * sqrt(det));
    T.mx[k] = px + bd->mux[kk];
    T.my[k] = py + bd->muy[kk];
    T.qa[k] = qa; T.qb[k] = qb; T.qc[k] = qc;
    T.ixx[k] = 1.0 / cxx;   // = qa - qb^2/qc
    T.iyy[k] = 1.0 / cyy;
    T.eq[k] = exp(-qc);
}

__device__ inline double wave_sum(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    return v;
}

// distance from m to the closed interval [lo, hi]
__device__ inline double dist_to_interval(double m, double lo, double hi) {
    return fmax(fmax(lo - m, m - hi), 0.0);
}

// ---- direct evaluator: sum over components of A exp(-q/2) at (x, y) -------------------------
__device__ inline double eval_direct(const CompTab &T, int K, double x, double y) {
    double s = 0.0;
    for (int k = 0; k < K; k++) {
        double dx = x - T.mx[k], dy = y - T.my[k];
        double q = T.qa[k] * dx * dx + 2.0 * T.qb[k] * dx * dy + T.qc[k] * dy * dy;
        s += T.A[k] * exp(-0.5 * q);
    }
    return s;
}

// ---- recurrence evaluator -------------------------------------------------------------------
// For a fixed column x the exponent of component k is a parabola in the row y:
//   E(y) = -1/2 (qa dx^2 + 2 qb dx dy + qc dy^2),  g(y) = A exp(E(y))
//   g(y+1) = g(y) r(y),  r(y) = exp(-(qb dx + qc dy + qc/2)),  r(y+1) = r(y) exp(-qc)
// A segment of L rows is seeded with two exp() and then costs 2 mul + 1 add per row.
// Underflow safety: a lane whose value matters anywhere in the segment (E >= -T there) has
// E >= -T - L sqrt(2 T qc) - qc L^2/2 at the seed row; L is chosen so that this stays above
// -680 (fp64 exp underflows gradually below -708), so a significant lane never starts from a
// flushed seed.  Insignificant lanes may start from 0 and stay 0: they are below e^-T anyway.
// r's exponent is clamped to +-680: it can only exceed that on lanes whose g is exactly 0.
#define REC_G 6           // components advanced together (independent chains = ILP)
#define REC_EMAX 680.0

__device__ inline int seg_len(double qc, double T) {
    // largest L with (L sqrt(qc/2) + sqrt(T))^2 <= REC_EMAX
    double u = sqrt(REC_EMAX) - sqrt(T);
    double L = u / sqrt(0.5 * qc);
    return (int)fmin(L, 4096.0);
}

// Accumulate source components [k0, k0+REC_G) over rows [ra, rb) of column x into acc (LDS
// column of this lane, stride TILE_W doubles).  `on` masks lanes outside the source box.
template <int G>
__device__ inline void rec_group(const CompTab &T, int k0, int kn, double x, int Y0, int ra, int rb,
                                 int L, bool on, double *__restrict__ acc_col) {
    double g[G], r[G], q[G];
    for (int sa = ra; sa < rb; sa += L) {
        int sb = min(sa + L, rb);
        double y0 = (double)(Y0 + sa);
#pragma unroll
        for (int i = 0; i < G; i++) {
            int k = k0 + i;
            if (i < kn) {
                double dx = x - T.mx[k], dy = y0 - T.my[k];
                double qa = T.qa[k], qb = T.qb[k], qc = T.qc[k];
                double e = -0.5 * (qa * dx * dx + 2.0 * qb * dx * dy + qc * dy * dy);
                double er = -(qb * dx + qc * dy + 0.5 * qc);
                er = fmin(fmax(er, -REC_EMAX), REC_EMAX);
                g[i] = on ? T.A[k] * exp(e) : 0.0;
                r[i] = exp(er);
                q[i] = T.eq[k];
            } else {
                g[i] = 0.0; r[i] = 0.0; q[i] = 0.0;
            }
        }
        for (int row = sa; row < sb; row++) {
            double s = 0.0;
#pragma unroll
            for (int i = 0; i < G; i++) {
                s += g[i];
                g[i] *= r[i];
                r[i] *= q[i];
            }
            acc_col[row * TILE_W] += s;
        }
    }
}

// ------------------------------------------------------------------------------------------
// k_render: one wave per (band, tile)
// ------------------------------------------------------------------------------------------
struct RenderArgs {
    const BandDev *bands;
    const SrcRec *recs;
    const int *lists;
    const int *tile_cnt;
    const int64_t *tile_off;
    const double *nelec;
    double *lambda;
    double *partials;
    int64_t S, capacity;
    int B, H, W, ntx, nty;
    int flags;        // CEL_RENDER_*
    int variant;      // 0 direct, 1 recurrence
    double tail_T;    // drop threshold (0 = never)
};

template <int TH>
__global__ void __launch_bounds__(64)
k_render(RenderArgs a) {
    __shared__ double acc[TH * TILE_W];
    __shared__ CompTab T;
    const int lane = threadIdx.x;
    const int tile = blockIdx.x;
    const int per_band = a.ntx * a.nty;
    const int b = tile / per_band;
    const int t = tile - b * per_band;
    const int ty = t / a.ntx, tx = t - ty * a.ntx;
    const int X0 = tx * TILE_W, Y0 = ty * TH;
    const int xi = X0 + lane;
    const double x = (double)xi;
    const BandDev *bd = a.bands + b;

#pragma unroll
    for (int r = 0; r < TH; r++) acc[r * TILE_W + lane] = 0.0;

    const int cnt = a.tile_cnt[tile];
    const int64_t off = a.tile_off[tile];
    const SrcRec *recs = a.recs + (int64_t)b * a.S;
    const double Tdrop = a.tail_T;

    for (int e = 0; e < cnt; e++) {
        int64_t at = off + e;
        if (at >= a.capacity) break;
        const int s = __builtin_amdgcn_readfirstlane(a.lists[at]);
        const SrcRec *rp = recs + s;
        const int type = rp->type;
        const int K = (type == 0) ? K_PSF : K_GAL;
        const int bx0 = rp->x0, bx1 = rp->x1, by0 = rp->y0, by1 = rp->y1;
        __syncthreads();   // previous source's table reads are done
        if (lane < K)
            build_comp(T, lane, type, rp->px, rp->py, rp->scale, rp->w00, rp->w01, rp->w11, rp->theta, bd);
        __syncthreads();
        const int ra = max(by0, Y0) - Y0, rb = min(by1, Y0 + TH) - Y0;
        const bool on = (xi >= bx0) && (xi < bx1);
        if (a.variant == 0) {
            for (int row = ra; row < rb; row++) {
                double v = eval_direct(T, K, x, (double)(Y0 + row));
                if (on) acc[row * TILE_W + lane] += v;
            }
        } else {
            // the part of this tile the source's box covers, for the drop test
            const double xa = (double)max(bx0, X0), xb = (double)(min(bx1, X0 + TILE_W) - 1);
            const double ya = (double)(Y0 + ra), yb = (double)(Y0 + rb - 1);
            for (int k0 = 0; k0 < K; k0 += REC_G) {
                const int kn = min(REC_G, K - k0);
                // group-uniform segment length and drop decision
                int L = 1 << 20;
                bool any = false;
                double qcmax = 0.0;
                for (int i = 0; i < kn; i++) {
                    int k = k0 + i;
                    double ddx = dist_to_interval(T.mx[k], xa, xb), ddy = dist_to_interval(T.my[k], ya, yb);
                    double qmin = fmax(ddx * ddx * T.ixx[k], ddy * ddy * T.iyy[k]);
                    bool keep = (Tdrop <= 0.0) || (0.5 * qmin <= Tdrop);
                    any = any || keep;
                    qcmax = fmax(qcmax, T.qc[k]);
                }
                if (!any) continue;
                L = seg_len(qcmax, Tdrop > 0.0 ? Tdrop : 100.0);
                if (L < 4) {
                    // pathologically sharp component: evaluate this group directly
                    for (int row = ra; row < rb; row++) {
                        double sum = 0.0;
                        for (int i = 0; i < kn; i++) {
                            int k = k0 + i;
                            double dx = x - T.mx[k], dy = (double)(Y0 + row) - T.my[k];
                            double q = T.qa[k] * dx * dx + 2.0 * T.qb[k] * dx * dy + T.qc[k] * dy * dy;
                            sum += T.A[k] * exp(-0.5 * q);
                        }
                        if (on) acc[row * TILE_W + lane] += sum;
                    }
                } else {
                    rec_group<REC_G>(T, k0, kn, x, Y0, ra, rb, L, on, acc + lane);
                }
            }
        }
    }

    // epilogue: lambda = eps + acc, written once (512-B coalesced rows); fused Poisson term
    const double eps = bd->eps;
    const bool store = !(a.flags & CEL_RENDER_NO_STORE);
    const bool ll = (a.flags & CEL_RENDER_LOGLIK) != 0;
    double part = 0.0;
    const int64_t plane = (int64_t)b * a.H * a.W;
    if (xi < a.W) {
#pragma unroll 4
        for (int r = 0; r < TH; r++) {
            int y = Y0 + r;
            if (y < a.H) {
                double lam = eps + acc[r * TILE_W + lane];
                int64_t idx = plane + (int64_t)y * a.W + xi;
                if (store) a.lambda[idx] = lam;
                if (ll) part += a.nelec[idx] * log(lam) - lam;
            }
        }
    }
    if (ll) {
        part = wave_sum(part);
        if (lane == 0) a.partials[tile] = part;
    }
}

// fixed-order reduction of the per-tile partials: one block per band
__global__ void __launch_bounds__(256)
k_reduce(const double *__restrict__ partials, int per_band, double *__restrict__ ll_band) {
    __shared__ double sm[256];
    int b = blockIdx.x;
    const double *p = partials + (int64_t)b * per_band;
    double s = 0.0, c = 0.0;   // Kahan per thread, fixed stride
    for (int i = threadIdx.x; i < per_band; i += 256) {
        double y = p[i] - c;
        double tsum = s + y;
        c = (tsum - s) - y;
        s = tsum;
    }
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) ll_band[b] = sm[0];
}

// ------------------------------------------------------------------------------------------
// k_stamps: one wave per (source, 64-column strip, row chunk) job
// ------------------------------------------------------------------------------------------
struct StampJob { int src; int x0; int y0; int y1; };   // strip starts at column x0; rows [y0,y1)

__global__ void __launch_bounds__(64)
k_stamps(const BandDev *__restrict__ bands, int band, const SrcRec *__restrict__ recs,
         const StampJob *__restrict__ jobs, const int4 *__restrict__ obox,
         const int64_t *__restrict__ offsets, int scaled, double *__restrict__ out) {
    __shared__ CompTab T;
    const int lane = threadIdx.x;
    const StampJob jb = jobs[blockIdx.x];
    const SrcRec *rp = recs + jb.src;
    const int4 ob = obox[jb.src];           // output box: x0, x1, y0, y1
    int type = rp->type;
    if (type < 0) type = -1 - type;         // caller-imposed limits: still a valid source kind
    const int K = (type == 0) ? K_PSF : K_GAL;
    const BandDev *bd = bands + band;
    if (lane < K)
        build_comp(T, lane, type, rp->px, rp->py, scaled ? rp->scale : 1.0, rp->w00, rp->w01, rp->w11,
                   rp->theta, bd);
    __syncthreads();
    const int xi = jb.x0 + lane;
    if (xi >= ob.y) return;
    const int nx = ob.y - ob.x;
    double *o = out + offsets[jb.src];
    for (int y = jb.y0; y < jb.y1; y++) {
        double v = eval_direct(T, K, (double)xi, (double)y);
        o[(int64_t)(y - ob.z) * nx + (xi - ob.x)] = v;
    }
}

// ------------------------------------------------------------------------------------------
// k_gmm: generic evaluator, one thread per point, components staged through LDS
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_gmm(const double *__restrict__ x, int64_t N, const double *__restrict__ comp /* K*6: A,mx,my,qa,qb,qc */,
      int K, double *__restrict__ probs) {
    __shared__ double sc[64 * 6];
    int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double px = 0.0, py = 0.0;
    if (n < N) { px = x[2 * n]; py = x[2 * n + 1]; }
    double s = 0.0;
    for (int k0 = 0; k0 < K; k0 += 64) {
        int kn = min(64, K - k0);
        __syncthreads();
        for (int i = threadIdx.x; i < kn * 6; i += blockDim.x) sc[i] = comp[(int64_t)k0 * 6 + i];
        __syncthreads();
        for (int k = 0; k < kn; k++) {
            double dx = px - sc[k * 6 + 1], dy = py - sc[k * 6 + 2];
            double q = sc[k * 6 + 3] * dx * dx + 2.0 * sc[k * 6 + 4] * dx * dy + sc[k * 6 + 5] * dy * dy;
            s += sc[k * 6 + 0] * exp(-0.5 * q);
        }
    }
    if (n < N) probs[n] = s;
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
struct Prof {
    hipEvent_t *ev = nullptr;   // pairs
    int cap = 0, used = 0;
    std::vector<int> kid;
    double sum_ms[CEL_K_COUNT] = {0};
    int64_t n[CEL_K_COUNT] = {0};
};

struct cel_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int variant = 1;
    double tail_T = 60.0;
    bool profile = false;
    Prof prof;
    double *pinned = nullptr;   // MAX_BANDS + 8 doubles of pinned host memory for readbacks
};

struct cel_images {
    cel_ctx *ctx = nullptr;
    int B = 0, H = 0, W = 0;   // H = rows held on the device (the window height)
    int full_H = 0, win_y0 = 0;  // the window is rows [win_y0, win_y0 + H) of a full_H-row frame
    int TH = 32, ntx = 0, nty = 0;
    cel_band hb[MAX_BANDS];
    BandDev *d_bands = nullptr;
    double *d_nelec = nullptr, *d_lambda = nullptr, *d_partials = nullptr, *d_llband = nullptr;
    bool have_nelec = false;
    // per-render scratch, grown on demand
    SrcRec *d_recs = nullptr;
    int4 *d_boxes = nullptr;
    int64_t recs_cap = 0;
    int *d_tile_cnt = nullptr;
    int64_t *d_tile_off = nullptr;
    unsigned long long *d_cursor = nullptr;   // [0] cursor, [1] overflow flag (as int)
    int *d_lists = nullptr;
    int64_t lists_cap = 0;
    double *d_stats = nullptr;
    int64_t last_S = 0;
    double last_entries = 0;
};

struct cel_sources {
    cel_ctx *ctx = nullptr;
    int64_t cap = 0, S = 0;
    int B = 0;
    int *d_type = nullptr;
    double *d_radec = nullptr, *d_counts = nullptr, *d_shape = nullptr;
};

static int prof_begin(cel_ctx *c, int k) {
    if (!c->profile) return -1;
    Prof &p = c->prof;
    if (p.used + 2 > p.cap) {
        int ncap = p.cap ? p.cap * 2 : 256;
        hipEvent_t *ne = (hipEvent_t *)realloc(p.ev, sizeof(hipEvent_t) * ncap);
        if (!ne) return -1;
        p.ev = ne;
        for (int i = p.cap; i < ncap; i++)
            if (hipEventCreate(&p.ev[i]) != hipSuccess) return -1;
        p.cap = ncap;
    }
    int i = p.used;
    p.used += 2;
    p.kid.push_back(k);
    (void)hipEventRecord(p.ev[i], c->stream);
    return i;
}
static void prof_end(cel_ctx *c, int i) {
    if (i >= 0) (void)hipEventRecord(c->prof.ev[i + 1], c->stream);
}
static void prof_collect(cel_ctx *c) {
    Prof &p = c->prof;
    for (int i = 0; i < p.used; i += 2) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.ev[i], p.ev[i + 1]) == hipSuccess) {
            int k = p.kid[i / 2];
            p.sum_ms[k] += ms;
            p.n[k] += 1;
        }
    }
    p.used = 0;
    p.kid.clear();
}

static double host_bounding_radius(const double *mu, const double *cov, int K, double error,
                                   const double *center) {
    double q = 1.0 - error;
    double rsq = -2.0 * log1p(-q);   // chi2.ppf(1 - error, 2)
    double best = -INFINITY;
    for (int i = 0; i < K; i++) {
        const double *c = cov + 4 * i;
        double s1 = sqrt(c[0]), s2 = sqrt(c[3]);
        double rho = c[1] / (s1 * s2);
        double A11 = s1, A21 = rho * s2, A22 = s2 * sqrt(1.0 - rho * rho);
        double An = 1.0 / rsq * (1.0 / (A11 * A11) + (A21 * A21) / (A22 * A22));
        double Bn = 1.0 / rsq * (-2.0 * A21 / (A11 * (A22 * A22)));
        double Cn = 1.0 / rsq * 1.0 / (A22 * A22);
        double maj = pow(0.5 * (An + Cn - sqrt(Bn * Bn + (An - Cn) * (An - Cn))), -0.5);
        double d0 = mu[2 * i] - (center ? center[0] : 0.0), d1 = mu[2 * i + 1] - (center ? center[1] : 0.0);
        double cand = maj + sqrt(d0 * d0 + d1 * d1);
        if (cand > best) best = cand;
    }
    return best;
}

static void band_to_dev(const cel_band &h, BandDev &d) {
    d.eps = h.eps;
    for (int k = 0; k < K_PSF; k++) {
        d.w[k] = h.w[k];
        d.mux[k] = h.mu[2 * k]; d.muy[k] = h.mu[2 * k + 1];
        d.cxx[k] = h.cov[4 * k]; d.cxy[k] = h.cov[4 * k + 1]; d.cyy[k] = h.cov[4 * k + 3];
    }
    for (int i = 0; i < 2; i++) { d.rho[i] = h.rho[i]; d.phi[i] = h.phi[i]; }
    for (int i = 0; i < 4; i++) { d.ups[i] = h.ups[i]; d.ups_inv[i] = h.ups_inv[i]; }
    d.R = h.R;
}

static int copy_in(void *dst, const void *src, size_t bytes, int mem, hipStream_t st) {
    if (bytes == 0) return CEL_OK;
    if (mem == CEL_DEVICE) {
        HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st));
    } else {
        HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));   // pageable source must stay valid
    }
    return CEL_OK;
}

static int copy_out(void *dst, const void *src, size_t bytes, int mem, hipStream_t st) {
    if (bytes == 0) return CEL_OK;
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, mem == CEL_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return CEL_OK;
}

extern "C" {

int cel_abi_version(void) { return CEL_ABI_VERSION; }
const char *cel_last_error(void) { return g_err; }

int cel_device_count(int *n) {
    if (!n) return fail(CEL_ERR_INVALID, "cel_device_count: null output");
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) c = 0;
    *n = c;
    return CEL_OK;
}

int cel_ctx_create(int device, void *stream, cel_ctx **out) {
    if (!out) return fail(CEL_ERR_INVALID, "cel_ctx_create: null output");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(CEL_ERR_NO_DEVICE, "no HIP device visible: this library has no CPU fallback");
    if (device < 0 || device >= n) return fail(CEL_ERR_INVALID, "device %d out of range [0,%d)", device, n);
    HIP_TRY(hipSetDevice(device));
    cel_ctx *c = new (std::nothrow) cel_ctx();
    if (!c) return fail(CEL_ERR_NOMEM, "out of host memory");
    c->device = device;
    if (stream) {
        c->stream = (hipStream_t)stream;
    } else {
        hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { delete c; return fail(CEL_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
        c->own_stream = true;
    }
    hipError_t e = hipHostMalloc((void **)&c->pinned, sizeof(double) * (MAX_BANDS + 8), hipHostMallocDefault);
    if (e != hipSuccess) { delete c; return fail(CEL_ERR_HIP, "hipHostMalloc: %s", hipGetErrorString(e)); }
    // profile constants, normalised as mixture_profiles.py:13,19
    double amp[K_PROF], var[K_PROF], se = 0.0, sd = 0.0;
    for (int i = 0; i < 6; i++) se += H_EXP_AMP[i];
    for (int i = 0; i < 8; i++) sd += H_DEV_AMP[i];
    for (int i = 0; i < 6; i++) { amp[i] = H_EXP_AMP[i] / se; var[i] = H_EXP_VAR[i]; }
    for (int i = 0; i < 8; i++) { amp[6 + i] = H_DEV_AMP[i] / sd; var[6 + i] = H_DEV_VAR[i]; }
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(c_prof_amp), amp, sizeof(amp)));
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(c_prof_var), var, sizeof(var)));
    *out = c;
    return CEL_OK;
}

int cel_ctx_destroy(cel_ctx *c) {
    if (!c) return CEL_OK;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (int i = 0; i < c->prof.cap; i++) (void)hipEventDestroy(c->prof.ev[i]);
    free(c->prof.ev);
    if (c->pinned) (void)hipHostFree(c->pinned);
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return CEL_OK;
}

int cel_ctx_set_stream(cel_ctx *c, void *stream) {
    if (!c) return fail(CEL_ERR_INVALID, "null context");
    (void)hipStreamSynchronize(c->stream);
    if (c->own_stream) { (void)hipStreamDestroy(c->stream); c->own_stream = false; }
    if (stream) {
        c->stream = (hipStream_t)stream;
    } else {
        HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        c->own_stream = true;
    }
    return CEL_OK;
}

int cel_ctx_synchronize(cel_ctx *c) {
    if (!c) return fail(CEL_ERR_INVALID, "null context");
    HIP_TRY(hipStreamSynchronize(c->stream));
    return CEL_OK;
}

int cel_ctx_set_option(cel_ctx *c, int key, double v) {
    if (!c) return fail(CEL_ERR_INVALID, "null context");
    switch (key) {
    case CEL_OPT_KERNEL:
        if (v != 0.0 && v != 1.0) return fail(CEL_ERR_INVALID, "CEL_OPT_KERNEL must be 0 or 1");
        c->variant = (int)v;
        return CEL_OK;
    case CEL_OPT_TAIL_LOG:
        if (!(v >= 0.0) || v > 300.0) return fail(CEL_ERR_INVALID, "CEL_OPT_TAIL_LOG must be in [0, 300]");
        c->tail_T = v;
        return CEL_OK;
    case CEL_OPT_PROFILE:
        c->profile = (v != 0.0);
        return CEL_OK;
    }
    return fail(CEL_ERR_INVALID, "unknown option %d", key);
}

int cel_ctx_get_option(cel_ctx *c, int key, double *v) {
    if (!c || !v) return fail(CEL_ERR_INVALID, "null argument");
    switch (key) {
    case CEL_OPT_KERNEL: *v = c->variant; return CEL_OK;
    case CEL_OPT_TAIL_LOG: *v = c->tail_T; return CEL_OK;
    case CEL_OPT_PROFILE: *v = c->profile ? 1.0 : 0.0; return CEL_OK;
    }
    return fail(CEL_ERR_INVALID, "unknown option %d", key);
}

// ---- images ---------------------------------------------------------------------------------
int cel_images_destroy(cel_images *im) {
    if (!im) return CEL_OK;
    (void)hipSetDevice(im->ctx->device);
    (void)hipStreamSynchronize(im->ctx->stream);
    void *ptrs[] = {im->d_bands, im->d_nelec, im->d_lambda, im->d_partials, im->d_llband, im->d_recs,
                    im->d_boxes, im->d_tile_cnt, im->d_tile_off, im->d_cursor, im->d_lists, im->d_stats};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    delete im;
    return CEL_OK;
}

int cel_images_create(cel_ctx *c, int B, int H, int W, const cel_band *bands, cel_images **out) {
    if (!c || !bands || !out) return fail(CEL_ERR_INVALID, "cel_images_create: null argument");
    if (B < 1 || B > MAX_BANDS) return fail(CEL_ERR_INVALID, "B=%d out of range [1,%d]", B, MAX_BANDS);
    if (H < 1 || W < 1 || (int64_t)H * W > (int64_t)1 << 34) return fail(CEL_ERR_INVALID, "bad image size %dx%d", H, W);
    HIP_TRY(hipSetDevice(c->device));
    cel_images *im = new (std::nothrow) cel_images();
    if (!im) return fail(CEL_ERR_NOMEM, "out of host memory");
    im->ctx = c; im->B = B; im->H = H; im->W = W;
    im->full_H = H; im->win_y0 = 0;
    im->TH = 32;
    im->ntx = (W + TILE_W - 1) / TILE_W;
    im->nty = (H + im->TH - 1) / im->TH;
    BandDev hb[MAX_BANDS];
    for (int b = 0; b < B; b++) {
        im->hb[b] = bands[b];
        for (int k = 0; k < K_PSF; k++) {
            const double *cv = bands[b].cov + 4 * k;
            double det = cv[0] * cv[3] - cv[1] * cv[2];
            if (!(cv[0] > 0) || !(cv[3] > 0) || !(det > 0)) {
                delete im;
                return fail(CEL_ERR_INVALID, "band %d: PSF component %d covariance is not positive definite", b, k);
            }
        }
        if (!(im->hb[b].R > 0.0))
            im->hb[b].R = host_bounding_radius(bands[b].mu, bands[b].cov, K_PSF, 0.001, nullptr);
        band_to_dev(im->hb[b], hb[b]);
    }
    size_t npix = (size_t)B * H * W;
    int T = B * im->ntx * im->nty;
    int rc = CEL_OK;
#define IM_TRY(expr)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) {                                                               \
            rc = fail(e_ == hipErrorOutOfMemory ? CEL_ERR_NOMEM : CEL_ERR_HIP, "%s: %s", #expr, \
                      hipGetErrorString(e_));                                                 \
            goto bad;                                                                         \
        }                                                                                     \
    } while (0)
    IM_TRY(hipMalloc((void **)&im->d_bands, sizeof(BandDev) * B));
    IM_TRY(hipMemcpy(im->d_bands, hb, sizeof(BandDev) * B, hipMemcpyHostToDevice));
    IM_TRY(hipMalloc((void **)&im->d_nelec, sizeof(double) * npix));
    IM_TRY(hipMalloc((void **)&im->d_lambda, sizeof(double) * npix));
    IM_TRY(hipMalloc((void **)&im->d_partials, sizeof(double) * T));
    IM_TRY(hipMalloc((void **)&im->d_llband, sizeof(double) * MAX_BANDS));
    IM_TRY(hipMalloc((void **)&im->d_tile_cnt, sizeof(int) * T));
    IM_TRY(hipMalloc((void **)&im->d_tile_off, sizeof(int64_t) * T));
    IM_TRY(hipMalloc((void **)&im->d_cursor, sizeof(unsigned long long) * 2));
    IM_TRY(hipMalloc((void **)&im->d_stats, sizeof(double) * 2));
    IM_TRY(hipMemsetAsync(im->d_lambda, 0, sizeof(double) * npix, c->stream));
#undef IM_TRY
    *out = im;
    return CEL_OK;
bad:
    cel_images_destroy(im);
    return rc;
}

int cel_images_set_nelec(cel_images *im, const double *nelec, int mem) {
    if (!im || !nelec) return fail(CEL_ERR_INVALID, "cel_images_set_nelec: null argument");
    HIP_TRY(hipSetDevice(im->ctx->device));
    int rc = copy_in(im->d_nelec, nelec, sizeof(double) * (size_t)im->B * im->H * im->W, mem, im->ctx->stream);
    if (rc == CEL_OK) im->have_nelec = true;
    return rc;
}

int cel_images_set_epsilon(cel_images *im, int band, double eps) {
    if (!im || band < 0 || band >= im->B) return fail(CEL_ERR_INVALID, "cel_images_set_epsilon: bad band");
    HIP_TRY(hipSetDevice(im->ctx->device));
    im->hb[band].eps = eps;
    im->ctx->pinned[MAX_BANDS] = eps;
    HIP_TRY(hipStreamSynchronize(im->ctx->stream));
    HIP_TRY(hipMemcpyAsync(&im->d_bands[band].eps, &im->ctx->pinned[MAX_BANDS], sizeof(double),
                           hipMemcpyHostToDevice, im->ctx->stream));
    HIP_TRY(hipStreamSynchronize(im->ctx->stream));
    return CEL_OK;
}

int cel_images_set_window(cel_images *im, int y0, int full_H) {
    if (!im) return fail(CEL_ERR_INVALID, "null images");
    if (y0 < 0 || full_H < 1 || (int64_t)y0 + im->H > full_H)
        return fail(CEL_ERR_INVALID, "window rows [%d, %d) do not fit a %d-row frame", y0, y0 + im->H, full_H);
    im->win_y0 = y0;
    im->full_H = full_H;
    return CEL_OK;
}

int cel_images_get_band(cel_images *im, int band, cel_band *out) {
    if (!im || !out || band < 0 || band >= im->B) return fail(CEL_ERR_INVALID, "cel_images_get_band: bad argument");
    *out = im->hb[band];
    return CEL_OK;
}

int cel_images_get_lambda(cel_images *im, double *out, int mem) {
    if (!im || !out) return fail(CEL_ERR_INVALID, "cel_images_get_lambda: null argument");
    HIP_TRY(hipSetDevice(im->ctx->device));
    return copy_out(out, im->d_lambda, sizeof(double) * (size_t)im->B * im->H * im->W, mem, im->ctx->stream);
}

int cel_images_device_ptrs(cel_images *im, void **nelec, void **lambda) {
    if (!im) return fail(CEL_ERR_INVALID, "null images");
    if (nelec) *nelec = im->d_nelec;
    if (lambda) *lambda = im->d_lambda;
    return CEL_OK;
}

// ---- sources --------------------------------------------------------------------------------
int cel_sources_destroy(cel_sources *s) {
    if (!s) return CEL_OK;
    (void)hipSetDevice(s->ctx->device);
    (void)hipStreamSynchronize(s->ctx->stream);
    void *ptrs[] = {s->d_type, s->d_radec, s->d_counts, s->d_shape};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    delete s;
    return CEL_OK;
}

int cel_sources_create(cel_ctx *c, int64_t capacity, int B, cel_sources **out) {
    if (!c || !out) return fail(CEL_ERR_INVALID, "cel_sources_create: null argument");
    if (capacity < 1 || capacity > ((int64_t)1 << 30)) return fail(CEL_ERR_INVALID, "bad capacity");
    if (B < 1 || B > MAX_BANDS) return fail(CEL_ERR_INVALID, "B=%d out of range", B);
    HIP_TRY(hipSetDevice(c->device));
    cel_sources *s = new (std::nothrow) cel_sources();
    if (!s) return fail(CEL_ERR_NOMEM, "out of host memory");
    s->ctx = c; s->cap = capacity; s->B = B;
    hipError_t e;
    if ((e = hipMalloc((void **)&s->d_type, sizeof(int) * capacity)) != hipSuccess ||
        (e = hipMalloc((void **)&s->d_radec, sizeof(double) * 2 * capacity)) != hipSuccess ||
        (e = hipMalloc((void **)&s->d_counts, sizeof(double) * B * capacity)) != hipSuccess ||
        (e = hipMalloc((void **)&s->d_shape, sizeof(double) * 4 * capacity)) != hipSuccess) {
        cel_sources_destroy(s);
        return fail(CEL_ERR_NOMEM, "hipMalloc(sources): %s", hipGetErrorString(e));
    }
    *out = s;
    return CEL_OK;
}

int cel_sources_set(cel_sources *s, int64_t S, const int32_t *type, const double *radec,
                    const double *counts, const double *shape, int mem) {
    if (!s || !type || !radec || !counts || !shape) return fail(CEL_ERR_INVALID, "cel_sources_set: null argument");
    if (S < 0 || S > s->cap) return fail(CEL_ERR_INVALID, "S=%lld exceeds capacity %lld", (long long)S, (long long)s->cap);
    HIP_TRY(hipSetDevice(s->ctx->device));
    hipStream_t st = s->ctx->stream;
    int rc;
    if ((rc = copy_in(s->d_type, type, sizeof(int) * S, mem, st))) return rc;
    if ((rc = copy_in(s->d_radec, radec, sizeof(double) * 2 * S, mem, st))) return rc;
    if ((rc = copy_in(s->d_counts, counts, sizeof(double) * s->B * S, mem, st))) return rc;
    if ((rc = copy_in(s->d_shape, shape, sizeof(double) * 4 * S, mem, st))) return rc;
    s->S = S;
    return CEL_OK;
}

// ---- prep + bin (shared by field and stamps) ------------------------------------------------
static int ensure_recs(cel_images *im, int64_t n) {
    if (n <= im->recs_cap) return CEL_OK;
    HIP_TRY(hipStreamSynchronize(im->ctx->stream));
    if (im->d_recs) (void)hipFree(im->d_recs);
    if (im->d_boxes) (void)hipFree(im->d_boxes);
    im->d_recs = nullptr; im->d_boxes = nullptr; im->recs_cap = 0;
    int64_t cap = n + n / 4 + 64;
    HIP_TRY(hipMalloc((void **)&im->d_recs, sizeof(SrcRec) * cap));
    HIP_TRY(hipMalloc((void **)&im->d_boxes, sizeof(int4) * cap));
    im->recs_cap = cap;
    return CEL_OK;
}

static int ensure_lists(cel_images *im, int64_t n) {
    if (n <= im->lists_cap) return CEL_OK;
    HIP_TRY(hipStreamSynchronize(im->ctx->stream));
    if (im->d_lists) (void)hipFree(im->d_lists);
    im->d_lists = nullptr; im->lists_cap = 0;
    HIP_TRY(hipMalloc((void **)&im->d_lists, sizeof(int) * n));
    im->lists_cap = n;
    return CEL_OK;
}

static double rsq_galaxy() {
    double q = 1.0 - 1e-5;           // celeste_galaxy_conditionals.py:207 error=1e-5
    return -2.0 * log1p(-q);         // scipy.stats.chi2.ppf(q, 2)
}

static int run_prep(cel_images *im, cel_sources *src) {
    cel_ctx *c = im->ctx;
    int64_t n = src->S * im->B;
    int rc = ensure_recs(im, n > 0 ? n : 1);
    if (rc) return rc;
    if (n == 0) return CEL_OK;
    int pi = prof_begin(c, CEL_K_PREP);
    hipLaunchKernelGGL(k_prep, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, im->d_bands, im->B,
                       im->full_H, im->W, im->win_y0, im->H, src->S, src->d_type, src->d_radec, src->d_counts, src->d_shape,
                       rsq_galaxy(), im->d_recs, im->d_boxes);
    prof_end(c, pi);
    HIP_TRY(hipGetLastError());
    return CEL_OK;
}

int cel_render_field(cel_images *im, cel_sources *src, int flags, double *ll_band, double *ll_total) {
    if (!im || !src) return fail(CEL_ERR_INVALID, "cel_render_field: null argument");
    if (src->ctx != im->ctx) return fail(CEL_ERR_INVALID, "images and sources belong to different contexts");
    if (src->B != im->B) return fail(CEL_ERR_INVALID, "sources carry %d bands, images %d", src->B, im->B);
    if ((ll_band || ll_total) && !(flags & CEL_RENDER_LOGLIK))
        return fail(CEL_ERR_INVALID, "log-likelihood outputs requested without CEL_RENDER_LOGLIK");
    if ((flags & CEL_RENDER_LOGLIK) && !im->have_nelec)
        return fail(CEL_ERR_INVALID, "CEL_RENDER_LOGLIK needs cel_images_set_nelec first");
    cel_ctx *c = im->ctx;
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    const int64_t S = src->S;
    const int T = im->B * im->ntx * im->nty;
    int rc = run_prep(im, src);
    if (rc) return rc;
    if (im->lists_cap == 0) {
        // first guess: every (band, source) touches ~6 tiles; grown on overflow below
        rc = ensure_lists(im, (S * im->B) * 6 + 1024);
        if (rc) return rc;
    }
    for (int attempt = 0; attempt < 8; attempt++) {
        HIP_TRY(hipMemsetAsync(im->d_cursor, 0, sizeof(unsigned long long) * 2, st));
        int pi = prof_begin(c, CEL_K_BIN);
        hipLaunchKernelGGL(k_bin, dim3(T), dim3(64), 0, st, im->d_boxes, S, im->ntx, im->nty, im->TH, 0,
                           im->d_tile_cnt, im->d_tile_off, im->d_cursor, (int *)nullptr, im->lists_cap,
                           (int *)(im->d_cursor + 1));
        hipLaunchKernelGGL(k_bin, dim3(T), dim3(64), 0, st, im->d_boxes, S, im->ntx, im->nty, im->TH, 1,
                           im->d_tile_cnt, im->d_tile_off, im->d_cursor, im->d_lists, im->lists_cap,
                           (int *)(im->d_cursor + 1));
        prof_end(c, pi);
        RenderArgs a;
        a.bands = im->d_bands; a.recs = im->d_recs; a.lists = im->d_lists; a.tile_cnt = im->d_tile_cnt;
        a.tile_off = im->d_tile_off; a.nelec = im->d_nelec; a.lambda = im->d_lambda; a.partials = im->d_partials;
        a.S = S; a.capacity = im->lists_cap; a.B = im->B; a.H = im->H; a.W = im->W; a.ntx = im->ntx; a.nty = im->nty;
        a.flags = flags; a.variant = c->variant; a.tail_T = c->tail_T;
        pi = prof_begin(c, CEL_K_RENDER);
        hipLaunchKernelGGL((k_render<32>), dim3(T), dim3(64), 0, st, a);
        prof_end(c, pi);
        if (flags & CEL_RENDER_LOGLIK) {
            pi = prof_begin(c, CEL_K_REDUCE);
            hipLaunchKernelGGL(k_reduce, dim3(im->B), dim3(256), 0, st, im->d_partials, im->ntx * im->nty, im->d_llband);
            prof_end(c, pi);
            HIP_TRY(hipMemcpyAsync(c->pinned, im->d_llband, sizeof(double) * im->B, hipMemcpyDeviceToHost, st));
        }
        // total list length + overflow flag ride back with the result
        HIP_TRY(hipMemcpyAsync(c->pinned + MAX_BANDS + 2, im->d_cursor, sizeof(unsigned long long) * 2,
                               hipMemcpyDeviceToHost, st));
        HIP_TRY(hipGetLastError());
        im->last_S = S;
        HIP_TRY(hipStreamSynchronize(st));
        unsigned long long total, ovf;
        memcpy(&total, c->pinned + MAX_BANDS + 2, sizeof(total));
        memcpy(&ovf, c->pinned + MAX_BANDS + 3, sizeof(ovf));
        im->last_entries = (double)total;
        if ((ovf & 0xffffffffull) == 0 && (int64_t)total <= im->lists_cap) break;
        rc = ensure_lists(im, (int64_t)total + (int64_t)total / 4 + 1024);   // rerun with room
        if (rc) return rc;
        if (attempt == 7) return fail(CEL_ERR_HIP, "tile lists kept overflowing");
    }
    if (flags & CEL_RENDER_LOGLIK) {
        double tot = 0.0;
        for (int b = 0; b < im->B; b++) {
            if (ll_band) ll_band[b] = c->pinned[b];
            tot += c->pinned[b];
        }
        if (ll_total) *ll_total = tot;
    }
    return CEL_OK;
}

int cel_field_stats(cel_images *im, double *n_srcpix, double *n_gauss, double *n_tile_entries) {
    if (!im) return fail(CEL_ERR_INVALID, "null images");
    cel_ctx *c = im->ctx;
    HIP_TRY(hipSetDevice(c->device));
    int64_t n = im->last_S * im->B;
    HIP_TRY(hipMemsetAsync(im->d_stats, 0, sizeof(double) * 2, c->stream));
    if (n > 0)
        hipLaunchKernelGGL(k_stats, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, im->d_recs, n, im->d_stats);
    HIP_TRY(hipMemcpyAsync(c->pinned + MAX_BANDS + 4, im->d_stats, sizeof(double) * 2, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(c->pinned + MAX_BANDS + 2, im->d_cursor, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (n_srcpix) *n_srcpix = c->pinned[MAX_BANDS + 4];
    if (n_gauss) *n_gauss = c->pinned[MAX_BANDS + 5];
    unsigned long long total;
    memcpy(&total, c->pinned + MAX_BANDS + 2, sizeof(total));
    if (n_tile_entries) *n_tile_entries = (double)total;
    return CEL_OK;
}

// ---- stamps ---------------------------------------------------------------------------------
// prep for ONE band: records are laid out [band][source]; the band's slice is reused.
int cel_stamp_boxes(cel_images *im, cel_sources *src, int band, int32_t *boxes, int32_t *status) {
    if (!im || !src || !boxes || !status) return fail(CEL_ERR_INVALID, "cel_stamp_boxes: null argument");
    if (band < 0 || band >= im->B) return fail(CEL_ERR_INVALID, "band %d out of range", band);
    if (src->B != im->B || src->ctx != im->ctx) return fail(CEL_ERR_INVALID, "sources do not match images");
    cel_ctx *c = im->ctx;
    HIP_TRY(hipSetDevice(c->device));
    int rc = run_prep(im, src);
    if (rc) return rc;
    im->last_S = src->S;
    int64_t S = src->S;
    std::vector<SrcRec> h((size_t)S);
    if (S) {
        HIP_TRY(hipMemcpyAsync(h.data(), im->d_recs + (int64_t)band * S, sizeof(SrcRec) * S, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    for (int64_t s = 0; s < S; s++) {
        boxes[4 * s + 0] = h[s].y0; boxes[4 * s + 1] = h[s].y1;
        boxes[4 * s + 2] = h[s].x0; boxes[4 * s + 3] = h[s].x1;
        status[s] = h[s].type >= 0 ? 1 : 0;
    }
    return CEL_OK;
}

int cel_render_stamps(cel_images *im, cel_sources *src, int band, int scaled, const int32_t *boxes_in,
                      const int64_t *offsets, double *out, int mem) {
    if (!im || !src || !offsets || !out) return fail(CEL_ERR_INVALID, "cel_render_stamps: null argument");
    if (band < 0 || band >= im->B) return fail(CEL_ERR_INVALID, "band %d out of range", band);
    if (src->B != im->B || src->ctx != im->ctx) return fail(CEL_ERR_INVALID, "sources do not match images");
    cel_ctx *c = im->ctx;
    HIP_TRY(hipSetDevice(c->device));
    int64_t S = src->S;
    if (S == 0) return CEL_OK;
    std::vector<int32_t> hb((size_t)S * 4), hs((size_t)S);
    int rc = cel_stamp_boxes(im, src, band, hb.data(), hs.data());
    if (rc) return rc;
    std::vector<int4> obox((size_t)S);
    std::vector<StampJob> jobs;
    const int ROWS = 64;
    for (int64_t s = 0; s < S; s++) {
        int y0, y1, x0, x1;
        bool ok;
        if (boxes_in) {
            y0 = boxes_in[4 * s]; y1 = boxes_in[4 * s + 1]; x0 = boxes_in[4 * s + 2]; x1 = boxes_in[4 * s + 3];
            ok = (y1 > y0 && x1 > x0);
        } else {
            y0 = hb[4 * s]; y1 = hb[4 * s + 1]; x0 = hb[4 * s + 2]; x1 = hb[4 * s + 3];
            ok = hs[s] != 0;
        }
        obox[s] = make_int4(x0, x1, y0, y1);
        if (!ok) continue;
        int64_t area = (int64_t)(y1 - y0) * (x1 - x0);
        if (offsets[s + 1] - offsets[s] != area)
            return fail(CEL_ERR_INVALID, "offsets[%lld+1]-offsets[%lld] = %lld but the stamp has %lld pixels",
                        (long long)s, (long long)s, (long long)(offsets[s + 1] - offsets[s]), (long long)area);
        for (int xs = x0; xs < x1; xs += TILE_W)
            for (int ys = y0; ys < y1; ys += ROWS)
                jobs.push_back(StampJob{(int)s, xs, ys, ys + ROWS < y1 ? ys + ROWS : y1});
    }
    if (jobs.empty()) return CEL_OK;
    int64_t total = offsets[S];
    StampJob *d_jobs = nullptr;
    int4 *d_obox = nullptr;
    int64_t *d_off = nullptr;
    double *d_out = nullptr;
    rc = CEL_OK;
    hipError_t e;
#define ST_TRY(expr)                                                                     \
    do {                                                                                 \
        e = (expr);                                                                      \
        if (e != hipSuccess) { rc = fail(CEL_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e)); goto done; } \
    } while (0)
    ST_TRY(hipMalloc((void **)&d_jobs, sizeof(StampJob) * jobs.size()));
    ST_TRY(hipMalloc((void **)&d_obox, sizeof(int4) * S));
    ST_TRY(hipMalloc((void **)&d_off, sizeof(int64_t) * (S + 1)));
    if (mem == CEL_DEVICE) d_out = out; else ST_TRY(hipMalloc((void **)&d_out, sizeof(double) * (total > 0 ? total : 1)));
    ST_TRY(hipMemcpyAsync(d_jobs, jobs.data(), sizeof(StampJob) * jobs.size(), hipMemcpyHostToDevice, c->stream));
    ST_TRY(hipMemcpyAsync(d_obox, obox.data(), sizeof(int4) * S, hipMemcpyHostToDevice, c->stream));
    ST_TRY(hipMemcpyAsync(d_off, offsets, sizeof(int64_t) * (S + 1), hipMemcpyHostToDevice, c->stream));
    {
        int pi = prof_begin(c, CEL_K_STAMPS);
        hipLaunchKernelGGL(k_stamps, dim3((unsigned)jobs.size()), dim3(64), 0, c->stream, im->d_bands, band,
                           im->d_recs + (int64_t)band * S, d_jobs, d_obox, d_off, scaled, d_out);
        prof_end(c, pi);
    }
    ST_TRY(hipGetLastError());
    if (mem != CEL_DEVICE) ST_TRY(hipMemcpyAsync(out, d_out, sizeof(double) * total, hipMemcpyDeviceToHost, c->stream));
    ST_TRY(hipStreamSynchronize(c->stream));
#undef ST_TRY
done:
    (void)hipStreamSynchronize(c->stream);
    if (d_jobs) (void)hipFree(d_jobs);
    if (d_obox) (void)hipFree(d_obox);
    if (d_off) (void)hipFree(d_off);
    if (mem != CEL_DEVICE && d_out) (void)hipFree(d_out);
    return rc;
}

// ---- generic evaluator ------------------------------------------------------------------------
int cel_gmm_like_2d(cel_ctx *c, const double *x, int64_t N, const double *ws, const double *mus,
                    const double *sigs, int K, double *probs, int mem) {
    if (!c || !x || !ws || !mus || !sigs || !probs) return fail(CEL_ERR_INVALID, "cel_gmm_like_2d: null argument");
    if (N < 0 || K < 1) return fail(CEL_ERR_INVALID, "Means, covariances and weights must have same first dimension!");
    if (N == 0) return CEL_OK;
    HIP_TRY(hipSetDevice(c->device));
    // gmm_like_fast.pyx:162-176: det, inverse and the normaliser are per-component scalars
    std::vector<double> comp((size_t)K * 6);
    const double log2pi = log(2.0 * PI_D);
    for (int k = 0; k < K; k++) {
        const double *s = sigs + 4 * k;
        double det = s[0] * s[3] - s[1] * s[2];
        comp[6 * k + 0] = exp(-log2pi - 0.5 * log(det)) * ws[k];
        comp[6 * k + 1] = mus[2 * k];
        comp[6 * k + 2] = mus[2 * k + 1];
        comp[6 * k + 3] = s[3] / det;
        comp[6 * k + 4] = -1 * s[1] / det;
        comp[6 * k + 5] = s[0] / det;
    }
    double *d_comp = nullptr, *d_x = nullptr, *d_p = nullptr;
    int rc = CEL_OK;
    hipError_t e;
#define G_TRY(expr)                                                                      \
    do {                                                                                 \
        e = (expr);                                                                      \
        if (e != hipSuccess) { rc = fail(CEL_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e)); goto done; } \
    } while (0)
    G_TRY(hipMalloc((void **)&d_comp, sizeof(double) * 6 * K));
    G_TRY(hipMemcpyAsync(d_comp, comp.data(), sizeof(double) * 6 * K, hipMemcpyHostToDevice, c->stream));
    if (mem == CEL_DEVICE) {
        d_x = const_cast<double *>(x);
        d_p = probs;
    } else {
        G_TRY(hipMalloc((void **)&d_x, sizeof(double) * 2 * N));
        G_TRY(hipMalloc((void **)&d_p, sizeof(double) * N));
        G_TRY(hipMemcpyAsync(d_x, x, sizeof(double) * 2 * N, hipMemcpyHostToDevice, c->stream));
    }
    {
        int pi = prof_begin(c, CEL_K_GMM);
        hipLaunchKernelGGL(k_gmm, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, c->stream, d_x, N, d_comp, K, d_p);
        prof_end(c, pi);
    }
    G_TRY(hipGetLastError());
    if (mem != CEL_DEVICE) G_TRY(hipMemcpyAsync(probs, d_p, sizeof(double) * N, hipMemcpyDeviceToHost, c->stream));
    G_TRY(hipStreamSynchronize(c->stream));
#undef G_TRY
done:
    (void)hipStreamSynchronize(c->stream);
    if (d_comp) (void)hipFree(d_comp);
    if (mem != CEL_DEVICE) {
        if (d_x) (void)hipFree(d_x);
        if (d_p) (void)hipFree(d_p);
    }
    return rc;
}

int cel_bounding_radius(const double *w, const double *mu, const double *cov, int K, double error,
                        const double *center, double *out) {
    (void)w;
    if (!mu || !cov || !out || K < 1) return fail(CEL_ERR_INVALID, "cel_bounding_radius: bad argument");
    if (!(error > 0.0 && error < 1.0)) return fail(CEL_ERR_INVALID, "error must be in (0,1)");
    *out = host_bounding_radius(mu, cov, K, error, center);
    return CEL_OK;
}

// ---- measurement ------------------------------------------------------------------------------
int cel_profile_reset(cel_ctx *c) {
    if (!c) return fail(CEL_ERR_INVALID, "null context");
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->prof.used = 0;
    c->prof.kid.clear();
    for (int k = 0; k < CEL_K_COUNT; k++) { c->prof.sum_ms[k] = 0.0; c->prof.n[k] = 0; }
    return CEL_OK;
}

int cel_profile_get(cel_ctx *c, int kernel, double *mean_ms, int64_t *launches) {
    if (!c || kernel < 0 || kernel >= CEL_K_COUNT) return fail(CEL_ERR_INVALID, "cel_profile_get: bad argument");
    HIP_TRY(hipStreamSynchronize(c->stream));
    prof_collect(c);
    if (mean_ms) *mean_ms = c->prof.n[kernel] ? c->prof.sum_ms[kernel] / (double)c->prof.n[kernel] : 0.0;
    if (launches) *launches = c->prof.n[kernel];
    return CEL_OK;
}

}  // extern "C"
