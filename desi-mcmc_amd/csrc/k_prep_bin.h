// k_prep_bin.h -- source preparation and tile binning kernels
#pragma once
#include "device_common.h"
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

// ---- pieces of k_prep that the fused small-field kernel (k_small_stars.h) evaluates itself: the same expressions, so the
// same positions and boxes ----
// equa2pixel (fits_image.py:166-174); cphi = cos(phi_1 pi / 180)
__device__ __forceinline__ void prep_pixel(const BandDev &bd, double ra, double dec, double cphi, double &px, double &py) {
    double s0 = (ra - bd.phi[0]) * cphi, s1 = dec - bd.phi[1];
    px = (bd.ups_inv[0] * s0 + bd.ups_inv[1] * s1) + bd.rho[0];
    py = (bd.ups_inv[2] * s0 + bd.ups_inv[3] * s1) + bd.rho[1];
}

// a star's record fields from its pixel position: celeste.py:130-140, the overlap test (with the reference's axis mix-up,
// Q1) + the int() box on the H x W frame; type 0, or -3 where the reference returns (None, None, None)
__device__ __forceinline__ void prep_star_box(const BandDev &bd, double px, double py, int H, int W, SrcRec &r) {
    const double BIG = 1073741824.0;
    bool miss = (px < -50 || px > 2.0 * H || py < -50 || px > 2.0 * W);
    if (miss || !(px == px) || !(py == py)) {
        r.type = -3;    // the reference returns (None, None, None) here, whatever the limits
    } else {
        double bound = bd.R;
        int lx = (int)clampd(px - bound, -BIG, BIG), hx = (int)clampd(px + bound + 1, -BIG, BIG);
        int ly = (int)clampd(py - bound, -BIG, BIG), hy = (int)clampd(py + bound + 1, -BIG, BIG);
        r.x0 = max(0, lx); r.x1 = min(hx, W);
        r.y0 = max(0, ly); r.y1 = min(hy, H);
    }
}

// row window [win_y0, win_y0 + win_h) of the H-row frame (strip partition across GPUs): boxes are formed against the FULL
// frame, then cut to the window and re-based, so a strip renders the same pixels the whole frame would; a record without
// a contribution gets an empty box and remembers its kind in a negative type
__device__ __forceinline__ void prep_window(SrcRec &r, double py, int win_y0, int win_h) {
    r.y0 = max(r.y0, win_y0) - win_y0;
    r.y1 = min(r.y1, win_y0 + win_h) - win_y0;
    r.py = py - (double)win_y0;
    if (r.type < 0 || r.x1 <= r.x0 || r.y1 <= r.y0) {
        r.x0 = r.x1 = r.y0 = r.y1 = 0;
        if (r.type >= 0) r.type = -1 - r.type;   // remember the kind, mark "no contribution"
    }
}

// what k_prep leaves for one (band, source) besides the record
__device__ __forceinline__ void prep_store(const SrcRec &r, int64_t i, SrcRec *__restrict__ recs, int4 *__restrict__ boxes,
                                           int *__restrict__ kind, int *__restrict__ status) {
    recs[i] = r;
    boxes[i] = make_int4(r.x0, r.x1, r.y0, r.y1);
    kind[i] = r.type < 0 ? 0 : (r.type == 0 ? K_PSF : K_GAL);
    // what cel_stamp_boxes / cel_source_boxes report: 1 = has a stamp, 0 = empty box, -1 = the
    // reference's overlap test fails (celeste.py:130-135)
    status[i] = r.type >= 0 ? 1 : (r.type == -3 ? -1 : 0);
}

// the record of source s in band b (entry i = b * S + s of the tables): k_prep's whole arithmetic, also called by the slice
// sampler's step kernel for the point it has just named (k_slice.h)
struct PrepArgs {
    const BandDev *bands;
    int B, H, W, win_y0, win_h;
    int64_t S;
    const int *type;
    const double *counts, *shape;
    double rsq_gal;
    SrcRec *recs; int4 *boxes; int *kind; int *status;      // recs == nullptr: nothing to do
    int nobox;
};
__device__ __forceinline__ void prep_one(const BandDev *__restrict__ bands, int b, int64_t s, int64_t i, int B, int H, int W, int win_y0, int win_h,
                                         const int *__restrict__ type, double ra, double dec, const double *__restrict__ counts,
                                         const double *__restrict__ shape, double rsq_gal, SrcRec *__restrict__ recs,
                                         int4 *__restrict__ boxes, int *__restrict__ kind, int *__restrict__ status, int nobox) {
    const BandDev &bd = bands[b];
    SrcRec r;
    memset(&r, 0, sizeof(r));
    int t = type[s];
    // equa2pixel (fits_image.py:166-174)
    double cphi = cos(bd.phi[1] / 180.0 * PI_D);
    double px, py;
    prep_pixel(bd, ra, dec, cphi, px, py);
    r.px = px; r.py = py;
    r.scale = counts[s * B + b];
    r.type = t;
    const double BIG = 1073741824.0;
    if (t == 0) {
        prep_star_box(bd, px, py, H, W, r);
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
        if (nobox) bound = (w00 == w00 && w11 == w11) ? 4.0 * (double)(H + W) : NAN;
        for (int k = 0; k < K_PSF && !nobox; k++) {
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
    } else if (t == 2) {
        // The older per-profile galaxy route, gen_galaxy_prof_psf_image
        // (celeste_galaxy_conditionals.py:134-182): the caller hands W = R R^T itself (:151; R comes
        // from gen_galaxy_transformation with the CONSTANT img.Ups_n, :33 -- not cd_at_pixel, Q9),
        // shape = (theta, W00, W01, W11).  theta = 1 / 0 selects the 'exp' / 'dev' profile alone.
        // Bound: calc_bounding_radius over the convolved components that carry weight, ERROR = 1e-5
        // (:160-161); box: the int() rule of :166-167 (as a star's); no overlap test on this route.
        double theta = shape[4 * s], w00 = shape[4 * s + 1], w01 = shape[4 * s + 2], w11 = shape[4 * s + 3];
        r.w00 = w00; r.w01 = w01; r.w11 = w11; r.theta = theta;
        r.type = 1;                       // downstream kernels see an ordinary galaxy record
        double rsq_inv = 1.0 / rsq_gal;
        double bound = -INFINITY;
        for (int k = 0; k < K_PSF; k++) {
            double dist = sqrt(bd.mux[k] * bd.mux[k] + bd.muy[k] * bd.muy[k]);
            for (int j = 0; j < K_PROF; j++) {
                if ((j < K_EXP) ? (theta == 0.0) : (theta == 1.0)) continue;
                double v = c_prof_var[j];
                bound = fmax(bound, comp_radius(v * w00 + bd.cxx[k], v * w01 + bd.cxy[k], v * w11 + bd.cyy[k],
                                                rsq_inv, dist));
            }
        }
        if (!(bound == bound) || !(px == px) || !(py == py)) {
            r.type = -1;
        } else {
            int lx = (int)clampd(px - bound, -BIG, BIG), hx = (int)clampd(px + bound + 1, -BIG, BIG);
            int ly = (int)clampd(py - bound, -BIG, BIG), hy = (int)clampd(py + bound + 1, -BIG, BIG);
            r.x0 = max(0, lx); r.x1 = min(hx, W);
            r.y0 = max(0, ly); r.y1 = min(hy, H);
        }
    } else {
        r.type = -1;
    }
    prep_window(r, py, win_y0, win_h);
    prep_store(r, i, recs, boxes, kind, status);
}

__global__ void __launch_bounds__(256)
k_prep(const BandDev *__restrict__ bands, int B, int H, int W, int win_y0, int win_h, int64_t S,
       const int *__restrict__ type, const double *__restrict__ radec,
       const double *__restrict__ counts, const double *__restrict__ shape, double rsq_gal,
       SrcRec *__restrict__ recs, int4 *__restrict__ boxes, int *__restrict__ kind, int *__restrict__ status,
       unsigned long long *__restrict__ cursor /* the binning pass's 4 cursors / flags, zeroed here (a memset of
                                                  its own cost 15 us of queue time per step), or nullptr */,
       const int *__restrict__ live /* [S] or nullptr: a source with live[s] < 0 is skipped (a retired slice chain:
                                       nothing reads its records this round) */,
       int nobox = 0 /* 1: the records feed conditional likelihoods on FIXED patch limits (the slice samplers' rounds): a
                        galaxy's own box -- the bounding radius over its 42 convolved components, most of this kernel's
                        arithmetic -- is not needed and is set to the whole window */) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 4 && cursor) cursor[i] = 0ull;
    if (i >= S * B) return;
    int b = (int)(i / S);
    int64_t s = i - (int64_t)b * S;
    if (live && live[s] < 0) return;
    prep_one(bands, b, s, i, B, H, W, win_y0, win_h, type, radec[2 * s], radec[2 * s + 1], counts, shape, rsq_gal, recs, boxes, kind, status, nobox);
}

// Gibbs resamples the sky level (models.py:156-160): one scalar, passed as a kernel argument so
// that the update is ordered on the stream without a host synchronisation
__global__ void k_set_eps(BandDev *__restrict__ bands, int band, double eps) { bands[band].eps = eps; }

// gen_galaxy_prof_psf_mixture_params (CelestePy/celeste_fast.pyx:100-140) for N sources that share
// the PSF and profile arrays: one thread per output component, PSF-major (idx = k * J + j):
//   weights = image_ws[k] * amp[j],  means = v_s + image_means[k],  covars = image_covars[k] + sigs[j] * W
__global__ void __launch_bounds__(256)
k_mixture_params(int64_t N, const double *__restrict__ Wm /* N*4 */, const double *__restrict__ v_s /* N*2 */,
                 const double *__restrict__ iw, const double *__restrict__ im, const double *__restrict__ ic, int Kp,
                 const double *__restrict__ amp, const double *__restrict__ sigs, int J,
                 double *__restrict__ weights, double *__restrict__ means, double *__restrict__ covars) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int K = Kp * J;
    if (i >= N * K) return;
    const int64_t n = i / K;
    const int c = (int)(i - n * K);
    const int k = c / J, j = c - k * J;
    weights[i] = iw[k] * amp[j];
    means[2 * i + 0] = v_s[2 * n + 0] + im[2 * k + 0];
    means[2 * i + 1] = v_s[2 * n + 1] + im[2 * k + 1];
#pragma unroll
    for (int q = 0; q < 4; q++) covars[4 * i + q] = ic[4 * k + q] + sigs[j] * Wm[4 * n + q];
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
// k_order: heaviest-first launch order of the tiles (counting sort on the work estimate)
// ------------------------------------------------------------------------------------------
// Tile order never changes results (every tile is written once); it only shortens the tail of
// k_render, whose tiles differ in work by orders of magnitude.  One block; 256 buckets.
// Inside a bucket the tiles land in atomic-arrival order: the launch order is not reproducible run
// to run (the results are).  A stable variant (per-wave slices, ranks among equal-bucket lanes by 64
// broadcasts per step) was built and measured: 43 us against 13 us -- 2 % of the step for nothing
// a result depends on -- and dropped.
__global__ void __launch_bounds__(1024)
k_order(const int *__restrict__ work, int T, int *__restrict__ order) {
    __shared__ int hist[256];
    __shared__ int red[1024];
    const int tid = threadIdx.x;
    int mx = 0;
    for (int i = tid; i < T; i += 1024) mx = max(mx, work[i]);
    red[tid] = mx;
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if (tid < o) red[tid] = max(red[tid], red[tid + o]);
        __syncthreads();
    }
    const long long wmax = red[0] > 0 ? red[0] : 1;
    for (int i = tid; i < T; i += 1024) atomicAdd(&hist[255 - (int)(((long long)work[i] * 255) / wmax)], 1);
    __syncthreads();
    if (tid == 0) {   // exclusive scan, bucket 0 = heaviest
        int run = 0;
        for (int k = 0; k < 256; k++) { int c = hist[k]; hist[k] = run; run += c; }
    }
    __syncthreads();
    for (int i = tid; i < T; i += 1024) {
        int bkt = 255 - (int)(((long long)work[i] * 255) / wmax);
        order[atomicAdd(&hist[bkt], 1)] = i;
    }
}
