// k_misc.h -- stamp and generic-evaluator kernels
#pragma once
// cel_sources_set_rows: n packed rows (type, radec, counts[B], shape) scattered to rows idx[] of the catalogue's arrays
__global__ void __launch_bounds__(256)
k_scatter_rows(int64_t n, int B, const int *__restrict__ idx, const int *__restrict__ type, const double *__restrict__ radec,
               const double *__restrict__ counts, const double *__restrict__ shape,
               int *__restrict__ d_type, double *__restrict__ d_radec, double *__restrict__ d_counts, double *__restrict__ d_shape) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t s = idx[i];
    d_type[s] = type[i];
    d_radec[2 * s] = radec[2 * i];
    d_radec[2 * s + 1] = radec[2 * i + 1];
    for (int b = 0; b < B; b++) d_counts[s * B + b] = counts[i * B + b];
    for (int k = 0; k < 4; k++) d_shape[4 * s + k] = shape[4 * i + k];
}

// cel_stamp_mass's short cut: the integer sums of the photon split (strictly inside every box) and of k_strict_totals (the
// boxes' first rows and columns) are the unit stamps' masses already -- up to what the split's drop rule left out: a component
// below eps e^-T on a half-tile is nothing beside the sky, but it is something of the stamp, the more the fainter the source.
// Measured against the mass kernel (tools/dbg/mass_shortcut_error.py, 2 000 sources at one counts / eps ratio each): the
// relative difference grows as eps / counts -- galaxies 1e-11 eps / counts at most (median 8e-13), stars 1.5e-13 -- so a
// galaxy is vouched for when counts >= eps / 16 and a star when counts >= eps / 1024 (at most 1.6e-10 off either way; a chain's
// faintest sources sit at a few hundredths of a sky pixel); the others (and a source without counts) are listed for the mass
// kernel proper.
#define MASS_VOUCH_GAL (1.0 / 16.0)
#define MASS_VOUCH_STAR (1.0 / 1024.0)
__global__ void __launch_bounds__(256)
k_mass_from_fx(int64_t n, int B, const unsigned long long *__restrict__ massfx, const double *__restrict__ counts /* [S][B] */,
               const int *__restrict__ type /* [S] */, const BandDev *__restrict__ bands, double *__restrict__ mass, int *__restrict__ todo, int *__restrict__ ntodo) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool need = false;
    if (i < n) {
        const int b = (int)(i % B);
        const double c = counts[i], eps = bands[b].eps;
#ifdef MASS_VOUCH_ALL      // tools/dbg/mass_shortcut_error.py: how far off the short cut is where it is NOT vouched for
        need = !(c > 0.0);
#else
        need = !(c >= ((type[i / B] == 0) ? MASS_VOUCH_STAR : MASS_VOUCH_GAL) * eps) || !(eps > 0.0) || !(c < 1e300);
#endif
        mass[i] = (double)massfx[i] * (1.0 / MASS_FX);
    }
    const unsigned long long m = __ballot(need);
    int base = 0;
    if ((threadIdx.x & 63) == 0 && m) base = atomicAdd(ntodo, __popcll(m));
    base = __shfl(base, 0);
    if (need) todo[base + __popcll(m & ((1ull << (threadIdx.x & 63)) - 1ull))] = (int)i;
}

#include "hw_source.h"
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
    // caller-imposed limits: still a valid source kind.  Record types: 0/1 star/galaxy with a
    // stamp, -1/-2 star/galaxy whose own box is empty, -3 star failing the overlap test
    if (type < 0) type = (type == -2) ? 1 : 0;
    const int K = (type == 0) ? K_PSF : K_GAL;
    const BandDev *bd = bands + band;
    if (lane < K)
    {
        Comp c = make_comp(lane, type, rp->px, rp->py, scaled ? rp->scale : 1.0, rp->w00, rp->w01, rp->w11,
                           rp->theta, bd);
        T.A[lane] = c.A; T.mx[lane] = c.mx; T.my[lane] = c.my;
        T.qa[lane] = c.qa; T.qb[lane] = c.qb; T.qc[lane] = c.qc;
    }
    __syncthreads();
    const int xi = jb.x0 + lane;
    if (xi >= ob.y) return;
    const int nx = ob.y - ob.x;
    double *o = out + offsets[jb.src];
    for (int y = jb.y0; y < jb.y1; y++) {
        double v = eval_direct(T, 0, K, (double)xi, (double)y, 1.0);
        o[(int64_t)(y - ob.z) * nx + (xi - ob.x)] = v;
    }
}

// k_stamps_hw: the same output from the column recurrence -- one wave per (source, 32-column x
// 64-row chunk) job renders the chunk into an LDS tile (hw_source.h, drop rule relative to the
// source itself: every stored pixel keeps a relative error below K e^-T) and stores it.
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2)))
k_stamps_hw(const BandDev *__restrict__ bands, int band, const SrcRec *__restrict__ recs,
            const StampJob *__restrict__ jobs, const int4 *__restrict__ obox,
            const int64_t *__restrict__ offsets, int scaled, double Tdrop, double *__restrict__ out) {
    __shared__ double acc[HW_TH * HW_TW];
    __shared__ CompTab T;
    __shared__ double et[64];
    const int lane = threadIdx.x;
    const int half = lane >> 5, col = lane & 31;
    const StampJob jb = jobs[blockIdx.x];
    const int4 ob = obox[jb.src];           // output box: x0, x1, y0, y1
    RecU rec = rec_unpack(rec_fetch(recs, jb.src, lane));
    if (rec.type < 0) rec.type = (rec.type == -2) ? 1 : 0;
    if (!scaled) rec.scale = 1.0;
    const BandDev *bd = bands + band;
    et[lane] = exp2((double)lane * (1.0 / 64.0));
#pragma unroll
    for (int r = 0; r < HW_TH / 2; r++) acc[r * 64 + lane] = 0.0;
    const LaneConst lc = lane_consts(lane, bd);
    const int xi = jb.x0 + col;
    const bool on = xi < ob.y;
    const int rb = jb.y1 - jb.y0;
    bool direct;
    const int Kk = hw_build(T, lc, rec, lane, (Tdrop > 0.0) ? HW_DROP_SELF : HW_DROP_NONE, Tdrop, 0.0, jb.y0, jb.x0,
                            min(ob.y, jb.x0 + HW_TW) - 1, 0, rb, direct, nullptr, et);
    hw_walk(T, et, Kk, (double)xi, jb.y0, 0, rb, on, direct, acc, lane);
    __syncthreads();
    if (!on) return;
    const int nx = ob.y - ob.x;
    double *o = out + offsets[jb.src] + (int64_t)(jb.y0 - ob.z) * nx + (xi - ob.x);
#pragma unroll 8
    for (int r = 0; r < HW_TH / 2; r++)
        if (2 * r + half < rb) o[(int64_t)(2 * r + half) * nx] = acc[r * 64 + lane];
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
// k_mog_ll: mog_loglike (CelestePy/util/dists/mog.py:5-21) -- the log-domain mixture evaluator
// ------------------------------------------------------------------------------------------
// out[n] = logsumexp_k( -q_k(x_n)/2 + lw[k] ),  lw[k] = -log(2 pi) - log(det_k)/2 + log(pi_k) formed by
// the caller exactly as the reference forms it (so a weight <= 0 arrives as NaN / -inf).
// One thread per point; the components (K*6 doubles: lw, mx, my, ia, ib2 = icov01 + icov10, ic) are
// staged through LDS 64 at a time; two sweeps (maximum, then the sum of exp(e - max)), as
// scipy's logsumexp does.
__global__ void __launch_bounds__(256)
k_mog_ll(const double *__restrict__ x, int64_t N, const double *__restrict__ comp, int K, double *__restrict__ out) {
    __shared__ double sc[64 * 6];
    int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double px = 0.0, py = 0.0;
    if (n < N) { px = x[2 * n]; py = x[2 * n + 1]; }
    double mx = -INFINITY, s = 0.0;
    bool bad = false;
    for (int pass = 0; pass < 2; pass++) {
        for (int k0 = 0; k0 < K; k0 += 64) {
            int kn = min(64, K - k0);
            __syncthreads();
            for (int i = threadIdx.x; i < kn * 6; i += blockDim.x) sc[i] = comp[(int64_t)k0 * 6 + i];
            __syncthreads();
            for (int k = 0; k < kn; k++) {
                double dx = px - sc[k * 6 + 1], dy = py - sc[k * 6 + 2];
                double q = sc[k * 6 + 3] * dx * dx + sc[k * 6 + 4] * dx * dy + sc[k * 6 + 5] * dy * dy;
                double e = -0.5 * q + sc[k * 6 + 0];
                if (pass == 0) { bad = bad || (e != e); mx = fmax(mx, e); }
                else s += exp(e - mx);
            }
        }
        if (pass == 0 && !(mx > -INFINITY && mx < INFINITY)) mx = 0.0;   // logsumexp: a non-finite maximum is replaced by 0
    }
    if (n < N) out[n] = bad ? NAN : log(s) + mx;
}
