// k_estep.h -- E-step sufficient statistics without materialising the (S+1, H, W) layers
//
// CelestePy's EM (celeste_em.py:38-88) builds gen_src_prob_layers (celeste.py:222-234), an
// (S+1) x H x W tensor of responsibilities eps/lambda and F_s/lambda, only to reduce it:
//   X~[n][s]  = sum_pixels nelec * F_s / lambda      photons source s is responsible for (:85)
//   noise[n]  = sum_pixels nelec * eps / lambda      photons the sky is responsible for  (:62)
//   mass[n][s]= sum_pixels unit stamp of s           fraction of s's light inside image n (:89)
// At 10 000 sources x 2048^2 that tensor is 335 GB per band; the three reductions need only the
// resident model image.  One 256-thread block per (source, band) re-evaluates the source's
// stamp on its own box (direct evaluator, exact) against lambda and nelec; a strided kernel
// reduces the sky term.  Both reductions run in a fixed order.
//
// k_estep_src_hw is the same reduction on the column recurrence (hw_source.h): one wave per
// (source, band), the box covered by 32 x 64 chunks, each rendered as the unit stamp into an LDS
// tile with the drop rule relative to the source itself (both sums are linear in the stamp, so
// their relative error stays below K e^-T), then reduced against nelec / lambda.
#pragma once
#include "hw_source.h"

__global__ void __launch_bounds__(256)
k_estep_src(const BandDev *__restrict__ bands, int B, int H, int W, int64_t S, const SrcRec *__restrict__ recs,
            const double *__restrict__ nelec, const double *__restrict__ lambda,
            double *__restrict__ xt /* S*B */, double *__restrict__ mass /* S*B */) {
    __shared__ CompTab T;
    __shared__ double red[256], red2[256];
    const int tid = threadIdx.x;
    const int64_t job = blockIdx.x;
    const int b = (int)(job % B);
    const int64_t s = job / B;
    const SrcRec *rp = recs + (int64_t)b * S + s;
    const int type = rp->type;
    if (type < 0) {
        if (tid == 0) { xt[job] = 0.0; mass[job] = 0.0; }
        return;
    }
    const BandDev *bd = bands + b;
    const int K = (type == 0) ? K_PSF : K_GAL;
    if (tid < K) {
        Comp c = make_comp(tid, type, rp->px, rp->py, 1.0, rp->w00, rp->w01, rp->w11, rp->theta, bd);
        T.A[tid] = c.A; T.mx[tid] = c.mx; T.my[tid] = c.my;
        T.qa[tid] = c.qa; T.qb[tid] = c.qb; T.qc[tid] = c.qc;
    }
    __syncthreads();
    const int x0 = rp->x0, y0 = rp->y0, nx = rp->x1 - rp->x0, ny = rp->y1 - rp->y0;
    const double counts = rp->scale;
    const int64_t plane = (int64_t)b * H * W;
    double a = 0.0, m = 0.0;
    const int n = nx * ny;
    for (int i = tid; i < n; i += 256) {
        int yy = i / nx, xx = i - yy * nx;
        double u = eval_direct(T, 0, K, (double)(x0 + xx), (double)(y0 + yy), 1.0);
        int64_t idx = plane + (int64_t)(y0 + yy) * W + (x0 + xx);
        a += nelec[idx] * (counts * u) / lambda[idx];
        m += u;
    }
    red[tid] = a; red2[tid] = m;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) { red[tid] += red[tid + o]; red2[tid] += red2[tid + o]; }
        __syncthreads();
    }
    if (tid == 0) { xt[job] = red[0]; mass[job] = red2[0]; }
}

__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2)))
k_estep_src_hw(const BandDev *__restrict__ bands, int B, int H, int W, int64_t S, const SrcRec *__restrict__ recs,
               const double *__restrict__ nelec, const double *__restrict__ lambda, double Tdrop,
               double *__restrict__ xt /* S*B */, double *__restrict__ mass /* S*B */) {
    __shared__ double acc[HW_TH * HW_TW];
    __shared__ CompTab T;
    __shared__ double et[64];
    const int lane = threadIdx.x;
    const int half = lane >> 5, col = lane & 31;
    const int64_t job = blockIdx.x;
    const int b = (int)(job % B);
    const int64_t s = job / B;
    RecU rec = rec_unpack(rec_fetch(recs + (int64_t)b * S, (int)s, lane));
    if (rec.type < 0) {
        if (lane == 0) { xt[job] = 0.0; mass[job] = 0.0; }
        return;
    }
    const BandDev *bd = bands + b;
    const double counts = rec.scale;
    rec.scale = 1.0;
    et[lane] = exp2((double)lane * (1.0 / 64.0));
    const LaneConst lc = lane_consts(lane, bd);
    const int dropmode = (Tdrop > 0.0) ? HW_DROP_SELF : HW_DROP_NONE;
    const int64_t plane = (int64_t)b * H * W;
    double a = 0.0, m = 0.0;
    for (int Y0 = rec.y0; Y0 < rec.y1; Y0 += HW_TH) {
        const int rb = min(HW_TH, rec.y1 - Y0);
        for (int X0 = rec.x0; X0 < rec.x1; X0 += HW_TW) {
            const int xi = X0 + col;
            const bool on = xi < rec.x1;
#pragma unroll
            for (int r = 0; r < HW_TH / 2; r++) acc[r * 64 + lane] = 0.0;
            bool direct;
            const int Kk = hw_build(T, lc, rec, lane, dropmode, Tdrop, 0.0, Y0, X0, min(rec.x1, X0 + HW_TW) - 1, 0, rb, direct, nullptr, et);
            hw_walk(T, et, Kk, (double)xi, Y0, 0, rb, on, direct, acc, lane);
            __syncthreads();
            const int64_t base = plane + (int64_t)Y0 * W + min(xi, rec.x1 - 1);
            for (int r0 = 0; r0 < HW_TH / 2 && 2 * r0 < rb; r0 += 8) {
                double ne[8], la[8];
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    const int64_t idx = base + (int64_t)min(2 * (r0 + r) + half, rb - 1) * W;
                    ne[r] = nelec[idx];
                    la[r] = lambda[idx];
                }
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    if (on && 2 * (r0 + r) + half < rb) {
                        const double u = acc[(r0 + r) * 64 + lane];
                        a += ne[r] * (counts * u) / la[r];
                        m += u;
                    }
                }
            }
            __syncthreads();
        }
    }
    a = wave_sum(a);
    m = wave_sum(m);
    if (lane == 0) { xt[job] = a; mass[job] = m; }
}

// sky responsibility: partial[b][blk] = sum over the block's pixel chunk of nelec * eps / lambda
__global__ void __launch_bounds__(256)
k_estep_noise(const BandDev *__restrict__ bands, int64_t npix, int nblk, const double *__restrict__ nelec,
              const double *__restrict__ lambda, double *__restrict__ partial) {
    __shared__ double red[256];
    const int b = blockIdx.x / nblk, blk = blockIdx.x - b * nblk;
    const double eps = bands[b].eps;
    const int64_t chunk = (npix + nblk - 1) / nblk;
    const int64_t lo = chunk * blk, hi = (lo + chunk < npix) ? lo + chunk : npix;
    const double *ne = nelec + (int64_t)b * npix, *la = lambda + (int64_t)b * npix;
    double a = 0.0;
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) a += ne[i] * eps / la[i];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// ---- the same reductions, walking TILES (round 2) ---------------------------------------------------
// k_estep_src_hw re-reads nelec and lambda under every source: 16 B x (sum of box areas) = 6.4 GB at
// config 3 where the images hold 0.34 GB (2.5 TB/s for 3.6 ms; 0.44 of the fp64 issue rate).  Here a
// wave owns a render tile, as in k_render_hw: it loads the tile's nelec / lambda ONCE (32 quotients
// per lane, in registers), then renders each source of the tile's list as a unit stamp into the LDS
// tile and reduces it against the quotients -- one (sum stamp * nelec / lambda, sum stamp) pair per list
// entry, written at the entry's position.  k_estep_gather then sums, for every (source, band), the
// entries of the tiles its box touches in tile order (the source is found in a tile's list by
// bisection: lists are ascending inside their star and galaxy parts), so the result is reproducible.
// The sky term (sum nelec * eps / lambda) falls out of the same pass.
struct EstepArgs {
    const BandDev *bands;
    const SrcRec *recs;
    const int *lists;
    const int *tile_cnt;
    const int64_t *tile_off;
    const int *order;
    const double *nelec, *lambda;
    double *partial;            // 2 per list entry: counts * sum(u * ne / lam), sum(u)
    double *noise_partial;      // per tile
    int64_t S, capacity;
    int B, H, W, ntx, nty;
    double tail_T;
};

__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2)))
k_estep_tiles(EstepArgs a) {
    __shared__ double acc[HW_TH * HW_TW];
    __shared__ CompTab T;
    __shared__ double et[64];
    const int lane = threadIdx.x;
    const int half = lane >> 5, col = lane & 31;
    const int tile = a.order ? a.order[blockIdx.x] : blockIdx.x;
    const int per_band = a.ntx * a.nty;
    const int b = tile / per_band;
    const int t = tile - b * per_band;
    const int ty = t / a.ntx, tx = t - ty * a.ntx;
    const int X0 = tx * HW_TW, Y0 = ty * HW_TH;
    const int xi = X0 + col;
    const BandDev *bd = a.bands + b;
    const double eps = bd->eps;
    // the tile's nelec / lambda, once: 32 quotients per lane in registers across the whole source loop -- 64 VGPRs beside the
    // ~200 of hw_build / hw_walk: 14 of them spill (60 B of scratch per lane, re-read once per source).  -DESTEP_W_LDS keeps
    // the quotients in a second 16 KB LDS tile instead: no scratch, and four waves per CU instead of eight (tools/ab_scratch.sh)
#ifdef ESTEP_W_LDS
    __shared__ double wl[HW_TH * HW_TW];
#define ESTEP_W(r) wl[(r) * 64 + lane]
#else
    double w[HW_TH / 2];
#define ESTEP_W(r) w[r]
#endif
    double noise = 0.0;
    {
        const int64_t base = (int64_t)b * a.H * a.W + (int64_t)(Y0 + half) * a.W + xi;
        double ne[HW_TH / 2], la[HW_TH / 2];
#pragma unroll
        for (int r = 0; r < HW_TH / 2; r++) {
            const bool in = (xi < a.W) && (Y0 + 2 * r + half < a.H);
            ne[r] = in ? a.nelec[base + (int64_t)(2 * r) * a.W] : 0.0;
            la[r] = in ? a.lambda[base + (int64_t)(2 * r) * a.W] : 1.0;
        }
#pragma unroll
        for (int r = 0; r < HW_TH / 2; r++) {
            ESTEP_W(r) = ne[r] / la[r];
            noise += ne[r] * eps / la[r];
        }
    }
    noise = wave_sum(noise);
    if (lane == 0) a.noise_partial[tile] = noise;
    const int cnt = a.tile_cnt[tile];
    const int64_t off = a.tile_off[tile];
    const int nent = (int)min((int64_t)cnt, a.capacity > off ? a.capacity - off : (int64_t)0);
    if (nent == 0) return;
    et[lane] = exp2((double)lane * (1.0 / 64.0));
#pragma unroll
    for (int r = 0; r < HW_TH / 2; r++) acc[r * 64 + lane] = 0.0;
    const SrcRec *recs = a.recs + (int64_t)b * a.S;
    const LaneConst lc = lane_consts(lane, bd);
    const int dropmode = (a.tail_T > 0.0) ? HW_DROP_SELF : HW_DROP_NONE;
    int idx64 = (lane < nent) ? a.lists[off + lane] : 0;
    int recw_next = rec_fetch(recs, __builtin_amdgcn_readlane(idx64, 0), lane);
    for (int e = 0; e < nent; e++) {
        const int recw = recw_next;
        if (e + 1 < nent) {
            if (((e + 1) & 63) == 0) idx64 = (e + 1 + lane < nent) ? a.lists[off + e + 1 + lane] : 0;
            recw_next = rec_fetch(recs, __builtin_amdgcn_readlane(idx64, (e + 1) & 63), lane);
        }
        RecU rec = rec_unpack(recw);
        const double counts = rec.scale;
        rec.scale = 1.0;                                    // the tile holds the unit stamp
        const int ra = max(rec.y0, Y0) - Y0, rb = min(rec.y1, Y0 + HW_TH) - Y0;
        const int xa = max(rec.x0, X0), xb = min(rec.x1, X0 + HW_TW) - 1;
        const bool on = (xi >= rec.x0) && (xi < rec.x1);
        bool direct;
        const int Kk = hw_build(T, lc, rec, lane, dropmode, a.tail_T, 0.0, Y0, xa, xb, ra, rb, direct, nullptr, et);
        hw_walk(T, et, Kk, (double)xi, Y0, ra, rb, on, direct, acc, lane);
        __syncthreads();
        double xt = 0.0, ms = 0.0;
#pragma unroll
        for (int r = 0; r < HW_TH / 2; r++) {               // static indices: w[] stays in registers
            if (2 * r + 1 >= ra && 2 * r < rb) {            // (wave-uniform) the row pair meets the box
                const double u = acc[r * 64 + lane];        // 0 outside the box's rows and columns
                acc[r * 64 + lane] = 0.0;                   // clean for the next source
                xt = fma(u, ESTEP_W(r), xt);
                ms += u;
            }
        }
        xt = wave_sum(xt);
        ms = wave_sum(ms);
        if (lane == 0) {
            a.partial[2 * (off + e)] = counts * xt;
            a.partial[2 * (off + e) + 1] = ms;
        }
    }
}

__global__ void __launch_bounds__(256)
k_estep_gather(const int4 *__restrict__ boxes /* B*S: x0, x1, y0, y1 */, const int *__restrict__ kind, int64_t S, int B, int ntx,
               int nty, const int *__restrict__ tile_cnt, const int *__restrict__ tile_nstar, const int64_t *__restrict__ tile_off,
               const int *__restrict__ lists, int64_t capacity, const double *__restrict__ partial,
               double *__restrict__ xt /* S*B */, double *__restrict__ mass /* S*B */) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;     // band-major, like the boxes
    if (i >= S * B) return;
    const int b = (int)(i / S);
    const int s = (int)(i - (int64_t)b * S);
    const int4 q = boxes[i];
    double a = 0.0, m = 0.0;
    if (q.y > q.x && q.w > q.z) {
        const bool star = (kind[i] == K_PSF);
        for (int ty = q.z / HW_TH; ty <= (q.w - 1) / HW_TH; ty++)
            for (int tx = q.x / HW_TW; tx <= (q.y - 1) / HW_TW; tx++) {
                const int tile = (b * nty + ty) * ntx + tx;
                const int64_t off = tile_off[tile];
                const int cnt = (int)min((int64_t)tile_cnt[tile], capacity > off ? capacity - off : (int64_t)0);
                const int ns = min(tile_nstar[tile], cnt);
                int lo = star ? 0 : ns, hi = star ? ns : cnt;         // the part of the list the source is in, ascending
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (lists[off + mid] < s) lo = mid + 1; else hi = mid;
                }
                const int64_t e = off + lo;
                a += partial[2 * e];
                m += partial[2 * e + 1];
            }
    }
    xt[(int64_t)s * B + b] = a;
    mass[(int64_t)s * B + b] = m;
}
