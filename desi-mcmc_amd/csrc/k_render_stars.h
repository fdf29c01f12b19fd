// k_render_stars.h -- the render kernel for tiles that hold nothing but stars (or nothing at all)
//
// Why a second kernel.  On a star field k_render_hw neither computes nor streams at a roof: a tile's phases
// (header -> list -> records, the star pass, nelec in / log / lambda out) run one behind the other inside a
// wave, and with the 16 KB fp64 accumulator tile only two waves fit a SIMD, so for 45 % of its cycles a wave
// sits in s_waitcnt with nobody to take the SIMD (profiles/r03_stars_pmc.json).  A star tile
// needs no component table and no general path, and the unit of work of the star pass is one COLUMN of one
// star -- columns are independent -- so this kernel takes the SAME 32 x 64 tile in NP parts of 32 / NP columns,
// one behind the other: the accumulator is 64 rows x 16 columns (8 KB), a wave needs ~12.6 KB of LDS and
// < 128 VGPRs, and three to four waves share a SIMD.  Nothing is seeded twice (a task belongs to one part), the
// tile's header, list and records are fetched once, and a part's epilogue (128-B row segments, four rows per
// wave-instruction) is half as long, so more of it hides under the neighbours' arithmetic.
//
// Launched by the host instead of k_render_hw when the catalogue holds no galaxy, and beside it (each kernel
// leaving the other's tiles alone) when galaxies are few; which kernel renders a tile depends on the call's
// inputs only.  A pixel receives its stars' terms in the order of the part's task list (data only).
#pragma once
#include "k_render_hw.h"


// a part's observed pixels into registers: 64 rows x CW columns at column Xa, 64 / CW rows per wave-instruction.
// INSIDE: the tile lies wholly inside the frame and the log-likelihood is wanted -- no condition on any load or
// store, so the compiler can count what is outstanding (see hw_epilogue).
template <int CW, bool INSIDE>
__device__ __forceinline__ void stars_nelec(const RenderArgs &a, int b, int Xa, int Y0, int lane, double (&ne)[HW_TH * CW / 64]) {
    constexpr int RPI = 64 / CW;
    const int c = lane % CW, rq = lane / CW;
    const int xi = Xa + c;
    const bool ll = (a.flags & CEL_RENDER_LOGLIK) != 0;
    const int64_t base = (int64_t)b * a.H * a.W + (int64_t)(Y0 + rq) * a.W + xi;
#pragma unroll
    for (int r = 0; r < HW_TH / RPI; r++) {
        if (INSIDE) ne[r] = a.nelec[base + (int64_t)(RPI * r) * a.W];
        else ne[r] = (ll && xi < a.W && Y0 + RPI * r + rq < a.H) ? a.nelec[base + (int64_t)(RPI * r) * a.W] : 0.0;
    }
}

// epilogue of one part: lambda = eps + acc written once, the Poisson terms summed per lane
template <int CW, bool INSIDE, bool STORE>
__device__ __forceinline__ double stars_epilogue(const RenderArgs &a, const double *__restrict__ acc, const double *__restrict__ lt,
                                                 double eps, int b, int Xa, int Y0, int lane, const double (&ne)[HW_TH * CW / 64]) {
    constexpr int RPI = 64 / CW;
    const int c = lane % CW, rq = lane / CW;
    const int xi = Xa + c;
    const int64_t base = (int64_t)b * a.H * a.W + (int64_t)(Y0 + rq) * a.W + xi;
    double part = 0.0;
    if (INSIDE) {
#pragma unroll
        for (int r = 0; r < HW_TH / RPI; r++) {
            const double lam = eps + acc[r * 64 + lane];
            if (STORE) a.lambda[base + (int64_t)(RPI * r) * a.W] = lam;
            part += ne[r] * log_tab(lam, lt) - lam;
        }
    } else if (xi < a.W) {
        const bool ll = (a.flags & CEL_RENDER_LOGLIK) != 0;
#pragma unroll
        for (int r = 0; r < HW_TH / RPI; r++) {
            if (Y0 + RPI * r + rq < a.H) {
                const double lam = eps + acc[r * 64 + lane];
                if (STORE) a.lambda[base + (int64_t)(RPI * r) * a.W] = lam;
                if (ll) part += ne[r] * log_tab(lam, lt) - lam;
            }
        }
    }
    return part;
}

// the host's rule (CEL_OPT_STAR_TILES = 1): frames with more tiles than k_render_hw has wave slots on the chip (256 CUs x 8).
// Measured (tools/star_tiles_threshold.py, star-only fields): 640 tiles 0.039 / 0.042 ms (general / this kernel), 1 440
// 0.034 / 0.036, 2 560 0.048 / 0.040, 5 760 0.080 / 0.070, 10 240 0.137 / 0.117.
#define STAR_TILES_MIN 2048

// flags bit (internal): this launch runs beside k_render_hw, which takes the tiles that hold a galaxy
#define CEL_RENDER_SPLIT_STARS 8

template <int ST_NP, bool PRE>        // column parts per tile; PRE: a part's nelec is requested BEFORE its star walk
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(ST_NP >= 4 ? 4 : (ST_NP == 2 ? 3 : 2))))
k_render_stars(RenderArgs a) {
    constexpr int ST_CW = HW_TW / ST_NP;   // columns per part
    __shared__ double acc[HW_TH * ST_CW];
    __shared__ StarTab ST;
    __shared__ double et[64];
    __shared__ double lt[128];
    const int lane = threadIdx.x;
    const unsigned long long t_start = a.cost ? wall_clock64() : 0ull;
    const int tile = a.order ? a.order[blockIdx.x] : blockIdx.x;
    const int per_band = a.ntx * a.nty;
    const int b = tile / per_band;
    const int t = tile - b * per_band;
    const int ty = t / a.ntx, tx = t - ty * a.ntx;
    const int X0 = tx * HW_TW, Y0 = ty * HW_TH;
    const BandDev *bd = a.bands + b;

    const int cnt = a.tile_cnt[tile];
    const int nstar_t = a.tile_nstar[tile];
    const int64_t off = a.tile_off[tile];
    if (cnt != nstar_t) return;                 // holds a galaxy: k_render_hw's
    if (cnt == 0) { hw_empty_tile(a, bd, tile, b, X0, Y0, lane, t_start); return; }

    et[lane] = exp2((double)lane * (1.0 / 64.0));
    lt[lane] = c_log_ic[lane];
    lt[64 + lane] = c_log_lc[lane];
    const SrcRec *recs = a.recs + (int64_t)b * a.S;
    const int strict = (a.flags >> 2) & 1;
    const int nstar = (int)min((int64_t)cnt, a.capacity > off ? a.capacity - off : (int64_t)0);
    __syncthreads();
    star_setup(ST, bd, et, lane);               // the host checked the one-segment condition for every band
    const double eps = bd->eps;
    unsigned d0 = 0;
    double part = 0.0;
    const bool in_frame = (X0 + HW_TW <= a.W) && (Y0 + HW_TH <= a.H);
    const bool inside = in_frame && (a.flags & CEL_RENDER_LOGLIK);
    const bool store = !(a.flags & CEL_RENDER_NO_STORE);
    if (nstar <= 64) star_stage(a, ST, recs, off, 0, nstar, lane, X0, Y0, strict);   // one batch: staged once for all parts
    for (int p = 0; p < ST_NP; p++) {
        const int Xa = X0 + p * ST_CW;
        __syncthreads();                        // the previous part's epilogue has read the accumulator, its tasks the task table
#pragma unroll
        for (int r = 0; r < HW_TH * ST_CW / 64; r++) acc[r * 64 + lane] = 0.0;
        double ne[HW_TH * ST_CW / 64];
        if (PRE && inside) stars_nelec<ST_CW, true>(a, b, Xa, Y0, lane, ne);
        if (nstar <= 64) {
            star_walk<false, ST_CW>(a, ST, et, acc, nstar, lane, Xa, Y0, strict, d0);
        } else {
            for (int base = 0; base < nstar; base += 64) {
                const int nb = min(64, nstar - base);
                star_stage(a, ST, recs, off, base, nb, lane, X0, Y0, strict);
                star_walk<false, ST_CW>(a, ST, et, acc, nb, lane, Xa, Y0, strict, d0);
            }
        }
        __syncthreads();
        if (inside) {
            if (!PRE) stars_nelec<ST_CW, true>(a, b, Xa, Y0, lane, ne);
            part += store ? stars_epilogue<ST_CW, true, true>(a, acc, lt, eps, b, Xa, Y0, lane, ne)
                          : stars_epilogue<ST_CW, true, false>(a, acc, lt, eps, b, Xa, Y0, lane, ne);
        } else if (in_frame) {                  // model images only: stores, none of them under a condition
            if (store) {
                constexpr int RPI = 64 / ST_CW;
                const int64_t base = (int64_t)b * a.H * a.W + (int64_t)(Y0 + lane / ST_CW) * a.W + Xa + lane % ST_CW;
#pragma unroll
                for (int r = 0; r < HW_TH / RPI; r++) a.lambda[base + (int64_t)(RPI * r) * a.W] = eps + acc[r * 64 + lane];
            }
        } else {
            stars_nelec<ST_CW, false>(a, b, Xa, Y0, lane, ne);
            part += store ? stars_epilogue<ST_CW, false, true>(a, acc, lt, eps, b, Xa, Y0, lane, ne)
                          : stars_epilogue<ST_CW, false, false>(a, acc, lt, eps, b, Xa, Y0, lane, ne);
        }
    }
    if (a.flags & CEL_RENDER_LOGLIK) {
        part = wave_sum(part);
        if (lane == 0) a.partials[tile] = part;
    }
    if (a.cost && lane == 0) a.cost[tile] = (int)min(wall_clock64() - t_start, 0x3fffffffull) + 1;
}
