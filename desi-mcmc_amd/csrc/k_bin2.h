// k_bin2.h -- tile binning: per-tile source lists, in source order
//
// k_bin_fine_blk<true> (the usual form): one 16-wave block per 256 x 256-pixel super-tile scans the band's S
//   boxes itself, keeps the ordered list of sources whose box touches the super-tile in LDS and writes,
//   for each of its 32-column x TH-row render tiles, the ordered source list, its length and a work
//   estimate for the heaviest-first launch order.
// Two-level form (a super-tile with more than BIN_CH candidates; sticky per image set):
//   k_bin_coarse writes the super-tile lists to global memory, k_bin_fine_blk<false> streams them through
//   LDS in chunks of BIN_CH.
// Both levels compact with ballot + prefix popcount, so every list is ordered: a render tile's list
// holds its STARS first (ascending source index), then its galaxies (ascending source index) -- the
// render kernel takes the stars of a tile through a batched 3-component path -- and the
// accumulation order in k_render, and with it every output bit, is reproducible.
// List SEGMENTS are placed with one atomicAdd per list (segment order in the buffer is
// arbitrary and irrelevant); no atomics touch list contents.
#pragma once
#include "device_common.h"

#define SUPER_W 256
#define SUPER_H 256

__device__ inline bool box_hits(int4 q, int X0, int X1, int Y0, int Y1) {
    return (q.x < X1) && (q.y > X0) && (q.z < Y1) && (q.w > Y0) && (q.y > q.x) && (q.w > q.z);
}

// grid = B * nsx * nsy blocks of 256 threads.  cursor[0]: coarse list cursor, cursor[1] (as int):
// overflow flag.
#define COARSE_WAVES 16
__global__ void __launch_bounds__(64 * COARSE_WAVES)
k_bin_coarse(const int4 *__restrict__ boxes, int64_t S, int nsx, int nsy, int *__restrict__ sup_cnt,
             int64_t *__restrict__ sup_off, unsigned long long *cursor, int *__restrict__ clist,
             int64_t capacity, int *overflow) {
    __shared__ int wcnt[COARSE_WAVES];
    __shared__ long long base_s;
    const int st = blockIdx.x;
    const int per_band = nsx * nsy;
    const int b = st / per_band;
    const int t = st - b * per_band;
    const int sy = t / nsx, sx = t - sy * nsx;
    const int X0 = sx * SUPER_W, X1 = X0 + SUPER_W, Y0 = sy * SUPER_H, Y1 = Y0 + SUPER_H;
    const int4 *bx = boxes + (int64_t)b * S;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // each wave owns a contiguous quarter of the sources, so concatenating the four
    // wave-ordered pieces keeps the whole list ordered
    const int64_t per = ((S + 64 * COARSE_WAVES - 1) / (64 * COARSE_WAVES)) * 64;
    const int64_t lo = per * wave, hi = (lo + per < S) ? lo + per : S;
    int count = 0;
    for (int64_t s0 = lo; s0 < hi; s0 += 64) {
        int64_t s = s0 + lane;
        bool hit = (s < hi) && box_hits(bx[s], X0, X1, Y0, Y1);
        count += __popcll(__ballot(hit));
    }
    if (lane == 0) wcnt[wave] = count;
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
        for (int w = 0; w < COARSE_WAVES; w++) tot += wcnt[w];
        sup_cnt[st] = tot;
        long long base = (long long)atomicAdd(cursor, (unsigned long long)tot);
        sup_off[st] = base;
        base_s = base;
    }
    __syncthreads();
    int64_t at0 = base_s;
    for (int w = 0; w < wave; w++) at0 += wcnt[w];
    int run = 0;
    for (int64_t s0 = lo; s0 < hi; s0 += 64) {
        int64_t s = s0 + lane;
        bool hit = (s < hi) && box_hits(bx[s], X0, X1, Y0, Y1);
        unsigned long long m = __ballot(hit);
        if (hit) {
            int64_t at = at0 + run + __popcll(m & ((1ull << lane) - 1ull));
            if (at < capacity) clist[at] = (int)s; else *overflow = 1;
        }
        run += __popcll(m);
    }
}

// Level 2: one block per super-tile stages the super-tile's candidates
// (box, component count, source index) through LDS in chunks of BIN_CH and its FINE_WAVES waves bin
// them into the super-tile's render tiles (wave w takes tiles w, w + FINE_WAVES, ...).  Pass 0 counts and
// reserves each tile's list segment, pass 1 fills it; with n <= BIN_CH (the usual case) the
// candidates are read from global memory once; every box test reads LDS.
#define BIN_CH 1024
// waves per block.  Round 6, from stamps inside the kernel (a -DBIN_STAMPS build): a 16-wave block takes 17 us (scan 3.1, second
// scan 5.6, counting pass 2.4, cursor atomic 2.2-3.3, filling pass 2.6-3.3) and only ONE fits a CU (78 VGPRs: 24 waves), so the
// 320 blocks of configs[2] ran in two rounds: 38 us.  Two 8-wave blocks fit (110 VGPRs: 16 waves per CU): one round of slower blocks, 33 us.
// ... so the host takes 8-wave blocks when the frame has more super-tiles than the device has CUs and 16-wave blocks
// otherwise (a rank's strip of an 8-way cut is 40 super-tiles: 19 us with 16 waves against 26 with 8).

// FUSED: the block finds its candidates itself -- its 16 waves scan the band's S boxes (each wave a
// contiguous slice, so the compacted list stays ordered) straight into the LDS staging arrays -- and
// k_bin_coarse, its list in global memory and a launch drop out of the step.  Possible while a
// super-tile has at most BIN_CH candidates (265 at config 3); a block that finds more raises bit 1 of
// the coarse overflow flag and the host goes back to the two-level form for these images.
template <bool FUSED, int FINE_WAVES /* waves per block: 8 or 16 */>
__global__ void __launch_bounds__(64 * FINE_WAVES)
k_bin_fine_blk(const int4 *__restrict__ boxes, const int *__restrict__ kind, int64_t S, int ntx, int nty, int TH,
               int TW, int nsx, int nsy, const int *__restrict__ sup_cnt, const int64_t *__restrict__ sup_off,
               const int *__restrict__ clist, int64_t ccap, int *__restrict__ tile_cnt, int *__restrict__ tile_nstar,
               int *__restrict__ tile_work, int64_t *__restrict__ tile_off, unsigned long long *cursor,
               int *__restrict__ lists, int64_t capacity, int *overflow, int *coarse_overflow) {
    constexpr int BIN_TPW = 32 / FINE_WAVES;      // max tiles per wave: (256/64) * (256/32) / FINE_WAVES
    __shared__ int4 sbox[BIN_CH];
    __shared__ int skind[BIN_CH];
    __shared__ int sid[BIN_CH];
    __shared__ int stcnt[FINE_WAVES * BIN_TPW], stoff[FINE_WAVES * BIN_TPW];
    __shared__ long long sbase;
    const int st = blockIdx.x;
    const int per_band_s = nsx * nsy;
    const int b = st / per_band_s;
    const int t = st - b * per_band_s;
    const int sy = t / nsx, sx = t - sy * nsx;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tpx = SUPER_W / TW, tpy = SUPER_H / TH;          // tiles per super-tile in x / y (<= 32 in all)
    const int4 *bx = boxes + (int64_t)b * S;
    const int *kd = kind + (int64_t)b * S;
    int n = 0;
    int64_t coff = 0;
    if (FUSED) {
        __shared__ int wcnt[FINE_WAVES];
        const int SX0 = sx * SUPER_W, SX1 = SX0 + SUPER_W, SY0 = sy * SUPER_H, SY1 = SY0 + SUPER_H;
        const int64_t per = ((S + 64 * FINE_WAVES - 1) / (64 * FINE_WAVES)) * 64;
        const int64_t lo = per * wave, hi = (lo + per < S) ? lo + per : S;
        // both scans issue SCAN_U box loads per lane before the first is used: the boxes come from L2 and a
        // block is alone on its CU, so a load per trip would be a latency per trip
        constexpr int SCAN_U = 8;
        int count = 0;
        for (int64_t s0 = lo; s0 < hi; s0 += 64 * SCAN_U) {
            int4 q[SCAN_U];
#pragma unroll
            for (int u = 0; u < SCAN_U; u++) {
                const int64_t s = s0 + 64 * u + lane;
                q[u] = (s < hi) ? bx[s] : make_int4(0, 0, 0, 0);          // an empty box hits nothing
            }
#pragma unroll
            for (int u = 0; u < SCAN_U; u++) count += __popcll(__ballot(box_hits(q[u], SX0, SX1, SY0, SY1)));
        }
        if (lane == 0) wcnt[wave] = count;
        __syncthreads();
        int at0 = 0;
        for (int w = 0; w < FINE_WAVES; w++) { if (w < wave) at0 += wcnt[w]; n += wcnt[w]; }
        if (n > BIN_CH) {                       // wave-uniform and block-uniform
            if (threadIdx.x == 0) atomicOr(coarse_overflow, 2);
            n = 0;                              // this attempt's lists of the super-tile stay empty; the host reruns
        } else if (count > 0) {
            int run = 0;
            for (int64_t s0 = lo; s0 < hi; s0 += 64 * SCAN_U) {
                int4 q[SCAN_U];
#pragma unroll
                for (int u = 0; u < SCAN_U; u++) {
                    const int64_t s = s0 + 64 * u + lane;
                    q[u] = (s < hi) ? bx[s] : make_int4(0, 0, 0, 0);
                }
#pragma unroll
                for (int u = 0; u < SCAN_U; u++) {
                    const int64_t s = s0 + 64 * u + lane;
                    const bool hit = box_hits(q[u], SX0, SX1, SY0, SY1);
                    const unsigned long long mk = __ballot(hit);
                    if (hit) {
                        const int at = at0 + run + __popcll(mk & ((1ull << lane) - 1ull));
                        sid[at] = (int)s; sbox[at] = q[u]; skind[at] = kd[s];
                    }
                    run += __popcll(mk);
                }
            }
        }
        __syncthreads();
    } else {
        n = sup_cnt[st];
        coff = sup_off[st];
        if (coff + n > ccap) n = (int)((ccap > coff) ? (ccap - coff) : 0);   // truncated coarse list (overflow is flagged)
    }
    int cnt[BIN_TPW], nst[BIN_TPW], work[BIN_TPW], run[BIN_TPW], rung[BIN_TPW];
    long long base[BIN_TPW];
#pragma unroll
    for (int j = 0; j < BIN_TPW; j++) { cnt[j] = 0; nst[j] = 0; work[j] = 0; run[j] = 0; rung[j] = 0; base[j] = 0; }

    for (int pass = 0; pass < 2; pass++) {
        for (int c0 = 0; c0 < n || (c0 == 0 && n == 0 && pass == 0); c0 += BIN_CH) {
            const int m = min(BIN_CH, n - c0);
            if (!FUSED && (pass == 0 || n > BIN_CH)) {
                __syncthreads();
                for (int i = threadIdx.x; i < m; i += 64 * FINE_WAVES) {
                    int s = clist[coff + c0 + i];
                    sid[i] = s;
                    sbox[i] = bx[s];
                    skind[i] = kd[s];
                }
                __syncthreads();
            }
#pragma unroll
            for (int j = 0; j < BIN_TPW; j++) {
                const int tl = wave + FINE_WAVES * j;          // local tile index
                if (tl >= tpx * tpy) continue;
                const int tx = sx * tpx + (tl % tpx), ty = sy * tpy + (tl / tpx);
                if (tx >= ntx || ty >= nty) continue;
                const int X0 = tx * TW, X1 = X0 + TW, Y0 = ty * TH, Y1 = Y0 + TH;
                for (int i0 = 0; i0 < m; i0 += 64) {
                    const int i = i0 + lane;
                    bool hit = false, star = false;
                    int4 q = make_int4(0, 0, 0, 0);
                    if (i < m) { q = sbox[i]; hit = box_hits(q, X0, X1, Y0, Y1); star = hit && (skind[i] == K_PSF); }
                    const unsigned long long mk = __ballot(hit);
                    const unsigned long long ms = __ballot(star);
                    if (pass == 0) {
                        if (hit) work[j] += skind[i] * (min(q.w, Y1) - max(q.z, Y0) + 18);
                        cnt[j] += __popcll(mk);
                        nst[j] += __popcll(ms);
                    } else {
                        if (hit) {
                            const unsigned long long below = (1ull << lane) - 1ull;
                            // stars fill the head of the segment, everything else follows them
                            int64_t at = star ? base[j] + run[j] + __popcll(ms & below)
                                              : base[j] + nst[j] + rung[j] + __popcll((mk & ~ms) & below);
                            if (at < capacity) lists[at] = sid[i]; else *overflow = 1;
                        }
                        run[j] += __popcll(ms);
                        rung[j] += __popcll(mk & ~ms);
                    }
                }
            }
            if (n == 0) break;
        }
        if (pass == 0) {
            // ONE cursor atomic per super-tile (a single address sustains only ~90 returning
            // atomics per microsecond: one per render tile cost 0.12 ms); the block's tiles get
            // consecutive segments by an LDS prefix sum
#pragma unroll
            for (int j = 0; j < BIN_TPW; j++) {
                const int tl = wave + FINE_WAVES * j;
                if (lane == 0 && tl < FINE_WAVES * BIN_TPW) stcnt[tl] = (tl < tpx * tpy) ? cnt[j] : 0;
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                int tot = 0;
                for (int k = 0; k < FINE_WAVES * BIN_TPW; k++) { int c = stcnt[k]; stoff[k] = tot; tot += c; }
                sbase = (long long)atomicAdd(cursor, (unsigned long long)tot);
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < BIN_TPW; j++) {
                const int tl = wave + FINE_WAVES * j;
                if (tl >= tpx * tpy) continue;
                base[j] = sbase + stoff[tl];
                const int tx = sx * tpx + (tl % tpx), ty = sy * tpy + (tl / tpx);
                if (tx >= ntx || ty >= nty) continue;
                const int tile = (b * nty + ty) * ntx + tx;
                int w = work[j];
                for (int o = 32; o > 0; o >>= 1) w += __shfl_down(w, o);
                if (lane == 0) {
                    tile_cnt[tile] = cnt[j];
                    tile_nstar[tile] = nst[j];
                    tile_work[tile] = w;
                    tile_off[tile] = base[j];
                }
            }
        }
    }
}

// ---- small catalogues: one wave per render tile, no super-tiles ----------------------------------------------
// With a few thousand sources the two-level machinery above is all latency (sixteen-wave blocks, three barriers, a global
// cursor: 35 us for 1 000 stars on 640 tiles, a third of that step).  Here every tile's wave tests the band's S boxes itself
// (S / 64 trips, the boxes from L2) and writes its list into a segment of its own, tile * S entries into the buffer: no
// atomics on list positions, no prefix sum between tiles, the same list format (stars first, each kind in source order),
// the same counts and work estimates.  The host takes it while S <= BIN_DIRECT_MAX_S and the segments fit the list buffer.
#define BIN_DIRECT_MAX_S 4096
__global__ void __launch_bounds__(64)
k_bin_direct(const int4 *__restrict__ boxes, const int *__restrict__ kind, int64_t S, int ntx, int nty, int TH, int TW,
             int *__restrict__ tile_cnt, int *__restrict__ tile_nstar, int *__restrict__ tile_work, int64_t *__restrict__ tile_off,
             unsigned long long *cursor, int *__restrict__ lists) {
    const int tile = blockIdx.x, lane = threadIdx.x;
    const int per_band = ntx * nty;
    const int b = tile / per_band;
    const int t = tile - b * per_band;
    const int ty = t / ntx, tx = t - ty * ntx;
    const int X0 = tx * TW, X1 = X0 + TW, Y0 = ty * TH, Y1 = Y0 + TH;
    const int4 *bx = boxes + (int64_t)b * S;
    const int *kd = kind + (int64_t)b * S;
    const int64_t base = (int64_t)tile * S;
    constexpr int U = 8;                    // box loads in flight per lane
    int cnt = 0, nst = 0, work = 0;
    for (int64_t s0 = 0; s0 < S; s0 += 64 * U) {
        int4 q[U];
        int k[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int64_t s = min(s0 + 64 * u + lane, S - 1);
            q[u] = bx[s];
            k[u] = kd[s];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const bool hit = (s0 + 64 * u + lane < S) && box_hits(q[u], X0, X1, Y0, Y1);
            if (hit) work += k[u] * (min(q[u].w, Y1) - max(q[u].z, Y0) + 18);
            cnt += __popcll(__ballot(hit));
            nst += __popcll(__ballot(hit && k[u] == K_PSF));
        }
    }
    int run = 0, rung = 0;
    for (int64_t s0 = 0; s0 < S && cnt > 0; s0 += 64 * U) {
        int4 q[U];
        int k[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int64_t s = min(s0 + 64 * u + lane, S - 1);
            q[u] = bx[s];
            k[u] = kd[s];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int64_t s = s0 + 64 * u + lane;
            const bool hit = (s < S) && box_hits(q[u], X0, X1, Y0, Y1);
            const bool star = hit && k[u] == K_PSF;
            const unsigned long long mk = __ballot(hit), ms = __ballot(star);
            if (hit) {
                const unsigned long long below = (1ull << lane) - 1ull;
                // stars fill the head of the segment, everything else follows them
                const int64_t at = star ? base + run + __popcll(ms & below) : base + nst + rung + __popcll((mk & ~ms) & below);
                lists[at] = (int)s;
            }
            run += __popcll(ms);
            rung += __popcll(mk & ~ms);
        }
    }
    for (int o = 32; o > 0; o >>= 1) work += __shfl_down(work, o);
    if (lane == 0) {
        tile_cnt[tile] = cnt;
        tile_nstar[tile] = nst;
        tile_work[tile] = work;
        tile_off[tile] = base;
        if (cnt) atomicAdd(cursor, (unsigned long long)cnt);      // the total list length, for cel_field_stats
    }
}
