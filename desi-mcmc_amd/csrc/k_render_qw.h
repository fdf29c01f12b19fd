// k_render_qw.h -- the render kernel on 16-column x 128-row tiles ("quarter-wave" layout)
//
// Same algorithm and the same 16 KB fp64 LDS accumulator as k_render_hw (k_render_hw.h); the 64
// lanes are four groups of 16: lane = 16 q + col, every quarter q works on its own group of
// components for the tile's 16 pixel columns.  Why: a recurrence is seeded once per (component,
// column, vertical tile the box spans) -- ~57 instructions against ~3 per row walked -- and every
// lane of a touched tile column walks the box's rows whether or not the column lies in the box.
// On the benchmark field's boxes (76 x 76 on average, tools/tile_geometry.py) 16 x 128 tiles need
// 40 % fewer seeds and 12 % fewer lane-rows than 32 x 64, for 20 % more (source, tile) pairs of
// set-up.  CEL_OPT_TILE_LAYOUT = 2.
#pragma once
#include "k_render_hw.h"

#define QW_TW 16
#define QW_TH 128

__global__ void __launch_bounds__(64)
k_render_qw(RenderArgs a) {
    __shared__ double acc[QW_TH * QW_TW];
    __shared__ CompTab T;
    __shared__ double et[64];
    const int lane = threadIdx.x;
    const int q = lane >> 4, col = lane & 15;
    const unsigned long long t_start = (a.timing || a.cost) ? wall_clock64() : 0ull;
    const int tile = a.order ? a.order[blockIdx.x] : blockIdx.x;
    const int per_band = a.ntx * a.nty;
    const int b = tile / per_band;
    const int t = tile - b * per_band;
    const int ty = t / a.ntx, tx = t - ty * a.ntx;
    const int X0 = tx * QW_TW, Y0 = ty * QW_TH;
    const int xi = X0 + col;
    const double x = (double)xi;
    const BandDev *bd = a.bands + b;
    const double eps = bd->eps;
    const bool store = !(a.flags & CEL_RENDER_NO_STORE);
    const bool ll = (a.flags & CEL_RENDER_LOGLIK) != 0;
    const int64_t plane = (int64_t)b * a.H * a.W;

    const int cnt = a.tile_cnt[tile];
    if (cnt == 0 && !a.timing) {
        // empty sky: pure streaming, one log per wave (see k_render_hw)
        double part = 0.0;
        const double leps = ll ? log(eps) : 0.0;
        if ((a.W & 15) == 0 && Y0 + QW_TH <= a.H) {
            // 16 B per lane: 8 rows of 8 lane-pairs per instruction
            const int cp = lane & 7, rq = lane >> 3;
            const int64_t b2 = plane + (int64_t)(Y0 + rq) * a.W + X0 + 2 * cp;
            double2 ne2[QW_TH / 8];
            if (ll) {
#pragma unroll
                for (int r = 0; r < QW_TH / 8; r++)
                    ne2[r] = *reinterpret_cast<const double2 *>(a.nelec + b2 + (int64_t)(8 * r) * a.W);
            }
#pragma unroll
            for (int r = 0; r < QW_TH / 8; r++) {
                if (store) *reinterpret_cast<double2 *>(a.lambda + b2 + (int64_t)(8 * r) * a.W) = make_double2(eps, eps);
                if (ll) part += (ne2[r].x * leps - eps) + (ne2[r].y * leps - eps);
            }
        } else if (xi < a.W) {
            const int64_t base = plane + (int64_t)(Y0 + q) * a.W + xi;
            double ne[QW_TH / 4];
#pragma unroll
            for (int r = 0; r < QW_TH / 4; r++)
                ne[r] = (ll && Y0 + 4 * r + q < a.H) ? a.nelec[base + (int64_t)(4 * r) * a.W] : 0.0;
#pragma unroll
            for (int r = 0; r < QW_TH / 4; r++) {
                if (Y0 + 4 * r + q < a.H) {
                    if (store) a.lambda[base + (int64_t)(4 * r) * a.W] = eps;
                    if (ll) part += ne[r] * leps - eps;
                }
            }
        }
        if (ll) {
            part = wave_sum(part);
            if (lane == 0) a.partials[tile] = part;
        }
        if (a.cost && lane == 0) a.cost[tile] = (int)min(wall_clock64() - t_start, 0x3fffffffull) + 1;
        return;
    }

    et[lane] = exp2((double)lane * (1.0 / 64.0));
#pragma unroll
    for (int r = 0; r < QW_TH / 4; r++) acc[r * 64 + lane] = 0.0;

    const int64_t off = a.tile_off[tile];
    const SrcRec *recs = a.recs + (int64_t)b * a.S;
    const double Tdrop = a.tail_T;
    const bool dropping = (a.variant != 0) && (Tdrop > 0.0) && (eps > 0.0);
    const float log_eps = dropping ? __logf((float)eps) : 0.0f;
    const int strict = (a.flags >> 2) & 1;

    unsigned dbg_comprows = 0, dbg_pairs = 0;   // only counted under CEL_OPT_TILE_TIMING
    const LaneConst lc = lane_consts(lane, bd);
    const int nent = (int)min((int64_t)cnt, a.capacity > off ? a.capacity - off : (int64_t)0);
    int idx64 = (lane < nent) ? a.lists[off + lane] : 0;
    int recw_next = (nent > 0) ? rec_fetch(recs, __builtin_amdgcn_readlane(idx64, 0), lane) : 0;

    for (int e = 0; e < nent; e++) {
        const int recw = recw_next;
        if (e + 1 < nent) {
            if (((e + 1) & 63) == 0) idx64 = (e + 1 + lane < nent) ? a.lists[off + e + 1 + lane] : 0;
            recw_next = rec_fetch(recs, __builtin_amdgcn_readlane(idx64, (e + 1) & 63), lane);
        }
        const RecU rec = rec_unpack(recw);
        const int K = (rec.type == 0) ? K_PSF : K_GAL;
        const int bx0 = rec.x0 + strict, bx1 = rec.x1, by0 = rec.y0 + strict, by1 = rec.y1;
        const int ra = max(by0, Y0) - Y0, rb = min(by1, Y0 + QW_TH) - Y0;
        const bool on = (xi >= bx0) && (xi < bx1);
        const double xa = (double)max(bx0, X0), xb = (double)(min(bx1, X0 + QW_TW) - 1);
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
                rlo = max(ra, (int)floorf(cy + ylo - 0.02f));
                rhi = min(rb, (int)ceilf(cy + yhi + 0.02f) + 1);
                keep = keep && (rhi > rlo);
            } else {
                keep = true;
            }
            Lk = seg_len(c.qc, fmin(fmax(Tk, 1.0), 300.0));
        }
        const unsigned long long km = __ballot(keep);
        const int Kk = __popcll(km);
        __syncthreads();   // previous source's table reads are done
        if (lane < 4) { T.gL[lane] = 4096; T.gr0[lane] = QW_TH; T.gr1[lane] = 0; }
        const int slot = slot_by_rows<4>(keep, rlo, rhi);
        if (keep) {
            const int p = slot;
            T.A[p] = c.A; T.mx[p] = c.mx; T.my[p] = c.my;
            T.qa[p] = c.qa * EXP_SCALE; T.qb[p] = c.qb * EXP_SCALE; T.qc[p] = c.qc * EXP_SCALE;
            T.eq[p] = exp_tab64(-c.qc * EXP_SCALE, et);
            const int gi = p / (4 * REC_G);       // one pass = four groups of <= REC_G components
            atomicMin(&T.gL[gi], Lk);
            atomicMin(&T.gr0[gi], rlo);
            atomicMax(&T.gr1[gi], rhi);
        }
        if (lane < HW_PAD) {   // zero components behind the table (amplitude 0, ratio 1)
            int p = Kk + lane;
            T.A[p] = 0.0; T.mx[p] = 0.0; T.my[p] = 0.0;
            T.qa[p] = 0.0; T.qb[p] = 0.0; T.qc[p] = 0.0;
            T.eq[p] = 1.0;
        }
        __syncthreads();
        if (a.variant == 0) {
            // direct evaluator: the quarters split the kept components
            const int kq = (Kk + 3) / 4;
            const int k0 = min(q * kq, Kk), k1 = min(k0 + kq, Kk);
            for (int row = ra; row < rb; row++) {
                double v = eval_direct(T, k0, k1, x, (double)(Y0 + row), 1.0 / EXP_SCALE);
                if (on) lds_add(&acc[row * QW_TW + col], v);
            }
            continue;
        }
        // passes of four groups: quarter q takes components [p0 + q G, p0 + (q + 1) G); what lies
        // behind the table's end is zero padding (at most 3 entries: G = ceil(R / 4))
        for (int p0 = 0; p0 < Kk; p0 += 4 * REC_G) {
            const int R = min(4 * REC_G, Kk - p0);
            const int G = (R + 3) / 4;
            const int gi = p0 / (4 * REC_G);
            const int L = __builtin_amdgcn_readfirstlane(T.gL[gi]);
            const int ga = __builtin_amdgcn_readfirstlane(T.gr0[gi]);
            const int gb = __builtin_amdgcn_readfirstlane(T.gr1[gi]);
            if (a.timing) { dbg_comprows += (unsigned)(gb - ga) * (unsigned)R; dbg_pairs += 1; }
            const int k0 = p0 + q * G;
            if (L < 4) {
                // pathologically sharp component: evaluate this pass directly
                const int k1 = min(k0 + G, p0 + R);
                for (int row = ga; row < gb; row++) {
                    double v = eval_direct(T, min(k0, p0 + R), k1, x, (double)(Y0 + row), 1.0 / EXP_SCALE);
                    if (on) lds_add(&acc[row * QW_TW + col], v);
                }
                continue;
            }
            double *colp = acc + col;
            switch (G) {
            case 6: rec_group_hw<6, QW_TW>(T, et, k0, x, Y0, ga, gb, L, on, colp); break;
            case 5: rec_group_hw<5, QW_TW>(T, et, k0, x, Y0, ga, gb, L, on, colp); break;
            case 4: rec_group_hw<4, QW_TW>(T, et, k0, x, Y0, ga, gb, L, on, colp); break;
            case 3: rec_group_hw<3, QW_TW>(T, et, k0, x, Y0, ga, gb, L, on, colp); break;
            case 2: rec_group_hw<2, QW_TW>(T, et, k0, x, Y0, ga, gb, L, on, colp); break;
            default: rec_group_hw<1, QW_TW>(T, et, k0, x, Y0, ga, gb, L, on, colp); break;
            }
        }
    }

    // epilogue: one wave-instruction covers four 128-B row segments (rows 4r .. 4r+3).
    // The component table is dead now: its LDS holds the log table (128 doubles) instead.
    __syncthreads();
    double *lt = reinterpret_cast<double *>(&T);
    lt[lane] = c_log_ic[lane];
    lt[64 + lane] = c_log_lc[lane];
    __syncthreads();
    double part = 0.0;
    if (xi < a.W) {
        double ne[QW_TH / 4];
        const int64_t base = plane + (int64_t)(Y0 + q) * a.W + xi;
#pragma unroll
        for (int r = 0; r < QW_TH / 4; r++)
            ne[r] = (ll && Y0 + 4 * r + q < a.H) ? a.nelec[base + (int64_t)(4 * r) * a.W] : 0.0;
#pragma unroll
        for (int r = 0; r < QW_TH / 4; r++) {
            if (Y0 + 4 * r + q < a.H) {
                double lam = eps + acc[r * 64 + lane];
                if (store) a.lambda[base + (int64_t)(4 * r) * a.W] = lam;
                if (ll) part += ne[r] * log_tab(lam, lt) - lam;
            }
        }
    }
    if (ll) {
        part = wave_sum(part);
        if (lane == 0) a.partials[tile] = part;
    }
    if (a.cost && lane == 0) a.cost[tile] = (int)min(wall_clock64() - t_start, 0x3fffffffull) + 1;
    if (a.timing && lane == 0) {
        a.timing[3 * (size_t)tile + 0] = t_start;
        a.timing[3 * (size_t)tile + 1] = wall_clock64();
        a.timing[3 * (size_t)tile + 2] = (unsigned long long)(unsigned)cnt | ((unsigned long long)dbg_pairs << 12) |
                                               ((unsigned long long)dbg_comprows << 32);
    }
}
