// k_render.h -- component tables, evaluators, the tile render kernel, the ll reduction
#pragma once
#include "device_common.h"

// ------------------------------------------------------------------------------------------
// component tables in LDS
// ------------------------------------------------------------------------------------------
// Per component (SoA over k; reads in the evaluators are wave-uniform LDS broadcasts):
//   A   = scale * weight / (2 pi sqrt(det))     mx, my = mean
//   qa, qb, qc = inverse covariance [[qa, qb], [qb, qc]]
//   eq  = exp(-qc): the row-to-row ratio of the recurrence's r
//   L   = longest underflow-safe recurrence segment for this component (rows)
#define CT_N (K_GAL + 12)   // + zero components the half-wave layout reads behind the table
struct CompTab {
    double A[CT_N], mx[CT_N], my[CT_N], qa[CT_N], qb[CT_N], qc[CT_N], eq[CT_N];
    // narrow types on purpose: accumulator tile + tables must stay <= 20 480 B per wave so that
    // 8 waves fit a CU's 160 KB of LDS (at 20 576 B only 7 did, and the kernel ran 6 % slower)
    short L[CT_N];            // <= 4096
    unsigned char r0[CT_N], r1[CT_N];   // tile rows [r0, r1) (<= 128) on which the component can exceed the drop level
    // per pair of groups (12 consecutive kept components, the unit the half-wave kernels walk):
    // shortest safe segment and the union of the row ranges, reduced with LDS min/max while the
    // table is written, so that the walk reads three values instead of looping over 12 entries
    int gL[4], gr0[4], gr1[4];
    // ... and the union of the row ranges of the pair's SMALLER half of the slots (the nested walk, rec_group_nested in
    // k_render_hw.h: slots are ordered by decreasing row count, so the later slots of a pair need fewer rows than its union)
    int gs0[4], gs1[4];
};

// fp32 reciprocal / square root / reciprocal square root as ONE instruction each (v_rcp_f32, v_sqrt_f32, v_rsq_f32: 1 ulp).
// Every use below either refines the value in fp64 (Newton steps) or rounds outwards by far more than an ulp.  The
// correctly rounded forms (__frcp_rn, __fsqrt_rn: what rounds 1-5 called) expand to the IEEE division / square-root sequences
// -- 11 and ~9 instructions: a dozen of them per (source, tile) set-up were 8 % of k_render_hw's instruction stream.
__device__ __forceinline__ float rcp_f32(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float sqrt_f32(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float rsq_f32(float x) { return __builtin_amdgcn_rsqf(x); }

struct Comp {   // one component in registers
    double A, mx, my, qa, qb, qc, ixx, iyy;   // ixx = 1/Sigma_xx, iyy = 1/Sigma_yy (marginals)
};

// Galaxy components are enumerated PSF-major and, inside one PSF component, by increasing
// profile variance, so that neighbouring k have similar widths: groups of the recurrence then
// share a segment length and are dropped together far from the centre.  (The reference's order
// is galaxy-major, mog.py:75-81; only the fp summation order differs, ~1e-16)
//   index into [exp0..exp5, dev0..dev7] sorted by variance:
__constant__ int c_prof_order[K_PROF] = {6, 7, 0, 8, 1, 9, 2, 10, 3, 11, 4, 12, 5, 13};

__device__ inline Comp make_comp(int k, int type, double px, double py, double scale, double w00,
                                 double w01, double w11, double theta, const BandDev *__restrict__ bd) {
    int kk = (type == 0) ? k : (k / K_PROF);
    double cxx = bd->cxx[kk], cxy = bd->cxy[kk], cyy = bd->cyy[kk], wt = bd->w[kk];
    if (type == 1) {
        int j = c_prof_order[k - kk * K_PROF];
        double var = c_prof_var[j];
        double amp = (j < K_EXP) ? theta * c_prof_amp[j] : (1.0 - theta) * c_prof_amp[j];
        cxx += var * w00; cxy += var * w01; cyy += var * w11;
        wt *= amp;
    }
    double det = cxx * cyy - cxy * cxy;
    double inv = 1.0 / det;
    Comp c;
    c.qa = cyy * inv; c.qb = -cxy * inv; c.qc = cxx * inv;
    c.A = scale * wt * (0.5 / PI_D) * sqrt(inv);      // weight / (2 pi sqrt(det))
    c.mx = px + bd->mux[kk];
    c.my = py + bd->muy[kk];
    // marginal precisions feed only the conservative drop test: fp32 reciprocals, shrunk by
    // 1e-6 so that rounding can only make the test keep more, never less
    c.ixx = (double)(rcp_f32((float)cxx) * 0.999999f);
    c.iyy = (double)(rcp_f32((float)cyy) * 0.999999f);
    return c;
}

// ---- the same, with every load hoisted out of the per-source loop ------------------------------
// What lane k needs from the band PSF and the profile tables depends only on k and on the source
// kind, so both roles are fetched once per tile into registers (LaneConst); per source nothing but
// the 128-byte record is read, and that record is fetched one source ahead, one dword per lane
// (lanes 0..31), and broadcast to scalar registers with v_readlane (RecU).  Without this every
// source of a tile started with three dependent global-memory round trips (list -> record ->
// PSF/profile gathers), ~1-2 us of an ~10 us source, which two waves per SIMD cannot hide.
struct LaneConst {
    double g_cxx, g_cxy, g_cyy, g_w, g_mux, g_muy, g_var, g_amp;   // galaxy role: PSF comp k/14, profile order[k%14]
    double s_cxx, s_cxy, s_cyy, s_w, s_mux, s_muy;                 // star role: PSF comp k (k < 3)
    bool g_exp;
};

__device__ inline LaneConst lane_consts(int lane, const BandDev *__restrict__ bd) {
    LaneConst lc;
    const int kk = min(lane / K_PROF, K_PSF - 1);
    const int j = c_prof_order[lane % K_PROF];
    lc.g_cxx = bd->cxx[kk]; lc.g_cxy = bd->cxy[kk]; lc.g_cyy = bd->cyy[kk];
    lc.g_w = bd->w[kk]; lc.g_mux = bd->mux[kk]; lc.g_muy = bd->muy[kk];
    lc.g_var = c_prof_var[j]; lc.g_amp = c_prof_amp[j]; lc.g_exp = (j < K_EXP);
    const int ks = min(lane, K_PSF - 1);
    lc.s_cxx = bd->cxx[ks]; lc.s_cxy = bd->cxy[ks]; lc.s_cyy = bd->cyy[ks];
    lc.s_w = bd->w[ks]; lc.s_mux = bd->mux[ks]; lc.s_muy = bd->muy[ks];
    return lc;
}

struct RecU {   // a source record in wave-uniform registers
    double px, py, scale, w00, w01, w11, theta;
    int x0, x1, y0, y1, type;
};

__device__ inline double rl_double(int v, int dw) {
    return __hiloint2double(__builtin_amdgcn_readlane(v, dw + 1), __builtin_amdgcn_readlane(v, dw));
}

// recw = dword (lane & 31) of the record, as loaded by rec_fetch
__device__ inline RecU rec_unpack(int recw) {
    RecU r;
    r.px = rl_double(recw, 0); r.py = rl_double(recw, 2); r.scale = rl_double(recw, 4);
    r.w00 = rl_double(recw, 6); r.w01 = rl_double(recw, 8); r.w11 = rl_double(recw, 10);
    r.theta = rl_double(recw, 12);
    r.x0 = __builtin_amdgcn_readlane(recw, 14); r.x1 = __builtin_amdgcn_readlane(recw, 15);
    r.y0 = __builtin_amdgcn_readlane(recw, 16); r.y1 = __builtin_amdgcn_readlane(recw, 17);
    r.type = __builtin_amdgcn_readlane(recw, 18);
    return r;
}

__device__ inline int rec_fetch(const SrcRec *__restrict__ recs, int s, int lane) {
    return reinterpret_cast<const int *>(recs + s)[lane & 31];
}

// 1/d and 1/sqrt(d) for the component tables: fp32 seed + two Newton steps (relative error ~1e-7 ->
// 1e-14 -> rounding), a dozen independent-ish instructions against the ~35 of a correctly rounded
// fp64 division followed by a correctly rounded square root, whose last bit nothing here needs.
// d is a covariance determinant: positive, far inside fp32's range.
__device__ inline void rcp_rsqrt(double d, double &inv, double &rsq) {
    double y = (double)rcp_f32((float)d);
    y = fma(y, fma(-d, y, 1.0), y);
    inv = fma(y, fma(-d, y, 1.0), y);
    double z = (double)rsq_f32((float)d);
    z = z * fma(-0.5 * d * z, z, 1.5);
    rsq = z * fma(-0.5 * d * z, z, 1.5);
}

// 1/d and 1/sqrt(d) alone, the same way (d positive and of moderate size: variances, determinants, 1 + t^2)
__device__ inline double rcp64(double d) {
    double y = (double)rcp_f32((float)d);
    y = fma(y, fma(-d, y, 1.0), y);
    return fma(y, fma(-d, y, 1.0), y);
}
__device__ inline double rsqrt64(double d) {
    double z = (double)rsq_f32((float)d);
    z = z * fma(-0.5 * d * z, z, 1.5);
    return z * fma(-0.5 * d * z, z, 1.5);
}

__device__ inline Comp make_comp_lc(const LaneConst &lc, const RecU &r) {
    double cxx, cxy, cyy, wt, mux, muy;
    if (r.type == 1) {
        double amp = lc.g_exp ? r.theta * lc.g_amp : (1.0 - r.theta) * lc.g_amp;
        cxx = lc.g_cxx + lc.g_var * r.w00; cxy = lc.g_cxy + lc.g_var * r.w01; cyy = lc.g_cyy + lc.g_var * r.w11;
        wt = lc.g_w * amp; mux = lc.g_mux; muy = lc.g_muy;
    } else {
        cxx = lc.s_cxx; cxy = lc.s_cxy; cyy = lc.s_cyy; wt = lc.s_w; mux = lc.s_mux; muy = lc.s_muy;
    }
    double det = cxx * cyy - cxy * cxy;
#ifdef COMP_IEEE_DIV
    double inv = 1.0 / det, rsq = sqrt(inv);
#else
    double inv, rsq;
    rcp_rsqrt(det, inv, rsq);
#endif
    Comp c;
    c.qa = cyy * inv; c.qb = -cxy * inv; c.qc = cxx * inv;
    c.A = r.scale * wt * (0.5 / PI_D) * rsq;
    c.mx = r.px + mux;
    c.my = r.py + muy;
    c.ixx = (double)(rcp_f32((float)cxx) * 0.999999f);
    c.iyy = (double)(rcp_f32((float)cyy) * 0.999999f);
    return c;
}

// Reserve `n` consecutive entries of a list for this lane (0: none) with ONE atomic per wave: a single address sustains
// only ~90 returning atomics per microsecond, and a list of 50 000 jobs built with one atomic each took 0.5 ms.
// All 64 lanes call it.  -> this lane's first entry
__device__ inline int wave_reserve(int *__restrict__ count, int n) {
    int incl = n;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(incl, o);
        if ((int)(threadIdx.x & 63) >= o) incl += up;
    }
    const int total = __shfl(incl, 63);
    int base = 0;
    if ((threadIdx.x & 63) == 0 && total > 0) base = atomicAdd(count, total);
    base = __shfl(base, 0);
    return base + incl - n;
}

__device__ inline double wave_sum(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    return v;
}

// The same sum by DPP moves (row shifts inside the rows of 16 lanes, then the two row broadcasts): VALU instructions instead
// of twelve ds_bpermute round trips -- for sums taken once per (source, tile) pair, where a wave has nothing else in flight
// to hide them behind.  The total arrives in LANE 63 only; a fixed tree: the same bits for the same inputs.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_take(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_lane63(double v) {
    v += dpp_take<0x111, 0xf>(v);      // row_shr:1
    v += dpp_take<0x112, 0xf>(v);      // row_shr:2
    v += dpp_take<0x114, 0xf>(v);      // row_shr:4
    v += dpp_take<0x118, 0xf>(v);      // row_shr:8: lane 15 of every row holds the row's sum
    v += dpp_take<0x142, 0xa>(v);      // row_bcast:15 into rows 1 and 3
    v += dpp_take<0x143, 0xc>(v);      // row_bcast:31 into rows 2 and 3: lane 63 holds the wave's
    return v;
}

// Minimum of the positive-definite form a x^2 + 2 b x y + c y^2 over the rectangle
// [x1,x2] x [y1,y2] (coordinates relative to the component mean).  0 when the mean is inside;
// otherwise the minimum lies on the boundary: the smallest of the four 1-D constrained edge
// minima.  The edge minimisers use fp32 reciprocals (second-order effect on the value); the
// result is shrunk by 1e-5 so that the drop test built on it can only err towards keeping.
__device__ inline double quad_min_rect(double a, double b, double c, double x1, double x2, double y1, double y2) {
    if (x1 <= 0.0 && x2 >= 0.0 && y1 <= 0.0 && y2 >= 0.0) return 0.0;
    const double rc = (double)rcp_f32((float)c), ra = (double)rcp_f32((float)a);
    double best = INFINITY;
#pragma unroll
    for (int i = 0; i < 2; i++) {
        double X = i ? x2 : x1;
        double t = fmin(fmax(-b * X * rc, y1), y2);
        best = fmin(best, a * X * X + (2.0 * b * X + c * t) * t);
        double Y = i ? y2 : y1;
        double u = fmin(fmax(-b * Y * ra, x1), x2);
        best = fmin(best, c * Y * Y + (2.0 * b * Y + a * u) * u);
    }
    return best * 0.99999;
}

// Rows on which a x^2 + 2 b x y + c y^2 <= R for SOME x in [x1, x2] (all relative to the component's
// centre): at fixed x the form's y-interval is (-b x -+ sqrt(c R - det x^2)) / c, det = a c - b^2; its
// upper end is concave in x with its maximum at x = -b sqrt(R / (det a)), the lower end convex with its
// minimum at +b sqrt(R / (det a)) -- on an interval, at those points clamped into it.  A tile on the
// flank of a wide component needs fewer rows than the component's full height (config 3: 9 % fewer
// component-rows).  fp32 reciprocal / square roots: the callers round outwards by more than their error.
__device__ inline void quad_rows_on_columns(double a, double b, double c, double R, double x1, double x2,
                                            float &ylo, float &yhi) {
    const double det = a * c - b * b;
    const double xs = b * (double)sqrt_f32((float)R * rcp_f32((float)(det * a)));
    const double xt = fmin(fmax(-xs, x1), x2), xb = fmin(fmax(xs, x1), x2);
    const float rc = rcp_f32((float)c);
    const float st = sqrt_f32(fmaxf((float)(c * R - det * xt * xt), 0.0f));
    const float sb = sqrt_f32(fmaxf((float)(c * R - det * xb * xb), 0.0f));
    yhi = ((float)(-b * xt) + st) * rc;
    ylo = ((float)(-b * xb) - sb) * rc;
}

// ---- exp() for the recurrence seeds -----------------------------------------------------------
// exp(x) = 2^e * 2^(j/64) * exp(r), x = (64 e + j) ln2/64 + r, |r| <= ln2/128: a 64-entry table
// of 2^(j/64) in LDS and a degree-5 polynomial (truncation r^6/720 < 4e-17), ~9 fp64 ops against
// ~17 + range checks for the library exp; error <= ~2 ulp.  The argument arrives already in units
// of ln2/64 (t = x * 64/ln2: the table builder folds that factor into the quadratic-form
// coefficients, which saves the scaling multiply and the two-step reduction per seed).
// f = t - rint(t) is exact; the powers of ln2/64 are folded into the polynomial's coefficients.
// Inputs are finite and <= 709*64/ln2 here; large negative inputs flush to 0 through ldexp.
__device__ inline double exp_tab64(double t, const double *__restrict__ et) {
    const double c1 = 1.0830424696249145e-02;     // (ln2/64)
    const double c2 = 5.864904955056169e-05;      // (ln2/64)^2 / 2
    const double c3 = 2.1173137155464774e-07;     // (ln2/64)^3 / 6
    const double c4 = 5.732851688640402e-10;      // (ln2/64)^4 / 24
    const double c5 = 1.2417843701716923e-12;     // (ln2/64)^5 / 120
    double n = rint(t);
    double f = t - n;
    int ni = (int)n;
    double p = fma(f, c5, c4);
    p = fma(p, f, c3);
    p = fma(p, f, c2);
    p = fma(p, f, c1);
    p = fma(p, f, 1.0);
    return ldexp(et[ni & 63] * p, ni >> 6);
}

// the same on a 256-entry table (t in units of ln2/256) with a cubic: |r| <= ln2/512, truncation r^4/24 < 1.4e-13 relative --
// two fma less, for the evaluations at the photons (k_patch_ll_nz: 14 VALU per galaxy component and photon with it), whose
// log-likelihood sums are tested to 1e-12
#define EXP_SCALE256 369.32993046757462703   // 256 / ln 2
__device__ inline double exp_tab256_p3(double t, const double *__restrict__ et256) {
    const double c1 = 2.7076061740622863e-03;     // (ln2/256)
    const double c2 = 3.6655655969101056e-06;     // (ln2/256)^2 / 2
    const double c3 = 3.3083026805413710e-09;     // (ln2/256)^3 / 6
    double n = rint(t);
    double f = t - n;
    int ni = (int)n;                              // saturates for t << 0: ldexp flushes to 0
    double p = fma(f, c3, c2);
    p = fma(p, f, c1);
    p = fma(p, f, 1.0);
    return ldexp(et256[ni & 255] * p, ni >> 8);
}

// ---- log() for the Poisson term of the epilogue -----------------------------------------------
// log(x) = e ln2 + lc[j] + log1p(r), x = 2^e m, m in [1,2), j = floor(64 (m - 1)),
// r = m * ic[j] - 1 with |r| <= 2^-7, log1p by its series to r^7 (truncation r^8/8 < 2e-18).
// ~19 fp64-rate instructions against ~95 for the library log (which carries double-double
// arithmetic for arguments this path never sees); error <= ~1.5 ulp.  x must be positive,
// finite and normal: lambda >= eps > 0 here.  lt = 128 doubles in LDS: ic[64] then lc[64].
__device__ inline double log_tab(double x, const double *__restrict__ lt) {
    const double LN2_HI = 0x1.62e42fee00000p-1;      // ln 2, upper 32 bits (e * HI is exact)
    const double LN2_LO = 0x1.a39ef35793c76p-33;     // ln 2 - HI
    if (!(x >= 2.2250738585072014e-308 && x <= 1.7976931348623157e308)) return log(x);   // 0, <0, denormal, inf, NaN
    int ex;
    double m = frexp(x, &ex) * 2.0;                  // [1, 2)
    ex -= 1;
    int j = (int)((m - 1.0) * 64.0);
    double r = fma(m, lt[j], -1.0);
    double p = fma(r, 1.0 / 7.0, -1.0 / 6.0);
    p = fma(p, r, 1.0 / 5.0);
    p = fma(p, r, -0.25);
    p = fma(p, r, 1.0 / 3.0);
    p = fma(p, r, -0.5);
    p = fma(p, r, 1.0);
    p *= r;
    double e = (double)ex;
    return fma(e, LN2_HI, lt[64 + j] + fma(e, LN2_LO, p));
}

// ---- direct evaluator: sum over components of A exp(-q/2) at (x, y) -------------------------
// qscale = 1 for a table holding the inverse covariance itself (stamps), 1/EXP_SCALE for the
// render kernel's table, whose qa/qb/qc carry the factor 64/ln2.
#define EXP_SCALE 92.332482616893656758   // 64 / ln 2
__device__ inline double eval_direct(const CompTab &T, int k0, int k1, double x, double y, double qscale) {
    double s = 0.0;
    for (int k = k0; k < k1; k++) {
        double dx = x - T.mx[k], dy = y - T.my[k];
        double q = T.qa[k] * dx * dx + 2.0 * T.qb[k] * dx * dy + T.qc[k] * dy * dy;
        s += T.A[k] * exp(-0.5 * qscale * q);
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
    // largest L with (L sqrt(qc/2) + sqrt(T))^2 <= REC_EMAX.  fp32 is ample for a row count;
    // the 0.999 keeps the rounding on the safe (shorter) side.  T <= 300 => u > 0.
    float u = 26.0768f - sqrt_f32((float)T);                 // sqrt(680) = 26.0768
    float L = u * rsq_f32(0.5f * (float)qc) * 0.999f;
    return (int)fminf(L, 4096.0f);
}

// acc += v as ONE LDS instruction (ds_add_f64, no return): no read-add-write dependency for the
// wave to wait on.  A tile is owned by a single wave and LDS operations of one wave execute in
// order, so the accumulation order -- and the result -- stays deterministic.
__device__ inline void lds_add(double *p, double v) { atomicAdd(p, v); }

// Accumulate components [k0, k0+G) over rows [ra, rb) of column x into acc_col (the LDS column of
// this lane, stride TILE_W doubles).  `on` masks lanes outside the source box.
template <int G>
__device__ inline void rec_group(const CompTab &T, const double *__restrict__ et, int k0, double x,
                                 int Y0, int ra, int rb, int L, bool on, double *__restrict__ acc_col) {
    double g[G], r[G], q[G];
    const double aon = on ? 1.0 : 0.0;
    for (int sa = ra; sa < rb; sa += L) {
        const int sb = min(sa + L, rb);
        const double y0 = (double)(Y0 + sa);
#pragma unroll
        for (int i = 0; i < G; i++) {
            const int k = k0 + i;
            // the table holds qa, qb, qc multiplied by 64/ln2 (EXP_SCALE): exponents come out
            // directly in the units exp_tab64 wants
            double dx = x - T.mx[k], dy = y0 - T.my[k];
            double qb = T.qb[k], qc = T.qc[k];
            double hx = qb * dx + qc * dy;                      // = -dE/dy (scaled)
            double e = -0.5 * (T.qa[k] * dx * dx + (qb * dx + hx) * dy);
            double er = fmin(fmax(-(hx + 0.5 * qc), -REC_EMAX * EXP_SCALE), REC_EMAX * EXP_SCALE);
            g[i] = (T.A[k] * aon) * exp_tab64(e, et);
            r[i] = exp_tab64(er, et);
            q[i] = T.eq[k];
        }
        // two rows per trip: the register rotation of g/r cancels (no v_mov copies)
        int row = sa;
        for (; row + 1 < sb; row += 2) {
            double s0 = g[0], s1, g1[G], r1[G];
#pragma unroll
            for (int i = 1; i < G; i++) s0 += g[i];
#pragma unroll
            for (int i = 0; i < G; i++) {
                g1[i] = g[i] * r[i];
                r1[i] = r[i] * q[i];
            }
            s1 = g1[0];
#pragma unroll
            for (int i = 1; i < G; i++) s1 += g1[i];
#pragma unroll
            for (int i = 0; i < G; i++) {
                g[i] = g1[i] * r1[i];
                r[i] = r1[i] * q[i];
            }
            lds_add(&acc_col[row * TILE_W], s0);
            lds_add(&acc_col[(row + 1) * TILE_W], s1);
        }
        if (row < sb) {
            double s0 = g[0];
#pragma unroll
            for (int i = 1; i < G; i++) s0 += g[i];
#pragma unroll
            for (int i = 0; i < G; i++) {
                g[i] *= r[i];
                r[i] *= q[i];
            }
            lds_add(&acc_col[row * TILE_W], s0);
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
    const int *tile_nstar; // how many of a tile's entries are stars: they come first in its list (k_bin2.h)
    const int64_t *tile_off;
    const int *order;     // tile launch order (heaviest first) or nullptr
    unsigned long long *timing;   // diagnostic: per-tile {start, end} wall clock (100 MHz) + XCC/CU id, or nullptr
    int *cost;            // out: every tile's measured duration (100 MHz ticks, >= 1): the next launch's order; or nullptr
    const double *nelec;
    double *lambda;
    double *partials;
    int64_t S, capacity;
    int B, H, W, ntx, nty;
    int flags;        // CEL_RENDER_*
    int variant;      // 0 direct, 1 recurrence
    double tail_T;    // drop threshold (0 = never)
    double *slabs;    // k_render_hw<, PARTS > 1>: PARTS accumulator tiles per render tile (16 KB each), and
    int *part_cnt;    //   the tiles' arrival counters (zero between launches)
    const int *dirty; // k_render_hw<false, 1>: per tile, != 0: render it; 0: its pixels and its partial are still those of the last
                      //   render (no changed source's box touches it); nullptr: every tile
};

// The tiles a box touches are marked dirty (32-column x 64-row tiles).  One thread per (changed row, band); called once with the
// boxes the changed sources HAD (before k_prep rewrites them) and once with those they have now.
#define DELTA_MAX 64
struct DeltaRows { int n; int idx[DELTA_MAX]; };
__global__ void __launch_bounds__(256)
k_mark_dirty(DeltaRows d, const int4 *__restrict__ boxes /* [B][S]: x0, x1, y0, y1 */, int64_t S, int B, int ntx, int nty, int TW, int TH,
             int *__restrict__ dirty) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= d.n * B) return;
    const int r = i / B, b = i - r * B;
    const int4 q = boxes[(int64_t)b * S + d.idx[r]];
    if (q.y <= q.x || q.w <= q.z) return;
    for (int ty = q.z / TH; ty <= (q.w - 1) / TH && ty < nty; ty++)
        for (int tx = q.x / TW; tx <= (q.y - 1) / TW && tx < ntx; tx++) dirty[(b * nty + ty) * ntx + tx] = 1;
}

template <int TH>
__global__ void __launch_bounds__(64)
k_render(RenderArgs a) {
    __shared__ double acc[TH * TILE_W];
    __shared__ CompTab T;
    __shared__ double et[64];
    const int lane = threadIdx.x;
    const unsigned long long t_start = a.timing ? wall_clock64() : 0ull;
    const int tile = a.order ? a.order[blockIdx.x] : blockIdx.x;
    const int per_band = a.ntx * a.nty;
    const int b = tile / per_band;
    const int t = tile - b * per_band;
    const int ty = t / a.ntx, tx = t - ty * a.ntx;
    const int X0 = tx * TILE_W, Y0 = ty * TH;
    const int xi = X0 + lane;
    const double x = (double)xi;
    const BandDev *bd = a.bands + b;

    et[lane] = exp2((double)lane * (1.0 / 64.0));
#pragma unroll
    for (int r = 0; r < TH; r++) acc[r * TILE_W + lane] = 0.0;

    const int cnt = a.tile_cnt[tile];
    const int64_t off = a.tile_off[tile];
    const SrcRec *recs = a.recs + (int64_t)b * a.S;
    // Drop rule: a component is skipped on this tile when its contribution is below
    // eps * e^-T everywhere on the part of the tile the source covers (eps = the band's sky
    // level, so the bound is relative to lambda >= eps).  T <= 0 or eps <= 0: never drop.
    const double Tdrop = a.tail_T;
    const double eps_sky = bd->eps;
    const bool dropping = (a.variant != 0) && (Tdrop > 0.0) && (eps_sky > 0.0);

    for (int e = 0; e < cnt; e++) {
        int64_t at = off + e;
        if (at >= a.capacity) break;
        const int s = __builtin_amdgcn_readfirstlane(a.lists[at]);
        const SrcRec *rp = recs + s;
        const int type = rp->type;
        const int K = (type == 0) ? K_PSF : K_GAL;
        const int bx0 = rp->x0, bx1 = rp->x1, by0 = rp->y0, by1 = rp->y1;
        const int ra = max(by0, Y0) - Y0, rb = min(by1, Y0 + TH) - Y0;
        const bool on = (xi >= bx0) && (xi < bx1);
        // the part of this tile the source's box covers, for the drop test
        const double xa = (double)max(bx0, X0), xb = (double)(min(bx1, X0 + TILE_W) - 1);
        const double ya = (double)(Y0 + ra), yb = (double)(Y0 + rb - 1);

        // lane k builds component k, decides whether it can matter on this tile, and the kept
        // components are compacted into the LDS table in k order
        bool keep = false;
        Comp c;
        int Lk = 0;
        int rlo = ra, rhi = rb;
        if (lane < K) {
            c = make_comp(lane, type, rp->px, rp->py, rp->scale, rp->w00, rp->w01, rp->w11, rp->theta, bd);
            // A e^E >= eps e^-T  <=>  E >= -(T + log(A/eps)) =: -Tk
            double Tk = dropping ? Tdrop + (double)__logf((float)(fabs(c.A) / eps_sky)) : 100.0;
            if (dropping) {
                // smallest value of the quadratic form on the covered rectangle
                double qmin = quad_min_rect(c.qa, c.qb, c.qc, xa - c.mx, xb - c.mx, ya - c.my, yb - c.my);
                keep = (0.5 * qmin <= Tk);
                // rows on which the component can matter at all: |dy| <= sqrt(2 Tk Sigma_yy)
                float half = sqrt_f32(2.0f * (float)fmax(Tk, 0.0) / (float)c.iyy) + 1.0f;
                rlo = max(ra, (int)floorf((float)(c.my - (double)Y0) - half));
                rhi = min(rb, (int)ceilf((float)(c.my - (double)Y0) + half) + 1);
                keep = keep && (rhi > rlo);
            } else {
                keep = true;
            }
            Lk = seg_len(c.qc, fmin(fmax(Tk, 1.0), 300.0));
        }
        const unsigned long long km = __ballot(keep);
        const int Kk = __popcll(km);
        __syncthreads();   // previous source's table reads are done
        if (keep) {
            int p = __popcll(km & ((1ull << lane) - 1ull));
            T.A[p] = c.A; T.mx[p] = c.mx; T.my[p] = c.my;
            T.qa[p] = c.qa * EXP_SCALE; T.qb[p] = c.qb * EXP_SCALE; T.qc[p] = c.qc * EXP_SCALE;
            T.eq[p] = exp(-c.qc);
            T.L[p] = Lk;
            T.r0[p] = rlo; T.r1[p] = rhi;
        }
        __syncthreads();
        if (a.variant == 0) {
            for (int row = ra; row < rb; row++) {
                double v = eval_direct(T, 0, Kk, x, (double)(Y0 + row), 1.0 / EXP_SCALE);
                if (on) acc[row * TILE_W + lane] += v;
            }
            continue;
        }
        for (int k0 = 0; k0 < Kk; k0 += REC_G) {
            const int kn = min(REC_G, Kk - k0);
            // the group walks the union of its components' row ranges with the shortest of their
            // safe segment lengths (neighbouring k have similar widths, so little is wasted)
            int L = T.L[k0], ga = T.r0[k0], gb = T.r1[k0];
            for (int i = 1; i < kn; i++) {
                L = min(L, T.L[k0 + i]);
                ga = min(ga, T.r0[k0 + i]);
                gb = max(gb, T.r1[k0 + i]);
            }
            L = __builtin_amdgcn_readfirstlane(L);
            ga = __builtin_amdgcn_readfirstlane(ga);
            gb = __builtin_amdgcn_readfirstlane(gb);
            if (L < 4) {
                // pathologically sharp component: evaluate this group directly
                for (int row = ga; row < gb; row++) {
                    double v = eval_direct(T, k0, k0 + kn, x, (double)(Y0 + row), 1.0 / EXP_SCALE);
                    if (on) acc[row * TILE_W + lane] += v;
                }
                continue;
            }
            double *col = acc + lane;
            switch (kn) {
            case 6: rec_group<6>(T, et, k0, x, Y0, ga, gb, L, on, col); break;
            case 5: rec_group<5>(T, et, k0, x, Y0, ga, gb, L, on, col); break;
            case 4: rec_group<4>(T, et, k0, x, Y0, ga, gb, L, on, col); break;
            case 3: rec_group<3>(T, et, k0, x, Y0, ga, gb, L, on, col); break;
            case 2: rec_group<2>(T, et, k0, x, Y0, ga, gb, L, on, col); break;
            default: rec_group<1>(T, et, k0, x, Y0, ga, gb, L, on, col); break;
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
    if (a.timing && lane == 0) {   // diagnostic build of the launch only (CEL_OPT_TILE_TIMING)
        a.timing[3 * (size_t)tile + 0] = t_start;
        a.timing[3 * (size_t)tile + 1] = wall_clock64();
        a.timing[3 * (size_t)tile + 2] = ((unsigned long long)tile << 32) | (unsigned)cnt;
    }
}

// fixed-order reduction of the per-tile partials: one block per band.  Only the tiles of tile rows [ty0, ty1) count (row
// width ntx; the whole band: ntx = per_band, rows [0, 1)): a rank's image set that holds a halo around the rows it OWNS
// (cel_images_set_noise_rows) adds its own rows' terms only -- the same terms, in the same order, as an image set of those rows
__global__ void __launch_bounds__(256)
k_reduce(const double *__restrict__ partials, int per_band, double *__restrict__ ll_band, int ntx, int ty0, int ty1) {
    __shared__ double sm[256];
    int b = blockIdx.x;
    const double *p = partials + (int64_t)b * per_band + (int64_t)ty0 * ntx;
    per_band = min(per_band, ty1 * ntx) - ty0 * ntx;
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
