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
//   k_bin_*   two-level binning (256 x 256 super-tiles, then render tiles) with ballot + prefix
//             popcount compaction: every tile's source list comes out in source order --
//             deterministic, no sort, no atomics on list contents; k_order launches the
//             heaviest tiles first.
//   k_render_hw / k_render
//             one wave per tile (32 x 64 with two component groups per column, or 64 x 32 with
//             one lane per column), coalesced row segments.  Gathers every
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
//   k_patch_ll / k_estep_* / k_photon_split
//             per-source conditional log-likelihoods, E-step reductions, Gibbs photon split.
// All arithmetic is fp64 on the vector ALU; MFMA is not used (no contraction in this path).

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <new>
#include <vector>
#include <type_traits>

#include "../../include/celeste_hip.h"

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

// A failing call leaves the function at once -- but asynchronous copies queued earlier in it may still be reading
// a local (a std::vector of staged arguments) or a caller's buffer: the error path drains the device first, so that
// nothing is in flight when those go out of scope.  (Error paths only; scratch_get fails through here as well.)
#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) {                                                               \
            (void)hipDeviceSynchronize();                                                     \
            return fail(e_ == hipErrorOutOfMemory ? CEL_ERR_NOMEM : CEL_ERR_HIP, "%s: %s (%s:%d)", \
                        #expr, hipGetErrorString(e_), __FILE__, __LINE__);                    \
        }                                                                                     \
    } while (0)

#include "device_common.h"
#include "k_prep_bin.h"
#include "k_bin2.h"
#include "k_render.h"
#include "k_render_hw.h"
#include "k_render_stars.h"
#include "k_small_stars.h"
#include "k_border.h"
#include "k_render_qw.h"
#include "k_misc.h"
#include "k_patch_ll.h"
#include "k_slice_gen.h"
#include "k_estep.h"
#include "k_split.h"
#include "k_slice.h"

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
struct Prof {
    // a ring of event pairs, created once when profiling is switched on: nothing is allocated inside
    // a timed loop, and finished pairs are read back a few at a time as new ones are handed out
    static const int PAIRS = 1024;
    hipEvent_t *ev = nullptr;   // 2 * PAIRS events; pair i = ev[2i], ev[2i+1]
    int kid[PAIRS];
    int head = 0, count = 0;    // next pair to hand out; pairs handed out and not yet read
    double sum_ms[CEL_K_COUNT] = {0};
    int64_t n[CEL_K_COUNT] = {0};
    unsigned seen[CEL_K_COUNT] = {0};   // launches of each kernel since the reset (level 3 times every fourth)
};

#ifndef SLICE_FUSE_DEFAULT
#define SLICE_FUSE_DEFAULT 0       // CEL_OPT_SLICE_FUSE: off.  Built and measured in round 6 (cel_slice_locations): no gain at any threshold
#endif
struct cel_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int variant = 1;
    // CEL_OPT_TAIL_LOG.  Two thresholds with different defaults: the FIELD render drops against the sky (24: a skipped
    // component is below eps * 4e-11; the benchmark field's log-likelihood keeps all 16 digits, its kernel takes 13 % less
    // time than at 32), the per-source kernels against the source's own value (32: their outputs feed acceptance ratios
    // and are tested to 1e-11).  Setting the option sets both; CEL_TAIL_LOG in the environment is the initial value of both.
    // (the environment value is held to the option's own range, [0, 300]; anything else is ignored)
    static bool env_tail_ok() { const char *e = getenv("CEL_TAIL_LOG"); return e && atof(e) >= 0.0 && atof(e) <= 300.0; }
    double tail_T = env_tail_ok() ? atof(getenv("CEL_TAIL_LOG")) : 32.0;
    double render_T = env_tail_ok() ? atof(getenv("CEL_TAIL_LOG")) : 24.0;
    int live = 0;               // image sets and source sets of this context that have not been destroyed (cel_ctx_destroy refuses)
    int split_full = 0;         // CEL_OPT_SPLIT_FULL_BOX
    int slice_fuse = SLICE_FUSE_DEFAULT;   // CEL_OPT_SLICE_FUSE
    int incremental = (getenv("CEL_INCREMENTAL") && atoi(getenv("CEL_INCREMENTAL")) == 0) ? 0 : 1;       // CEL_OPT_INCREMENTAL
    int tile_parts = (getenv("CEL_TILE_PARTS") && (atoi(getenv("CEL_TILE_PARTS")) == 1 || atoi(getenv("CEL_TILE_PARTS")) == 2 || atoi(getenv("CEL_TILE_PARTS")) == 4))
                         ? atoi(getenv("CEL_TILE_PARTS")) : 0;       // CEL_OPT_TILE_PARTS (the env var: the initial value, for A/B runs)
    int profile = 0;          // CEL_OPT_PROFILE: 0 off, 1 every kernel, 2 the evaluating kernels only
    int star_tiles = (getenv("CEL_STAR_TILES") && atoi(getenv("CEL_STAR_TILES")) >= 0 && atoi(getenv("CEL_STAR_TILES")) <= 3)
                         ? atoi(getenv("CEL_STAR_TILES")) : 1;       // CEL_OPT_STAR_TILES (the env var: the initial value, for test runs)
    int n_cu = 256;           // compute units of the device (set at creation): the binning pass sizes its blocks by it
    int tile_order = 1;       // 0 = launch k_render tiles in index order, 1 = heaviest first by the last render's measured tile durations (estimate when none), 2 = heaviest first by the estimate only
    int tile_rows = 32;       // rows per render tile (32 or 64), read when an image set is created
    bool tile_timing = false; // diagnostic: k_render stamps each tile's start/end wall clock
    double nz_bias = getenv("CEL_NZ_BIAS") ? atof(getenv("CEL_NZ_BIAS")) : 4.0;   // see k_nz_layout (the env var: experiments only)
    int split_reuse = getenv("CEL_SPLIT_REUSE") ? std::min(std::max(atoi(getenv("CEL_SPLIT_REUSE")), 0), 2) : 2;      // CEL_OPT_SPLIT_REUSE (the env var: the initial value, for A/B runs): 1 = the split's totals from a model image already on the device (k_border.h), 2 = ... and the stamp masses from the split's own sums
    bool mass_reuse_of() const { return split_reuse >= 2; }
    int nz_force = 0;         // CEL_OPT_PHOTON_LISTS: 0 = per patch, whichever is estimated cheaper; 1 = every patch at its photons; 2 = never
    int debug = 0;            // CEL_OPT_DEBUG: timing-only ablation bits handed to the render kernel (results are wrong when set)
    int tile_layout = 1;      // 0: 64 x tile_rows tiles, one lane per column (k_render)
                              // 1: 32 x 64 tiles, two component groups per column (k_render_hw)
    Prof prof;
    double *pinned = nullptr;   // MAX_BANDS + 16 doubles of pinned host memory for readbacks
    hipEvent_t slice_ev[2] = {nullptr, nullptr};     // cel_slice_locations: one per batch of rounds in flight
    // grow-only device scratch for the small per-call buffers of the stamp / patch-ll entry
    // points (a hipMalloc + hipFree pair per call costs more than the kernels they bracket)
    void *scratch[8] = {nullptr};
    size_t scratch_cap[8] = {0};
};

static int scratch_get(cel_ctx *c, int slot, size_t bytes, void **out) {
    if (bytes == 0) bytes = 8;
    if (bytes > c->scratch_cap[slot]) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (c->scratch[slot]) (void)hipFree(c->scratch[slot]);
        c->scratch[slot] = nullptr; c->scratch_cap[slot] = 0;
        size_t cap = bytes + bytes / 2 + 256;
        HIP_TRY(hipMalloc(&c->scratch[slot], cap));
        c->scratch_cap[slot] = cap;
    }
    *out = c->scratch[slot];
    return CEL_OK;
}

struct cel_images {
    cel_ctx *ctx = nullptr;
    int B = 0, H = 0, W = 0;   // H = rows held on the device (the window height)
    int full_H = 0, win_y0 = 0;  // the window is rows [win_y0, win_y0 + H) of a full_H-row frame
    int noise_y0 = 0, noise_y1 = 0x7fffffff;     // window rows whose sky photons cel_photon_split's noise sums count
    int TW = 64, TH = 32, ntx = 0, nty = 0;   // render tile geometry (fixed at creation)
    cel_band hb[MAX_BANDS];
    BandDev *d_bands = nullptr;
    double *d_nelec = nullptr, *d_lambda = nullptr, *d_partials = nullptr, *d_llband = nullptr;
    bool have_nelec = false;
    // per-render scratch, grown on demand
    double *d_slabs = nullptr;       // k_render_hw<, PARTS>: PARTS accumulator slabs per render tile, allocated with the first such launch
    int *d_part_cnt = nullptr;       // ... and the tiles' arrival counters
    int slabs_parts = 0;
    SrcRec *d_recs = nullptr;
    int4 *d_boxes = nullptr;
    int *d_kind = nullptr;
    int *d_status = nullptr;       // per (band, source): 1 has a stamp, 0 empty box, -1 overlap-test miss
    int64_t recs_cap = 0;
    // which sources the records on the device belong to (cel_sources::gen, unique per cel_sources_set),
    // and the host copy of their boxes / status that cel_stamp_boxes / cel_source_boxes hand out:
    // a stamp call (boxes, then stamps) runs k_prep and the D2H once, not twice
    uint64_t recs_gen = 0, hbox_gen = 0;
    // which sources the model image in d_lambda (full boxes, the current sky levels) and the tile lists belong to: what lets
    // the photon split take its totals from that image (k_border.h); 0 = not valid
    uint64_t lambda_gen = 0, lists_gen = 0;
    unsigned long long *d_massfx = nullptr;   // the split's integer stamp-mass sums (k_split.h, k_border.h), per (source, band)
    int64_t massfx_cap = 0;
    uint64_t massfx_gen = 0;                  // ... hold the masses of the catalogue of this generation (0: of none)
    int *d_mass_todo = nullptr;               // (source, band) jobs the mass kernel proper still has to do + their count behind them
    std::vector<int4> h_boxes;
    std::vector<int> h_status;
    int *d_tile_cnt = nullptr, *d_tile_nstar = nullptr, *d_tile_work = nullptr, *d_order = nullptr;
    int *d_tile_cost = nullptr;   // measured duration of every tile in the last render (the next render's launch order)
    int64_t cost_S = -1;          // number of sources that render had (-1: nothing measured yet)
    bool bin_two_level = false;   // a super-tile once held more than BIN_CH candidates: coarse lists in global memory from then on
    int64_t mass_pending = -1;       // doubles waiting in d_mass between cel_stamp_mass_begin and _end (-1: none)
    int64_t mass_todo_S = -1;        // >= 0: that _begin took the short cut on a catalogue of so many sources; _end finishes its leftovers
    uint64_t mass_gen = 0;           // ... of the catalogue of this generation, with the to-do list at mass_todo_ptr: what _end checks
    const int *mass_todo_ptr = nullptr; //  before it launches on the leftovers (a call in between may have re-run k_prep for another
                                     //     catalogue or re-allocated the split's buffers)
    double *d_mass = nullptr;        // (a scratch slot of the context: not owned)
    long long *d_btot = nullptr;     // per-1024-entries totals of the patch / list layout scans
    bool nelec_u16 = false;          // every observed pixel in 0 ... 65 535: the split's 16-bit photons-left plane
    bool star_one_segment = false;   // every band passes star_setup's test: k_render_stars may take star tiles
    int64_t order_S = -1;         // d_order already holds the heaviest-first order of those costs (sorted behind that render's readback)
    hipEvent_t ev_step = nullptr; // marks a step's readback copy: the host waits for it, not for the sort queued behind it
    int64_t *d_tile_off = nullptr;
    unsigned long long *d_cursor = nullptr;   // fine cursor, fine overflow, coarse cursor, coarse overflow
    int *d_lists = nullptr;
    int64_t lists_cap = 0;
    int nsx = 0, nsy = 0;                     // 256 x 256 super-tiles of the coarse binning level
    int *d_sup_cnt = nullptr;
    int64_t *d_sup_off = nullptr;
    int *d_clist = nullptr;
    int64_t clist_cap = 0;
    unsigned long long *d_timing = nullptr;   // CEL_OPT_TILE_TIMING diagnostic stamps, 3 per tile
    // device-resident sample patches of the last resident photon split (source-major, index s*B+b)
    int *d_samp = nullptr;      // the resident photon split's sample patches: int32 photon counts, source-major
    int4 *d_snz = nullptr;      // nonzero rectangles of the resident sample patches (k_patch_nzbox)
    double *d_ssum = nullptr;   // photons per (source, band) of the resident split, summed by the split kernel itself
    bool ssum_valid = false;
    // host copies of the last resident split's sums and patch offsets (pinned), made INSIDE cel_photon_split before its last
    // wait: cel_samples_fetch hands them over without touching the stream, on which the photon lists are still being compacted
    double *h_ssum = nullptr;
    int64_t *h_soff = nullptr;
    int64_t hsum_cap = 0;
    bool hsum_valid = false;
    // photon lists of the resident split (k_nz_layout / k_nz_compact): per patch the pixels that hold a photon
    int *d_nnz = nullptr, *d_nzmode = nullptr;
    int64_t *d_nzoff = nullptr;
    NzEntry *d_nzlist = nullptr;
    int64_t nzlist_cap = 0;
    bool nz_valid = false;
    double *d_rate = nullptr;   // per-pixel total rates of the photon split (strict boxes), B*H*W, on first use
    bool rate_in_lambda = false;    // the last split read its totals from the model image itself (CEL_OPT_SPLIT_FULL_BOX with a current image)
    int64_t samp_cap = 0;
    int4 *d_sbox = nullptr;
    int64_t *d_soff = nullptr;
    int64_t slay_cap = 0, samp_S = 0, samp_total = 0;
    double *d_stats = nullptr;
    // device-resident slice sampler (cel_slice_locations): chain state, proposal set, per-round outputs
    void *d_slice = nullptr;
    size_t slice_cap = 0;
    cel_sources *slice_prop = nullptr;
    void *d_sgen = nullptr;     // the general slice sampler's state (cel_slice_sample)
    size_t sgen_cap = 0;
    cel_sources *sgen_prop = nullptr;
    int64_t last_S = 0;
    double last_entries = 0;
    bool counted = false;            // in ctx->live
    bool nelec_shared = false;       // cel_images_device_ptrs handed the observed pixels' device pointer out
    uint64_t partials_gen = 0;       // the per-tile Poisson partials in d_partials are those of this catalogue generation's render (0: not)
    uint64_t lambda_uid = 0;         // the catalogue OBJECT whose generation lambda_gen is (the incremental render compares row stamps of the same object only)
    double lambda_T = -1.0;          // ... and the drop threshold that image was rendered at
    int lambda_parts = 0;            // ... with so many blocks per tile
    int *d_dirty = nullptr;          // per tile: touched by a changed source's box (the incremental render)
    int64_t last_dirty = -1;         // tiles the last render rendered incrementally (-1: it rendered every tile)
    // the one-launch path of a small star field (k_small_stars.h): per-block partials + per-band arrival counters
    double *d_small = nullptr, *h_small = nullptr;      // one buffer: pinned host memory and its device address
    double *d_small_consts = nullptr;
    unsigned long long small_seq = 0;
    bool llband_on_host = false;  // the last render's per-band sums were formed on the host (the small path): in h_llband
    double h_llband[MAX_BANDS] = {0};
    bool small_off = false;       // a part once held more stars than the kernel stages: the general path from then on
};

static std::atomic<uint64_t> g_source_gen{0};

struct cel_sources {
    cel_ctx *ctx = nullptr;
    bool counted = false;          // in ctx->live
    int64_t cap = 0, S = 0;
    int B = 0;
    uint64_t gen = 0;              // changes with every cel_sources_set (process-wide counter)
    uint64_t uid = 0;              // which catalogue object this is (process-wide, never reused: a generation alone does not say whose it is)
    uint64_t full_gen = 0;         // ... the generation of the last change of the WHOLE catalogue (cel_sources_set, a sampler's update)
    std::vector<uint64_t> row_gen; // per row: the generation of its last change by cel_sources_set_rows (empty: none since full_gen)
    int64_t n_gal = -1;            // entries that are not stars (type != 0); -1 = unknown (types set from device memory)
    int *d_type = nullptr;
    double *d_radec = nullptr, *d_counts = nullptr, *d_shape = nullptr;
    std::vector<int32_t> h_type;   // the types as last set from host memory (cel_sources_set_rows keeps n_gal exact with them)
};

// The launch order of the tiles matters only when there are more tiles than the chip runs at once (2048 waves of the
// general kernel): below that every tile starts at once whatever the order, and a sort + its launch is all cost.
// A frame of few tiles is rendered by several one-wave blocks per tile (k_render_hw<, PARTS>, k_render_hw.h): 4 while even
// that many fit the chip's 2 048 wave slots at once (512 tiles), 2 up to 3 072 tiles (CEL_OPT_TILE_PARTS: 0 = this rule,
// 1 / 2 / 4 = always that many).  Measured on one rank's strip of the benchmark field cut 8 ways (1 280 tiles): one wave per
// tile 0.37 ms, two 0.21, four 0.25-0.29 -- every part pays its tile set-up and slab traffic, and 5 120 blocks are 2.5
// rounds of the chip.  (A tile whose list is short uses fewer of its parts: one per HW_PART_ENTRIES entries.)  The 32 x 64 layout's general kernel only; the star-tile kernels and the diagnostic instantiation
// keep one wave per tile.
static inline int tile_parts_of(const cel_ctx *c, const cel_images *im) {
    if (im->TW != 32 || c->variant == 0) return 1;
    if (c->tile_parts) return c->tile_parts;
    const int64_t T = (int64_t)im->B * im->ntx * im->nty;
    return T <= 512 ? 4 : (T <= 3072 ? 2 : 1);
}
static inline int tile_order_of(const cel_ctx *c, const cel_images *im) {
    return ((int64_t)im->B * im->ntx * im->nty * tile_parts_of(c, im) > 2048) ? c->tile_order : 0;
}

static bool prof_alloc(Prof &p) {
    if (p.ev) return true;
    p.ev = (hipEvent_t *)calloc(2 * Prof::PAIRS, sizeof(hipEvent_t));
    if (!p.ev) return false;
    for (int i = 0; i < 2 * Prof::PAIRS; i++)
        if (hipEventCreate(&p.ev[i]) != hipSuccess) return false;
    return true;
}
// read the oldest outstanding pair; `wait`: block until it has completed
static bool prof_harvest_one(Prof &p, bool wait) {
    if (p.count == 0) return false;
    const int t = (p.head - p.count + Prof::PAIRS) % Prof::PAIRS;
    if (wait) (void)hipEventSynchronize(p.ev[2 * t + 1]);
    else if (hipEventQuery(p.ev[2 * t + 1]) != hipSuccess) return false;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, p.ev[2 * t], p.ev[2 * t + 1]) == hipSuccess) {
        p.sum_ms[p.kid[t]] += ms;
        p.n[p.kid[t]] += 1;
    }
    p.count--;
    return true;
}
// reserve a pair for kernel k; returns the index of its first event, -1 when profiling is off
static int prof_slot(cel_ctx *c, int k) {
    if (!c->profile) return -1;
    // level 2: the small kernels around a render (prep, binning, reduction) go unbracketed -- an event pair costs the host
    // ~10 us per launch, 3 % of a 1.35 ms step when every kernel carries one
    if (c->profile >= 2 && (k == CEL_K_PREP || k == CEL_K_BIN || k == CEL_K_REDUCE)) return -1;
    // level 3: a SAMPLE of the evaluating launches -- every fourth of a kernel -- for steps so short that the pair's ~10 us
    // of host time is a fifth of what is being measured (configs[1]: a 44 us step)
    if (c->profile == 3 && (c->prof.seen[k]++ & 3) != 0) return -1;
    Prof &p = c->prof;
    if (!prof_alloc(p)) return -1;
    if (p.count == Prof::PAIRS) prof_harvest_one(p, true);
    else if (p.count > Prof::PAIRS / 2) { if (prof_harvest_one(p, false)) prof_harvest_one(p, false); }
    const int t = p.head;
    p.head = (p.head + 1) % Prof::PAIRS;
    p.count++;
    p.kid[t] = k;
    return 2 * t;
}
static int prof_begin(cel_ctx *c, int k) {
    const int i = prof_slot(c, k);
    if (i >= 0) (void)hipEventRecord(c->prof.ev[i], c->stream);
    return i;
}
static void prof_end(cel_ctx *c, int i) {
    if (i >= 0) (void)hipEventRecord(c->prof.ev[i + 1], c->stream);
}
// The step's own kernels (prep, binning, render, reduce) are timed WITHOUT marker packets: an event
// pair is reserved here and handed to hipExtLaunchKernelGGL, which stamps it from the dispatch's own
// start / completion signals.  A hipEventRecord between two kernels opens a ~10 us bubble in the
// queue (rocprofv3 kernel trace): four of them per 1.6 ms step were 2 % of what was being measured.
#define EV0(c, i) ((i) >= 0 ? (c)->prof.ev[(i)] : (hipEvent_t) nullptr)
#define EV1(c, i) ((i) >= 0 ? (c)->prof.ev[(i) + 1] : (hipEvent_t) nullptr)
// launch with optional start / stop events attached to the dispatch
#define LAUNCH_EV(kernel, grid, block, st, e0, e1, ...)                                              \
    do {                                                                                             \
        hipEvent_t e0_ = (e0), e1_ = (e1);                                                           \
        if (e0_ || e1_) hipExtLaunchKernelGGL(kernel, grid, block, 0, st, e0_, e1_, 0, __VA_ARGS__); \
        else hipLaunchKernelGGL(kernel, grid, block, 0, st, __VA_ARGS__);                            \
    } while (0)
static void prof_collect(cel_ctx *c) {   // after a stream synchronise: everything outstanding has completed
    while (prof_harvest_one(c->prof, true)) {}
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
    {
        int ncu = 0;
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && ncu > 0) c->n_cu = ncu;
    }
    hipError_t e = hipHostMalloc((void **)&c->pinned, sizeof(double) * (MAX_BANDS + 16), hipHostMallocDefault);
    if (e != hipSuccess) { delete c; return fail(CEL_ERR_HIP, "hipHostMalloc: %s", hipGetErrorString(e)); }
    // profile constants, normalised as mixture_profiles.py:13,19
    double amp[K_PROF], var[K_PROF], se = 0.0, sd = 0.0;
    for (int i = 0; i < 6; i++) se += H_EXP_AMP[i];
    for (int i = 0; i < 8; i++) sd += H_DEV_AMP[i];
    for (int i = 0; i < 6; i++) { amp[i] = H_EXP_AMP[i] / se; var[i] = H_EXP_VAR[i]; }
    for (int i = 0; i < 8; i++) { amp[6 + i] = H_DEV_AMP[i] / sd; var[6 + i] = H_DEV_VAR[i]; }
    double ic[64], lc[64];
    for (int j = 0; j < 64; j++) {
        long double cj = 1.0L + ((long double)j + 0.5L) / 64.0L;
        ic[j] = (double)(1.0L / cj);
        lc[j] = (double)(-logl((long double)ic[j]));
    }
    if ((e = hipMemcpyToSymbol(HIP_SYMBOL(c_prof_amp), amp, sizeof(amp))) != hipSuccess ||
        (e = hipMemcpyToSymbol(HIP_SYMBOL(c_prof_var), var, sizeof(var))) != hipSuccess ||
        (e = hipMemcpyToSymbol(HIP_SYMBOL(c_log_ic), ic, sizeof(ic))) != hipSuccess ||
        (e = hipMemcpyToSymbol(HIP_SYMBOL(c_log_lc), lc, sizeof(lc))) != hipSuccess) {
        (void)cel_ctx_destroy(c);           // (the stream and the pinned block go with it)
        return fail(CEL_ERR_HIP, "hipMemcpyToSymbol: %s", hipGetErrorString(e));
    }
    *out = c;
    return CEL_OK;
}

int cel_ctx_destroy(cel_ctx *c) {
    if (!c) return CEL_OK;
    if (c->live > 0)        // their destructors synchronise this context's stream: destroy them first
        return fail(CEL_ERR_INVALID, "cel_ctx_destroy: %d image / source sets of this context are still alive", c->live);
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->prof.ev)
        for (int i = 0; i < 2 * Prof::PAIRS; i++)
            if (c->prof.ev[i]) (void)hipEventDestroy(c->prof.ev[i]);
    free(c->prof.ev);
    for (int k = 0; k < 2; k++) if (c->slice_ev[k]) (void)hipEventDestroy(c->slice_ev[k]);
    if (c->pinned) (void)hipHostFree(c->pinned);
    for (void *p : c->scratch)
        if (p) (void)hipFree(p);
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
        if (v != v) { c->tail_T = 32.0; c->render_T = 24.0; return CEL_OK; }       // NaN: the defaults
        if (!(v >= 0.0) || v > 300.0) return fail(CEL_ERR_INVALID, "CEL_OPT_TAIL_LOG must be in [0, 300] (NaN: the defaults)");
        c->tail_T = v;
        c->render_T = v;
        return CEL_OK;
    case CEL_OPT_TAIL_LOG_SOURCE:
        if (v != v) { c->tail_T = 32.0; return CEL_OK; }
        if (!(v >= 0.0) || v > 300.0) return fail(CEL_ERR_INVALID, "CEL_OPT_TAIL_LOG_SOURCE must be in [0, 300] (NaN: the default)");
        c->tail_T = v;
        return CEL_OK;
    case CEL_OPT_INCREMENTAL:
        if (v != 0.0 && v != 1.0) return fail(CEL_ERR_INVALID, "CEL_OPT_INCREMENTAL must be 0 or 1");
        c->incremental = (int)v;
        return CEL_OK;
    case CEL_OPT_SPLIT_FULL_BOX:
        if (v != 0.0 && v != 1.0) return fail(CEL_ERR_INVALID, "CEL_OPT_SPLIT_FULL_BOX must be 0 or 1");
        c->split_full = (int)v;
        return CEL_OK;
    case CEL_OPT_SLICE_FUSE:
        if (!(v >= 0.0 && v <= 1e9) || v != floor(v)) return fail(CEL_ERR_INVALID, "CEL_OPT_SLICE_FUSE must be 0, 1 or a block count");
        c->slice_fuse = (int)v;
        return CEL_OK;
    case CEL_OPT_TILE_PARTS:
        if (v != 0.0 && v != 1.0 && v != 2.0 && v != 4.0) return fail(CEL_ERR_INVALID, "CEL_OPT_TILE_PARTS must be 0 (by the frame's size), 1, 2 or 4");
        c->tile_parts = (int)v;
        return CEL_OK;
    case CEL_OPT_PROFILE:
        if (v != 0.0 && v != 1.0 && v != 2.0 && v != 3.0) return fail(CEL_ERR_INVALID, "CEL_OPT_PROFILE must be 0, 1, 2 or 3");
        if (v != 0.0 && !prof_alloc(c->prof)) return fail(CEL_ERR_HIP, "CEL_OPT_PROFILE: cannot create the timing events");
        c->profile = (int)v;
        return CEL_OK;
    case CEL_OPT_TILE_ORDER:
        if (v != 0.0 && v != 1.0 && v != 2.0) return fail(CEL_ERR_INVALID, "CEL_OPT_TILE_ORDER must be 0, 1 or 2");
        c->tile_order = (int)v;
        return CEL_OK;
    case CEL_OPT_TILE_ROWS:
        if (v != 32.0 && v != 64.0) return fail(CEL_ERR_INVALID, "CEL_OPT_TILE_ROWS must be 32 or 64");
        c->tile_rows = (int)v;
        return CEL_OK;
    case CEL_OPT_TILE_TIMING:
        c->tile_timing = (v != 0.0);
        return CEL_OK;
    case CEL_OPT_TILE_LAYOUT:
        if (v != 0.0 && v != 1.0 && v != 2.0) return fail(CEL_ERR_INVALID, "CEL_OPT_TILE_LAYOUT must be 0, 1 or 2");
        c->tile_layout = (int)v;
        return CEL_OK;
    case CEL_OPT_PHOTON_LISTS:
        if (v != 0.0 && v != 1.0 && v != 2.0) return fail(CEL_ERR_INVALID, "CEL_OPT_PHOTON_LISTS must be 0, 1 or 2");
        c->nz_force = (int)v;
        return CEL_OK;
    case CEL_OPT_SPLIT_REUSE:
        if (v != 0.0 && v != 1.0 && v != 2.0) return fail(CEL_ERR_INVALID, "CEL_OPT_SPLIT_REUSE must be 0, 1 or 2");
        c->split_reuse = (int)v;
        return CEL_OK;
    case CEL_OPT_STAR_TILES:
        if (v != 0.0 && v != 1.0 && v != 2.0 && v != 3.0) return fail(CEL_ERR_INVALID, "CEL_OPT_STAR_TILES must be 0, 1, 2 or 3");
        c->star_tiles = (int)v;
        return CEL_OK;
    case CEL_OPT_DEBUG:
        if (!(v >= 0.0) || v > 4095.0) return fail(CEL_ERR_INVALID, "CEL_OPT_DEBUG must be in [0, 4095]");
#ifndef CEL_ABLATE
        // the shipped library carries no ablation path: only the two result-preserving diagnostic bits exist
        if (((int)v) & ~(64 | 128))
            return fail(CEL_ERR_INVALID, "CEL_OPT_DEBUG: the timing-only ablation bits exist only in a -DCEL_ABLATE build "
                                         "(make -C desi-mcmc_amd/csrc ablate); this library accepts 64 and 128");
#endif
        c->debug = (int)v;
        return CEL_OK;
    }
    return fail(CEL_ERR_INVALID, "unknown option %d", key);
}

int cel_ctx_get_option(cel_ctx *c, int key, double *v) {
    if (!c || !v) return fail(CEL_ERR_INVALID, "null argument");
    switch (key) {
    case CEL_OPT_KERNEL: *v = c->variant; return CEL_OK;
    case CEL_OPT_TAIL_LOG: *v = c->render_T; return CEL_OK;
    case CEL_OPT_TAIL_LOG_SOURCE: *v = c->tail_T; return CEL_OK;
    case CEL_OPT_TILE_PARTS: *v = c->tile_parts; return CEL_OK;
    case CEL_OPT_INCREMENTAL: *v = c->incremental; return CEL_OK;
    case CEL_OPT_SPLIT_FULL_BOX: *v = c->split_full; return CEL_OK;
    case CEL_OPT_SLICE_FUSE: *v = c->slice_fuse; return CEL_OK;
    case CEL_OPT_PROFILE: *v = (double)c->profile; return CEL_OK;
    case CEL_OPT_TILE_ORDER: *v = (double)c->tile_order; return CEL_OK;
    case CEL_OPT_TILE_ROWS: *v = c->tile_rows; return CEL_OK;
    case CEL_OPT_TILE_TIMING: *v = c->tile_timing ? 1.0 : 0.0; return CEL_OK;
    case CEL_OPT_TILE_LAYOUT: *v = c->tile_layout; return CEL_OK;
    case CEL_OPT_PHOTON_LISTS: *v = c->nz_force; return CEL_OK;
    case CEL_OPT_STAR_TILES: *v = c->star_tiles; return CEL_OK;
    case CEL_OPT_SPLIT_REUSE: *v = c->split_reuse; return CEL_OK;
    case CEL_OPT_DEBUG: *v = c->debug; return CEL_OK;
    }
    return fail(CEL_ERR_INVALID, "unknown option %d", key);
}

// ---- images ---------------------------------------------------------------------------------
int cel_images_destroy(cel_images *im) {
    if (!im) return CEL_OK;
    (void)hipSetDevice(im->ctx->device);
    (void)hipStreamSynchronize(im->ctx->stream);
    void *ptrs[] = {im->d_bands, im->d_nelec, im->d_lambda, im->d_partials, im->d_llband, im->d_recs,
                    im->d_boxes, im->d_kind, im->d_status, im->d_tile_cnt, im->d_tile_nstar, im->d_tile_work, im->d_tile_cost, im->d_order, im->d_tile_off, im->d_lists, im->d_stats,
                    im->d_sup_cnt, im->d_sup_off, im->d_clist, im->d_timing, im->d_samp, im->d_sbox, im->d_soff, im->d_rate, im->d_snz, im->d_ssum,
                    im->d_nnz, im->d_nzmode, im->d_nzoff, im->d_nzlist, im->d_btot, im->d_slabs, im->d_part_cnt, im->d_dirty};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    if (im->d_slice) (void)hipFree(im->d_slice);
    if (im->slice_prop) cel_sources_destroy(im->slice_prop);
    if (im->sgen_prop) cel_sources_destroy(im->sgen_prop);
    if (im->d_sgen) (void)hipFree(im->d_sgen);
    if (im->h_small) (void)hipHostFree(im->h_small);
    if (im->d_small_consts) (void)hipFree(im->d_small_consts);
    if (im->d_massfx) (void)hipFree(im->d_massfx);
    if (im->h_ssum) (void)hipHostFree(im->h_ssum);
    if (im->h_soff) (void)hipHostFree(im->h_soff);
    if (im->d_mass_todo) (void)hipFree(im->d_mass_todo);
    if (im->ev_step) (void)hipEventDestroy(im->ev_step);
    if (im->counted) im->ctx->live--;
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
    if (c->tile_layout == 1) { im->TW = HW_TW; im->TH = HW_TH; }
    else if (c->tile_layout == 2) { im->TW = QW_TW; im->TH = QW_TH; }
    else { im->TW = TILE_W; im->TH = c->tile_rows; }
    im->ntx = (W + im->TW - 1) / im->TW;
    im->nty = (H + im->TH - 1) / im->TH;
    im->nsx = (W + SUPER_W - 1) / SUPER_W;
    im->nsy = (H + SUPER_H - 1) / SUPER_H;
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
    im->star_one_segment = true;
    for (int b = 0; b < B; b++)
        for (int k = 0; k < K_PSF; k++) {      // star_setup's test (k_render_hw.h), on the host
            const BandDev &d = hb[b];
            const double inv = 1.0 / (d.cxx[k] * d.cyy[k] - d.cxy[k] * d.cxy[k]);
            const double rb_ = d.R + 2.0;
            const double emax = 0.5 * quad_max_rect_hw(d.cyy[k] * inv, -d.cxy[k] * inv, d.cxx[k] * inv, -rb_ - d.mux[k], rb_ - d.mux[k],
                                                       -rb_ - d.muy[k], rb_ - d.muy[k]);
            if (!(emax <= 0.999 * STAR_EMAX)) im->star_one_segment = false;
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
    // per-band sums and the binning cursors share one buffer: they ride back to the host in one copy
    IM_TRY(hipMalloc((void **)&im->d_llband, sizeof(double) * MAX_BANDS + sizeof(unsigned long long) * 4));
    im->d_cursor = reinterpret_cast<unsigned long long *>(im->d_llband + MAX_BANDS);
    IM_TRY(hipMalloc((void **)&im->d_tile_cnt, sizeof(int) * T));
    IM_TRY(hipMalloc((void **)&im->d_tile_nstar, sizeof(int) * T));
    IM_TRY(hipMalloc((void **)&im->d_tile_work, sizeof(int) * T));
    IM_TRY(hipMalloc((void **)&im->d_tile_cost, sizeof(int) * T));
    IM_TRY(hipMalloc((void **)&im->d_order, sizeof(int) * T));
    IM_TRY(hipMalloc((void **)&im->d_tile_off, sizeof(int64_t) * T));
    IM_TRY(hipMalloc((void **)&im->d_sup_cnt, sizeof(int) * B * im->nsx * im->nsy));
    IM_TRY(hipMalloc((void **)&im->d_sup_off, sizeof(int64_t) * B * im->nsx * im->nsy));
    IM_TRY(hipMalloc((void **)&im->d_stats, sizeof(double) * 2));
    IM_TRY(hipMemsetAsync(im->d_lambda, 0, sizeof(double) * npix, c->stream));
#undef IM_TRY
    im->counted = true;
    c->live++;
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
    if (rc != CEL_OK) return rc;
    im->have_nelec = true;
    im->partials_gen = 0;                 // the tiles' Poisson partials were formed against the old pixels (the incremental render keeps none of them)
    // the image's range: 0 ... 65 535 everywhere lets the photon split keep its photons-left plane in 16 bits (k_split.h)
    im->nelec_u16 = false;
    {
        const int NB = 512;
        double *d_rng = nullptr;
        HIP_TRY(hipMalloc((void **)&d_rng, sizeof(double) * 3 * NB));
        hipLaunchKernelGGL(k_nelec_range, dim3(NB), dim3(256), 0, im->ctx->stream, (const double *)im->d_nelec,
                           (int64_t)im->B * im->H * im->W, d_rng);
        std::vector<double> h((size_t)3 * NB);
        hipError_t e = hipMemcpyAsync(h.data(), d_rng, sizeof(double) * 3 * NB, hipMemcpyDeviceToHost, im->ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(im->ctx->stream);
        (void)hipFree(d_rng);
        HIP_TRY(e);
        double lo = INFINITY, hi = -INFINITY, bad = 0.0;
        for (int k = 0; k < NB; k++) { lo = fmin(lo, h[3 * k]); hi = fmax(hi, h[3 * k + 1]); bad += h[3 * k + 2]; }
        im->nelec_u16 = (bad == 0.0) && (lo >= 0.0) && (hi <= 65535.0);
    }
    return CEL_OK;
}

int cel_images_set_epsilon(cel_images *im, int band, double eps) {
    if (!im || band < 0 || band >= im->B) return fail(CEL_ERR_INVALID, "cel_images_set_epsilon: bad band");
    HIP_TRY(hipSetDevice(im->ctx->device));
    im->hb[band].eps = eps;
    im->lambda_gen = 0;                   // the model image on the device was rendered with the old sky level
    im->partials_gen = 0;                 // ... and the tiles' Poisson partials against it
    // stream-ordered, no host synchronisation (Gibbs calls this per band per sweep)
    hipLaunchKernelGGL(k_set_eps, dim3(1), dim3(1), 0, im->ctx->stream, im->d_bands, band, eps);
    HIP_TRY(hipGetLastError());
    return CEL_OK;
}

int cel_images_set_window(cel_images *im, int y0, int full_H) {
    if (!im) return fail(CEL_ERR_INVALID, "null images");
    if (y0 < 0 || full_H < 1 || (int64_t)y0 + im->H > full_H)
        return fail(CEL_ERR_INVALID, "window rows [%d, %d) do not fit a %d-row frame", y0, y0 + im->H, full_H);
    im->win_y0 = y0;
    im->full_H = full_H;
    im->recs_gen = im->hbox_gen = 0;      // boxes are cut to the window
    im->lambda_gen = im->lists_gen = 0;
    im->massfx_gen = 0;
    return CEL_OK;
}

int cel_images_set_noise_rows(cel_images *im, int y0, int y1) {
    if (!im) return fail(CEL_ERR_INVALID, "null images");
    if (y0 < 0 || y1 < y0) return fail(CEL_ERR_INVALID, "cel_images_set_noise_rows: rows [%d, %d)", y0, y1);
    im->noise_y0 = y0;
    im->noise_y1 = y1;
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
    if (nelec) {        // the caller may write it, at any time: no assumption about its range any more, no Poisson partials kept
        *nelec = im->d_nelec; im->nelec_u16 = false; im->nelec_shared = true; im->partials_gen = 0;
    }
    if (lambda) *lambda = im->d_lambda;
    return CEL_OK;
}

int cel_images_loglik_device(cel_images *im, void **ll_band) {
    if (!im || !ll_band) return fail(CEL_ERR_INVALID, "cel_images_loglik_device: null argument");
    if (im->llband_on_host) {
        // the one-launch path of a small star field adds its blocks' partials on the host: put the sums where every other
        // render leaves them
        cel_ctx *c = im->ctx;
        HIP_TRY(hipSetDevice(c->device));
        HIP_TRY(hipMemcpyAsync(im->d_llband, im->h_llband, sizeof(double) * im->B, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        im->llband_on_host = false;
    }
    *ll_band = im->d_llband;
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
    if (s->counted) s->ctx->live--;
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
    s->uid = ++g_source_gen;
    hipError_t e;
    if ((e = hipMalloc((void **)&s->d_type, sizeof(int) * capacity)) != hipSuccess ||
        (e = hipMalloc((void **)&s->d_radec, sizeof(double) * 2 * capacity)) != hipSuccess ||
        (e = hipMalloc((void **)&s->d_counts, sizeof(double) * B * capacity)) != hipSuccess ||
        (e = hipMalloc((void **)&s->d_shape, sizeof(double) * 4 * capacity)) != hipSuccess) {
        cel_sources_destroy(s);
        return fail(CEL_ERR_NOMEM, "hipMalloc(sources): %s", hipGetErrorString(e));
    }
    s->counted = true;
    s->ctx->live++;
    *out = s;
    return CEL_OK;
}

int cel_sources_set(cel_sources *s, int64_t S, const int32_t *type, const double *radec,
                    const double *counts, const double *shape, int mem) {
    if (!s || !type || !radec || !counts || !shape) return fail(CEL_ERR_INVALID, "cel_sources_set: null argument");
    if (S < 0 || S > s->cap) return fail(CEL_ERR_INVALID, "S=%lld exceeds capacity %lld", (long long)S, (long long)s->cap);
    HIP_TRY(hipSetDevice(s->ctx->device));
    hipStream_t st = s->ctx->stream;
    if (S > 0) {
        const hipMemcpyKind kind = (mem == CEL_DEVICE) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
        HIP_TRY(hipMemcpyAsync(s->d_type, type, sizeof(int) * S, kind, st));
        HIP_TRY(hipMemcpyAsync(s->d_radec, radec, sizeof(double) * 2 * S, kind, st));
        HIP_TRY(hipMemcpyAsync(s->d_counts, counts, sizeof(double) * s->B * S, kind, st));
        HIP_TRY(hipMemcpyAsync(s->d_shape, shape, sizeof(double) * 4 * S, kind, st));
        if (mem != CEL_DEVICE) HIP_TRY(hipStreamSynchronize(st));   // pageable sources must stay valid: one sync for the four
    }
    s->S = S;
    s->gen = s->full_gen = ++g_source_gen;
    s->row_gen.clear();
    s->n_gal = -1;
    s->h_type.clear();
    if (mem != CEL_DEVICE) {
        s->n_gal = 0;
        for (int64_t i = 0; i < S; i++) s->n_gal += (type[i] != 0);
        s->h_type.assign(type, type + S);
    }
    return CEL_OK;
}

// n rows of the catalogue replaced (host arrays, packed: idx[n], type[n], radec[n][2], counts[n][B], shape[n][4]): what a
// caller that changed ONE source between two evaluations uploads instead of the whole catalogue (the RJ moves and slice
// steps of CelestePy/util/infer/mcmc_transitions.py:37-152 call celeste_likelihood after every such change)
int cel_sources_set_rows(cel_sources *s, int64_t n, const int32_t *idx, const int32_t *type, const double *radec,
                         const double *counts, const double *shape) {
    if (!s || n < 0 || (n > 0 && (!idx || !type || !radec || !counts || !shape)))
        return fail(CEL_ERR_INVALID, "cel_sources_set_rows: null argument");
    for (int64_t i = 0; i < n; i++)
        if (idx[i] < 0 || idx[i] >= s->S) return fail(CEL_ERR_INVALID, "cel_sources_set_rows: row %d outside the catalogue's %lld", idx[i], (long long)s->S);
    if (n == 0) return CEL_OK;
    {   // a row named twice would be written by two threads of k_scatter_rows at once: the last writer is undefined
        std::vector<int32_t> seen(idx, idx + n);
        std::sort(seen.begin(), seen.end());
        for (int64_t i = 1; i < n; i++)
            if (seen[i] == seen[i - 1]) return fail(CEL_ERR_INVALID, "cel_sources_set_rows: row %d is named twice", seen[i]);
    }
    cel_ctx *c = s->ctx;
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    const int B = s->B;
    // one staging buffer: idx, type (ints), then radec, counts, shape (doubles)
    const size_t ints = ((size_t)(2 * n) * sizeof(int) + 7) & ~(size_t)7;
    const size_t bytes = ints + sizeof(double) * (size_t)n * (2 + B + 4);
    char *d = nullptr;
    int rc = scratch_get(c, 7, bytes, (void **)&d);
    if (rc) return rc;
    std::vector<char> h(bytes);
    memcpy(h.data(), idx, sizeof(int) * n);
    memcpy(h.data() + sizeof(int) * n, type, sizeof(int) * n);
    double *hd = reinterpret_cast<double *>(h.data() + ints);
    memcpy(hd, radec, sizeof(double) * 2 * n);
    memcpy(hd + 2 * n, counts, sizeof(double) * B * n);
    memcpy(hd + (2 + B) * n, shape, sizeof(double) * 4 * n);
    HIP_TRY(hipMemcpyAsync(d, h.data(), bytes, hipMemcpyHostToDevice, st));
    const double *dd = reinterpret_cast<const double *>(d + ints);
    hipLaunchKernelGGL(k_scatter_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, B, (const int *)d, (const int *)d + n,
                       dd, dd + 2 * n, dd + (2 + B) * n, s->d_type, s->d_radec, s->d_counts, s->d_shape);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));          // the pageable staging vector goes out of scope
    if ((int64_t)s->h_type.size() == s->S) {
        for (int64_t i = 0; i < n; i++) {
            s->n_gal += (type[i] != 0) - (s->h_type[idx[i]] != 0);
            s->h_type[idx[i]] = type[i];
        }
    } else {
        s->n_gal = -1;
    }
    s->gen = ++g_source_gen;
    // which rows changed since when: a render whose image set still holds the model image of an earlier generation of THIS
    // catalogue renders only the tiles these rows' boxes touch (render_impl: the incremental path)
    if ((int64_t)s->row_gen.size() != s->S) s->row_gen.assign((size_t)s->S, 0);
    for (int64_t i = 0; i < n; i++) s->row_gen[idx[i]] = s->gen;
    return CEL_OK;
}

// ---- prep + bin (shared by field and stamps) ------------------------------------------------
static int ensure_recs(cel_images *im, int64_t n) {
    if (n <= im->recs_cap) return CEL_OK;
    HIP_TRY(hipStreamSynchronize(im->ctx->stream));
    if (im->d_recs) (void)hipFree(im->d_recs);
    if (im->d_boxes) (void)hipFree(im->d_boxes);
    if (im->d_kind) (void)hipFree(im->d_kind);
    if (im->d_status) (void)hipFree(im->d_status);
    im->d_recs = nullptr; im->d_boxes = nullptr; im->d_kind = nullptr; im->d_status = nullptr; im->recs_cap = 0;
    im->recs_gen = im->hbox_gen = 0;
    int64_t cap = n + n / 4 + 64;
    HIP_TRY(hipMalloc((void **)&im->d_recs, sizeof(SrcRec) * cap));
    HIP_TRY(hipMalloc((void **)&im->d_boxes, sizeof(int4) * cap));
    HIP_TRY(hipMalloc((void **)&im->d_kind, sizeof(int) * cap));
    HIP_TRY(hipMalloc((void **)&im->d_status, sizeof(int) * cap));
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

static int ensure_clist(cel_images *im, int64_t n) {
    if (n <= im->clist_cap) return CEL_OK;
    HIP_TRY(hipStreamSynchronize(im->ctx->stream));
    if (im->d_clist) (void)hipFree(im->d_clist);
    im->d_clist = nullptr; im->clist_cap = 0;
    HIP_TRY(hipMalloc((void **)&im->d_clist, sizeof(int) * n));
    im->clist_cap = n;
    return CEL_OK;
}

static double rsq_galaxy() {
    double q = 1.0 - 1e-5;           // celeste_galaxy_conditionals.py:207 error=1e-5
    return -2.0 * log1p(-q);         // scipy.stats.chi2.ppf(q, 2)
}

static int run_prep(cel_images *im, cel_sources *src, const int *d_live = nullptr, int nobox = 0) {
    cel_ctx *c = im->ctx;
    int64_t n = src->S * im->B;
    int rc = ensure_recs(im, n > 0 ? n : 1);
    if (rc) return rc;
    if (n == 0) { im->recs_gen = src->gen; im->last_S = 0; return CEL_OK; }
    int pi = prof_slot(c, CEL_K_PREP);
    LAUNCH_EV(k_prep, dim3((unsigned)((n + 255) / 256)), dim3(256), c->stream, EV0(c, pi), EV1(c, pi), im->d_bands, im->B,
              im->full_H, im->W, im->win_y0, im->H, src->S, src->d_type, src->d_radec, src->d_counts, src->d_shape,
              rsq_galaxy(), im->d_recs, im->d_boxes, im->d_kind, im->d_status, im->d_cursor, d_live, nobox);
    HIP_TRY(hipGetLastError());
    im->recs_gen = (d_live || nobox) ? 0 : src->gen;      // a partial table is nobody else's
    im->last_S = src->S;
    return CEL_OK;
}

// host copies of the boxes and status of every (band, source) of `src` (20 B each, not the 128-B
// records); reused until the sources or the window change
static int host_boxes(cel_images *im, cel_sources *src) {
    if (im->hbox_gen == src->gen && src->gen != 0 && im->recs_gen == src->gen) return CEL_OK;
    int rc = CEL_OK;
    if (im->recs_gen != src->gen || src->gen == 0) rc = run_prep(im, src);
    if (rc) return rc;
    const int64_t n = src->S * im->B;
    im->h_boxes.resize((size_t)n);
    im->h_status.resize((size_t)n);
    if (n) {
        hipStream_t st = im->ctx->stream;
        HIP_TRY(hipMemcpyAsync(im->h_boxes.data(), im->d_boxes, sizeof(int4) * n, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(im->h_status.data(), im->d_status, sizeof(int) * n, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    im->hbox_gen = src->gen;
    return CEL_OK;
}

// internal render flag: sources take part only strictly inside their boxes on the low side
// (x > x0, y > y0), the photon split's membership rule (celeste_sample_sources.pyx:50-51)
#define CEL_RENDER_STRICT 4
// internal render flag: the caller reads the tile lists afterwards (the E-step's tile walk): the general path
#define CEL_RENDER_KEEP_LISTS 16
static int render_impl(cel_images *im, cel_sources *src, int flags, double *ll_band, double *ll_total, double *lambda_out);

int cel_render_field(cel_images *im, cel_sources *src, int flags, double *ll_band, double *ll_total) {
    if (flags & ~(CEL_RENDER_LOGLIK | CEL_RENDER_NO_STORE)) return fail(CEL_ERR_INVALID, "cel_render_field: unknown flag bits");
    return render_impl(im, src, flags, ll_band, ll_total, nullptr);
}

// A small star field in one launch (k_small_stars.h): prep, binning, render and the per-band reduction.  *done = false when a
// part of a tile held more stars than the kernel stages (the caller then takes the general path, from now on).
static int render_small_stars(cel_images *im, cel_sources *src, int flags, bool *done) {
    cel_ctx *c = im->ctx;
    hipStream_t st = c->stream;
    const int64_t S = src->S;
    const int B = im->B;
    *done = false;
    int rc = ensure_recs(im, S * B);
    if (rc) return rc;
    const int nblk = B * im->ntx * im->nty * SMALL_NB, nblk_band = im->ntx * im->nty * SMALL_NB;
    if (!im->h_small) {
        // the blocks' partials and the overflow word behind them live in pinned, device-mapped, coherent HOST memory: the
        // kernel stores them over PCIe itself (5 KB at configs[1]) and the step needs no copy command behind the kernel
        // -- a D2H copy of this size cost the step ~8 us of queue latency
        HIP_TRY(hipHostMalloc((void **)&im->h_small, sizeof(double) * (nblk + 1), hipHostMallocMapped | hipHostMallocCoherent));
        memset(im->h_small, 0, sizeof(double) * (nblk + 1));
        HIP_TRY(hipHostGetDevicePointer((void **)&im->d_small, im->h_small, 0));
        // the bands' star-pass constants, once (the PSF and the WCS of an image set do not change)
        HIP_TRY(hipMalloc((void **)&im->d_small_consts, sizeof(double) * SMALL_CONSTS * B));
        hipLaunchKernelGGL(k_small_consts, dim3(B), dim3(64), 0, st, (const BandDev *)im->d_bands, im->d_small_consts);
    }
    RenderArgs a;
    memset(&a, 0, sizeof(a));
    a.bands = im->d_bands; a.recs = im->d_recs; a.nelec = im->d_nelec; a.lambda = im->d_lambda; a.partials = im->d_small;
    a.S = S; a.B = B; a.H = im->H; a.W = im->W; a.ntx = im->ntx; a.nty = im->nty;
    a.flags = flags; a.variant = c->variant; a.tail_T = c->render_T;
    SmallArgs x;
    x.radec = src->d_radec; x.counts = src->d_counts;
    x.recs = im->d_recs; x.boxes = im->d_boxes; x.kind = im->d_kind; x.status = im->d_status;
    x.partials = im->d_small;
    x.flag = reinterpret_cast<unsigned long long *>(im->d_small + nblk);
    x.stamp = 0x8000000000000000ull | ++im->small_seq;
    x.full_H = im->full_H; x.win_y0 = im->win_y0;
    x.consts = im->d_small_consts;
    {
        int mul = (int)(nblk_band * 0.381966) | 1;
        auto gcd = [](int a, int b) { while (b) { int t = a % b; a = b; b = t; } return a; };
        while (mul > 1 && gcd(mul, nblk_band) != 1) mul += 2;
        if (nblk_band < 4) mul = 1;
        x.perm_mul = mul % nblk_band ? mul % nblk_band : 1;
        x.perm_add = nblk_band / 3 + 1;
    }
    // (Dealing the tiles to the block slots by the star counts of the call before -- heaviest first, each to the free slot of
    // its band whose CU holds the least -- was built on top of the shuffle and measured 23.1-23.3 us against 23.4-23.7: a
    // block's duration correlates with its star count at 0.3 only.  Removed.)
    x.stamps = nullptr;
    static const char *stamp_path = getenv("CEL_SMALL_STAMPS");     // diagnostic: dump every block's phase stamps of each call
    unsigned long long *d_stamps = nullptr;
    if (stamp_path) { HIP_TRY(hipMalloc((void **)&d_stamps, sizeof(unsigned long long) * 8 * nblk)); x.stamps = d_stamps; }
    int pi = prof_slot(c, CEL_K_SMALL_STARS);
    LAUNCH_EV(k_small_stars, dim3((unsigned)nblk), dim3(64 * SMALL_NWV), st, EV0(c, pi), EV1(c, pi), a, x);
    const bool ll = (flags & CEL_RENDER_LOGLIK) != 0;
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));
    if (d_stamps) {
        std::vector<unsigned long long> hs((size_t)8 * nblk);
        (void)hipMemcpy(hs.data(), d_stamps, sizeof(unsigned long long) * 8 * nblk, hipMemcpyDeviceToHost);
        (void)hipFree(d_stamps);
        if (FILE *fp = fopen(stamp_path, "wb")) { fwrite(hs.data(), sizeof(unsigned long long), hs.size(), fp); fclose(fp); }
    }
    unsigned long long cur0;
    memcpy(&cur0, im->h_small + nblk, sizeof(cur0));
    if (cur0 == x.stamp) { im->small_off = true; return CEL_OK; }
    if (ll) {
        // a band's partials in a fixed order -- eight interleaved Kahan sums (index mod 8: independent chains for the host's
        // pipeline; one chain of 512 dependent adds per band was 13 us of the step), added in order: the same bits
        // whatever order the blocks ran in
        for (int b = 0; b < B; b++) {
            const double *pb = im->h_small + (size_t)b * nblk_band;
            double sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, comp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            int i = 0;
            for (; i + 8 <= nblk_band; i += 8)
                for (int k = 0; k < 8; k++) {
                    const double y = pb[i + k] - comp[k];
                    const double t = sum[k] + y;
                    comp[k] = (t - sum[k]) - y;
                    sum[k] = t;
                }
            for (int k = 0; i < nblk_band; i++, k++) {
                const double y = pb[i] - comp[k];
                const double t = sum[k] + y;
                comp[k] = (t - sum[k]) - y;
                sum[k] = t;
            }
            c->pinned[b] = ((sum[0] + sum[1]) + (sum[2] + sum[3])) + ((sum[4] + sum[5]) + (sum[6] + sum[7]));
            im->h_llband[b] = c->pinned[b];
        }
        im->llband_on_host = true;              // d_llband does not hold this render's sums (cel_images_loglik_device uploads them)
    }
    im->recs_gen = src->gen;                    // the kernel wrote k_prep's records, boxes and status words
    im->lists_gen = 0;                          // ... and no tile lists
    if (!(flags & CEL_RENDER_NO_STORE)) im->lambda_gen = 0;
    im->last_S = S;
    im->cost_S = -1; im->order_S = -1;
    im->last_entries = 0;
    *done = true;
    return CEL_OK;
}

static int render_impl(cel_images *im, cel_sources *src, int flags, double *ll_band, double *ll_total, double *lambda_out) {
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
    int rc;
    // a small star field (configs[1]): one launch instead of four (k_small_stars.h) -- unless a caller needs the tile lists
    // (the photon split, the E-step), an image of its own or the diagnostics
    const bool keep_lists = (flags & CEL_RENDER_KEEP_LISTS) != 0;
    flags &= ~CEL_RENDER_KEEP_LISTS;
    // the rows this image set OWNS (cel_images_set_noise_rows; default: all): the log-likelihood adds their tiles only
    const bool own_rows = im->noise_y0 > 0 || im->noise_y1 < im->H;
    int own_ty0 = 0, own_ty1 = im->nty;
    if (own_rows && (flags & CEL_RENDER_LOGLIK)) {
        if (im->noise_y0 % im->TH != 0 || (im->noise_y1 % im->TH != 0 && im->noise_y1 < im->H))
            return fail(CEL_ERR_INVALID, "the owned rows [%d, %d) must begin and end on render-tile rows (%d) for a log-likelihood",
                        im->noise_y0, im->noise_y1, im->TH);
        own_ty0 = im->noise_y0 / im->TH;
        own_ty1 = std::min(im->nty, (im->noise_y1 + im->TH - 1) / im->TH);
    }
    if (S > 0 && S <= SMALL_MAX_S && T <= STAR_TILES_MIN && im->TW == HW_TW && c->variant != 0 && c->star_tiles == 1 &&
        !c->tile_timing && !(c->debug & ~64) && src->n_gal == 0 && im->star_one_segment && !im->small_off && !lambda_out &&
        !keep_lists && !(flags & CEL_RENDER_STRICT) && !own_rows) {
        bool done = false;
        if ((rc = render_small_stars(im, src, flags, &done))) return rc;
        if (done) {
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
    }
    // The INCREMENTAL render (round 5).  The reference's single-source moves (util/infer/mcmc_transitions.py:37-152) evaluate the
    // whole field's likelihood after changing ONE source.  When this image set still holds the model image, records, lists and
    // per-tile Poisson partials of an earlier generation of THIS catalogue and only a few rows changed since
    // (cel_sources_set_rows stamps them), only the tiles that the changed sources' boxes touch -- the boxes they had and the
    // boxes they have now -- are rendered again, each from its complete list: the same arithmetic in the same order as the full
    // render, so pixels, partials and log-likelihoods are the full render's bit for bit; every other tile's pixels and partial
    // are still valid.  Source preparation and binning run in full (66 us at configs[2]); a render with nothing changed, or
    // after a whole-catalogue upload, is a full render -- nothing is ever answered from a cache.
    const bool diag_r = c->tile_timing || (c->debug & ~64);
    const bool stars_only_r = im->TW == HW_TW && !diag_r && c->star_tiles && src->n_gal == 0 && im->star_one_segment &&
                              c->variant != 0 && (c->star_tiles == 2 || T > STAR_TILES_MIN);
    DeltaRows delta;
    delta.n = 0;
    bool incr = c->incremental && !lambda_out && !(flags & (CEL_RENDER_NO_STORE | CEL_RENDER_STRICT)) && im->TW == HW_TW &&
                c->variant != 0 && !diag_r && !stars_only_r && tile_parts_of(c, im) == 1 && S > 0 && S == im->last_S &&
                im->lambda_gen != 0 && im->lambda_uid == src->uid && im->lambda_T == c->render_T && im->lambda_parts == 1 && im->lambda_gen != src->gen &&
                im->lambda_gen >= src->full_gen &&
                im->lists_gen == im->lambda_gen && im->recs_gen == im->lambda_gen && (int64_t)src->row_gen.size() == S &&
                (!(flags & CEL_RENDER_LOGLIK) || (im->partials_gen == im->lambda_gen && !im->nelec_shared));
    if (incr) {
        for (int64_t s = 0; s < S && incr; s++)
            if (src->row_gen[(size_t)s] > im->lambda_gen) {
                if (delta.n == DELTA_MAX) incr = false;
                else delta.idx[delta.n++] = (int)s;
            }
        if (delta.n == 0) incr = false;
    }
    if (incr) {
        if (!im->d_dirty) HIP_TRY(hipMalloc((void **)&im->d_dirty, sizeof(int) * (size_t)T));
        HIP_TRY(hipMemsetAsync(im->d_dirty, 0, sizeof(int) * (size_t)T, st));
        // the tiles the changed sources' OLD boxes touch (before k_prep rewrites the boxes) ...
        hipLaunchKernelGGL(k_mark_dirty, dim3((unsigned)((delta.n * im->B + 255) / 256)), dim3(256), 0, st, delta, (const int4 *)im->d_boxes, S, im->B,
                           im->ntx, im->nty, im->TW, im->TH, im->d_dirty);
    }
    if (flags & CEL_RENDER_LOGLIK) im->partials_gen = 0;
    im->lists_gen = 0;
    if (!lambda_out && !(flags & CEL_RENDER_NO_STORE)) im->lambda_gen = 0;
    rc = run_prep(im, src);
    if (rc) return rc;
    if (incr)       // ... and the tiles their new boxes touch
        hipLaunchKernelGGL(k_mark_dirty, dim3((unsigned)((delta.n * im->B + 255) / 256)), dim3(256), 0, st, delta, (const int4 *)im->d_boxes, S, im->B,
                           im->ntx, im->nty, im->TW, im->TH, im->d_dirty);
    im->last_dirty = -1;
    int parts_used = 1;
    if (im->lists_cap == 0) {
        // first guess: every (band, source) touches ~6 tiles; grown on overflow below
        rc = ensure_lists(im, (S * im->B) * 6 + 1024);
        if (rc) return rc;
    }
    if (im->clist_cap == 0) {
        // first guess: every (band, source) touches ~3 super-tiles; grown on overflow below
        rc = ensure_clist(im, (S * im->B) * 3 + 1024);
        if (rc) return rc;
    }
    const int NS = im->B * im->nsx * im->nsy;
    const int tile_order = tile_order_of(c, im);
    // a small catalogue is binned by one wave per tile into per-tile segments of S entries (k_bin_direct)
    const bool bin_direct = S > 0 && S <= BIN_DIRECT_MAX_S && (int64_t)T * S <= ((int64_t)1 << 25);
    if (bin_direct && (rc = ensure_lists(im, (int64_t)T * S))) return rc;
    for (int attempt = 0; attempt < 8; attempt++) {
        // d_cursor: [0] fine cursor, [1] fine overflow, [2] coarse cursor, [3] coarse overflow;
        // zeroed by k_prep (no memset in the queue), by hand only when that did not run or on a retry
        if (attempt > 0 || S * im->B == 0) HIP_TRY(hipMemsetAsync(im->d_cursor, 0, sizeof(unsigned long long) * 4, st));
        // one event pair over the binning kernels: start on the first, stop on the last
        int pi = prof_slot(c, CEL_K_BIN);
        // the order by the previous render's measured durations was sorted behind that render's readback
        // (below): nothing to do here then
        const bool order_ready = (tile_order == 1 && im->order_S == S && im->cost_S == S);
        const hipEvent_t bin_ev1 = (tile_order && !order_ready) ? (hipEvent_t) nullptr : EV1(c, pi);
        // a small catalogue: one wave per tile (k_bin_direct).  Otherwise one binning kernel while no super-tile holds more
        // than BIN_CH candidates, the two-level form after that
        if (bin_direct) {
            LAUNCH_EV(k_bin_direct, dim3(T), dim3(64), st, EV0(c, pi), bin_ev1, im->d_boxes, im->d_kind, S, im->ntx, im->nty, im->TH, im->TW,
                      im->d_tile_cnt, im->d_tile_nstar, im->d_tile_work, im->d_tile_off, im->d_cursor, im->d_lists);
        } else if (im->bin_two_level) {
            LAUNCH_EV(k_bin_coarse, dim3(NS), dim3(64 * COARSE_WAVES), st, EV0(c, pi), (hipEvent_t) nullptr, im->d_boxes, S, im->nsx, im->nsy,
                      im->d_sup_cnt, im->d_sup_off, im->d_cursor + 2, im->d_clist, im->clist_cap, (int *)(im->d_cursor + 3));
            // (16-wave blocks while every super-tile gets a CU of its own, 8-wave blocks -- two to a CU -- beyond that: k_bin2.h)
            if (NS > c->n_cu)
                LAUNCH_EV((k_bin_fine_blk<false, 8>), dim3(NS), dim3(64 * 8), st, (hipEvent_t) nullptr, bin_ev1,
                          im->d_boxes, im->d_kind, S, im->ntx, im->nty, im->TH, im->TW,
                          im->nsx, im->nsy, im->d_sup_cnt, im->d_sup_off, im->d_clist, im->clist_cap, im->d_tile_cnt,
                          im->d_tile_nstar, im->d_tile_work, im->d_tile_off, im->d_cursor, im->d_lists, im->lists_cap,
                          (int *)(im->d_cursor + 1), (int *)(im->d_cursor + 3));
            else
                LAUNCH_EV((k_bin_fine_blk<false, 16>), dim3(NS), dim3(64 * 16), st, (hipEvent_t) nullptr, bin_ev1,
                          im->d_boxes, im->d_kind, S, im->ntx, im->nty, im->TH, im->TW,
                          im->nsx, im->nsy, im->d_sup_cnt, im->d_sup_off, im->d_clist, im->clist_cap, im->d_tile_cnt,
                          im->d_tile_nstar, im->d_tile_work, im->d_tile_off, im->d_cursor, im->d_lists, im->lists_cap,
                          (int *)(im->d_cursor + 1), (int *)(im->d_cursor + 3));
        } else if (NS > c->n_cu) {
            LAUNCH_EV((k_bin_fine_blk<true, 8>), dim3(NS), dim3(64 * 8), st, EV0(c, pi), bin_ev1,
                      im->d_boxes, im->d_kind, S, im->ntx, im->nty, im->TH, im->TW,
                      im->nsx, im->nsy, im->d_sup_cnt, im->d_sup_off, im->d_clist, im->clist_cap, im->d_tile_cnt,
                      im->d_tile_nstar, im->d_tile_work, im->d_tile_off, im->d_cursor, im->d_lists, im->lists_cap,
                      (int *)(im->d_cursor + 1), (int *)(im->d_cursor + 3));
        } else {
            LAUNCH_EV((k_bin_fine_blk<true, 16>), dim3(NS), dim3(64 * 16), st, EV0(c, pi), bin_ev1,
                      im->d_boxes, im->d_kind, S, im->ntx, im->nty, im->TH, im->TW,
                      im->nsx, im->nsy, im->d_sup_cnt, im->d_sup_off, im->d_clist, im->clist_cap, im->d_tile_cnt,
                      im->d_tile_nstar, im->d_tile_work, im->d_tile_off, im->d_cursor, im->d_lists, im->lists_cap,
                      (int *)(im->d_cursor + 1), (int *)(im->d_cursor + 3));
        }
        if (tile_order && !order_ready)
            // heaviest first: by the durations the tiles had in the previous render when that was
            // of the same source count (an MCMC chain changes little from one evaluation to the
            // next), by the binning pass's estimate otherwise.  The order never changes results.
            LAUNCH_EV(k_order, dim3(1), dim3(1024), st, (hipEvent_t) nullptr, EV1(c, pi),
                      (const int *)((im->cost_S == S && tile_order == 1) ? im->d_tile_cost : im->d_tile_work), T, im->d_order);
        RenderArgs a;
        a.bands = im->d_bands; a.recs = im->d_recs; a.lists = im->d_lists; a.tile_cnt = im->d_tile_cnt;
        a.tile_nstar = im->d_tile_nstar;
        a.tile_off = im->d_tile_off; a.nelec = im->d_nelec; a.lambda = lambda_out ? lambda_out : im->d_lambda; a.partials = im->d_partials;
        a.S = S; a.capacity = im->lists_cap; a.B = im->B; a.H = im->H; a.W = im->W; a.ntx = im->ntx; a.nty = im->nty;
        a.flags = flags | (c->debug << 8); a.variant = c->variant; a.tail_T = c->render_T; a.order = tile_order ? im->d_order : nullptr;
        a.timing = nullptr;
        a.dirty = incr ? im->d_dirty : nullptr;
        a.cost = (im->TW == HW_TW || im->TW == QW_TW) ? im->d_tile_cost : nullptr;
        if (c->tile_timing) {
            if (!im->d_timing) HIP_TRY(hipMalloc((void **)&im->d_timing, sizeof(unsigned long long) * 3 * T));
            a.timing = im->d_timing;
        }
        // a catalogue without galaxies: the star-tile kernel (k_render_stars.h), when the frame has more tiles than the
        // general kernel has wave slots (STAR_TILES_MIN; measured break-even, tools/star_tiles_threshold.py); the
        // instantiation with counters / time stamps / ablations exists for the general kernel only
        const bool diag = a.timing || (c->debug & ~64);
        const int parts = diag ? 1 : tile_parts_of(c, im);
        parts_used = (im->TW == HW_TW && !diag) ? parts : 1;
        a.slabs = nullptr; a.part_cnt = nullptr;
        if (parts > 1) {
            if (im->slabs_parts < parts) {
                HIP_TRY(hipStreamSynchronize(st));
                if (im->d_slabs) (void)hipFree(im->d_slabs);
                im->d_slabs = nullptr; im->slabs_parts = 0;
                HIP_TRY(hipMalloc((void **)&im->d_slabs, sizeof(double) * HW_TH * HW_TW * (size_t)T * parts));
                if (!im->d_part_cnt) {
                    HIP_TRY(hipMalloc((void **)&im->d_part_cnt, sizeof(int) * (size_t)T));
                    HIP_TRY(hipMemsetAsync(im->d_part_cnt, 0, sizeof(int) * (size_t)T, st));
                }
                im->slabs_parts = parts;
            }
            a.slabs = im->d_slabs; a.part_cnt = im->d_part_cnt;
        }
        const bool stars_only = im->TW == HW_TW && !diag && c->star_tiles && src->n_gal == 0 && im->star_one_segment &&
                                c->variant != 0 && (c->star_tiles == 2 || T > STAR_TILES_MIN);    // (1 and 3: the rule; 3 = without the one-launch small path)
        if (stars_only) parts_used = 0;         // (the star-tile kernel adds a pixel's stars in another order: not the general kernel's bits)
        pi = prof_slot(c, stars_only ? CEL_K_RENDER_STARS : CEL_K_RENDER);
        if (im->TW == QW_TW)
            LAUNCH_EV(k_render_qw, dim3(T), dim3(64), st, EV0(c, pi), EV1(c, pi), a);
        else if (im->TW == HW_TW) {
            // the production instantiation has no diagnostic code in it; counters, time stamps and (CEL_ABLATE
            // builds) ablations live in the second one
            if (diag) LAUNCH_EV(k_render_hw<true>, dim3(T), dim3(64), st, EV0(c, pi), EV1(c, pi), a);
            else if (stars_only) LAUNCH_EV((k_render_stars<2, false>), dim3(T), dim3(64), st, EV0(c, pi), EV1(c, pi), a);
            else if (parts > 1) {
                const unsigned grid = (unsigned)(((T + 7) / 8) * 8 * parts);
                if (parts == 4) LAUNCH_EV((k_render_hw<false, 4>), dim3(grid), dim3(64), st, EV0(c, pi), EV1(c, pi), a);
                else LAUNCH_EV((k_render_hw<false, 2>), dim3(grid), dim3(64), st, EV0(c, pi), EV1(c, pi), a);
            }
            else LAUNCH_EV(k_render_hw<false>, dim3(T), dim3(64), st, EV0(c, pi), EV1(c, pi), a);
        }
        else if (im->TH == 64)
            LAUNCH_EV((k_render<64>), dim3(T), dim3(64), st, EV0(c, pi), EV1(c, pi), a);
        else
            LAUNCH_EV((k_render<32>), dim3(T), dim3(64), st, EV0(c, pi), EV1(c, pi), a);
        if (flags & CEL_RENDER_LOGLIK) {
            pi = prof_slot(c, CEL_K_REDUCE);
            LAUNCH_EV(k_reduce, dim3(im->B), dim3(256), st, EV0(c, pi), EV1(c, pi), (const double *)im->d_partials, im->ntx * im->nty, im->d_llband,
                      im->ntx, own_ty0, own_ty1);
            im->llband_on_host = false;
        }
        // the per-band sums, the total list length and the overflow flags ride back in ONE copy
        HIP_TRY(hipMemcpyAsync(c->pinned, im->d_llband, sizeof(double) * MAX_BANDS + sizeof(unsigned long long) * 4,
                               hipMemcpyDeviceToHost, st));
        HIP_TRY(hipGetLastError());
        im->last_S = S;
        // An MCMC chain renders the same number of sources again: sort this render's tile durations into
        // the next render's launch order NOW, behind the readback the host is waiting for, instead of
        // in front of the next render (13 us + a launch gap per step).  The host waits for the copy only.
        const bool post_order = (tile_order == 1 && a.cost != nullptr);
        if (post_order) {
            if (!im->ev_step) HIP_TRY(hipEventCreateWithFlags(&im->ev_step, hipEventDisableTiming));
            HIP_TRY(hipEventRecord(im->ev_step, st));
            hipLaunchKernelGGL(k_order, dim3(1), dim3(1024), 0, st, (const int *)im->d_tile_cost, T, im->d_order);
            HIP_TRY(hipEventSynchronize(im->ev_step));
        } else {
            HIP_TRY(hipStreamSynchronize(st));
        }
        unsigned long long cur[4];
        memcpy(cur, c->pinned + MAX_BANDS, sizeof(cur));
        const bool fine_ok = (cur[1] & 0xffffffffull) == 0 && (int64_t)cur[0] <= im->lists_cap;
        const bool too_dense = (cur[3] & 2ull) != 0;        // a super-tile with more candidates than the one-kernel form stages
        const bool coarse_ok = (cur[3] & 0xffffffffull) == 0 && (int64_t)cur[2] <= im->clist_cap;
        if (coarse_ok) im->last_entries = (double)cur[0];
        if (fine_ok && coarse_ok) {
            im->cost_S = a.cost ? S : -1; im->order_S = post_order ? S : -1;
            im->lists_gen = src->gen;
            if (!lambda_out && !(flags & (CEL_RENDER_NO_STORE | CEL_RENDER_STRICT)) && im->TW == HW_TW && c->variant != 0) {
                im->lambda_gen = src->gen;
                im->lambda_uid = src->uid;
                im->lambda_T = c->render_T;
                // a render by the DIAG instantiation (tile timing, ablation bits) vouches for nothing: with an ablation bit set its
                // pixels and partials are documented as wrong, and the incremental path would take them as a base
                im->lambda_parts = diag ? 0 : parts_used;
                // (a render WITHOUT the log-likelihood vouches for no partials: those in the buffer may be of another sky
                // level or drop threshold although the catalogue's generation is the same -- found by tools/dbg/incremental_stress.py)
                im->partials_gen = (flags & CEL_RENDER_LOGLIK) ? src->gen : 0;
            }
            if (incr) im->last_dirty = -2;          // (counted on request: cel_debug_last_render)
            break;
        }
        im->cost_S = -1;
        im->order_S = -1;
        // rerun with room (a truncated coarse list also truncates the fine counts)
        if (too_dense) { im->bin_two_level = true; continue; }
        if (!coarse_ok) rc = ensure_clist(im, (int64_t)cur[2] + (int64_t)cur[2] / 4 + 1024);
        if (!rc && !fine_ok) rc = ensure_lists(im, (int64_t)cur[0] + (int64_t)cur[0] / 4 + 1024);
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

int cel_debug_split_rates(cel_images *im, double *out) {
    if (!im || !out) return fail(CEL_ERR_INVALID, "cel_debug_split_rates: null argument");
    if (!im->d_rate) return fail(CEL_ERR_INVALID, "no totals image: cel_photon_split on the recurrence kernels has not run");
    HIP_TRY(hipSetDevice(im->ctx->device));
    return copy_out(out, im->rate_in_lambda ? im->d_lambda : im->d_rate, sizeof(double) * (size_t)im->B * im->H * im->W, CEL_HOST, im->ctx->stream);
}

int cel_debug_last_render(cel_images *im, int64_t *dirty_tiles) {
    if (!im || !dirty_tiles) return fail(CEL_ERR_INVALID, "cel_debug_last_render: null argument");
    *dirty_tiles = -1;
    if (im->last_dirty == -1 || !im->d_dirty) return CEL_OK;
    HIP_TRY(hipSetDevice(im->ctx->device));
    const int T = im->B * im->ntx * im->nty;
    std::vector<int> h((size_t)T);
    int rc = copy_out(h.data(), im->d_dirty, sizeof(int) * (size_t)T, CEL_HOST, im->ctx->stream);
    if (rc) return rc;
    int64_t n = 0;
    for (int v : h) n += (v != 0);
    *dirty_tiles = n;
    return CEL_OK;
}

int cel_debug_tile_timing(cel_images *im, uint64_t *out, int64_t *n_tiles) {
    if (!im || !n_tiles) return fail(CEL_ERR_INVALID, "cel_debug_tile_timing: null argument");
    const int T = im->B * im->ntx * im->nty;
    *n_tiles = T;
    if (!out) return CEL_OK;
    if (!im->d_timing) return fail(CEL_ERR_INVALID, "no timing recorded: set CEL_OPT_TILE_TIMING before rendering");
    HIP_TRY(hipSetDevice(im->ctx->device));
    return copy_out(out, im->d_timing, sizeof(unsigned long long) * 3 * T, CEL_HOST, im->ctx->stream);
}

int cel_field_stats(cel_images *im, double *n_srcpix, double *n_gauss, double *n_tile_entries) {
    if (!im) return fail(CEL_ERR_INVALID, "null images");
    cel_ctx *c = im->ctx;
    HIP_TRY(hipSetDevice(c->device));
    int64_t n = im->last_S * im->B;
    HIP_TRY(hipMemsetAsync(im->d_stats, 0, sizeof(double) * 2, c->stream));
    if (n > 0)
        hipLaunchKernelGGL(k_stats, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, im->d_recs, n, im->d_stats);
    HIP_TRY(hipMemcpyAsync(c->pinned + MAX_BANDS + 6, im->d_stats, sizeof(double) * 2, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (n_srcpix) *n_srcpix = c->pinned[MAX_BANDS + 6];
    if (n_gauss) *n_gauss = c->pinned[MAX_BANDS + 7];
    if (n_tile_entries) *n_tile_entries = im->last_entries;
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
    int rc = host_boxes(im, src);
    if (rc) return rc;
    const int64_t S = src->S;
    const int4 *hb = im->h_boxes.data() + (int64_t)band * S;     // x0, x1, y0, y1
    const int *hs = im->h_status.data() + (int64_t)band * S;
    for (int64_t s = 0; s < S; s++) {
        boxes[4 * s + 0] = hb[s].z; boxes[4 * s + 1] = hb[s].w;
        boxes[4 * s + 2] = hb[s].x; boxes[4 * s + 3] = hb[s].y;
        status[s] = hs[s];
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
    if (offsets[0] != 0) return fail(CEL_ERR_INVALID, "cel_render_stamps: offsets[0] must be 0");
    std::vector<int32_t> hb((size_t)S * 4), hs((size_t)S);
    int rc = cel_stamp_boxes(im, src, band, hb.data(), hs.data());
    if (rc) return rc;
    std::vector<int4> obox((size_t)S);
    std::vector<StampJob> jobs;
    std::vector<int64_t> skipped;   // sources without a stamp whose slot in `out` is not empty
    const int ROWS = 64;
    const int strip_w = (c->variant == 0) ? TILE_W : HW_TW;     // the recurrence kernel works on 32 x 64 chunks
    for (int64_t s = 0; s < S; s++) {
        int y0, y1, x0, x1;
        bool ok;
        if (boxes_in) {
            y0 = boxes_in[4 * s]; y1 = boxes_in[4 * s + 1]; x0 = boxes_in[4 * s + 2]; x1 = boxes_in[4 * s + 3];
            ok = (y1 > y0 && x1 > x0) && hs[s] != -1;   // an overlap-test miss is None whatever the limits
        } else {
            y0 = hb[4 * s]; y1 = hb[4 * s + 1]; x0 = hb[4 * s + 2]; x1 = hb[4 * s + 3];
            ok = hs[s] > 0;
        }
        obox[s] = make_int4(x0, x1, y0, y1);
        if (offsets[s + 1] < offsets[s]) return fail(CEL_ERR_INVALID, "offsets must not decrease (source %lld)", (long long)s);
        if (!ok) {
            // the reference returns (None, None, None) here; the slot the caller sized for it is
            // zero-filled so that nothing stale can be read from it
            if (offsets[s + 1] > offsets[s]) skipped.push_back(s);
            continue;
        }
        int64_t area = (int64_t)(y1 - y0) * (x1 - x0);
        if (offsets[s + 1] - offsets[s] != area)
            return fail(CEL_ERR_INVALID, "offsets[%lld+1]-offsets[%lld] = %lld but the stamp has %lld pixels",
                        (long long)s, (long long)s, (long long)(offsets[s + 1] - offsets[s]), (long long)area);
        for (int xs = x0; xs < x1; xs += strip_w)
            for (int ys = y0; ys < y1; ys += ROWS)
                jobs.push_back(StampJob{(int)s, xs, ys, ys + ROWS < y1 ? ys + ROWS : y1});
    }
    int64_t total = offsets[S];
    if (jobs.empty() && (skipped.empty() || mem != CEL_DEVICE)) {
        if (mem != CEL_DEVICE)
            for (int64_t s : skipped) memset(out + offsets[s], 0, sizeof(double) * (size_t)(offsets[s + 1] - offsets[s]));
        return CEL_OK;
    }
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
    if ((rc = scratch_get(c, 4, sizeof(StampJob) * jobs.size(), (void **)&d_jobs)) ||
        (rc = scratch_get(c, 5, sizeof(int4) * S, (void **)&d_obox)) ||
        (rc = scratch_get(c, 6, sizeof(int64_t) * (S + 1), (void **)&d_off)))
        return rc;
    if (mem == CEL_DEVICE) d_out = out;
    else if ((rc = scratch_get(c, 7, sizeof(double) * (total > 0 ? total : 1), (void **)&d_out))) return rc;
    for (int64_t s : skipped)
        ST_TRY(hipMemsetAsync(d_out + offsets[s], 0, sizeof(double) * (size_t)(offsets[s + 1] - offsets[s]), c->stream));
    if (jobs.empty()) goto sync;
    ST_TRY(hipMemcpyAsync(d_jobs, jobs.data(), sizeof(StampJob) * jobs.size(), hipMemcpyHostToDevice, c->stream));
    ST_TRY(hipMemcpyAsync(d_obox, obox.data(), sizeof(int4) * S, hipMemcpyHostToDevice, c->stream));
    ST_TRY(hipMemcpyAsync(d_off, offsets, sizeof(int64_t) * (S + 1), hipMemcpyHostToDevice, c->stream));
    {
        int pi = prof_begin(c, CEL_K_STAMPS);
        if (c->variant == 0)
            hipLaunchKernelGGL(k_stamps, dim3((unsigned)jobs.size()), dim3(64), 0, c->stream, im->d_bands, band,
                               im->d_recs + (int64_t)band * S, d_jobs, d_obox, d_off, scaled, d_out);
        else
            hipLaunchKernelGGL(k_stamps_hw, dim3((unsigned)jobs.size()), dim3(64), 0, c->stream, im->d_bands, band,
                               im->d_recs + (int64_t)band * S, d_jobs, d_obox, d_off, scaled, c->tail_T, d_out);
        prof_end(c, pi);
    }
    ST_TRY(hipGetLastError());
    if (mem != CEL_DEVICE) ST_TRY(hipMemcpyAsync(out, d_out, sizeof(double) * total, hipMemcpyDeviceToHost, c->stream));
sync:
    ST_TRY(hipStreamSynchronize(c->stream));
#undef ST_TRY
done:
    (void)hipStreamSynchronize(c->stream);
    return rc;
}

// ---- per-source conditional log-likelihoods ---------------------------------------------------
int cel_patch_loglik_multi(cel_images *im, cel_sources *src, const int32_t *owner, int64_t NB,
                           const int32_t *boxes, const int64_t *offsets, const double *data, int mem, int mode,
                           double *ll_out) {
    if (!im || !src || !ll_out) return fail(CEL_ERR_INVALID, "cel_patch_loglik: null argument");
    if (src->B != im->B || src->ctx != im->ctx) return fail(CEL_ERR_INVALID, "sources do not match images");
    if (mode != 0 && mode != 1 && mode != 2 && mode != 4)
        return fail(CEL_ERR_INVALID, "mode must be 0 (conditional), 1 (isolated), 2 (patch Poisson) or 4 (Poisson on a background plane)");
    // resident form: boxes == offsets == data == NULL -> the patches of the last resident photon
    // split (mode 0) or the observed image on those boxes (mode 1); NB must be that split's S
    const bool resident = (!boxes && !offsets && !data);
    if (!resident && (!boxes || !offsets)) return fail(CEL_ERR_INVALID, "cel_patch_loglik: null boxes / offsets");
    if (resident && mode > 1) return fail(CEL_ERR_INVALID, "the resident form scores mode 0 or 1");
    const int nplanes = (mode == 4) ? 2 : 1;               // mode 4: data plane, then background plane
    if (resident && (im->samp_S <= 0 || NB != im->samp_S))
        return fail(CEL_ERR_INVALID, "resident form needs a resident photon split of NB = %lld sources (have %lld)",
                    (long long)NB, (long long)im->samp_S);
    if (NB < 1 || (NB > 1 && !owner)) return fail(CEL_ERR_INVALID, "cel_patch_loglik: NB patch sets need an owner array");
    if (resident && mode == 1 && !im->have_nelec) return fail(CEL_ERR_INVALID, "the isolated form needs cel_images_set_nelec");
    cel_ctx *c = im->ctx;
    HIP_TRY(hipSetDevice(c->device));
    const int B = im->B;
    const int64_t P = src->S, nb = NB * B;
    if (P == 0) return CEL_OK;
    if (!resident && offsets[0] != 0) return fail(CEL_ERR_INVALID, "cel_patch_loglik: offsets[0] must be 0");
    std::vector<int4> hbox((size_t)(resident ? 0 : nb));
    for (int64_t i = 0; i < (resident ? 0 : nb); i++) {
        int y0 = boxes[4 * i], y1 = boxes[4 * i + 1], x0 = boxes[4 * i + 2], x1 = boxes[4 * i + 3];
        int64_t area = (y1 > y0 && x1 > x0) ? (int64_t)(y1 - y0) * (x1 - x0) : 0;
        if (offsets[i + 1] - offsets[i] != area * nplanes)
            return fail(CEL_ERR_INVALID, "patch set %lld band %d: offsets give %lld patch values, the box has %lld pixels x %d plane(s)",
                        (long long)(i / B), (int)(i % B), (long long)(offsets[i + 1] - offsets[i]), (long long)area, nplanes);
        if (area > 0 && (y0 < 0 || x0 < 0 || y1 > im->H || x1 > im->W))
            return fail(CEL_ERR_INVALID, "patch set %lld band %d: patch limits outside the image", (long long)(i / B), (int)(i % B));
        hbox[(size_t)i] = make_int4(x0, x1, y0, y1);
    }
    if (owner)
        for (int64_t p = 0; p < P; p++)
            if (owner[p] < 0 || owner[p] >= NB) return fail(CEL_ERR_INVALID, "owner[%lld] = %d out of range", (long long)p, owner[p]);
    if (!resident && offsets[nb] > 0 && !data) return fail(CEL_ERR_INVALID, "cel_patch_loglik: null data");
    int rc = run_prep(im, src);
    if (rc) return rc;
    im->last_S = P;
    int4 *d_box = nullptr;
    int64_t *d_off = nullptr;
    int *d_owner = nullptr;
    double *d_data = nullptr, *d_out = nullptr;
    // a call with few proposals is a handful of one-wave jobs as long as their longest: each is dealt to PLL_PARTS blocks
    // (the values do not depend on it: the kernel sums its chunks in PLL_PARTS classes either way)
    // (with photon lists a job always owns PLL_PARTS slots: the kernel that scores at the photons writes its four class sums)
    const bool nzl = resident && mode == 0 && c->variant != 0 && im->nz_valid;
    const int nsplit = (c->variant != 0 && mode == 0 && P * B <= 8192) ? PLL_PARTS : 1;
    const int nparts = nzl ? PLL_PARTS : nsplit;
    std::vector<double> hout((size_t)(P * B * nparts));
    hipError_t e;
#define PL_TRY(expr)                                                                     \
    do {                                                                                 \
        e = (expr);                                                                      \
        if (e != hipSuccess) { rc = fail(CEL_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e)); goto done; } \
    } while (0)
    int4 *d_nz = nullptr;
    if ((rc = scratch_get(c, 0, sizeof(int4) * (resident ? 0 : 2 * nb) + sizeof(int) * (owner ? P : 0), (void **)&d_box)) ||
        (rc = scratch_get(c, 1, sizeof(int64_t) * (nb + 1), (void **)&d_off)) ||
        (rc = scratch_get(c, 2, sizeof(double) * P * B * nparts, (void **)&d_out)))
        goto done;
    if (owner) {
        d_owner = reinterpret_cast<int *>(d_box + (resident ? 0 : 2 * nb));
        PL_TRY(hipMemcpyAsync(d_owner, owner, sizeof(int) * P, hipMemcpyHostToDevice, c->stream));
    }
    if (!resident) d_nz = d_box + nb;
    if (resident) {
        d_nz = im->d_snz;
        d_box = im->d_sbox;
        d_off = im->d_soff;
        d_data = nullptr;                                 // mode 0: the int32 patches (below); mode 1 reads nelec on the boxes
    } else {
        PL_TRY(hipMemcpyAsync(d_box, hbox.data(), sizeof(int4) * nb, hipMemcpyHostToDevice, c->stream));
        PL_TRY(hipMemcpyAsync(d_off, offsets, sizeof(int64_t) * (nb + 1), hipMemcpyHostToDevice, c->stream));
    }
    if (resident) {
    } else if (mem == CEL_DEVICE) {
        d_data = const_cast<double *>(data);
    } else {
        if ((rc = scratch_get(c, 3, sizeof(double) * (offsets[nb] > 0 ? offsets[nb] : 1), (void **)&d_data))) goto done;   // copies are queued: leave through the sync
        if (offsets[nb] > 0)
            PL_TRY(hipMemcpyAsync(d_data, data, sizeof(double) * offsets[nb], hipMemcpyHostToDevice, c->stream));
    }
    {
        int pi = prof_begin(c, CEL_K_PATCH_LL);
        // the resident split's patches are int32 photon counts (mode 0); everything a caller hands over, and the observed
        // image the resident mode 1 reads, is double
        const int *i_data = (resident && mode == 0) ? im->d_samp : nullptr;
        if (c->variant == 0) {
            if (i_data)
                hipLaunchKernelGGL(k_patch_ll<int>, dim3((unsigned)(P * B)), dim3(256), 0, c->stream, im->d_bands, B, P, im->d_recs,
                                   d_owner, d_box, d_off, i_data, im->d_nelec, im->H, im->W, mode, d_out);
            else
                hipLaunchKernelGGL(k_patch_ll<double>, dim3((unsigned)(P * B)), dim3(256), 0, c->stream, im->d_bands, B, P, im->d_recs,
                                   d_owner, d_box, d_off, (const double *)d_data, im->d_nelec, im->H, im->W, mode, d_out);
        } else if (mode == 0) {
            if (!resident)
                hipLaunchKernelGGL(k_patch_nzbox<double>, dim3((unsigned)nb), dim3(64), 0, c->stream, d_box, d_off, (const double *)d_data, d_nz);
            if (i_data) {
                hipLaunchKernelGGL((k_patch_ll_hw<0, int>), dim3((unsigned)(P * B * nsplit)), dim3(64), 0, c->stream, im->d_bands, B, P, im->d_recs,
                                   d_owner, d_box, d_off, i_data, im->d_nelec, im->H, im->W, d_nz, c->tail_T, d_out,
                                   (const int *)nullptr, nsplit, (const int *)nullptr, (const int *)(nzl ? im->d_nzmode : nullptr), 0, nparts);
                if (nzl)                    // ... and the proposals whose patch is scored at its photons (the others return at once)
                    hipLaunchKernelGGL(k_patch_ll_nz<false>, dim3((unsigned)(P * B)), dim3(64), 0, c->stream, im->d_bands, B, P, im->d_recs,
                                       (const int *)d_owner, (const int4 *)d_box, (const int4 *)d_nz, (const int *)im->d_nzmode,
                                       (const int64_t *)im->d_nzoff, (const NzEntry *)im->d_nzlist, d_out, (const int *)nullptr,
                                       (const int *)nullptr, (const SliceFuse *)nullptr);
            } else
                hipLaunchKernelGGL((k_patch_ll_hw<0, double>), dim3((unsigned)(P * B * nparts)), dim3(64), 0, c->stream, im->d_bands, B, P, im->d_recs,
                                   d_owner, d_box, d_off, (const double *)d_data, im->d_nelec, im->H, im->W, d_nz, c->tail_T, d_out,
                                   (const int *)nullptr, nparts, (const int *)nullptr);
        } else if (mode == 2)
            hipLaunchKernelGGL((k_patch_ll_hw<2, double>), dim3((unsigned)(P * B)), dim3(64), 0, c->stream, im->d_bands, B, P, im->d_recs,
                               d_owner, d_box, d_off, (const double *)d_data, im->d_nelec, im->H, im->W, (const int4 *)nullptr, c->tail_T, d_out,
                               (const int *)nullptr, 1, (const int *)nullptr);
        else if (mode == 4)
            hipLaunchKernelGGL((k_patch_ll_hw<4, double>), dim3((unsigned)(P * B)), dim3(64), 0, c->stream, im->d_bands, B, P, im->d_recs,
                               d_owner, d_box, d_off, (const double *)d_data, im->d_nelec, im->H, im->W, (const int4 *)nullptr, c->tail_T, d_out,
                               (const int *)nullptr, 1, (const int *)nullptr);
        else
            hipLaunchKernelGGL((k_patch_ll_hw<1, double>), dim3((unsigned)(P * B)), dim3(64), 0, c->stream, im->d_bands, B, P, im->d_recs,
                               d_owner, d_box, d_off, (const double *)d_data, im->d_nelec, im->H, im->W, (const int4 *)nullptr, c->tail_T, d_out,
                               (const int *)nullptr, 1, (const int *)nullptr);
        prof_end(c, pi);
    }
    PL_TRY(hipGetLastError());
    PL_TRY(hipMemcpyAsync(hout.data(), d_out, sizeof(double) * P * B * nparts, hipMemcpyDeviceToHost, c->stream));
    PL_TRY(hipStreamSynchronize(c->stream));
    for (int64_t p = 0; p < P; p++) {
        double s = 0.0;
        for (int b = 0; b < B; b++) {                       // band order, like the reference's image loop
            const double *q = hout.data() + (size_t)(p * B + b) * nparts;
            double x = q[0];
            for (int k = 1; k < nparts; k++) x += q[k];     // a job's parts first, in order (as k_slice_step does)
            s += x;
        }
        ll_out[p] = s;
    }
#undef PL_TRY
done:
    (void)hipStreamSynchronize(c->stream);
    return rc;
}

int cel_patch_loglik(cel_images *im, cel_sources *src, const int32_t *boxes, const int64_t *offsets,
                     const double *data, int mem, int mode, double *ll_out) {
    return cel_patch_loglik_multi(im, src, nullptr, 1, boxes, offsets, data, mem, mode, ll_out);
}

// (Round 4 measured this kernel on a SECOND stream beside the photon split -- its inputs are ready before the split when the
// chain's trace render came last: the split slowed down by 1.35 ms for the 1.7 ms taken off the flux step and the sweep was
// 0.9 ms LONGER than with the two kernels one behind the other; two VALU- and LDS-bound kernels have nothing to give each
// other.  Removed again.)
int cel_stamp_mass_ready(cel_images *im, cel_sources *src, int *ready) {
    if (!im || !src || !ready) return fail(CEL_ERR_INVALID, "cel_stamp_mass_ready: null argument");
    *ready = (src->ctx == im->ctx && src->B == im->B && im->ctx->mass_reuse_of() && src->gen != 0 && im->massfx_gen == src->gen &&
              src->S * im->B <= im->massfx_cap) ? 1 : 0;
    return CEL_OK;
}

int cel_stamp_mass_begin(cel_images *im, cel_sources *src) {
    if (!im || !src) return fail(CEL_ERR_INVALID, "cel_stamp_mass: null argument");
    if (src->B != im->B || src->ctx != im->ctx) return fail(CEL_ERR_INVALID, "sources do not match images");
    cel_ctx *c = im->ctx;
    HIP_TRY(hipSetDevice(c->device));
    const int B = im->B;
    const int64_t S = src->S;
    im->mass_pending = -1;
    im->mass_todo_S = -1;
    if (S == 0) { im->mass_pending = 0; return CEL_OK; }
    int rc = CEL_OK;
    const bool from_split = c->mass_reuse_of() && src->gen != 0 && im->massfx_gen == src->gen && S * B <= im->massfx_cap;
    if (!from_split || im->recs_gen != src->gen) rc = run_prep(im, src);
    if (rc) return rc;
    double *d_out = nullptr;
    if ((rc = scratch_get(c, 2, sizeof(double) * S * B, (void **)&d_out))) return rc;
    int pi = prof_begin(c, CEL_K_MASS);
    if (from_split) {
        // the photon split that has just run on this catalogue (with k_strict_totals before it) summed every unit stamp it
        // evaluated: those sums are the masses; the mass kernel proper runs on the few jobs the short cut does not vouch for
        int *d_ntodo = im->d_mass_todo + im->massfx_cap;
        HIP_TRY(hipMemsetAsync(d_ntodo, 0, sizeof(int), c->stream));
        hipLaunchKernelGGL(k_mass_from_fx, dim3((unsigned)((S * B + 255) / 256)), dim3(256), 0, c->stream, S * B, B,
                           (const unsigned long long *)im->d_massfx, (const double *)src->d_counts, (const int *)src->d_type, (const BandDev *)im->d_bands, d_out,
                           im->d_mass_todo, d_ntodo);
        // (how many: read in cel_stamp_mass_end, which launches the mass kernel on exactly those -- none at all in a field
        // without very faint sources; a launch of S B blocks that find nothing to do cost 0.15 ms of the flux step)
        HIP_TRY(hipMemcpyAsync(c->pinned + MAX_BANDS + 14, d_ntodo, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        im->mass_todo_S = S;
        im->mass_gen = src->gen; im->mass_todo_ptr = im->d_mass_todo;
    } else {
        hipLaunchKernelGGL((k_patch_ll_hw<3, double>), dim3((unsigned)(S * B)), dim3(64), 0, c->stream, im->d_bands, B, S, im->d_recs,
                           (const int *)nullptr, (const int4 *)nullptr, (const int64_t *)nullptr, (const double *)nullptr,
                           (const double *)nullptr, im->H, im->W, (const int4 *)nullptr, c->tail_T, d_out);
    }
    prof_end(c, pi);
    HIP_TRY(hipGetLastError());
    im->mass_pending = S * B;
    im->d_mass = d_out;
    return CEL_OK;
}

// the second half of cel_stamp_mass without the copy: the leftovers' kernel queued, im->d_mass complete in stream order; *n_out
// = values (0: nothing pending was there to finish)
static int mass_finish(cel_images *im, int64_t *n_out) {
    if (im->mass_pending < 0) return fail(CEL_ERR_INVALID, "cel_stamp_mass_end without a cel_stamp_mass_begin");
    cel_ctx *c = im->ctx;
    HIP_TRY(hipSetDevice(c->device));
    const int64_t n = im->mass_pending;
    im->mass_pending = -1;
    const int64_t S_todo = im->mass_todo_S;
    im->mass_todo_S = -1;
    *n_out = n;
    if (n == 0) return CEL_OK;
    if (S_todo >= 0) {                  // the short cut's leftovers (cel_stamp_mass_begin)
        HIP_TRY(hipStreamSynchronize(c->stream));
        int ntodo = 0;
        memcpy(&ntodo, c->pinned + MAX_BANDS + 14, sizeof(int));
        if (ntodo > 0) {
            // Calls that came between _begin and _end (the header lists what may: cel_gamma_streams, cel_samples_fetch,
            // cel_images_set_epsilon) leave the records and the to-do list alone.  Anything that re-ran k_prep for another
            // catalogue, or a photon split that re-allocated its buffers, is caught here instead of scoring stale records:
            if (im->mass_todo_ptr != im->d_mass_todo || im->recs_gen != im->mass_gen)
                return fail(CEL_ERR_INVALID, "cel_stamp_mass_end: the source records or the photon split were rebuilt since cel_stamp_mass_begin");
            hipLaunchKernelGGL((k_patch_ll_hw<3, double>), dim3((unsigned)ntodo), dim3(64), 0, c->stream, im->d_bands, im->B, S_todo, im->d_recs,
                               (const int *)nullptr, (const int4 *)nullptr, (const int64_t *)nullptr, (const double *)nullptr,
                               (const double *)nullptr, im->H, im->W, (const int4 *)nullptr, c->tail_T, im->d_mass,
                               (const int *)im->d_mass_todo);
            HIP_TRY(hipGetLastError());
        }
    }
    return CEL_OK;
}

int cel_stamp_mass_end(cel_images *im, double *mass) {
    if (!im || !mass) return fail(CEL_ERR_INVALID, "cel_stamp_mass_end: null argument");
    int64_t n = 0;
    int rc = mass_finish(im, &n);
    if (rc || n == 0) return rc;
    return copy_out(mass, im->d_mass, sizeof(double) * n, CEL_HOST, im->ctx->stream);
}

// Source.resample_fluxes for the whole catalogue on the device (k_flux_step): the stamp masses (cel_stamp_mass's own path, short
// cut and leftovers), the Gamma variates and the new fluxes without a host round trip; the catalogue's expected counts are
// rewritten in place.  One D2H of S * 5 doubles + S ints at the end.
int cel_flux_conditionals(cel_images *im, cel_sources *src, uint64_t seed, double a0, double b0, const int32_t *band_letter,
                          const double *calib, const double *kappa, double *flux_new, int32_t *active) {
    if (!im || !src || !band_letter || !calib || !kappa || !flux_new || !active) return fail(CEL_ERR_INVALID, "cel_flux_conditionals: null argument");
    if (src->B != im->B || src->ctx != im->ctx) return fail(CEL_ERR_INVALID, "sources do not match images");
    if (im->samp_S <= 0 || im->samp_S != src->S || !im->ssum_valid)
        return fail(CEL_ERR_INVALID, "cel_flux_conditionals needs a resident photon split of these %lld sources (have %lld)",
                    (long long)src->S, (long long)im->samp_S);
    if (!(a0 > 0.0) || !(b0 > 0.0) || !(a0 < 1e300) || !(b0 < 1e300)) return fail(CEL_ERR_INVALID, "cel_flux_conditionals: a0 and b0 must be positive and finite");
    cel_ctx *c = im->ctx;
    const int B = im->B;
    const int64_t S = src->S;
    for (int b = 0; b < B; b++) {
        if (band_letter[b] < 0 || band_letter[b] > 4) return fail(CEL_ERR_INVALID, "cel_flux_conditionals: band letter %d of image %d outside ugriz", band_letter[b], b);
        if (!(calib[b] > 0.0) || !(kappa[b] > 0.0)) return fail(CEL_ERR_INVALID, "cel_flux_conditionals: calib and kappa must be positive");
    }
    HIP_TRY(hipSetDevice(c->device));
    int rc = cel_stamp_mass_begin(im, src);
    if (rc) return rc;
    int64_t n = 0;
    if ((rc = mass_finish(im, &n))) return rc;
    // the per-image constants ride in one small upload: letter[B] | ratio[B] | calib[B] | kappa[B], then the outputs
    char *d = nullptr;
    const size_t cbytes = (size_t)MAX_BANDS * (sizeof(int) + 3 * sizeof(double));
    if ((rc = scratch_get(c, 6, cbytes + sizeof(double) * 5 * S + sizeof(int) * S + 64, (void **)&d))) return rc;
    struct { int letter[MAX_BANDS]; double ratio[MAX_BANDS], calib[MAX_BANDS], kappa[MAX_BANDS]; } hc;
    memset(&hc, 0, sizeof(hc));
    for (int b = 0; b < B; b++) { hc.letter[b] = band_letter[b]; hc.ratio[b] = kappa[b] / calib[b]; hc.calib[b] = calib[b]; hc.kappa[b] = kappa[b]; }
    static_assert(sizeof(hc) == MAX_BANDS * (sizeof(int) + 3 * sizeof(double)), "packed");
    HIP_TRY(hipMemcpyAsync(d, &hc, sizeof(hc), hipMemcpyHostToDevice, c->stream));      // (pageable: staged before the call returns)
    const int *d_letter = reinterpret_cast<const int *>(d);
    const double *d_ratio = reinterpret_cast<const double *>(d + sizeof(int) * MAX_BANDS);
    double *d_flux = reinterpret_cast<double *>(d + cbytes);
    int *d_act = reinterpret_cast<int *>(d + cbytes + sizeof(double) * 5 * S);
    if (S > 0) {
        hipLaunchKernelGGL(k_flux_step, dim3((unsigned)((S * 5 + 255) / 256)), dim3(256), 0, c->stream, S, B, (const double *)im->d_ssum,
                           (const double *)im->d_mass, (const int64_t *)im->d_soff, d_letter, d_ratio, d_ratio + MAX_BANDS, d_ratio + 2 * MAX_BANDS,
                           a0, b0, (unsigned long long)seed, src->d_counts, d_flux, d_act);
        HIP_TRY(hipGetLastError());
        // the catalogue on the device has new counts: a new generation (no row stamps: everything may have changed)
        src->gen = src->full_gen = ++g_source_gen;
        src->row_gen.clear();
        HIP_TRY(hipMemcpyAsync(flux_new, d_flux, sizeof(double) * 5 * S, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(active, d_act, sizeof(int) * S, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return CEL_OK;
}

// celeste_mcmc.gamma_by_stream on the device (k_gamma_streams): n standard Gamma(a[i]) variates, element i from its own
// streams keyed by (seed, i).  Host arrays in and out.
int cel_gamma_streams(cel_ctx *c, int64_t n, const double *a, uint64_t seed, double *out) {
    if (!c || n < 0 || (n > 0 && (!a || !out))) return fail(CEL_ERR_INVALID, "cel_gamma_streams: bad argument");
    if (n == 0) return CEL_OK;
    for (int64_t i = 0; i < n; i++)
        if (!(a[i] > 0.0) || !(a[i] < 1e300)) return fail(CEL_ERR_INVALID, "cel_gamma_streams: shape parameter %lld is not positive and finite", (long long)i);
    HIP_TRY(hipSetDevice(c->device));
    double *d = nullptr;
    int rc = scratch_get(c, 6, sizeof(double) * 2 * n, (void **)&d);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(d, a, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_gamma_streams, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, n, (const double *)d,
                       (unsigned long long)seed, d + n);
    HIP_TRY(hipGetLastError());
    return copy_out(out, d + n, sizeof(double) * n, CEL_HOST, c->stream);
}

int cel_stamp_mass(cel_images *im, cel_sources *src, double *mass) {
    if (!mass) return fail(CEL_ERR_INVALID, "cel_stamp_mass: null argument");
    int rc = cel_stamp_mass_begin(im, src);
    return rc ? rc : cel_stamp_mass_end(im, mass);
}

// ---- lock-step slice sampling of the locations, on the device -----------------------------------
int cel_slice_locations(cel_images *im, cel_sources *src, const int32_t *chain_ids, double sigma, uint64_t seed,
                        int max_rounds, double *radec_out, double *llh_out, int64_t *stats) {
    if (!im || !src) return fail(CEL_ERR_INVALID, "cel_slice_locations: null argument");
    if (src->B != im->B || src->ctx != im->ctx) return fail(CEL_ERR_INVALID, "sources do not match images");
    if (im->samp_S <= 0 || im->samp_S != src->S)
        return fail(CEL_ERR_INVALID, "cel_slice_locations needs a resident photon split of these %lld sources (have %lld)",
                    (long long)src->S, (long long)im->samp_S);
    if (!(sigma > 0.0) || max_rounds < 1) return fail(CEL_ERR_INVALID, "cel_slice_locations: sigma and max_rounds must be positive");
    cel_ctx *c = im->ctx;
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    const int B = im->B;
    const int64_t S = src->S;
    // one allocation, carved: 2 x u64, 10 x f64 (x and x0 are 2 per chain), 4 x i32 per chain + owner + ll (S*B) + chain ids + 2 ints
    // blocks per (chain, band) job in rounds with at most SLICE_SPLIT_JOBS jobs left (measured at config 3: 4 blocks
    // below 2048 jobs 29.4 ms per location step, below 8192 jobs 29.0; 8 blocks 29.4; without 30.3)
    const int SLICE_SPLIT = PLL_PARTS, SLICE_SPLIT_JOBS = 8192;
    const size_t per_chain = 2 * 8 + 10 * 8 + 4 * 4 + 4 + (size_t)B * 8 * SLICE_SPLIT + 4 + (size_t)B * 4 * 6 * SLICE_SPLIT + 3 * 4;
    const size_t need = per_chain * (size_t)S + 128 + 2 * sizeof(SliceFuse) + 16;
    if (need > im->slice_cap) {
        HIP_TRY(hipStreamSynchronize(st));
        if (im->d_slice) (void)hipFree(im->d_slice);
        im->d_slice = nullptr; im->slice_cap = 0;
        HIP_TRY(hipMalloc(&im->d_slice, need + need / 4));
        im->slice_cap = need + need / 4;
    }
    if (!im->slice_prop || im->slice_prop->cap < S) {
        if (im->slice_prop) cel_sources_destroy(im->slice_prop);
        im->slice_prop = nullptr;
        int rc0 = cel_sources_create(c, S + S / 4 + 16, B, &im->slice_prop);
        if (rc0) return rc0;
    }
    cel_sources *prop = im->slice_prop;
    char *p = (char *)im->d_slice;
    SliceState ss;
    ss.key = (unsigned long long *)p; p += 8 * S;
    ss.count = (unsigned long long *)p; p += 8 * S;
    ss.x = (double *)p; p += 16 * S;
    ss.x0 = (double *)p; p += 16 * S;
    ss.lower = (double *)p; p += 8 * S;
    ss.upper = (double *)p; p += 8 * S;
    ss.log_u = (double *)p; p += 8 * S;
    ss.llh_s = (double *)p; p += 8 * S;
    ss.new_z = (double *)p; p += 8 * S;
    ss.new_llh = (double *)p; p += 8 * S;
    double *d_ll = (double *)p; p += 8 * S * B * SLICE_SPLIT;
    ss.phase = (int *)p; p += 4 * S;
    ss.kdir = (int *)p; p += 4 * S;
    ss.first = (int *)p; p += 4 * S;
    ss.steps = (int *)p; p += 4 * S;
    int *d_owner = (int *)p; p += 4 * S;
    int *d_ids = (int *)p; p += 4 * S;
    // per (job, part) block entries (k_job_work): work estimates, the heaviest-first block lists of a full round, and the
    // running chains' blocks of the late rounds -- for the jobs scored densely and for those scored at their photons
    int *d_work = (int *)p; p += 4 * S * B * SLICE_SPLIT;
    int *d_jobs = (int *)p; p += 4 * S * B * SLICE_SPLIT;
    int *d_live = (int *)p; p += 4 * S * B * SLICE_SPLIT;
    int *d_work_nz = (int *)p; p += 4 * S * B * SLICE_SPLIT;
    int *d_jobs_nz = (int *)p; p += 4 * S * B * SLICE_SPLIT;
    int *d_live_nz = (int *)p; p += 4 * S * B * SLICE_SPLIT;
    // the fused rounds (SliceFuse, k_slice_state.h): per chain its ticket counter and the blocks the full / the live lists hold for it
    int *d_tick = (int *)p; p += 4 * S;
    int *d_need_full = (int *)p; p += 4 * S;
    int *d_need_live = (int *)p; p += 4 * S;
    int *d_flags = (int *)p;            // [0] chains still running, [1] error bits, [2] likelihood evaluations so far, [3] rounds with work,
                                        // [4] / [5] live dense / photon-list jobs of the batch, [6..7] byte counter, [8] / [9] dense / photon-list jobs;
                                        // a call that may fuse rounds: [6] chains still running (k_slice_live_jobs, per batch: a fused round keeps no
                                        // count), [11] / [12] evaluations / rounds with work (k_slice_bytes, at the end)
    // the proposal set: this catalogue with the locations rewritten every round
    HIP_TRY(hipMemcpyAsync(prop->d_type, src->d_type, sizeof(int) * S, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(prop->d_counts, src->d_counts, sizeof(double) * B * S, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(prop->d_shape, src->d_shape, sizeof(double) * 4 * S, hipMemcpyDeviceToDevice, st));
    prop->S = S;
    prop->n_gal = -1;
    if (chain_ids) {
        HIP_TRY(hipMemcpyAsync(d_ids, chain_ids, sizeof(int) * S, hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    HIP_TRY(hipMemsetAsync(d_flags, 0, sizeof(int) * 13, st));      // [0..3] and [10]: see k_slice.h; the rest is set where it is used
    if (c->slice_fuse) HIP_TRY(hipMemsetAsync(d_tick, 0, sizeof(int) * 2 * S, st));    // the tickets and the full lists' block counts
    const unsigned g256 = (unsigned)((S + 255) / 256);
    hipLaunchKernelGGL(k_slice_init, dim3(g256), dim3(256), 0, st, ss, S, src->d_radec, chain_ids ? d_ids : (const int *)nullptr,
                       im->d_soff, B, (unsigned long long)seed, sigma, d_owner);
    // the (chain, band) jobs of a round, heaviest first (the photon rectangles are fixed for the call)
    // The blocks of a round.  With the recurrence kernels every (chain, band) job is one block, or PLL_PARTS blocks when it
    // is long (k_job_work), listed heaviest first; with photon lists a patch is scored either densely (k_patch_ll_hw<0>) or
    // at its photons (k_patch_ll_nz): two lists.  The photon rectangles, lists and routes are fixed for the call.
    const bool use_nz = im->nz_valid && c->variant != 0;
    int64_t n_dense = S * B, n_nz = 0;
    if (c->variant != 0) {
        const int nent = (int)(S * B * SLICE_SPLIT);
        hipLaunchKernelGGL(k_job_work, dim3((unsigned)((nent + 255) / 256)), dim3(256), 0, st, src->d_type, im->d_snz, S, B, d_work,
                           (const int *)(use_nz ? im->d_nzmode : nullptr), (const int *)im->d_nnz, use_nz ? d_work_nz : (int *)nullptr,
                           (use_nz && c->slice_fuse) ? d_need_full : (int *)nullptr);
        // members compacted by the whole GPU, then ordered heaviest first by one block (the live-list arrays are free now)
        HIP_TRY(hipMemsetAsync(d_flags + 8, 0, sizeof(int) * 2, st));
        hipLaunchKernelGGL(k_list_members, dim3((unsigned)((nent + 255) / 256)), dim3(256), 0, st, d_work, nent, d_live, d_live_nz, d_flags + 8);
        hipLaunchKernelGGL(k_order_compact, dim3(1), dim3(1024), 0, st, (const int *)d_live, (const int *)d_live_nz, (const int *)(d_flags + 8), d_jobs);
        if (use_nz) {
            hipLaunchKernelGGL(k_list_members, dim3((unsigned)((nent + 255) / 256)), dim3(256), 0, st, d_work_nz, nent, d_live, d_live_nz, d_flags + 9);
            hipLaunchKernelGGL(k_order_compact, dim3(1), dim3(1024), 0, st, (const int *)d_live, (const int *)d_live_nz, (const int *)(d_flags + 9), d_jobs_nz);
        }
        int *h_cnt = reinterpret_cast<int *>(c->pinned + MAX_BANDS + 2);
        h_cnt[1] = 0;
        HIP_TRY(hipMemcpyAsync(h_cnt, d_flags + 8, sizeof(int) * (use_nz ? 2 : 1), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        n_dense = h_cnt[0]; n_nz = use_nz ? h_cnt[1] : 0;
    }
    // One launch per round: every patch of the call is scored at its photons (no dense job: the usual case, and config 5's), so
    // the block that finishes a chain's last job of a round steps the chain itself (SliceFuse).  With dense jobs in the call the
    // chains are stepped by k_slice_step as before (the dense kernel carries no ticket).  CEL_OPT_SLICE_FUSE = 0: never fused.
    // MEASURED, AND OFF BY DEFAULT (round 6, kernel traces of ten sweeps per setting, the same chains: the location step's span
    // on the device 12.67 ms unfused; fused in rounds of at most 4 096 blocks 12.65, 12 288: 12.70, 32 768: 12.9, every round:
    // 13.25).  A ticket is a returning atomic every block waits for before it leaves (~2 us of a wave slot) and the stepping block
    // holds its slot ~7 us longer: on a full round (50 000 blocks on 4 096 slots) 45 us more kernel time against the 27 us of the
    // step launch and its two gaps; in the late rounds the step's latency moves from a launch of its own to the end of the
    // likelihood launch and the round gains under 10 us.  c->slice_fuse = N: rounds of at most N blocks are fused (1: every round).
    const bool fuse_ok = use_nz && c->slice_fuse && n_dense == 0 && n_nz > 0;
    bool fused = false;                               // of the batch being queued
    int64_t rounds = 0, evals = 0, queued = 0, live = S;
    if (chain_ids) {                    // chains with a negative id are another rank's: they never run here
        live = 0;
        for (int64_t s = 0; s < S; s++) live += chain_ids[s] >= 0;
        if (live < 1) live = 1;         // a launch needs a block; k_slice_live_jobs finds nothing to run
    }
    int rc = CEL_OK;
    // The chains advance on the device alone (propose -> records -> likelihoods -> consume), so rounds are queued SLICE_BATCH
    // at a time and the flags read once per batch -- and TWO batches are kept in flight: batch k + 1 is queued before the
    // host waits for batch k's flags, so the queue does not drain while the host takes its turn (with one batch in flight
    // a kernel trace shows the device idle for 50-100 us at every batch boundary; on an idle host the step measures the
    // same either way, 12.6-13.0 ms -- the second batch is there for the hosts that wake up late).
    // A round queued after the last chain has finished scores nothing (every job retires at its first instruction) and
    // is not counted: the price is at most one batch of empty launches at the end.
    const int SLICE_BATCH = 4;                       // (6, 8 and 12 measure the same sweep)
    const int ostr = (c->variant != 0) ? SLICE_SPLIT : 1;        // slots per job in d_ll: the recurrence kernels always fill all four
    // the blocks of a batch: the full heaviest-first lists while (nearly) every chain runs, afterwards the running chains'
    // blocks (k_slice_live_jobs, built behind the batch before).  The host sizes a launch by the newest counts it has READ --
    // those of the batch before the one whose lists the launch walks; chains only retire, so those are upper bounds (times
    // PLL_PARTS where the lists in between began to deal every job), and the kernels stop at the device's own count.
    int64_t live_dense = -1, live_nz = -1;                       // < 0: no live lists read yet
    int *h_flag_slot[2] = {reinterpret_cast<int *>(c->pinned + MAX_BANDS + 2), reinterpret_cast<int *>(c->pinned + MAX_BANDS + 8)};
    PrepArgs pa;                                                 // what run_prep(im, prop, d_owner, 1) hands k_prep
    pa.bands = im->d_bands; pa.B = B; pa.H = im->full_H; pa.W = im->W; pa.win_y0 = im->win_y0; pa.win_h = im->H; pa.S = S;
    pa.type = prop->d_type; pa.counts = prop->d_counts; pa.shape = prop->d_shape; pa.rsq_gal = rsq_galaxy();
    pa.recs = im->d_recs; pa.boxes = im->d_boxes; pa.kind = im->d_kind; pa.status = im->d_status; pa.nobox = 1;
    if ((rc = ensure_recs(im, S * B > 0 ? S * B : 1))) return rc;   // (before the pointers are taken)
    pa.recs = im->d_recs; pa.boxes = im->d_boxes; pa.kind = im->d_kind; pa.status = im->d_status;
    // the fused rounds' arguments, in device memory: [0] for the full lists, [1] for the live lists (they differ in `need`)
    SliceFuse *d_fz = reinterpret_cast<SliceFuse *>(((uintptr_t)(d_flags + 13) + 15) & ~(uintptr_t)15);
    if (fuse_ok) {
        SliceFuse fz[2];
        memset(fz, 0, sizeof(fz));
        for (int k = 0; k < 2; k++) {
            fz[k].tick = d_tick; fz[k].need = k ? d_need_live : d_need_full; fz[k].st = ss; fz[k].ll_pb = d_ll; fz[k].nparts = SLICE_SPLIT;
            fz[k].B = B; fz[k].sigma = sigma; fz[k].flags = d_flags; fz[k].prop_radec = prop->d_radec; fz[k].owner = d_owner; fz[k].pa = pa;
        }
        HIP_TRY(hipMemcpyAsync(d_fz, fz, sizeof(fz), hipMemcpyHostToDevice, st));      // (pageable: staged before the call returns)
    }
    if (!c->slice_ev[0]) {
        HIP_TRY(hipEventCreateWithFlags(&c->slice_ev[0], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&c->slice_ev[1], hipEventDisableTiming));
    }
    int last_par[2] = {0, 0};                                    // parity of a batch's last round: which slot holds its running count
    int deal_of[2] = {0, 0};                                     // whether a batch's live lists dealt every job
    int deal_read = 0;                                           // ... of the batch whose counts live_dense / live_nz are
    int qb = 0, rb = 0;                                          // batches queued / read
    bool finished = false;
    for (;;) {
        while (qb - rb < 2 && queued < max_rounds && !finished) {
            const int nb = (int)std::min<int64_t>(SLICE_BATCH, (int64_t)max_rounds - queued);
            const bool use_live = live_dense >= 0 && (live_dense + live_nz) * 4 < (n_dense + n_nz) * 3;
            // the lists this batch walks are those of batch qb - 1; the counts in hand those of batch rb - 1 <= qb - 1
            const int grow = (use_live && qb >= 1 && deal_of[(qb - 1) & 1] && !deal_read) ? SLICE_SPLIT : 1;
            const int64_t cap_blocks = S * B * SLICE_SPLIT;
            const int64_t gd = use_live ? std::min<int64_t>(live_dense * grow, cap_blocks) : n_dense;
            const int64_t gn = !use_nz ? 0 : (use_live ? std::min<int64_t>(live_nz * grow, cap_blocks) : n_nz);
            fused = fuse_ok && (c->slice_fuse == 1 || gn <= (int64_t)c->slice_fuse);
            for (int k = 0; k < nb; k++) {
                // the first round's points; every later round's were named by the step kernel of the round before
                if (queued == 0) hipLaunchKernelGGL(k_slice_propose, dim3(g256), dim3(256), 0, st, ss, S, prop->d_radec, d_owner, d_flags);
                prop->gen = prop->full_gen = ++g_source_gen;
                // the first round's records; every later round's were written by the step kernel that named its points
                if (queued == 0 && (rc = run_prep(im, prop, d_owner, 1))) return rc;      // the patch limits are fixed: no boxes
                if (c->variant == 0) {
                    int pi = prof_slot(c, CEL_K_PATCH_LL);
                    LAUNCH_EV(k_patch_ll<int>, dim3((unsigned)(S * B)), dim3(256), st, EV0(c, pi), EV1(c, pi), im->d_bands, B, S, im->d_recs,
                              d_owner, im->d_sbox, im->d_soff, (const int *)im->d_samp, im->d_nelec, im->H, im->W, 0, d_ll);
                } else {
                    if (gd > 0) {
                        int pi = prof_slot(c, CEL_K_PATCH_LL);
                        LAUNCH_EV((k_patch_ll_hw<0, int>), dim3((unsigned)gd), dim3(64), st, EV0(c, pi), EV1(c, pi),
                                  im->d_bands, B, S, im->d_recs,
                                  d_owner, im->d_sbox, im->d_soff, (const int *)im->d_samp, im->d_nelec, im->H, im->W, im->d_snz, c->tail_T, d_ll,
                                  (const int *)(use_live ? d_live : d_jobs), 1, (const int *)(use_live ? d_flags + 4 : nullptr),
                                  (const int *)(use_nz ? im->d_nzmode : nullptr), 1, SLICE_SPLIT);
                    }
                    if (gn > 0) {
                        int pi = prof_slot(c, CEL_K_PATCH_LL);
                        const SliceFuse *fq = fused ? d_fz + (use_live ? 1 : 0) : nullptr;
                        if (fq)
                            LAUNCH_EV(k_patch_ll_nz<true>, dim3((unsigned)gn), dim3(64), st, EV0(c, pi), EV1(c, pi), im->d_bands, B, S, im->d_recs,
                                      (const int *)d_owner, (const int4 *)im->d_sbox, (const int4 *)im->d_snz, (const int *)im->d_nzmode,
                                      (const int64_t *)im->d_nzoff, (const NzEntry *)im->d_nzlist, d_ll,
                                      (const int *)(use_live ? d_live_nz : d_jobs_nz), (const int *)(use_live ? d_flags + 5 : nullptr), fq);
                        else
                            LAUNCH_EV(k_patch_ll_nz<false>, dim3((unsigned)gn), dim3(64), st, EV0(c, pi), EV1(c, pi), im->d_bands, B, S, im->d_recs,
                                      (const int *)d_owner, (const int4 *)im->d_sbox, (const int4 *)im->d_snz, (const int *)im->d_nzmode,
                                      (const int64_t *)im->d_nzoff, (const NzEntry *)im->d_nzlist, d_ll,
                                      (const int *)(use_live ? d_live_nz : d_jobs_nz), (const int *)(use_live ? d_flags + 5 : nullptr), fq);
                    }
                }
                if (!fused)
                hipLaunchKernelGGL(k_slice_step, dim3((unsigned)((S + 63) / 64)), dim3(64 * B), 0, st, ss, S, B, ostr, d_ll, sigma, d_flags, (int)queued, prop->d_radec, d_owner,
                                   (k == nb - 1 && c->variant != 0 && !fuse_ok) ? 1 : 0, pa);
                queued++;
            }
            const int slot = qb & 1;
            deal_of[slot] = 0;
            if (c->variant != 0) {          // the running chains' blocks, for the next batch; few chains left: every job dealt
                deal_of[slot] = (live * B <= SLICE_SPLIT_JOBS) ? 1 : 0;
                if (fuse_ok) {                   // (otherwise the batch's last k_slice_step cleared the two counts)
                    HIP_TRY(hipMemsetAsync(d_flags + 4, 0, sizeof(int) * 3, st));
                    HIP_TRY(hipMemsetAsync(d_need_live, 0, sizeof(int) * S, st));
                }
                hipLaunchKernelGGL(k_slice_live_jobs, dim3((unsigned)((S * B + 255) / 256)), dim3(256), 0, st, ss, S, B, d_live, d_flags + 4,
                                   (const int *)(use_nz ? im->d_nzmode : nullptr), d_live_nz, d_flags + 5, (const int *)im->d_nnz,
                                   (const int4 *)im->d_snz, deal_of[slot], fuse_ok ? d_need_live : (int *)nullptr, fuse_ok ? d_flags + 6 : (int *)nullptr);
            }
            HIP_TRY(hipMemcpyAsync(h_flag_slot[slot], d_flags, sizeof(int) * 11, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipEventRecord(c->slice_ev[slot], st));
            HIP_TRY(hipGetLastError());
            last_par[slot] = (int)((queued - 1) & 1);
            qb++;
        }
        if (rb == qb) break;            // nothing in flight: max_rounds reached
        const int slot = rb & 1;
        HIP_TRY(hipEventSynchronize(c->slice_ev[slot]));
        rb++;
        const int *hf = h_flag_slot[slot];
        const int running = fuse_ok ? hf[6] : hf[last_par[slot] ? 10 : 0], err = hf[1];    // the batch's last round's slot; a call that may fuse: k_slice_live_jobs' count
        live = running;
        if (c->variant != 0) { live_dense = hf[4]; live_nz = hf[5]; deal_read = deal_of[slot]; }
        if (err & 3) {
            (void)hipStreamSynchronize(st);                      // the batch still in flight works on this call's buffers
            return fail(CEL_ERR_INVALID, (err & 1) ? "Slice sampler got a NaN" : "Slice sampler shrank to zero!");
        }
        if (!fuse_ok) { evals = hf[2]; rounds = hf[3]; }
        if (running == 0) finished = true;                       // whatever is still queued scores nothing
        else if (queued >= max_rounds && rb == qb)
            return fail(CEL_ERR_INVALID, "cel_slice_locations: %d rounds without every chain finishing", max_rounds);
        if (finished) break;
    }
    // the new locations replace the catalogue's
    HIP_TRY(hipMemcpyAsync(src->d_radec, ss.x, sizeof(double) * 2 * S, hipMemcpyDeviceToDevice, st));
    src->gen = src->full_gen = ++g_source_gen;
    src->row_gen.clear();
    if (radec_out) HIP_TRY(hipMemcpyAsync(radec_out, ss.x, sizeof(double) * 2 * S, hipMemcpyDeviceToHost, st));
    if (llh_out) HIP_TRY(hipMemcpyAsync(llh_out, ss.new_llh, sizeof(double) * S, hipMemcpyDeviceToHost, st));
    unsigned long long *d_bytes = reinterpret_cast<unsigned long long *>(((uintptr_t)(d_flags + 6) + 7) & ~(uintptr_t)7);
    if (stats) {
        // algorithmic bytes of the call: a chain made 1 + steps[s] evaluations (the first direction's level, then one
        // per shrink step), each walking its photon rectangles
        HIP_TRY(hipMemsetAsync(d_bytes, 0, sizeof(unsigned long long), st));
        hipLaunchKernelGGL(k_slice_bytes, dim3(g256), dim3(256), 0, st, ss, S, B, (const int4 *)im->d_snz, d_bytes,
                           (const int *)(use_nz ? im->d_nzmode : nullptr), (const int64_t *)im->d_nzoff, fuse_ok ? d_flags + 11 : (int *)nullptr);
        HIP_TRY(hipMemcpyAsync(h_flag_slot[0] + 6, d_bytes, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
        if (fuse_ok) HIP_TRY(hipMemcpyAsync(h_flag_slot[1], d_flags + 11, sizeof(int) * 2, hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(hipStreamSynchronize(st));
    if (stats) {
        if (fuse_ok) { evals = h_flag_slot[1][0]; rounds = h_flag_slot[1][1]; }    // what the per-round counts of the unfused rounds add up to
        stats[0] = rounds; stats[1] = evals;
        stats[2] = (int64_t)(*reinterpret_cast<unsigned long long *>(h_flag_slot[0] + 6));
        stats[3] = queued;
    }
    return CEL_OK;
}

// The general slice sampler (k_slice_gen.h): random directions, stepping out by doubling, D = 2 or 4.
int cel_slice_sample(cel_images *im, cel_sources *src, int param, const int32_t *chain_ids, const double *dirs, int numdir,
                     int step_out, int max_steps_out, double sigma, double phi_max, uint64_t seed, int max_rounds,
                     double *x_out, double *llh_out, int64_t *stats) {
    if (!im || !src) return fail(CEL_ERR_INVALID, "cel_slice_sample: null argument");
    if (src->B != im->B || src->ctx != im->ctx) return fail(CEL_ERR_INVALID, "sources do not match images");
    if (im->samp_S <= 0 || im->samp_S != src->S)
        return fail(CEL_ERR_INVALID, "cel_slice_sample needs a resident photon split of these %lld sources (have %lld)",
                    (long long)src->S, (long long)im->samp_S);
    if (param != 0 && param != 1) return fail(CEL_ERR_INVALID, "cel_slice_sample: param must be 0 (location) or 1 (shape)");
    if (!(sigma > 0.0) || max_rounds < 1 || max_steps_out < 0) return fail(CEL_ERR_INVALID, "cel_slice_sample: sigma and max_rounds must be positive");
    const int D = param ? 4 : 2;
    const int compwise = dirs ? 0 : 1;
    const int ndir = compwise ? D : numdir;
    if (ndir < 1 || ndir > 64) return fail(CEL_ERR_INVALID, "cel_slice_sample: numdir must be in [1, 64]");
    cel_ctx *c = im->ctx;
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    const int B = im->B;
    const int64_t S = src->S;
    const int ostr = (c->variant != 0) ? PLL_PARTS : 1;
    // one allocation, carved: 2 x u64, (3 D + 13) x f64, (4 + D) x i32 per chain, the directions, 2 owners, 2 B ostr log-likelihood slots
    const size_t per_chain = 2 * 8 + (size_t)(3 * D + 13) * 8 + (size_t)(4 + D) * 4 + (size_t)ndir * D * 8 + 2 * 4 + (size_t)2 * B * ostr * 8 + 4 +
                             (size_t)2 * 2 * B * PLL_PARTS * 4;
    const size_t need = per_chain * (size_t)S + 256;
    if (need > im->sgen_cap) {
        HIP_TRY(hipStreamSynchronize(st));
        if (im->d_sgen) (void)hipFree(im->d_sgen);
        im->d_sgen = nullptr; im->sgen_cap = 0;
        HIP_TRY(hipMalloc(&im->d_sgen, need + need / 4));
        im->sgen_cap = need + need / 4;
    }
    if (!im->sgen_prop || im->sgen_prop->cap < 2 * S) {
        if (im->sgen_prop) cel_sources_destroy(im->sgen_prop);
        im->sgen_prop = nullptr;
        int rc0 = cel_sources_create(c, 2 * S + S / 2 + 16, B, &im->sgen_prop);
        if (rc0) return rc0;
    }
    cel_sources *prop = im->sgen_prop;
    char *p = (char *)im->d_sgen;
    SliceGen g;
    SliceState rs;
    rs.key = (unsigned long long *)p; p += 8 * S;
    rs.count = (unsigned long long *)p; p += 8 * S;
    g.key = rs.key; g.count = rs.count;
    g.x = (double *)p; p += 8 * S * D;
    g.x0 = (double *)p; p += 8 * S * D;
    g.dir = (double *)p; p += 8 * S * D;
    g.lower = (double *)p; p += 8 * S;
    g.upper = (double *)p; p += 8 * S;
    g.log_u = (double *)p; p += 8 * S;
    g.llh_s = (double *)p; p += 8 * S;
    g.new_z = (double *)p; p += 8 * S;
    g.new_llh = (double *)p; p += 8 * S;
    g.start_lower = (double *)p; p += 8 * S;
    g.start_upper = (double *)p; p += 8 * S;
    g.acc_L = (double *)p; p += 8 * S;
    g.acc_U = (double *)p; p += 8 * S;
    g.pri = (double *)p; p += 16 * S;
    double *d_dirs = (double *)p; p += 8 * S * ndir * D;
    double *d_ll = (double *)p; p += 8 * 2 * S * B * ostr;
    g.phase = (int *)p; p += 4 * S;
    g.kdir = (int *)p; p += 4 * S;
    g.l_out = (int *)p; p += 4 * S;
    g.u_out = (int *)p; p += 4 * S;
    g.order = (int *)p; p += 4 * S * D;
    int *d_owner = (int *)p; p += 8 * S;
    int *d_ids = (int *)p; p += 4 * S;
    int *d_list = (int *)p; p += 4 * 2 * S * B * PLL_PARTS;          // the running chains' blocks (k_sg_live_jobs): dense, at the photons
    int *d_list_nz = (int *)p; p += 4 * 2 * S * B * PLL_PARTS;
    int *d_flags = (int *)(((uintptr_t)p + 15) & ~(uintptr_t)15);
    g.D = D; g.ndir = ndir; g.compwise = compwise; g.step_out = step_out ? 1 : 0; g.max_steps_out = max_steps_out; g.param = param;
    g.sigma = sigma; g.phi_max = param ? phi_max : 0.0; g.dirs = compwise ? nullptr : d_dirs;
    if (!compwise) HIP_TRY(hipMemcpyAsync(d_dirs, dirs, sizeof(double) * S * ndir * D, hipMemcpyHostToDevice, st));
    if (chain_ids) HIP_TRY(hipMemcpyAsync(d_ids, chain_ids, sizeof(int) * S, hipMemcpyHostToDevice, st));
    if (!compwise || chain_ids) HIP_TRY(hipStreamSynchronize(st));          // pageable sources must stay valid
    HIP_TRY(hipMemsetAsync(d_flags, 0, sizeof(int) * 8, st));
    const unsigned g256 = (unsigned)((S + 255) / 256);
    // the proposal set: every chain's source twice, the sampled parameter rewritten every round
    hipLaunchKernelGGL(k_sg_fill, dim3((unsigned)((2 * S + 255) / 256)), dim3(256), 0, st, S, B, (const int *)src->d_type,
                       (const double *)src->d_radec, (const double *)src->d_counts, (const double *)src->d_shape,
                       prop->d_type, prop->d_radec, prop->d_counts, prop->d_shape);
    prop->S = 2 * S;
    prop->n_gal = -1;
    hipLaunchKernelGGL(k_sg_init, dim3(g256), dim3(256), 0, st, g, rs, S, (const double *)(param ? src->d_shape : src->d_radec),
                       chain_ids ? (const int *)d_ids : (const int *)nullptr, (const int64_t *)im->d_soff, B, (const int *)src->d_type,
                       (unsigned long long)seed);
    const bool use_nz = im->nz_valid && c->variant != 0;
    int64_t rounds = 0, evals = 0, queued = 0;
    int *h_flags = reinterpret_cast<int *>(c->pinned + MAX_BANDS + 2);
    int rc = CEL_OK;
    const int64_t P = 2 * S;
    const int BATCH = 4;
    const int64_t DEAL_ALL_BELOW = 8192;          // (chain, slot, band) jobs: fewer than the GPU has wave slots for -- deal every job
    int64_t live = S;                             // chains running at the last readback
    // the first batch's block lists (every later one is built behind the batch before it); their counts come back now
    int64_t n_dense = P * B, n_nz = 0;
    if (c->variant != 0) {
        hipLaunchKernelGGL(k_sg_live_jobs, dim3((unsigned)((S * B + 255) / 256)), dim3(256), 0, st, g, S, B,
                           (const int *)(use_nz ? im->d_nzmode : nullptr), (const int *)im->d_nnz, (const int4 *)im->d_snz, 0,
                           d_list, d_flags + 4, d_list_nz, d_flags + 5);
        HIP_TRY(hipMemcpyAsync(h_flags, d_flags, sizeof(int) * 6, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        n_dense = h_flags[4]; n_nz = h_flags[5];
    }
    for (;;) {
        const int nb = (int)std::min<int64_t>(BATCH, (int64_t)max_rounds - queued);
        for (int k = 0; k < nb; k++) {
            hipLaunchKernelGGL(k_sg_propose, dim3(g256), dim3(256), 0, st, g, rs, S, param ? prop->d_shape : prop->d_radec, d_owner, d_flags,
                               queued == 0 ? 1 : 0);
            prop->gen = prop->full_gen = ++g_source_gen;
            if ((rc = run_prep(im, prop, d_owner, 1))) return rc;      // the patch limits are fixed: no boxes
            if (c->variant == 0) {
                int pi = prof_slot(c, CEL_K_PATCH_LL);
                LAUNCH_EV(k_patch_ll<int>, dim3((unsigned)(P * B)), dim3(256), st, EV0(c, pi), EV1(c, pi), im->d_bands, B, P, im->d_recs,
                          (const int *)d_owner, im->d_sbox, im->d_soff, (const int *)im->d_samp, im->d_nelec, im->H, im->W, 0, d_ll);
            } else {
                if (n_dense > 0) {
                    int pi = prof_slot(c, CEL_K_PATCH_LL);
                    LAUNCH_EV((k_patch_ll_hw<0, int>), dim3((unsigned)n_dense), dim3(64), st, EV0(c, pi), EV1(c, pi), im->d_bands, B, P, im->d_recs,
                              (const int *)d_owner, im->d_sbox, im->d_soff, (const int *)im->d_samp, im->d_nelec, im->H, im->W, im->d_snz, c->tail_T, d_ll,
                              (const int *)d_list, 1, (const int *)nullptr, (const int *)(use_nz ? im->d_nzmode : nullptr), 1, PLL_PARTS);
                }
                if (n_nz > 0) {
                    int pi = prof_slot(c, CEL_K_PATCH_LL);
                    LAUNCH_EV(k_patch_ll_nz<false>, dim3((unsigned)n_nz), dim3(64), st, EV0(c, pi), EV1(c, pi), im->d_bands, B, P, im->d_recs,
                              (const int *)d_owner, (const int4 *)im->d_sbox, (const int4 *)im->d_snz, (const int *)im->d_nzmode,
                              (const int64_t *)im->d_nzoff, (const NzEntry *)im->d_nzlist, d_ll, (const int *)d_list_nz, (const int *)nullptr, (const SliceFuse *)nullptr);
                }
            }
            hipLaunchKernelGGL(k_sg_consume, dim3(g256), dim3(256), 0, st, g, rs, S, B, ostr, (const double *)d_ll, d_flags);
            queued++;
        }
        if (c->variant != 0) {          // the next batch's blocks: the chains still running now
            HIP_TRY(hipMemsetAsync(d_flags + 4, 0, sizeof(int) * 2, st));
            hipLaunchKernelGGL(k_sg_live_jobs, dim3((unsigned)((S * B + 255) / 256)), dim3(256), 0, st, g, S, B,
                               (const int *)(use_nz ? im->d_nzmode : nullptr), (const int *)im->d_nnz, (const int4 *)im->d_snz,
                               (live * 2 * B <= DEAL_ALL_BELOW) ? 1 : 0, d_list, d_flags + 4, d_list_nz, d_flags + 5);
        }
        HIP_TRY(hipMemcpyAsync(h_flags, d_flags, sizeof(int) * 6, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(st));
        const int running = h_flags[0], err = h_flags[1];
        if (c->variant != 0) { n_dense = h_flags[4]; n_nz = h_flags[5]; }
        live = running;
        if (err & 1) return fail(CEL_ERR_INVALID, "Slice sampler got a NaN");
        if (err & 2) return fail(CEL_ERR_INVALID, "Slice sampler shrank to zero!");
        evals = h_flags[2];
        rounds = h_flags[3];
        if (running == 0) break;
        if (queued >= max_rounds) return fail(CEL_ERR_INVALID, "cel_slice_sample: %d rounds without every chain finishing", max_rounds);
    }
    // the new states replace the catalogue's
    HIP_TRY(hipMemcpyAsync(param ? src->d_shape : src->d_radec, g.x, sizeof(double) * D * S, hipMemcpyDeviceToDevice, st));
    src->gen = src->full_gen = ++g_source_gen; src->row_gen.clear();
    if (x_out) HIP_TRY(hipMemcpyAsync(x_out, g.x, sizeof(double) * D * S, hipMemcpyDeviceToHost, st));
    if (llh_out) HIP_TRY(hipMemcpyAsync(llh_out, g.new_llh, sizeof(double) * S, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (stats) { stats[0] = rounds; stats[1] = evals; stats[2] = 0; stats[3] = queued; }
    return CEL_OK;
}

// ---- photon split -------------------------------------------------------------------------------
int cel_source_boxes(cel_images *im, cel_sources *src, int32_t *boxes, int32_t *status) {
    if (!im || !src || !boxes || !status) return fail(CEL_ERR_INVALID, "cel_source_boxes: null argument");
    if (src->B != im->B || src->ctx != im->ctx) return fail(CEL_ERR_INVALID, "sources do not match images");
    cel_ctx *c = im->ctx;
    HIP_TRY(hipSetDevice(c->device));
    int rc = host_boxes(im, src);
    if (rc) return rc;
    const int64_t n = src->S * im->B;
    for (int64_t i = 0; i < n; i++) {
        const int4 b = im->h_boxes[(size_t)i];
        boxes[4 * i + 0] = b.z; boxes[4 * i + 1] = b.w;
        boxes[4 * i + 2] = b.x; boxes[4 * i + 3] = b.y;
        status[i] = im->h_status[(size_t)i];
    }
    return CEL_OK;
}

int cel_photon_split(cel_images *im, cel_sources *src, uint64_t seed, const int64_t *offsets, double *samp,
                     int mem, double *noise) {
    if (!im || !src) return fail(CEL_ERR_INVALID, "cel_photon_split: null argument");
    const bool resident = (offsets == nullptr);
    if (!resident && !samp) return fail(CEL_ERR_INVALID, "cel_photon_split: offsets given without an output buffer");
    if (!im->have_nelec) return fail(CEL_ERR_INVALID, "cel_photon_split needs cel_images_set_nelec first");
    if (im->TW * im->TH != 2048) return fail(CEL_ERR_INVALID, "cel_photon_split needs 2048-pixel render tiles");
    cel_ctx *c = im->ctx;
    if (src->ctx != im->ctx || src->B != im->B) return fail(CEL_ERR_INVALID, "sources do not match images");
    HIP_TRY(hipSetDevice(c->device));
    // records + tile lists for exactly these sources come from a render.  Recurrence form: that
    // render is the one the split needs anyway -- every pixel's total rate, an image of its own
    // with the split's strict boxes (the model image of cel_images_get_lambda is left alone).
    // Direct form: a plain render (its kernel accumulates the totals itself).
    const bool hw = (c->variant != 0) && (im->TW == HW_TW);
    bool use_massfx = false;
    const double *full_rate = nullptr;
    int rc;
    if (!hw) im->partials_gen = 0;              // (the direct form keeps its noise partials in the render's buffer)
    if (hw) {
        if (!im->d_rate) HIP_TRY(hipMalloc((void **)&im->d_rate, sizeof(double) * (size_t)im->B * im->H * im->W));
        im->massfx_gen = 0;
        const bool current = src->gen != 0 && im->lambda_gen == src->gen && im->lists_gen == src->gen && im->recs_gen == src->gen;
        if (c->split_full) {
            // CEL_OPT_SPLIT_FULL_BOX: a source takes part on its WHOLE box, so the totals are the model image itself -- the one
            // on the device when it is these sources', a render into the totals image otherwise
            if (c->split_reuse && current) { full_rate = im->d_lambda; rc = CEL_OK; }
            else rc = render_impl(im, src, 0, nullptr, nullptr, im->d_rate);
        } else if (c->split_reuse && current) {
            const int64_t nm = src->S * im->B;
            if (resident && c->mass_reuse_of() && nm > 0) {
                // both kernels of this path also sum every unit stamp they evaluate: together the stamps' masses (cel_stamp_mass)
                if (nm > im->massfx_cap) {
                    HIP_TRY(hipStreamSynchronize(c->stream));
                    if (im->d_massfx) (void)hipFree(im->d_massfx);
                    if (im->d_mass_todo) (void)hipFree(im->d_mass_todo);
                    im->d_massfx = nullptr; im->d_mass_todo = nullptr; im->massfx_cap = 0;
                    const int64_t cap = nm + nm / 4 + 64;
                    HIP_TRY(hipMalloc((void **)&im->d_massfx, sizeof(unsigned long long) * cap));
                    HIP_TRY(hipMalloc((void **)&im->d_mass_todo, sizeof(int) * (cap + 1)));
                    im->massfx_cap = cap;
                }
                HIP_TRY(hipMemsetAsync(im->d_massfx, 0, sizeof(unsigned long long) * nm, c->stream));
                use_massfx = true;
            }
            // the model image, records and tile lists of exactly these sources are on the device (the chain's trace render
            // came last): the totals are that image minus every source's first box row and column (k_border.h)
            RenderArgs a;
            memset(&a, 0, sizeof(a));
            a.bands = im->d_bands; a.recs = im->d_recs; a.lists = im->d_lists; a.tile_cnt = im->d_tile_cnt; a.tile_off = im->d_tile_off;
            a.lambda = im->d_lambda; a.S = src->S; a.capacity = im->lists_cap; a.B = im->B; a.H = im->H; a.W = im->W;
            a.ntx = im->ntx; a.nty = im->nty;
            int pi = prof_slot(c, CEL_K_TOTALS);
            LAUNCH_EV(k_strict_totals, dim3((unsigned)(im->B * im->ntx * im->nty)), dim3(64), c->stream, EV0(c, pi), EV1(c, pi), a, im->d_rate,
                      use_massfx ? im->d_massfx : (unsigned long long *)nullptr);
            HIP_TRY(hipGetLastError());
            rc = CEL_OK;
        } else {
            rc = render_impl(im, src, CEL_RENDER_STRICT, nullptr, nullptr, im->d_rate);
        }
    } else {
        rc = cel_render_field(im, src, 0, nullptr, nullptr);
    }
    if (rc) return rc;
    const int B = im->B;
    const int64_t S = src->S, n = S * B;
    const int T = B * im->ntx * im->nty;
    int64_t total = 0;
    int64_t *d_off = nullptr;
    void *d_samp = nullptr;             // resident: int32 photon counts; a caller's buffer: doubles
    if (resident) {
        // patch boxes + offsets laid out on the device; only the total size comes back
        if (n + 1 > im->slay_cap) {
            HIP_TRY(hipStreamSynchronize(c->stream));
            if (im->d_sbox) (void)hipFree(im->d_sbox);
            if (im->d_soff) (void)hipFree(im->d_soff);
            if (im->d_snz) (void)hipFree(im->d_snz);
            if (im->d_ssum) (void)hipFree(im->d_ssum);
            if (im->d_nnz) (void)hipFree(im->d_nnz);
            if (im->d_nzmode) (void)hipFree(im->d_nzmode);
            if (im->d_nzoff) (void)hipFree(im->d_nzoff);
            im->d_nnz = nullptr; im->d_nzmode = nullptr; im->d_nzoff = nullptr;
            im->d_sbox = nullptr; im->d_soff = nullptr; im->d_snz = nullptr; im->d_ssum = nullptr; im->slay_cap = 0;
            int64_t cap = n + n / 4 + 64;
            HIP_TRY(hipMalloc((void **)&im->d_sbox, sizeof(int4) * cap));
            HIP_TRY(hipMalloc((void **)&im->d_snz, sizeof(int4) * cap));
            HIP_TRY(hipMalloc((void **)&im->d_soff, sizeof(int64_t) * cap));
            HIP_TRY(hipMalloc((void **)&im->d_ssum, sizeof(double) * cap));
            HIP_TRY(hipMalloc((void **)&im->d_nnz, sizeof(int) * cap));
            HIP_TRY(hipMalloc((void **)&im->d_nzmode, sizeof(int) * cap));
            HIP_TRY(hipMalloc((void **)&im->d_nzoff, sizeof(int64_t) * cap));
            if (im->d_btot) (void)hipFree(im->d_btot);
            im->d_btot = nullptr;
            HIP_TRY(hipMalloc((void **)&im->d_btot, sizeof(long long) * (size_t)(cap / 1024 + 2)));
            im->slay_cap = cap;
        }
        {
            const unsigned nblk = (unsigned)((n + 1023) / 1024);
            hipLaunchKernelGGL(k_samp_layout, dim3(nblk), dim3(1024), 0, c->stream, im->d_recs, S, B, im->d_sbox, im->d_soff, im->d_btot);
            hipLaunchKernelGGL(k_scan_totals, dim3(1), dim3(1024), 0, c->stream, im->d_btot, (int)nblk, im->d_soff + n);
            hipLaunchKernelGGL(k_scan_apply, dim3(nblk), dim3(1024), 0, c->stream, im->d_soff, n, (const long long *)im->d_btot);
        }
        HIP_TRY(hipMemcpyAsync(c->pinned + MAX_BANDS + 2, im->d_soff + n, sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        memcpy(&total, c->pinned + MAX_BANDS + 2, sizeof(total));
        if (total > im->samp_cap) {
            if (im->d_samp) (void)hipFree(im->d_samp);
            im->d_samp = nullptr; im->samp_cap = 0;
            int64_t cap = total + total / 8 + 1024;
            HIP_TRY(hipMalloc((void **)&im->d_samp, sizeof(int) * cap));
            im->samp_cap = cap;
        }
        d_off = im->d_soff;
        d_samp = im->d_samp;
        im->samp_S = S;
        im->samp_total = total;
    } else {
        // the caller's layout against the sources' own boxes (cel_source_boxes): a patch too small for its box would be
        // written past its end
        if (offsets[0] != 0) return fail(CEL_ERR_INVALID, "cel_photon_split: offsets[0] must be 0");
        if ((rc = host_boxes(im, src))) return rc;
        for (int64_t sidx = 0; sidx < S; sidx++)
            for (int b = 0; b < B; b++) {
                const int4 bx = im->h_boxes[(size_t)((int64_t)b * S + sidx)];
                const int64_t area = (im->h_status[(size_t)((int64_t)b * S + sidx)] > 0) ? (int64_t)(bx.y - bx.x) * (bx.w - bx.z) : 0;
                const int64_t i = sidx * B + b;
                if (offsets[i + 1] - offsets[i] != area)
                    return fail(CEL_ERR_INVALID, "cel_photon_split: offsets give source %lld band %d %lld values, its box has %lld pixels (cel_source_boxes)",
                                (long long)sidx, b, (long long)(offsets[i + 1] - offsets[i]), (long long)area);
            }
        total = offsets[n];
        if ((rc = scratch_get(c, 1, sizeof(int64_t) * (n + 1), (void **)&d_off))) return rc;
        HIP_TRY(hipMemcpyAsync(d_off, offsets, sizeof(int64_t) * (n + 1), hipMemcpyHostToDevice, c->stream));
        if (mem == CEL_DEVICE) d_samp = samp;
        else if ((rc = scratch_get(c, 3, sizeof(double) * (total > 0 ? total : 1), &d_samp))) return rc;
    }
    // resident + recurrence form: the kernel writes every interior pixel and reduces the photon
    // rectangles itself; otherwise zero the buffer and (resident) find the rectangles afterwards
    const bool fused_nz = resident && hw && n > 0;
    if (resident) im->ssum_valid = fused_nz;
    // photon lists: pixel coordinates are packed into 16 bits each
    const bool lists = fused_nz && im->W < 65536 && (im->win_y0 + im->H) < 65536 && c->nz_force != 2;
    if (resident) im->nz_valid = false;
    if (fused_nz) HIP_TRY(hipMemsetAsync(im->d_ssum, 0, sizeof(double) * n, c->stream));
    if (lists) HIP_TRY(hipMemsetAsync(im->d_nnz, 0, sizeof(int) * n, c->stream));
    if (fused_nz)
        hipLaunchKernelGGL(k_samp_prepare<int>, dim3((unsigned)((n + 4 * SAMP_PREP_PER_WAVE - 1) / (4 * SAMP_PREP_PER_WAVE))), dim3(256), 0, c->stream,
                           im->d_sbox, im->d_soff, im->d_samp, im->d_snz, n);
    else if (total > 0)
        HIP_TRY(hipMemsetAsync(d_samp, 0, (resident ? sizeof(int) : sizeof(double)) * total, c->stream));
    {
        SplitArgs a;
        a.bands = im->d_bands; a.recs = im->d_recs; a.lists = im->d_lists; a.tile_cnt = im->d_tile_cnt;
        a.tile_off = im->d_tile_off; a.nelec = im->d_nelec; a.offsets = d_off; a.samp = d_samp;
        a.partials = im->d_partials; a.S = S; a.capacity = im->lists_cap; a.B = B; a.H = im->H; a.W = im->W;
        a.ntx = im->ntx; a.nty = im->nty; a.TW = im->TW; a.TH = im->TH; a.seed = seed;
        a.win_y0 = im->win_y0; a.full_H = im->full_H;
        a.noise_y0 = im->noise_y0; a.noise_y1 = im->noise_y1;
        a.rate_img = full_rate ? full_rate : im->d_rate; a.tail_T = c->tail_T; a.nz = fused_nz ? im->d_snz : nullptr;
        a.strict = c->split_full ? 0 : 1;
        im->rate_in_lambda = full_rate != nullptr;
        a.order = (hw && tile_order_of(c, im)) ? im->d_order : nullptr;
        a.sums = fused_nz ? im->d_ssum : nullptr;
        a.nnz = lists ? im->d_nnz : nullptr;
        a.massfx = (use_massfx && hw && resident) ? im->d_massfx : nullptr;
        a.debug = c->debug;
        if (hw && (rc = scratch_get(c, 2, sizeof(double) * 2 * (size_t)T, (void **)&a.partials))) return rc;
        int pi = prof_begin(c, CEL_K_SPLIT);
        if (hw && resident) {
            if (im->nelec_u16) hipLaunchKernelGGL((k_photon_split_hw<int, unsigned short>), dim3(2 * T), dim3(64), 0, c->stream, a);
            else hipLaunchKernelGGL((k_photon_split_hw<int, int>), dim3(2 * T), dim3(64), 0, c->stream, a);
        } else if (hw) {
            if (im->nelec_u16) hipLaunchKernelGGL((k_photon_split_hw<double, unsigned short>), dim3(2 * T), dim3(64), 0, c->stream, a);
            else hipLaunchKernelGGL((k_photon_split_hw<double, int>), dim3(2 * T), dim3(64), 0, c->stream, a);
        }
        else if (resident) hipLaunchKernelGGL(k_photon_split<int>, dim3(T), dim3(64), 0, c->stream, a);
        else hipLaunchKernelGGL(k_photon_split<double>, dim3(T), dim3(64), 0, c->stream, a);
        prof_end(c, pi);
        hipLaunchKernelGGL(k_reduce, dim3(B), dim3(256), 0, c->stream, a.partials, (hw ? 2 : 1) * im->ntx * im->nty, im->d_llband,
                           (hw ? 2 : 1) * im->ntx * im->nty, 0, 1);
        if (resident && n > 0 && !fused_nz)   // where each patch's photons are: the conditional likelihoods evaluate only there
            hipLaunchKernelGGL(k_patch_nzbox<int>, dim3((unsigned)n), dim3(64), 0, c->stream, im->d_sbox, im->d_soff, (const int *)im->d_samp, im->d_snz);
    }
    if (lists)      // where each patch's photon list starts, and whether its likelihood is cheaper at the photons or densely
    {
        const unsigned nblk = (unsigned)((n + 1023) / 1024);
        hipLaunchKernelGGL(k_nz_layout, dim3(nblk), dim3(1024), 0, c->stream, (const int *)im->d_nnz, (const int4 *)im->d_snz,
                           (const int *)src->d_type, S, B, c->nz_force, c->nz_bias, im->d_nzoff, im->d_nzmode, im->d_btot);
        hipLaunchKernelGGL(k_scan_totals, dim3(1), dim3(1024), 0, c->stream, im->d_btot, (int)nblk, im->d_nzoff + n);
        hipLaunchKernelGGL(k_scan_apply, dim3(nblk), dim3(1024), 0, c->stream, im->d_nzoff, n, (const long long *)im->d_btot);
    }
    HIP_TRY(hipMemcpyAsync(c->pinned, im->d_llband, sizeof(double) * B, hipMemcpyDeviceToHost, c->stream));
    if (lists) HIP_TRY(hipMemcpyAsync(c->pinned + MAX_BANDS + 2, im->d_nzoff + n, sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
    if (resident) im->hsum_valid = false;
    if (fused_nz) {
        // what a Gibbs sweep asks for next (the photons per source: the flux and sky steps; the patch areas) rides back now:
        // cel_samples_fetch then needs no wait of its own, and the list compaction queued below runs under the host's work
        if (n + 1 > im->hsum_cap) {
            if (im->h_ssum) (void)hipHostFree(im->h_ssum);
            if (im->h_soff) (void)hipHostFree(im->h_soff);
            im->h_ssum = nullptr; im->h_soff = nullptr; im->hsum_cap = 0;
            const int64_t cap = n + n / 4 + 64;
            HIP_TRY(hipHostMalloc((void **)&im->h_ssum, sizeof(double) * cap, hipHostMallocDefault));
            HIP_TRY(hipHostMalloc((void **)&im->h_soff, sizeof(int64_t) * cap, hipHostMallocDefault));
            im->hsum_cap = cap;
        }
        HIP_TRY(hipMemcpyAsync(im->h_ssum, im->d_ssum, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(im->h_soff, im->d_soff, sizeof(int64_t) * (n + 1), hipMemcpyDeviceToHost, c->stream));
    }
    if (!resident && mem != CEL_DEVICE && total > 0)
        HIP_TRY(hipMemcpyAsync(samp, d_samp, sizeof(double) * total, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (fused_nz) im->hsum_valid = true;
    if (noise) for (int b = 0; b < B; b++) noise[b] = c->pinned[b];
    if (use_massfx && hw && resident) im->massfx_gen = src->gen;
    if (lists) {
        int64_t nn = 0;
        memcpy(&nn, c->pinned + MAX_BANDS + 2, sizeof(nn));
        if (nn > im->nzlist_cap) {
            if (im->d_nzlist) (void)hipFree(im->d_nzlist);
            im->d_nzlist = nullptr; im->nzlist_cap = 0;
            const int64_t cap = nn + nn / 4 + 1024;
            HIP_TRY(hipMalloc((void **)&im->d_nzlist, sizeof(NzEntry) * cap));
            im->nzlist_cap = cap;
        }
        // stream-ordered: whatever scores against these patches next runs behind it
        hipLaunchKernelGGL(k_nz_compact, dim3((unsigned)n), dim3(64), 0, c->stream, (const int4 *)im->d_sbox, (const int64_t *)im->d_soff,
                           (const int *)im->d_samp, (const int4 *)im->d_snz, (const int64_t *)im->d_nzoff, im->d_nzlist);
        HIP_TRY(hipGetLastError());
        im->nz_valid = true;
    }
    return CEL_OK;
}

int cel_samples_info(cel_images *im, int64_t *S, int64_t *total) {
    if (!im || !S || !total) return fail(CEL_ERR_INVALID, "cel_samples_info: null argument");
    *S = im->samp_S;
    *total = im->samp_total;
    return CEL_OK;
}

int cel_samples_fetch(cel_images *im, int32_t *boxes, int64_t *offsets, double *data, double *sums) {
    if (!im) return fail(CEL_ERR_INVALID, "cel_samples_fetch: null argument");
    if (im->samp_S <= 0) return fail(CEL_ERR_INVALID, "no resident photon split: call cel_photon_split with offsets = NULL first");
    cel_ctx *c = im->ctx;
    HIP_TRY(hipSetDevice(c->device));
    const int64_t n = im->samp_S * im->B;
    int rc = CEL_OK;
    if (!boxes && !data && im->hsum_valid && n + 1 <= im->hsum_cap) {        // host copies made by the split itself: no wait
        if (offsets) memcpy(offsets, im->h_soff, sizeof(int64_t) * (n + 1));
        if (sums) memcpy(sums, im->h_ssum, sizeof(double) * n);
        return CEL_OK;
    }
    if (boxes) {
        std::vector<int4> hb((size_t)n);
        if ((rc = copy_out(hb.data(), im->d_sbox, sizeof(int4) * n, CEL_HOST, c->stream))) return rc;
        for (int64_t i = 0; i < n; i++) {
            boxes[4 * i] = hb[i].z; boxes[4 * i + 1] = hb[i].w; boxes[4 * i + 2] = hb[i].x; boxes[4 * i + 3] = hb[i].y;
        }
    }
    if (offsets && (rc = copy_out(offsets, im->d_soff, sizeof(int64_t) * (n + 1), CEL_HOST, c->stream))) return rc;
    if (data && im->samp_total > 0) {           // int32 on the device, doubles across the ABI
        std::vector<int> hs((size_t)im->samp_total);
        if ((rc = copy_out(hs.data(), im->d_samp, sizeof(int) * im->samp_total, CEL_HOST, c->stream))) return rc;
        for (int64_t i = 0; i < im->samp_total; i++) data[i] = (double)hs[(size_t)i];
    }
    if (sums) {
        if (im->ssum_valid) {       // the split kernel summed them itself (exact: integer-valued)
            if ((rc = copy_out(sums, im->d_ssum, sizeof(double) * n, CEL_HOST, c->stream))) return rc;
        } else {
            double *d_sums = nullptr;
            if ((rc = scratch_get(c, 2, sizeof(double) * n, (void **)&d_sums))) return rc;
            hipLaunchKernelGGL(k_patch_sums<int>, dim3((unsigned)n), dim3(256), 0, c->stream, im->d_soff, (const int *)im->d_samp, d_sums);
            if ((rc = copy_out(sums, d_sums, sizeof(double) * n, CEL_HOST, c->stream))) return rc;
        }
    }
    return CEL_OK;
}

int cel_samples_photon_rects(cel_images *im, int32_t *rects) {
    if (!im || !rects) return fail(CEL_ERR_INVALID, "cel_samples_photon_rects: null argument");
    if (im->samp_S <= 0 || !im->d_snz)
        return fail(CEL_ERR_INVALID, "no resident photon split: call cel_photon_split with offsets = NULL first");
    cel_ctx *c = im->ctx;
    HIP_TRY(hipSetDevice(c->device));
    const int64_t n = im->samp_S * im->B;
    std::vector<int4> hb((size_t)n);
    int rc = copy_out(hb.data(), im->d_snz, sizeof(int4) * n, CEL_HOST, c->stream);
    if (rc) return rc;
    for (int64_t i = 0; i < n; i++) {
        const bool held = hb[i].y > hb[i].x && hb[i].w > hb[i].z;         // (the kernels mark an empty patch in two ways)
        rects[4 * i] = held ? hb[i].z : 0; rects[4 * i + 1] = held ? hb[i].w : 0;
        rects[4 * i + 2] = held ? hb[i].x : 0; rects[4 * i + 3] = held ? hb[i].y : 0;
    }
    return CEL_OK;
}

int cel_debug_binomial(cel_ctx *c, int64_t n, double p, uint64_t seed, int64_t N, int64_t *out) {
    if (!c || !out || N < 0 || n < 0) return fail(CEL_ERR_INVALID, "cel_debug_binomial: bad argument");
    if (N == 0) return CEL_OK;
    HIP_TRY(hipSetDevice(c->device));
    long long *d = nullptr;
    HIP_TRY(hipMalloc((void **)&d, sizeof(long long) * N));
    hipLaunchKernelGGL(k_binomial_draws, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, c->stream, (long long)n, p,
                       (unsigned long long)seed, N, d);
    int rc = copy_out(out, d, sizeof(long long) * N, CEL_HOST, c->stream);
    (void)hipFree(d);
    return rc;
}

// ---- E-step statistics -------------------------------------------------------------------------
int cel_estep_stats(cel_images *im, cel_sources *src, double *xtilde, double *mass, double *noise) {
    if (!im || !src) return fail(CEL_ERR_INVALID, "cel_estep_stats: null argument");
    if (!im->have_nelec) return fail(CEL_ERR_INVALID, "cel_estep_stats needs cel_images_set_nelec first");
    cel_ctx *c = im->ctx;
    // lambda for exactly these sources must be resident (this also runs k_prep for them)
    int rc = render_impl(im, src, CEL_RENDER_KEEP_LISTS, nullptr, nullptr, nullptr);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(c->device));
    const int B = im->B;
    const int64_t S = src->S;
    const int nblk = im->ntx * im->nty;                 // d_partials holds B * nblk doubles
    im->partials_gen = 0;                               // ... of this call's sky term from here on
    double *d_x = nullptr, *d_m = nullptr, *d_part = nullptr;
    // recurrence evaluator on the 32 x 64 layout: the tile-walking form (the render above left the
    // lists and boxes of exactly these sources on the device); CEL_OPT_DEBUG bit 64 keeps the per-source form
    const bool tiles = (c->variant != 0) && (im->TW == HW_TW) && !(c->debug & 64) && S > 0;
    std::vector<double> hx((size_t)(S * B)), hm((size_t)(S * B));
    hipError_t e;
#define ES_TRY(expr)                                                                     \
    do {                                                                                 \
        e = (expr);                                                                      \
        if (e != hipSuccess) { rc = fail(CEL_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e)); goto done; } \
    } while (0)
    if (S > 0) {
        // the context's scratch arena (slots 4/5), not a hipMalloc per call: EM iterates this
        if ((rc = scratch_get(c, 4, sizeof(double) * S * B, (void **)&d_x)) ||
            (rc = scratch_get(c, 5, sizeof(double) * S * B, (void **)&d_m))) goto done;
        if (tiles && (rc = scratch_get(c, 6, sizeof(double) * 2 * (size_t)im->lists_cap, (void **)&d_part))) goto done;
        int pi = prof_begin(c, CEL_K_ESTEP);
        if (c->variant == 0)
            hipLaunchKernelGGL(k_estep_src, dim3((unsigned)(S * B)), dim3(256), 0, c->stream, im->d_bands, B, im->H, im->W,
                               S, im->d_recs, im->d_nelec, im->d_lambda, d_x, d_m);
        else if (tiles) {
            // walk the render tiles: nelec / lambda are read once, every list entry gets its pair of sums,
            // the gather adds a source's entries in tile order
            EstepArgs ea;
            ea.bands = im->d_bands; ea.recs = im->d_recs; ea.lists = im->d_lists; ea.tile_cnt = im->d_tile_cnt;
            ea.tile_off = im->d_tile_off; ea.order = tile_order_of(c, im) ? im->d_order : nullptr;
            ea.nelec = im->d_nelec; ea.lambda = im->d_lambda; ea.partial = d_part; ea.noise_partial = im->d_partials;
            ea.S = S; ea.capacity = im->lists_cap; ea.B = B; ea.H = im->H; ea.W = im->W; ea.ntx = im->ntx; ea.nty = im->nty;
            ea.tail_T = c->tail_T;
            hipLaunchKernelGGL(k_estep_tiles, dim3((unsigned)(B * nblk)), dim3(64), 0, c->stream, ea);
            hipLaunchKernelGGL(k_estep_gather, dim3((unsigned)((S * B + 255) / 256)), dim3(256), 0, c->stream, im->d_boxes, im->d_kind,
                               S, B, im->ntx, im->nty, im->d_tile_cnt, im->d_tile_nstar, im->d_tile_off, im->d_lists, im->lists_cap,
                               d_part, d_x, d_m);
        } else
            hipLaunchKernelGGL(k_estep_src_hw, dim3((unsigned)(S * B)), dim3(64), 0, c->stream, im->d_bands, B, im->H, im->W,
                               S, im->d_recs, im->d_nelec, im->d_lambda, c->tail_T, d_x, d_m);
        prof_end(c, pi);
        ES_TRY(hipMemcpyAsync(hx.data(), d_x, sizeof(double) * S * B, hipMemcpyDeviceToHost, c->stream));
        ES_TRY(hipMemcpyAsync(hm.data(), d_m, sizeof(double) * S * B, hipMemcpyDeviceToHost, c->stream));
    }
    if (!tiles)      // the tile walk has left the sky term's per-tile sums in d_partials
        hipLaunchKernelGGL(k_estep_noise, dim3(B * nblk), dim3(256), 0, c->stream, im->d_bands, (int64_t)im->H * im->W, nblk,
                           im->d_nelec, im->d_lambda, im->d_partials);
    hipLaunchKernelGGL(k_reduce, dim3(B), dim3(256), 0, c->stream, im->d_partials, nblk, im->d_llband, nblk, 0, 1);
    ES_TRY(hipMemcpyAsync(c->pinned, im->d_llband, sizeof(double) * B, hipMemcpyDeviceToHost, c->stream));
    ES_TRY(hipGetLastError());
    ES_TRY(hipStreamSynchronize(c->stream));
    if (xtilde) memcpy(xtilde, hx.data(), sizeof(double) * S * B);
    if (mass) memcpy(mass, hm.data(), sizeof(double) * S * B);
    if (noise) for (int b = 0; b < B; b++) noise[b] = c->pinned[b];
#undef ES_TRY
done:
    (void)hipStreamSynchronize(c->stream);
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

int cel_mog_loglike(cel_ctx *c, const double *x, int64_t N, const double *means, const double *icovs,
                    const double *logw, int K, double *out, int mem) {
    if (!c || !x || !means || !icovs || !logw || !out) return fail(CEL_ERR_INVALID, "cel_mog_loglike: null argument");
    if (N < 0 || K < 1) return fail(CEL_ERR_INVALID, "cel_mog_loglike: bad sizes");
    if (N == 0) return CEL_OK;
    HIP_TRY(hipSetDevice(c->device));
    std::vector<double> comp((size_t)K * 6);
    for (int k = 0; k < K; k++) {
        comp[6 * k + 0] = logw[k];
        comp[6 * k + 1] = means[2 * k];
        comp[6 * k + 2] = means[2 * k + 1];
        comp[6 * k + 3] = icovs[4 * k];
        comp[6 * k + 4] = icovs[4 * k + 1] + icovs[4 * k + 2];     // the einsum keeps both off-diagonal terms (mog.py:15-16)
        comp[6 * k + 5] = icovs[4 * k + 3];
    }
    double *d_comp = nullptr, *d_x = nullptr, *d_o = nullptr;
    int rc;
    if ((rc = scratch_get(c, 4, sizeof(double) * 6 * K, (void **)&d_comp))) return rc;
    HIP_TRY(hipMemcpyAsync(d_comp, comp.data(), sizeof(double) * 6 * K, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));      // comp is a local
    if (mem == CEL_DEVICE) {
        d_x = const_cast<double *>(x);
        d_o = out;
    } else {
        if ((rc = scratch_get(c, 5, sizeof(double) * 2 * N, (void **)&d_x)) ||
            (rc = scratch_get(c, 6, sizeof(double) * N, (void **)&d_o))) return rc;
        HIP_TRY(hipMemcpyAsync(d_x, x, sizeof(double) * 2 * N, hipMemcpyHostToDevice, c->stream));
    }
    int pi = prof_begin(c, CEL_K_GMM);
    hipLaunchKernelGGL(k_mog_ll, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, c->stream, d_x, N, d_comp, K, d_o);
    prof_end(c, pi);
    HIP_TRY(hipGetLastError());
    if (mem != CEL_DEVICE) HIP_TRY(hipMemcpyAsync(out, d_o, sizeof(double) * N, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return CEL_OK;
}

int cel_galaxy_mixture_params(cel_ctx *c, int64_t N, const double *W, const double *v_s, const double *image_ws,
                              const double *image_means, const double *image_covars, int K_psf, const double *amp,
                              const double *sigs, int J, double *weights, double *means, double *covars) {
    if (!c || !W || !v_s || !image_ws || !image_means || !image_covars || !amp || !sigs || !weights || !means || !covars)
        return fail(CEL_ERR_INVALID, "cel_galaxy_mixture_params: null argument");
    if (N < 0 || K_psf < 1 || J < 1 || K_psf > 4096 || J > 4096) return fail(CEL_ERR_INVALID, "cel_galaxy_mixture_params: bad sizes");
    if (N == 0) return CEL_OK;
    HIP_TRY(hipSetDevice(c->device));
    const int64_t K = (int64_t)K_psf * J, n = N * K;
    // inputs in one upload: W (4N), v_s (2N), ws (Kp), means (2Kp), covars (4Kp), amp (J), sigs (J)
    const size_t nin = (size_t)(6 * N + 7 * K_psf + 2 * J);
    std::vector<double> hin(nin);
    double *q = hin.data();
    memcpy(q, W, sizeof(double) * 4 * N); q += 4 * N;
    memcpy(q, v_s, sizeof(double) * 2 * N); q += 2 * N;
    memcpy(q, image_ws, sizeof(double) * K_psf); q += K_psf;
    memcpy(q, image_means, sizeof(double) * 2 * K_psf); q += 2 * K_psf;
    memcpy(q, image_covars, sizeof(double) * 4 * K_psf); q += 4 * K_psf;
    memcpy(q, amp, sizeof(double) * J); q += J;
    memcpy(q, sigs, sizeof(double) * J);
    double *d_in = nullptr, *d_out = nullptr;
    int rc;
    if ((rc = scratch_get(c, 4, sizeof(double) * nin, (void **)&d_in)) ||
        (rc = scratch_get(c, 5, sizeof(double) * 7 * n, (void **)&d_out))) return rc;
    HIP_TRY(hipMemcpyAsync(d_in, hin.data(), sizeof(double) * nin, hipMemcpyHostToDevice, c->stream));
    const double *dW = d_in, *dv = dW + 4 * N, *dws = dv + 2 * N, *dmu = dws + K_psf, *dcv = dmu + 2 * K_psf,
                 *damp = dcv + 4 * K_psf, *dsig = damp + J;
    hipLaunchKernelGGL(k_mixture_params, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, N, dW, dv, dws, dmu, dcv,
                       K_psf, damp, dsig, J, d_out, d_out + n, d_out + 3 * n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(weights, d_out, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(means, d_out + n, sizeof(double) * 2 * n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(covars, d_out + 3 * n, sizeof(double) * 4 * n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return CEL_OK;
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
    c->prof.head = 0;
    c->prof.count = 0;
    for (int k = 0; k < CEL_K_COUNT; k++) { c->prof.sum_ms[k] = 0.0; c->prof.n[k] = 0; c->prof.seen[k] = 0; }
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
