// k_slice.h -- lock-step slice sampling of source locations, state machine on the device
//
// Source.resample_location (CelestePy/sources.py:308-319) = slicesample(u, location_likelihood, ...)
// (CelestePy/util/infer/slicesample.py:89-227) with the options that call uses: component-wise,
// no stepping out.  One chain per source; every round each unfinished chain names the point it
// needs next (written into the proposal set's radec), cel_patch_loglik's kernels score all of them against
// the resident photon patches, and k_slice_step advances the chains and names their next points.
// Nothing but one counter crosses PCIe per round.
//
// The arithmetic and the random streams are those of the host engine
// (desi-mcmc_amd/util/infer/slicesample.py: one SplitMix64 stream per chain keyed by (seed, chain id),
// drawn in the reference's order), so a chain takes the same trajectory on either engine.
#pragma once
#include "device_common.h"
#include "k_slice_state.h"

// ---- Gamma variates on per-element counter-based streams -----------------------------------------------------------------
// celeste_mcmc.gamma_by_stream on the device: the flux conditionals' Gamma(a_0 + photons) draws of Source.resample_fluxes
// (CelestePy/sources.py:341-345), one per (source, band), each from its own SplitMix64 streams keyed by (seed, element) --
// uniforms on `key`, normals (Box-Muller) on `nkey` -- exactly as util/infer/slicesample.ChainStreams hands them out, so an
// element's draw does not depend on which others are drawn beside it, by which rank, or in what order.  Marsaglia & Tsang
// (2000) with the squeeze test, the same decisions in the same order as the numpy version; the values agree with it to
// rounding (cos and log are the device's).  50 000 elements: ~10 us; on the host the draws took 1.7 ms on a good day.
// one standard Gamma(a) variate of element i's streams (what k_gamma_streams writes to out[i])
__device__ inline double gamma_stream_draw(double ai, unsigned long long seed, int64_t i) {
#pragma clang fp contract(off)
    const unsigned long long key = sl_mix(seed ^ ((unsigned long long)i * 0xD1342543DE82EF95ull));
    const unsigned long long nkey = sl_mix(key ^ 0xA0761D6478BD642Full);
    unsigned long long cnt = 0ull, ncnt = 0ull;
    const bool boost = ai < 1.0;
    const double aa = boost ? ai + 1.0 : ai;
    const double d = aa - 1.0 / 3.0;
    const double c = 1.0 / sqrt(9.0 * d);
    double res = 0.0;
    for (int guard = 0; guard < 1000; guard++) {
        const double u1 = ((double)(sl_mix(nkey + ncnt * 0x9E3779B97F4A7C15ull) >> 11) + 0.5) * (1.0 / 9007199254740992.0);
        const double u2 = ((double)(sl_mix(nkey + (ncnt + 1ull) * 0x9E3779B97F4A7C15ull) >> 11) + 0.5) * (1.0 / 9007199254740992.0);
        ncnt += 2ull;
        const double x = sqrt(-2.0 * log(u1)) * cos(2.0 * PI_D * u2);
        const double u = ((double)(sl_mix(key + cnt * 0x9E3779B97F4A7C15ull) >> 11) + 0.5) * (1.0 / 9007199254740992.0);
        cnt += 1ull;
        const double t = 1.0 + c * x;
        const double v = t * t * t;
        const double x2 = x * x;
        bool ok = (v > 0.0) && (u < 1.0 - 0.0331 * x2 * x2);
        if (!ok && v > 0.0) ok = log(u) < 0.5 * x2 + d * (1.0 - v + log(v));
        if (ok) { res = d * v; break; }
    }
    if (boost) {
        const double u = ((double)(sl_mix(key + cnt * 0x9E3779B97F4A7C15ull) >> 11) + 0.5) * (1.0 / 9007199254740992.0);
        res *= pow(u, 1.0 / ai);
    }
    return res;
}

__global__ void __launch_bounds__(256)
k_gamma_streams(int64_t n, const double *__restrict__ a, unsigned long long seed, double *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = gamma_stream_draw(a[i], seed, i);
}

// ---- the flux conditionals of a whole catalogue on the device (round 6) -------------------------------------------------------
// Source.resample_fluxes (CelestePy/sources.py:321-349) for every source and band letter at once: element (s, L) of the (S, 5)
// flux table gets  Gamma(a0 + photons of s in the images of letter L) / (b0 + sum over those images of mass * kappa / calib),
// the Gamma variate from element s * 5 + L's own streams (gamma_stream_draw), every operation in the order the host form
// (celeste_mcmc.ModelGibbs.resample_fluxes) takes it -- the values are the host's bit for bit.  The photons (the resident
// split's sums), the stamp masses (cel_stamp_mass's buffer) and the patch layout are on the device already; the new expected
// counts go straight into the catalogue's device array.  A source without any patch (`active` 0) is left alone, as the
// reference leaves it (sources.py:243).
__global__ void __launch_bounds__(256)
k_flux_step(int64_t S, int B, const double *__restrict__ sums /* [S][B] */, const double *__restrict__ mass /* [S][B] */,
            const int64_t *__restrict__ soff /* [S*B + 1]: a patch exists where soff grows */, const int *__restrict__ letter /* [B]: 0..4 */,
            const double *__restrict__ ratio /* [B]: kappa / calib, as the host divides them */, const double *__restrict__ calib,
            const double *__restrict__ kappa, double a0, double b0, unsigned long long seed,
            double *__restrict__ counts /* [S][B]: the catalogue's expected photons, rewritten for active sources */,
            double *__restrict__ flux_new /* [S][5] */, int *__restrict__ active /* [S] */) {
#pragma clang fp contract(off)
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S * 5) return;
    const int64_t s = i / 5;
    const int L = (int)(i - s * 5);
    bool act = false;
    double cnt = 0.0, psf = 0.0;
    for (int b = 0; b < B; b++) {
        const int64_t j = s * B + b;
        const bool has = soff[j + 1] > soff[j];
        act = act || has;
        if (letter[b] == L) {
            cnt += sums[j];                                   // band_counts[:, L] += sums[:, b]
            const double m = mass[j] * (has ? 1.0 : 0.0);     // mass = m * has_patch
            psf += m * ratio[b];                              // psf_sums[:, L] += mass[:, b] * (kappa[b] / calib[b])
        }
    }
    const double a_n = a0 + cnt;
    const double g = gamma_stream_draw(a_n, seed, i);
    const double fnew = g * (1.0 / (b0 + psf));
    flux_new[i] = fnew;
    if (L == 0) active[s] = act ? 1 : 0;
    if (act) {
        for (int b = 0; b < B; b++)
            if (letter[b] == L) counts[s * B + b] = fnew / calib[b] * kappa[b];       // ModelGibbs.counts: fl / calib * kappa
    }
}

__global__ void __launch_bounds__(256)
k_slice_init(SliceState st, int64_t S, const double *__restrict__ radec, const int *__restrict__ chain_ids,
             const int64_t *__restrict__ soff, int B, unsigned long long seed, double sigma, int *__restrict__ owner) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    const unsigned long long id = (unsigned long long)(chain_ids ? chain_ids[s] : (int)s);
    st.key[s] = sl_mix(seed ^ (id * 0xD1342543DE82EF95ull));
    st.count[s] = 0ull;
    st.x[2 * s] = radec[2 * s];
    st.x[2 * s + 1] = radec[2 * s + 1];
    st.steps[s] = 0;
    st.new_llh[s] = NAN;
    // a random order of the two axes: argsort (stable) of two uniforms (slicesample.py:214-221)
    const double u0 = sl_uniform(st, s), u1 = sl_uniform(st, s);
    st.first[s] = (u1 < u0) ? 1 : 0;
    st.kdir[s] = 0;
    // a source without any sample patch is left alone (the reference asserts there, sources.py:243); so is one
    // whose chain id is negative: it is another rank's to update (one chain dealt over several GPUs)
    const bool has_patch = soff[(s + 1) * B] > soff[s * B] && !(chain_ids && chain_ids[s] < 0);
    if (has_patch) sl_start_direction(st, s, sigma);
    else st.phase[s] = SL_FINAL;
    owner[s] = has_patch ? (int)s : -1;
}

// Flags of a call (ints): [0] / [10] chains still running after the last even / odd round, [1] error bits, [2] likelihood
// evaluations so far, [3] rounds that had a chain to score.  The host queues rounds in batches and reads the flags once per
// batch, so the device counts: round r's consume kernel adds its running chains to slot r & 1, and -- the other slot holding
// the complete count of round r - 1 -- books round r as a round with work when that count is positive, then clears it for
// round r + 1.

// the first round of a call: every chain's first point
__global__ void __launch_bounds__(256)
k_slice_propose(SliceState st, int64_t S, double *__restrict__ prop_radec, int *__restrict__ owner, int *__restrict__ flags) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s == 0) { flags[3] += 1; flags[0] = 0; flags[10] = 0; }
    if (s >= S) return;
    sl_propose_chain(st, s, prop_radec, owner);
}

// Consume round `round`'s log-likelihoods (per (chain, band), summed in band order as the host sums them) and name the
// point every chain that is still running needs next: one kernel per round instead of a consume and a propose (a chain's
// next point depends on its own state only).
// Launch: 64 chains per block; with pa.recs, 64 * B threads -- the block's first wave runs the chains, then every thread
// writes the record of one (chain, band) of the points just named.
__global__ void __launch_bounds__(1024)
k_slice_step(SliceState st, int64_t S, int B, int nparts /* blocks per (chain, band) job: their partial sums are added first, in order */,
             const double *__restrict__ ll_pb, double sigma, int *__restrict__ flags, int round,
             double *__restrict__ prop_radec, int *__restrict__ owner,
             int zero_live /* the last round of a batch: flags[4], flags[5] -- the live lists' lengths, which this batch's likelihood launches
                              have read -- are cleared for k_slice_live_jobs, instead of a fill launch of their own (13 per sweep) */,
             PrepArgs pa /* pa.recs != nullptr: the named point's records (k_prep's own arithmetic, prep_one) are written here too --
                            the round's k_prep launch, 7 us of kernel and as much of queue, is gone */) {
#pragma clang fp contract(off)
    __shared__ double s_ra[64], s_dec[64];
    __shared__ int s_own[64];
    const bool chain_thread = threadIdx.x < 64;
    const int64_t s = chain_thread ? (int64_t)blockIdx.x * 64 + threadIdx.x : S;        // (the other waves: no chain)
    int *const n_active = flags + ((round & 1) ? 10 : 0);
    int *const err = flags + 1;
    if (s == 0 && zero_live) { flags[4] = 0; flags[5] = 0; }
    if (s == 0 && round > 0) {
        int *const prev = flags + ((round & 1) ? 0 : 10);        // complete: round - 1's kernel has finished
        if (*prev > 0) flags[3] += 1;
        *prev = 0;                                               // round + 1 adds here
    }
    bool active = false, scored = false;
    if (s < S) scored = sl_consume_chain(st, s, B, nparts, ll_pb, sigma, err, active);
    const unsigned long long m = __ballot(active), me = __ballot(scored);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(n_active, __popcll(m));
    if ((threadIdx.x & 63) == 0 && me) atomicAdd(err + 1, __popcll(me));         // evaluations so far (int: < 2^31 per call)
    if (chain_thread) s_own[threadIdx.x] = -1;
    if (s < S) {
        sl_propose_chain(st, s, prop_radec, owner);                              // the next round's point (a finished chain retires)
        s_ra[threadIdx.x] = prop_radec[2 * s]; s_dec[threadIdx.x] = prop_radec[2 * s + 1]; s_own[threadIdx.x] = owner[s];
    }
    if (!pa.recs) return;
    __syncthreads();
    const int ci = threadIdx.x / pa.B, b = threadIdx.x - ci * pa.B;             // one (chain, band) per thread
    const int64_t sc = (int64_t)blockIdx.x * 64 + ci;
    if (ci < 64 && sc < S && s_own[ci] >= 0)
        prep_one(pa.bands, b, sc, (int64_t)b * pa.S + sc, pa.B, pa.H, pa.W, pa.win_y0, pa.win_h, pa.type, s_ra[ci], s_dec[ci], pa.counts, pa.shape,
                 pa.rsq_gal, pa.recs, pa.boxes, pa.kind, pa.status, pa.nobox);
}

// the (chain, band) jobs of the chains that are still running, in no particular order (they all start at once:
// the list is only built when it is shorter than the GPU has wave slots).
// Entries are the likelihood kernels' block descriptors job << 3 | part << 1 | dealt: a long job -- or, when few chains
// are left and a round lasts as long as its longest block, every job (deal_all) -- is dealt to PLL_PARTS blocks.  With
// photon lists: two lists, the jobs scored densely and the jobs scored at their photons, each for its kernel.  Built
// behind a batch of rounds for the next one, so that the counts ride back with the batch's flags and the next launches
// are exactly as large as their lists.
__global__ void __launch_bounds__(256)
k_slice_live_jobs(SliceState st, int64_t S, int B, int *__restrict__ list, int *__restrict__ count,
                  const int *__restrict__ nzmode, int *__restrict__ list_nz, int *__restrict__ count_nz,
                  const int *__restrict__ nnz, const int4 *__restrict__ nzbox, int deal_all,
                  int *__restrict__ need = nullptr /* per chain, zeroed by the caller: the blocks these lists hold for it (SliceFuse) */,
                  int *__restrict__ running = nullptr /* zeroed by the caller: chains still running (the fused rounds keep no per-round count) */) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool alive = i < S * B && st.phase[i / B] != SL_FINAL;
    if (running) {
        const unsigned long long rm = __ballot(alive && (i % B) == 0);
        if ((threadIdx.x & 63) == 0 && rm) atomicAdd(running, __popcll(rm));
    }
    bool nz = false, deal = deal_all != 0;
    if (alive) {
        nz = nzmode && nzmode[i];
        if (!deal) {
            if (nz) deal = nnz[i] > 2048;               // NZ_SPLIT_PHOTONS (k_patch_ll.h)
            else {
                const int4 q = nzbox[i];
                deal = (q.y > q.x && q.w > q.z) && (long long)((q.y - q.x + 31) / 32) * ((q.w - q.z + 63) / 64) > 6;     // PLL_SPLIT_CHUNKS
            }
        }
    }
    const int per = deal ? 4 : 1;
    const int at_d = wave_reserve(count, (alive && !nz) ? per : 0);
    const int at_n = nzmode ? wave_reserve(count_nz, (alive && nz) ? per : 0) : 0;
    if (!alive) return;
    if (need) atomicAdd(&need[i / B], per);
    int *dst = nz ? list_nz : list;
    const int at = nz ? at_n : at_d;
    for (int part = 0; part < per; part++) dst[at + part] = (int)(i << 3) | (part << 1) | (deal ? 1 : 0);
}

// algorithmic HBM bytes of a finished call (what bench.py prices the location step's kernel against): a chain that
// ran made 1 + steps[s] evaluations, each of which reads, per band, one 128-B record and the 4-B photon counts of
// the rectangle that holds the source's photons (k_patch_nzbox)
__global__ void __launch_bounds__(256)
k_slice_bytes(SliceState st, int64_t S, int B, const int4 *__restrict__ nzbox, unsigned long long *__restrict__ bytes,
              const int *__restrict__ nzmode, const int64_t *__restrict__ nzoff,
              int *__restrict__ evals_rounds = nullptr /* [0] += evaluations of the call, [1] = max over chains = rounds with work (the fused rounds count neither as they go) */) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long v = 0ull;
    if (evals_rounds) {
        int ev = (s < S && st.new_llh[s] == st.new_llh[s]) ? 1 + st.steps[s] : 0, mx = ev;
        for (int o = 32; o > 0; o >>= 1) { ev += __shfl_xor(ev, o); mx = max(mx, __shfl_xor(mx, o)); }
        if ((threadIdx.x & 63) == 0 && ev) { atomicAdd(evals_rounds, ev); atomicMax(evals_rounds + 1, mx); }
    }
    if (s < S && st.new_llh[s] == st.new_llh[s]) {          // NaN: the chain never ran
        unsigned long long per = 0ull;
        for (int b = 0; b < B; b++) {
            const int4 q = nzbox[s * B + b];
            const long long area = (q.y > q.x && q.w > q.z) ? (long long)(q.y - q.x) * (q.w - q.z) : 0;
            // a patch scored at its photons reads its list (8 B per photon-holding pixel) instead of the rectangle (4 B per pixel)
            if (nzmode && nzmode[s * B + b]) per += 8ull * (unsigned long long)(nzoff[s * B + b + 1] - nzoff[s * B + b]) + 128ull;
            else per += 4ull * (unsigned long long)area + 128ull;
        }
        v = per * (unsigned long long)(1 + st.steps[s]);
    }
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(bytes, v);
}
