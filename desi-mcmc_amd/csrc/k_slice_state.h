// k_slice_state.h -- the device slice sampler's per-chain state and the step of ONE chain (consume a round's log-likelihoods,
// name the next point, write its records): shared by k_slice_step (k_slice.h: one launch per round for every chain) and by the
// likelihood kernel itself (k_patch_ll_nz, k_patch_ll.h): the block that finishes a chain's LAST job of a round steps the chain
// there and then (SliceFuse below), so that a round of the location step is one launch instead of three.
#pragma once
#include "device_common.h"
#include "k_prep_bin.h"

#define SL_LEVEL 0
#define SL_SHRINK 4
#define SL_FINAL 7

struct SliceState {            // SoA over chains
    unsigned long long *key, *count;
    double *x;                 // current location (ra, dec), 2 per chain
    double *x0;                // location the current direction started from, 2 per chain
    double *lower, *upper, *log_u, *llh_s, *new_z, *new_llh;
    int *phase, *kdir, *first; // first = axis of the chain's first direction (0 or 1)
    int *steps;                // shrink steps taken (diagnostic)
};

__device__ inline unsigned long long sl_mix(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull;
    unsigned long long z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__device__ inline double sl_uniform(const SliceState &st, int64_t s) {
    const unsigned long long c = st.count[s];
    st.count[s] = c + 1ull;
    const unsigned long long z = sl_mix(st.key[s] + c * 0x9E3779B97F4A7C15ull);
    return ((double)(z >> 11) + 0.5) * (1.0 / 9007199254740992.0);
}

// slicesample.py:142-146: the interval about the current point and the random part of the level
__device__ inline void sl_start_direction(const SliceState &st, int64_t s, double sigma) {
#pragma clang fp contract(off)
    st.x0[2 * s] = st.x[2 * s];
    st.x0[2 * s + 1] = st.x[2 * s + 1];
    const double up = sigma * sl_uniform(st, s);
    st.upper[s] = up;
    st.lower[s] = up - sigma;
    st.log_u[s] = log(sl_uniform(st, s));
    st.phase[s] = SL_LEVEL;
}

// the point an unfinished chain needs next -> the proposal set's radec; owner[s] = -1 retires a chain
__device__ __forceinline__ void sl_propose_chain(const SliceState &st, int64_t s, double *__restrict__ prop_radec, int *__restrict__ owner) {
#pragma clang fp contract(off)
    const int ph = st.phase[s];
    if (ph == SL_FINAL) { owner[s] = -1; return; }
    double z = 0.0;
    if (ph == SL_SHRINK) {
        z = (st.upper[s] - st.lower[s]) * sl_uniform(st, s) + st.lower[s];      // slicesample.py:172
        st.new_z[s] = z;
        st.steps[s] += 1;
    }
    const int axis = st.kdir[s] == 0 ? st.first[s] : 1 - st.first[s];
    const double d0 = axis == 0 ? 1.0 : 0.0, d1 = axis == 1 ? 1.0 : 0.0;
    prop_radec[2 * s] = st.x0[2 * s] + z * d0;
    prop_radec[2 * s + 1] = st.x0[2 * s + 1] + z * d1;
    owner[s] = (int)s;
}


// Consume one round's log-likelihoods of chain s (per (chain, band) slot sums added in slot order, bands in band order: the
// order the host engine adds them in) and advance the chain (slicesample.py:146-203).  -> the chain was scored this round;
// `active` = it still runs.  One thread per chain.
__device__ __forceinline__ bool sl_consume_chain(const SliceState &st, int64_t s, int B, int nparts, const double *__restrict__ ll_pb,
                                                 double sigma, int *__restrict__ err, bool &active, bool coherent = false) {
#pragma clang fp contract(off)
    active = false;
    const int ph = st.phase[s];
    if (ph == SL_FINAL) return false;
    double v = 0.0;
    for (int b = 0; b < B; b++) {
        const double *q = ll_pb + (s * B + b) * nparts;
        // coherent: the slots were written by OTHER blocks of the launch that is still running (possibly on another XCD, whose L2
        // this one does not snoop): agent-scope loads
        double x = coherent ? __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : q[0];
        for (int k = 1; k < nparts; k++) x += coherent ? __hip_atomic_load(q + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : q[k];
        v += x;
    }
    if (ph == SL_LEVEL) {
        st.llh_s[s] = st.log_u[s] + v;                                   // slicesample.py:146
        st.phase[s] = SL_SHRINK;
    } else {
        if (v != v) atomicOr(err, 1);                                    // "Slice sampler got a NaN"
        const double z = st.new_z[s];
        if (v > st.llh_s[s]) {                                           // accepted (:177; no doubling, nothing to test)
            st.new_llh[s] = v;
            const int axis = st.kdir[s] == 0 ? st.first[s] : 1 - st.first[s];
            const double d0 = axis == 0 ? 1.0 : 0.0, d1 = axis == 1 ? 1.0 : 0.0;
            st.x[2 * s] = st.x0[2 * s] + z * d0;                        // :203
            st.x[2 * s + 1] = st.x0[2 * s + 1] + z * d1;
            const int k = st.kdir[s] + 1;
            st.kdir[s] = k;
            if (k >= 2) st.phase[s] = SL_FINAL;
            else {
                // the second axis starts where the first ended: its level needs the log-likelihood
                // of a point that has just been scored (the reference evaluates it again and gets
                // the same number), so the chain goes straight to shrinking
                sl_start_direction(st, s, sigma);
                st.llh_s[s] = st.log_u[s] + v;
                st.phase[s] = SL_SHRINK;
            }
        } else if (z < 0.0) {
            st.lower[s] = z;
        } else if (z > 0.0) {
            st.upper[s] = z;
        } else {
            atomicOr(err, 2);                                            // "Slice sampler shrank to zero!"
            st.phase[s] = SL_FINAL;
        }
    }
    active = st.phase[s] != SL_FINAL;
    return true;
}

// ---- the step fused into the likelihood kernel (round 6) -------------------------------------------------------------------
// A round of the location step used to be three dependent launches: the likelihood kernel, k_slice_step (11 us: every chain's
// consume + propose + records), and their two gaps -- 25 us per round beyond the likelihoods, 52 rounds per sweep.  A chain's
// step needs nothing but the chain's OWN jobs of the round, so the block that finishes the chain's last job does it: every block
// of a chain draws a ticket from the chain's counter when its slot sums are out; the block that draws ticket need[s] - 1 --
// whichever it is -- resets the counter, adds the slots in their fixed order and runs the step: lane 0 the chain, lanes 0..B-1
// the records of the point it named.  The hand-off carries no fence: 50 000 blocks per round each writing back and invalidating
// their XCD's L2 (what an agent-scope release / acquire pair costs on this chip) took the round from 0.5 to 1.3 ms.  Instead
// the slot sums THEMSELVES travel at agent scope -- sl_put: relaxed atomic stores (write-through), drained (vmcnt(0)) before the
// relaxed fetch_add; the stepper reads them with relaxed agent-scope loads -- and everything else the stepper touches (the
// chain's state, its records) is written by one block per round and read in later launches only.  The arithmetic, the random stream and the order of every sum are k_slice_step's: a chain
// takes the same trajectory, bit for bit, fused or not (tests/test_gibbs.py).  Nobody waits for anybody: a launch ends when
// its last block has ended, as before.
struct SliceFuse {
    int *tick;                 // per chain: blocks of this round that have finished; nullptr: no fused step (every other caller)
    const int *need;           // per chain: the blocks the launch's list holds for it
    SliceState st;
    const double *ll_pb;       // the slots of every (chain, band) job, nparts each
    int nparts, B;
    double sigma;
    int *flags;                // [1] error bits
    double *prop_radec;
    int *owner;
    PrepArgs pa;
};

// a slot sum of a likelihood block: plain for every caller but the fused rounds
__device__ __forceinline__ void sl_put(double *q, double v, bool coherent) {
    if (coherent) __hip_atomic_store(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *q = v;
}

// all 64 lanes of the block call it after their slot stores (sl_put, coherent); returns in every lane
__device__ __forceinline__ void sl_fused_step(const SliceFuse &fz, int64_t p, int lane) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the wave's slot stores have been acknowledged
    int last = 0;
    if (lane == 0) {
        const int t = __hip_atomic_fetch_add(&fz.tick[p], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = (t == fz.need[p] - 1) ? 1 : 0;
    }
    if (!__builtin_amdgcn_readfirstlane(last)) return;
    double ra = 0.0, dec = 0.0;
    int own = -1;
    if (lane == 0) {
        __hip_atomic_store(&fz.tick[p], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // for the next round
        bool active;
        sl_consume_chain(fz.st, p, fz.B, fz.nparts, fz.ll_pb, fz.sigma, fz.flags + 1, active, true);
        sl_propose_chain(fz.st, p, fz.prop_radec, fz.owner);
        ra = fz.prop_radec[2 * p]; dec = fz.prop_radec[2 * p + 1]; own = fz.owner[p];
    }
    own = __builtin_amdgcn_readfirstlane(own);
    if (own < 0 || !fz.pa.recs) return;
    ra = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(ra)), __builtin_amdgcn_readfirstlane(__double2loint(ra)));
    dec = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(dec)), __builtin_amdgcn_readfirstlane(__double2loint(dec)));
    if (lane < fz.pa.B)
        prep_one(fz.pa.bands, lane, p, (int64_t)lane * fz.pa.S + p, fz.pa.B, fz.pa.H, fz.pa.W, fz.pa.win_y0, fz.pa.win_h, fz.pa.type, ra, dec,
                 fz.pa.counts, fz.pa.shape, fz.pa.rsq_gal, fz.pa.recs, fz.pa.boxes, fz.pa.kind, fz.pa.status, fz.pa.nobox);
}
