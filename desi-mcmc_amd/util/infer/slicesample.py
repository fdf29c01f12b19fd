"""Slice sampling: `slicesample` of CelestePy/util/infer/slicesample.py:89-227 -- same arguments,
same algorithm -- restated as a LOCK-STEP state machine over many chains.

The reference calls its log-probability one point at a time; Source.resample_location
(CelestePy/sources.py:308-319) does that 10-50 times per source per sweep, and every call is one
source's conditional likelihood.  Here S chains (one per source) advance together: in every
round each unfinished chain names the point(s) it needs next, ALL of them are evaluated by one
call of `logprob_batch` (one device launch of cel_patch_loglik_multi for a whole catalogue),
and the chains consume their values; a chain that has finished one direction starts its next in the
following round without waiting for the others.  There is no per-chain Python.

Algorithm per chain (slicesample.py:114-203, component-wise or random directions):
    for each direction:                                   upper = sigma * U;  lower = upper - sigma
        llh_s = log(U) + logprob(x)                       (slice level)
        step out (optional): doubling (:150-157) or by sigma (:158-164), at most max_steps_out
        shrink (:167-190): new_z ~ U(lower, upper); accept when logprob > llh_s (and, after
            doubling, the interval passes `acceptable`, :119-131); else the interval shrinks to new_z
Random numbers: one counter-based stream per chain (SplitMix64 of (seed, chain id, counter)), so
a chain's trajectory does not depend on which other chains run beside it.  The reference draws
from numpy's global MT19937; handed the draws a reference run made (`rng=`), the engine returns
that run's samples bit for bit (tests/test_slicesample.py against tests/golden/slicesample.npz).

Reference behaviour kept on purpose: `upper_bound` / `lower_bound` are checked at the start
(:206-211) and turned into per-direction bounds (:134-139) that are never used afterwards; the
`step` keyword of Source.resample_location is not an argument of slicesample, so sigma stays at
its default 1.0 there.
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix(x):
    """SplitMix64 finaliser on uint64 arrays."""
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


class ChainStreams(object):
    """Two streams per chain: the uniforms, u(chain, k) = SplitMix64(SplitMix64(seed ^ chain * C) + k), and -- keyed off
    the same chain key -- the normals that make random directions.  The normals have a stream of their own so that a
    chain's k-th direction does not depend on how many uniforms its earlier directions used: the device engine
    (cel_slice_sample) is handed all directions up front, computed here (log, cos and sqrt enter the positions)."""

    def __init__(self, seed, chain_ids):
        ids = np.asarray(chain_ids, dtype=np.uint64)
        with np.errstate(over="ignore"):
            self.key = _splitmix((np.uint64(int(seed) & 0xFFFFFFFFFFFFFFFF) ^ (ids * np.uint64(0xD1342543DE82EF95))) & _M64)
            self.nkey = _splitmix(self.key ^ np.uint64(0xA0761D6478BD642F))
        self.count = np.zeros(ids.shape[0], dtype=np.uint64)
        self.ncount = np.zeros(ids.shape[0], dtype=np.uint64)

    def uniform(self, idx):
        """next uniform in (0, 1) of the chains idx (53 random bits, never 0)"""
        with np.errstate(over="ignore"):
            z = _splitmix((self.key[idx] + self.count[idx] * np.uint64(0x9E3779B97F4A7C15)) & _M64)
        self.count[idx] += np.uint64(1)
        return ((z >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)

    def _nuniform(self, idx):
        with np.errstate(over="ignore"):
            z = _splitmix((self.nkey[idx] + self.ncount[idx] * np.uint64(0x9E3779B97F4A7C15)) & _M64)
        self.ncount[idx] += np.uint64(1)
        return ((z >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)

    def normal(self, idx):
        """next standard normal of the chains idx (Box-Muller on two numbers of the chain's normal stream)"""
        u1, u2 = self._nuniform(idx), self._nuniform(idx)
        return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)

    def directions(self, numdir, D):
        """(S, numdir, D): every chain's next `numdir` random unit directions, exactly as slicesample_lockstep draws them
        one by one (slicesample.py:224-226)"""
        every = np.arange(self.key.shape[0])
        out = np.empty((every.size, numdir, D))
        for k in range(numdir):
            dr = np.stack([self.normal(every) for _ in range(D)], axis=1)
            out[:, k] = dr / np.sqrt(np.sum(dr ** 2, axis=1, keepdims=True))
        return out


# phases of a chain inside one direction
_P_LEVEL, _P_OUT_DOUBLE, _P_OUT_LEFT, _P_OUT_RIGHT, _P_SHRINK, _P_ACCEPT, _P_DONE, _P_FINAL = 0, 1, 2, 3, 4, 5, 6, 7


def slicesample_lockstep(init_x, logprob_batch, sigma=1.0, step_out=True, max_steps_out=1000, compwise=True,
                         numdir=2, doubling_step=True, upper_bound=np.inf, lower_bound=-np.inf, seed=0,
                         chain_ids=None, stats=None, rng=None, accept="reference"):
    """One slicesample() update of S chains at once.

    init_x          (S, D) current states
    logprob_batch   callable(idx (n,), X (n, D)) -> (n,) log-probabilities: chain idx[i] evaluated at
                    X[i].  A chain may appear more than once in one call (the two ends of its
                    interval while stepping out).
    other arguments as slicesample (slicesample.py:110-118); seed / chain_ids name the random streams
    stats           optional dict: receives 'rounds', 'evals', 'max_steps_in'
    accept          "reference" (default: `acceptable` as the reference writes it) or "neal" (the doubling procedure's
                    acceptance test with Neal's sticky flag: see _accept_advance)
    rng             optional replacement of the per-chain streams (`uniform(idx)`, `normal(idx)`): how the
                    tests hand the engine the very draws the reference made (tests/golden/slicesample.npz)
    -> (new_x (S, D), new_llh (S,))"""
    X = np.array(init_x, dtype=np.float64, copy=True)
    if X.ndim != 2:
        raise ValueError("init_x must be (S, D)")
    S, D = X.shape
    ub = np.broadcast_to(np.asarray(upper_bound, dtype=np.float64), (S, D)) if np.ndim(upper_bound) else np.full((S, D), float(upper_bound))
    lb = np.broadcast_to(np.asarray(lower_bound, dtype=np.float64), (S, D)) if np.ndim(lower_bound) else np.full((S, D), float(lower_bound))
    assert np.all(X < ub), "init_x >= ub"                     # slicesample.py:206-209
    assert np.all(X > lb), "init_x <= lb"
    if rng is None:
        rng = ChainStreams(seed, np.arange(S) if chain_ids is None else chain_ids)
    every = np.arange(S)
    # directions of this update: a random order of the axes (:214-221) or numdir random unit vectors (:223-228)
    if compwise:
        order = np.argsort(np.stack([rng.uniform(every) for _ in range(D)], axis=1), axis=1, kind="stable")
        ndir = D
    else:
        ndir = int(numdir)
    new_llh = np.full(S, np.nan)
    n_rounds = n_evals = 0
    # per-chain state of the direction a chain is working on.  Chains do NOT wait for each other
    # between directions: one that has finished its first axis starts its second in the next
    # round, so an update takes max over chains of (evaluations of the chain) rounds, not the sum
    # over directions of the slowest chain of each.
    kdir = np.zeros(S, dtype=np.int64)
    direction = np.zeros((S, D))
    x0 = X.copy()
    upper, lower, log_u, llh_s = np.zeros(S), np.zeros(S), np.zeros(S), np.zeros(S)
    phase = np.full(S, _P_LEVEL)
    l_out = np.zeros(S, dtype=np.int64)
    u_out = np.zeros(S, dtype=np.int64)
    start_lower, start_upper = np.zeros(S), np.zeros(S)
    new_z, acc_L, acc_U = np.zeros(S), np.zeros(S), np.zeros(S)
    sticky = np.zeros(S, dtype=bool) if accept == "neal" else None
    steps_in = np.zeros(S, dtype=np.int64)
    max_in = 0

    def start_direction(idx):
        if idx.size == 0:
            return
        if compwise:
            direction[idx] = 0.0
            direction[idx, order[idx, kdir[idx]]] = 1.0
        else:
            dr = np.stack([rng.normal(idx) for _ in range(D)], axis=1)
            direction[idx] = dr / np.sqrt(np.sum(dr ** 2, axis=1, keepdims=True))
        x0[idx] = X[idx]
        upper[idx] = sigma * rng.uniform(idx)              # :142-143
        lower[idx] = upper[idx] - sigma
        log_u[idx] = np.log(rng.uniform(idx))              # :146, the level's random part
        phase[idx] = _P_LEVEL
        l_out[idx] = 0
        u_out[idx] = 0
        steps_in[idx] = 0

    def enter_shrink(idx):
        start_lower[idx] = lower[idx]
        start_upper[idx] = upper[idx]
        phase[idx] = _P_SHRINK

    if ndir > 0:
        start_direction(every)
    else:
        phase[:] = _P_FINAL
    while True:
        act = np.nonzero(phase != _P_FINAL)[0]
        if act.size == 0:
            break
        ph = phase[act]
        # ---- which points does every chain need this round? --------------------------------
        single = act[(ph == _P_LEVEL) | (ph == _P_OUT_LEFT) | (ph == _P_OUT_RIGHT) | (ph == _P_SHRINK)]
        double = act[(ph == _P_OUT_DOUBLE) | (ph == _P_ACCEPT)]
        zs = np.zeros(single.size)
        p1 = phase[single]
        zs[p1 == _P_OUT_LEFT] = lower[single[p1 == _P_OUT_LEFT]]
        zs[p1 == _P_OUT_RIGHT] = upper[single[p1 == _P_OUT_RIGHT]]
        sh = single[p1 == _P_SHRINK]
        if sh.size:
            new_z[sh] = (upper[sh] - lower[sh]) * rng.uniform(sh) + lower[sh]     # :172
            steps_in[sh] += 1
            zs[p1 == _P_SHRINK] = new_z[sh]
        p2 = phase[double]
        za = np.where(p2 == _P_OUT_DOUBLE, lower[double], acc_L[double])
        zb = np.where(p2 == _P_OUT_DOUBLE, upper[double], acc_U[double])
        idx = np.concatenate([single, double, double]) if double.size else single
        z = np.concatenate([zs, za, zb]) if double.size else zs
        vals = np.asarray(logprob_batch(idx, x0[idx] + z[:, None] * direction[idx]), dtype=np.float64)
        n_rounds += 1
        n_evals += idx.size
        v1 = vals[:single.size]
        va = vals[single.size:single.size + double.size]
        vb = vals[single.size + double.size:]

        # ---- consume ------------------------------------------------------------------------
        m = p1 == _P_LEVEL                                 # llh_s = log(U) + logprob(x)   (:146)
        c = single[m]
        if c.size:
            llh_s[c] = log_u[c] + v1[m]
            if step_out:
                phase[c] = _P_OUT_DOUBLE if doubling_step else _P_OUT_LEFT
            else:
                enter_shrink(c)
        m = p1 == _P_OUT_LEFT                              # :159-161
        c = single[m]
        if c.size:
            go = (v1[m] > llh_s[c]) & (l_out[c] < max_steps_out)
            l_out[c[go]] += 1
            lower[c[go]] -= sigma
            phase[c[~go]] = _P_OUT_RIGHT
        m = p1 == _P_OUT_RIGHT                             # :162-164
        c = single[m]
        if c.size:
            go = (v1[m] > llh_s[c]) & (u_out[c] < max_steps_out)
            u_out[c[go]] += 1
            upper[c[go]] += sigma
            enter_shrink(c[~go])
        m = p2 == _P_OUT_DOUBLE                            # :151-157
        c = double[m]
        if c.size:
            go = ((va[m] > llh_s[c]) | (vb[m] > llh_s[c])) & ((l_out[c] + u_out[c]) < max_steps_out)
            g = c[go]
            if g.size:
                left = rng.uniform(g) < 0.5
                width = upper[g] - lower[g]
                l_out[g[left]] += 1
                lower[g[left]] -= width[left]
                u_out[g[~left]] += 1
                upper[g[~left]] += width[~left]
            enter_shrink(c[~go])
        m = p1 == _P_SHRINK                                # :173-190
        c = single[m]
        if c.size:
            v = v1[m]
            if np.any(np.isnan(v)):
                raise Exception("Slice sampler got a NaN")
            inside = v > llh_s[c]
            # accepted unless the doubled interval has to be tested (:177, :119-131)
            need = inside & ((start_upper[c] - start_lower[c]) > 1.1 * sigma)
            ok = inside & ~need
            new_llh[c[ok]] = v[ok]
            phase[c[ok]] = _P_DONE
            t = c[need]
            if t.size:
                new_llh[t] = v[need]
                acc_L[t] = start_lower[t]
                acc_U[t] = start_upper[t]
                phase[t] = _P_ACCEPT
                _accept_advance(t, new_z, llh_s, acc_L, acc_U, sigma, phase, None, None, sticky)
            r = c[~inside]
            if r.size:
                if np.any(new_z[r] == 0.0):
                    raise Exception("Slice sampler shrank to zero!")
                neg = new_z[r] < 0
                lower[r[neg]] = new_z[r[neg]]
                upper[r[~neg]] = new_z[r[~neg]]
        m = p2 == _P_ACCEPT                                # the halving test of `acceptable`
        c = double[m]
        if c.size:
            _accept_advance(c, new_z, llh_s, acc_L, acc_U, sigma, phase, va[m], vb[m], sticky)
            rej = c[phase[c] == _P_SHRINK]
            if rej.size:                                   # not acceptable: shrink as a rejection (:180-183)
                neg = new_z[rej] < 0
                lower[rej[neg]] = new_z[rej[neg]]
                upper[rej[~neg]] = new_z[rej[~neg]]
        # ---- chains that finished a direction: move, then on to their next one ---------------
        fin = act[phase[act] == _P_DONE]
        if fin.size:
            X[fin] = x0[fin] + new_z[fin, None] * direction[fin]          # :203
            max_in = max(max_in, int(steps_in[fin].max()))
            kdir[fin] += 1
            last = kdir[fin] >= ndir
            phase[fin[last]] = _P_FINAL
            nxt = fin[~last]
            if nxt.size:
                # the next direction starts where this one ended: its level needs logprob at a point
                # that has just been scored (the reference evaluates it again and gets the same number)
                start_direction(nxt)
                llh_s[nxt] = log_u[nxt] + new_llh[nxt]
                if step_out:
                    phase[nxt] = _P_OUT_DOUBLE if doubling_step else _P_OUT_LEFT
                else:
                    enter_shrink(nxt)
    if stats is not None:
        stats.update(rounds=n_rounds, evals=n_evals, max_steps_in=max_in)
    return X, new_llh


def _accept_advance(c, new_z, llh_s, acc_L, acc_U, sigma, phase, vL, vU, sticky=None):
    """`acceptable` (slicesample.py:119-131) for the chains c, resumed with the values vL, vU of
    logprob at the interval ends the previous halving asked for (None: first entry).  Chains leave
    with phase _P_DONE (acceptable), _P_SHRINK (not acceptable) or stay in _P_ACCEPT with the next
    pair of ends to evaluate in acc_L / acc_U.
    sticky (a bool array over all chains, or None): Neal's test (2003, fig. 6) keeps its flag D once a halving has parted the
    new point from the start point and checks the ends at EVERY halving from then on; the reference recomputes `splits` per
    halving (:124) and checks only where the new point lies in the upper half again -- it accepts points Neal's test
    rejects, and its doubling updates do not leave the target exactly invariant (DESIGN Q20; the calibration test shows it).
    The default reproduces the reference; accept="neal" in the sampler's options passes the array."""
    todo = c
    if vL is None and sticky is not None:
        sticky[c] = False
    if vL is not None:
        # the halving that asked for these values had `splits` true: reject when both ends are below the level
        bad = (llh_s[c] >= vU) & (llh_s[c] >= vL)
        phase[c[bad]] = _P_SHRINK
        todo = c[~bad]
    while todo.size:
        wide = (acc_U[todo] - acc_L[todo]) > 1.1 * sigma
        phase[todo[~wide]] = _P_DONE
        todo = todo[wide]
        if not todo.size:
            break
        L, U, z = acc_L[todo], acc_U[todo], new_z[todo]
        middle = 0.5 * (L + U)
        splits = ((middle > 0) & (z >= middle)) | ((middle <= 0) & (z < middle))
        lo = z < middle
        acc_U[todo[lo]] = middle[lo]
        acc_L[todo[~lo]] = middle[~lo]
        # chains whose halving split off the start point need logprob at both new ends: they wait
        # for the next round; the others halve again at once
        if sticky is not None:
            sticky[todo] |= splits
            splits = sticky[todo]
        todo = todo[~splits]


def slicesample(init_x, logprob, *logprob_args, **slice_sample_args):
    """generate a new sample from a probability density using slice sampling
    -- CelestePy/util/infer/slicesample.py:89-227 (same arguments; `seed` may be added).

    init_x : array (D,) or float;  logprob : callable, lprob = logprob(x, *logprob_args)
    Returns (new_x, new_llh)."""
    kw = dict(sigma=slice_sample_args.get('sigma', 1.0), step_out=slice_sample_args.get('step_out', True),
              max_steps_out=slice_sample_args.get('max_steps_out', 1000), compwise=slice_sample_args.get('compwise', True),
              numdir=slice_sample_args.get('numdir', 2), doubling_step=slice_sample_args.get('doubling_step', True),
              upper_bound=slice_sample_args.get('upper_bound', np.inf), lower_bound=slice_sample_args.get('lower_bound', -np.inf))
    seed = slice_sample_args.get('seed')
    if seed is None:
        seed = int(np.random.randint(0, 2 ** 31 - 1))
    scalar = isinstance(init_x, float) or isinstance(init_x, np.number)
    x = np.array([init_x], dtype=np.float64) if scalar else np.asarray(init_x, dtype=np.float64)
    for k in ('upper_bound', 'lower_bound'):
        if np.ndim(kw[k]):
            kw[k] = np.asarray(kw[k], dtype=np.float64)[None, :]

    def batch(idx, pts):
        return np.array([logprob(p, *logprob_args) for p in pts], dtype=np.float64)

    new_x, new_llh = slicesample_lockstep(x[None, :], batch, seed=seed, **kw)
    if scalar:
        return float(new_x[0, 0]), float(new_llh[0])
    return new_x[0], float(new_llh[0])
