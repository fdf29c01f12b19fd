"""Scalar restatement of the reference's slice sampler, one chain at a time.

TEST INFRASTRUCTURE ONLY (like the rest of oracle/): imported by tests/test_slicesample.py as the
checker of the product's lock-step engine (desi-mcmc_amd/util/infer/slicesample.py).  The product
never imports this module.

Follows CelestePy/util/infer/slicesample.py:89-227 (`slicesample`: direction_slice :114-203 with
`acceptable` :119-131, the component-wise / random-direction drivers :213-228) statement by statement;
the only change is where the random numbers come from: every `npr.rand()` / `npr.randn()` of the
reference is a draw from ONE stream handed in by the caller (the product's per-chain stream), in the
reference's order, so that a chain of the lock-step engine can be compared with it draw for draw.
`npr.shuffle(ordering)` (:216) becomes a stable argsort of one uniform per axis, as the engine does.
Parity status: PINNED.  tests/golden/make_golden.py (`gen_slicesample`) runs the reference's own
`slicesample` in the build container with its module-level `npr` replaced by a recorder and stores,
per call, every draw the reference made (in call order) and the `(x, llh)` it returned
(tests/golden/slicesample.npz).  tests/test_slicesample.py feeds those draws to `scalar_slicesample`
through `ReplayStream` and requires the reference's `(x, llh)` bit for bit, for every option set.
`TARGETS` are the closed-form log-densities both sides evaluate (the same code, so the same bits).
"""
import numpy as np

_CI2 = np.linalg.inv(np.array([[2.0, 0.8], [0.8, 1.0]]))
_MU2 = np.array([0.3, -1.0])
_A4 = np.array([[1.0, 0.3, 0.0, -0.2], [0.3, 0.5, 0.1, 0.0], [0.0, 0.1, 2.0, 0.4], [-0.2, 0.0, 0.4, 0.8]])
_CI4 = np.linalg.inv(_A4)


def _gauss(x):
    d = np.asarray(x, dtype=np.float64) - _MU2
    return float(-0.5 * (d @ _CI2 @ d))


def _bimodal(x):
    x = np.asarray(x, dtype=np.float64)
    return float(np.logaddexp(-0.5 * np.sum((x - 2.0) ** 2) / 0.3, -0.5 * np.sum((x + 2.0) ** 2) / 0.5))


def _gauss4(x):
    d = np.asarray(x, dtype=np.float64)
    return float(-0.5 * (d @ _CI4 @ d))


def _halfgauss(x):
    """the bounded demo target of slicesample.py:261-264"""
    x = np.asarray(x, dtype=np.float64)
    if np.any(x <= 0.0):
        return -np.inf
    return float(-0.5 * np.sum(x ** 2))


TARGETS = {"gauss": _gauss, "bimodal": _bimodal, "gauss4": _gauss4, "halfgauss": _halfgauss}

# kinds of a recorded draw (tests/golden/slicesample.npz): npr.rand(), one element of npr.randn(D), one
# sort key of npr.shuffle(ordering) -- the shuffled order is stored as the D keys whose stable argsort
# gives it, key[ordering[i]] = (i + 0.5) / D, which is the form the restatement consumes
DRAW_RAND, DRAW_RANDN, DRAW_SHUFFLE_KEY = 0, 1, 2


def shuffle_keys(ordering):
    ordering = np.asarray(ordering, dtype=np.int64)
    keys = np.empty(ordering.size)
    keys[ordering] = (np.arange(ordering.size) + 0.5) / ordering.size
    return keys


class ReplayStream(object):
    """Hands recorded draws back in order, through the interface of the product's ChainStreams
    (`uniform(idx)`, `normal(idx)` for ONE chain); refuses a draw of the wrong kind or past the end."""

    def __init__(self, kinds, vals):
        self.kinds, self.vals, self.pos = np.asarray(kinds), np.asarray(vals, dtype=np.float64), 0

    def _next(self, allowed, idx):
        if np.size(idx) != 1:
            raise AssertionError("a replay stream carries one chain")
        if self.pos >= self.vals.size:
            raise AssertionError("more draws requested than the reference made")
        if int(self.kinds[self.pos]) not in allowed:
            raise AssertionError("draw %d: reference made kind %d, restatement asks for %s" % (self.pos, self.kinds[self.pos], allowed))
        v = self.vals[self.pos]
        self.pos += 1
        return np.array([v])

    def uniform(self, idx):
        return self._next((DRAW_RAND, DRAW_SHUFFLE_KEY), idx)

    def normal(self, idx):
        return self._next((DRAW_RANDN,), idx)

    def exhausted(self):
        return self.pos == self.vals.size


def scalar_slicesample(init_x, logprob, stream, chain, sigma=1.0, step_out=True, max_steps_out=1000, compwise=True,
                       numdir=2, doubling_step=True):
    """slicesample.py:114-228, one chain, uniforms taken from stream `chain` in the reference's order"""
    one = np.array([chain])

    def rand():
        return stream.uniform(one)[0]

    def randn():
        return stream.normal(one)[0]

    def direction_slice(direction, init_x):
        def dir_logprob(z):
            return logprob(direction * z + init_x)

        def acceptable(z, llh_s, L, U):
            while (U - L) > 1.1 * sigma:
                middle = 0.5 * (L + U)
                splits = (middle > 0 and z >= middle) or (middle <= 0 and z < middle)
                if z < middle:
                    U = middle
                else:
                    L = middle
                if splits and llh_s >= dir_logprob(U) and llh_s >= dir_logprob(L):
                    return False
            return True
        upper = sigma * rand()
        lower = upper - sigma
        llh_s = np.log(rand()) + dir_logprob(0.0)
        l_steps_out = u_steps_out = 0
        if step_out:
            if doubling_step:
                while (dir_logprob(lower) > llh_s or dir_logprob(upper) > llh_s) and (l_steps_out + u_steps_out) < max_steps_out:
                    if rand() < 0.5:
                        l_steps_out += 1
                        lower -= (upper - lower)
                    else:
                        u_steps_out += 1
                        upper += (upper - lower)
            else:
                while dir_logprob(lower) > llh_s and l_steps_out < max_steps_out:
                    l_steps_out += 1
                    lower -= sigma
                while dir_logprob(upper) > llh_s and u_steps_out < max_steps_out:
                    u_steps_out += 1
                    upper += sigma
        start_upper, start_lower = upper, lower
        while True:
            new_z = (upper - lower) * rand() + lower
            new_llh = dir_logprob(new_z)
            if new_llh > llh_s and acceptable(new_z, llh_s, start_lower, start_upper):
                break
            elif new_z < 0:
                lower = new_z
            elif new_z > 0:
                upper = new_z
            else:
                raise Exception("Slice sampler shrank to zero!")
        return new_z * direction + init_x, new_llh
    dims = init_x.shape[0]
    if compwise:
        ordering = np.argsort([rand() for _ in range(dims)], kind="stable")
        new_x = init_x.copy()
        for d in ordering:
            direction = np.zeros(dims)
            direction[d] = 1.0
            new_x, new_llh = direction_slice(direction, new_x)
    else:
        new_x = init_x
        for d in range(numdir):
            direction = np.array([randn() for _ in range(dims)])
            direction = direction / np.sqrt(np.sum(direction ** 2))
            new_x, new_llh = direction_slice(direction, new_x)
    return new_x, new_llh
