"""CPU tests of the lock-step slice sampler (desi-mcmc_amd/util/infer/slicesample.py) against a
scalar restatement of the reference's algorithm (CelestePy/util/infer/slicesample.py:89-227) fed
with the same per-chain random stream, and for the distribution it leaves invariant."""
import numpy as np
import pytest

import desi_mcmc_amd  # noqa: F401
from desi_mcmc_amd.util.infer.slicesample import ChainStreams, slicesample, slicesample_lockstep


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


def _targets():
    Ci = np.linalg.inv(np.array([[2.0, 0.8], [0.8, 1.0]]))
    mu = np.array([0.3, -1.0])

    def gauss(x):
        d = np.atleast_2d(x) - mu
        return -0.5 * np.einsum("ni,ij,nj->n", d, Ci, d)

    def bimodal(x):
        x = np.atleast_2d(x)
        return np.logaddexp(-0.5 * np.sum((x - 2.0) ** 2, axis=1) / 0.3, -0.5 * np.sum((x + 2.0) ** 2, axis=1) / 0.5)
    return gauss, bimodal


@pytest.mark.parametrize("kw", [
    dict(sigma=1.0, step_out=True, doubling_step=True),
    dict(sigma=0.4, step_out=True, doubling_step=False),
    dict(sigma=25.0, step_out=False),                       # what Source.resample_location runs (sigma >> posterior width)
    dict(sigma=0.7, step_out=True, doubling_step=True, compwise=False, numdir=3),
    dict(sigma=0.3, step_out=True, doubling_step=True, max_steps_out=3),
])
def test_lockstep_equals_scalar_restatement_chain_by_chain(kw):
    """Every chain of a lock-step batch follows exactly the trajectory the reference's scalar
    algorithm takes with that chain's random stream, for both targets and every option set."""
    for f in _targets():
        S = 37
        x0 = np.random.RandomState(1).randn(S, 2)
        for sweep in range(3):
            seed = 1000 + sweep
            X, ll = slicesample_lockstep(x0, lambda idx, P: f(P), seed=seed, **kw)
            for c in range(S):
                st = ChainStreams(seed, np.arange(S))
                xs, ls = scalar_slicesample(x0[c].copy(), lambda p: float(f(p)[0]), st, c, **kw)
                assert np.array_equal(X[c], xs), (c, X[c], xs)
                assert ll[c] == ls
            x0 = X


def test_chain_trajectory_does_not_depend_on_the_batch():
    gauss, _ = _targets()
    x0 = np.random.RandomState(2).randn(50, 2)
    X, ll = slicesample_lockstep(x0, lambda idx, P: gauss(P), sigma=1.0, seed=7)
    sub = np.array([3, 11, 40])
    Xs, lls = slicesample_lockstep(x0[sub], lambda idx, P: gauss(P), sigma=1.0, seed=7, chain_ids=sub)
    assert np.array_equal(Xs, X[sub]) and np.array_equal(lls, ll[sub])


def test_logprob_batch_receives_chain_indices():
    """the batch callable is told which chain every point belongs to (the device path scores each
    proposal against its own source's photon patch)"""
    centres = np.array([[0.0, 0.0], [10.0, -5.0], [-3.0, 7.0]])
    seen = []

    def f(idx, P):
        seen.append(idx.copy())
        return -0.5 * np.sum((P - centres[idx]) ** 2, axis=1)
    X = centres + 0.1
    for it in range(200):
        X, _ = slicesample_lockstep(X, f, sigma=2.0, seed=it)
    assert np.all(np.abs(X - centres) < 6.0)
    assert any(len(np.unique(i)) < len(i) for i in seen)          # step-out rounds score both interval ends


def test_invariant_distribution_moments():
    gauss, _ = _targets()
    S = 3000
    X = np.zeros((S, 2))
    for it in range(25):
        X, _ = slicesample_lockstep(X, lambda idx, P: gauss(P), sigma=1.0, seed=50 + it)
    np.testing.assert_allclose(X.mean(axis=0), [0.3, -1.0], atol=0.08)
    np.testing.assert_allclose(np.cov(X.T), [[2.0, 0.8], [0.8, 1.0]], atol=0.15)


def test_scalar_api_and_bounds_checks():
    gauss, _ = _targets()
    x, ll = slicesample(np.array([0.1, 0.2]), lambda p: float(gauss(p)[0]), sigma=1.0, seed=3)
    assert x.shape == (2,) and np.isfinite(ll) and ll == float(gauss(x)[0])
    xs, lls = slicesample(0.5, lambda p: -0.5 * float(p[0]) ** 2, seed=4)
    assert isinstance(xs, float)
    with pytest.raises(AssertionError):                    # slicesample.py:206-211
        slicesample(np.array([1.0, 1.0]), lambda p: 0.0, upper_bound=np.array([0.5, 2.0]))
    with pytest.raises(Exception, match="NaN"):
        slicesample(np.array([0.0]), lambda p: float("nan") if abs(p[0]) > 1e-9 else 0.0, step_out=False, seed=1)
