"""CPU tests of the lock-step slice sampler (desi-mcmc_amd/util/infer/slicesample.py) against the
oracle's scalar restatement of the reference's algorithm (oracle/slicesample_oracle.py, following
CelestePy/util/infer/slicesample.py:89-227) fed with the same per-chain random stream, and for the
distribution it leaves invariant."""
import numpy as np
import pytest

import desi_mcmc_amd  # noqa: F401
from desi_mcmc_amd.util.infer.slicesample import ChainStreams, slicesample, slicesample_lockstep
from oracle.slicesample_oracle import scalar_slicesample


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
