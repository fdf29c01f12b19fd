"""CPU tests of the lock-step slice sampler (desi-mcmc_amd/util/infer/slicesample.py) against the
oracle's scalar restatement of the reference's algorithm (oracle/slicesample_oracle.py, following
CelestePy/util/infer/slicesample.py:89-227) fed with the same per-chain random stream, and for the
distribution it leaves invariant."""
import json
import os

import numpy as np
import pytest

import desi_mcmc_amd  # noqa: F401
from desi_mcmc_amd.util.infer.slicesample import ChainStreams, slicesample, slicesample_lockstep
from oracle.slicesample_oracle import DRAW_SHUFFLE_KEY, TARGETS, ReplayStream, scalar_slicesample

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "slicesample.npz")


def _golden_cases():
    g = np.load(GOLD)
    for ci in range(int(g["ncases"])):
        yield ci, {k[len("c%d_" % ci):]: g[k] for k in g.files if k.startswith("c%d_" % ci)}


def test_restatement_reproduces_the_reference_run_bit_for_bit():
    """THE PIN of oracle/slicesample_oracle.py: fed the draws the reference's own slicesample made
    (recorded by tests/golden/make_golden.py::gen_slicesample while running
    CelestePy/util/infer/slicesample.py:89-227 in the build container), the scalar restatement
    consumes exactly those draws, kind by kind, and returns the reference's (x, llh) bit for bit --
    15 option sets x 12 successive calls, including the call of sources.py:315-319, the 4-D
    random-direction call of celeste_mcmc.py:229-239 and the bounded demo of slicesample.py:261-276."""
    ncase = 0
    for ci, c in _golden_cases():
        kw = json.loads(str(c["kw"]))
        kw.pop("lower_bound", None)                     # computed and never applied by the reference (:134-139)
        kw.pop("upper_bound", None)
        f = TARGETS[str(c["target"])]
        x = c["x0"].copy()
        off = c["draw_off"]
        for it in range(c["x"].shape[0]):
            st = ReplayStream(c["draw_kind"][off[it]:off[it + 1]], c["draw_val"][off[it]:off[it + 1]])
            x, llh = scalar_slicesample(x.copy(), f, st, 0, **kw)
            assert st.exhausted(), (ci, it, st.pos, off[it + 1] - off[it])
            assert np.array_equal(x, c["x"][it]), (ci, it, x, c["x"][it])
            assert llh == c["llh"][it], (ci, it)
        # the stored sort keys are the reference's shuffles (one per component-wise call)
        keys = c["draw_val"][c["draw_kind"] == DRAW_SHUFFLE_KEY]
        if keys.size:
            D = c["x0"].size
            assert np.array_equal(np.argsort(keys.reshape(-1, D), axis=1, kind="stable"), c["perms"])
        ncase += 1
    assert ncase == 15


def test_lockstep_engine_reproduces_the_reference_run_bit_for_bit():
    """The PRODUCT engine itself, handed the reference's recorded draws through its `rng=` seam,
    returns the reference's (x, llh) bit for bit: same draws in the same order, same arithmetic."""
    for ci, c in _golden_cases():
        kw = json.loads(str(c["kw"]))
        f = TARGETS[str(c["target"])]
        x = c["x0"].copy()
        off = c["draw_off"]
        for it in range(c["x"].shape[0]):
            st = ReplayStream(c["draw_kind"][off[it]:off[it + 1]], c["draw_val"][off[it]:off[it + 1]])
            X, ll = slicesample_lockstep(x[None, :], lambda idx, P: np.array([f(p) for p in P]), rng=st, **kw)
            x = X[0]
            assert st.exhausted(), (ci, it)
            assert np.array_equal(x, c["x"][it]), (ci, it, x, c["x"][it])
            assert ll[0] == c["llh"][it], (ci, it)


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


def test_shape_prior_matches_the_reference():
    """galaxy_shape_prior_constrained (celeste_galaxy_conditionals.py:268-275), the log-prior slice_sample_skew adds to
    the conditional likelihood: support and values as the reference's own functions gave them (200 points in and
    out of the support; phi bounded by pi as the reference bounds it).  The default bound is the renderer's unit."""
    from desi_mcmc_amd.celeste_galaxy_conditionals import galaxy_shape_prior_constrained as prior
    g = np.load(GOLD)
    th, want = g["prior_th"], g["prior_lp"]
    got = prior(th[:, 0], th[:, 1], th[:, 2], th[:, 3], phi_max=np.pi)
    assert np.array_equal(got, want) and 40 < np.isfinite(want).sum() < 160
    for t, w in zip(th[:20], want[:20]):
        assert prior(*t, phi_max=np.pi) == w                       # scalar form
    assert np.isfinite(prior(0.5, 1.0, 90.0, 0.5)) and prior(0.5, 1.0, 90.0, 0.5, phi_max=np.pi) == -np.inf
    assert prior(0.5, 1.0, 180.0, 0.5) == -np.inf and prior(1.0, 1.0, 90.0, 0.5) == -np.inf


def test_gamma_by_stream_is_a_gamma_sampler_and_order_free():
    """the flux step's Gamma variates (Marsaglia & Tsang on per-element counter-based streams): the first two moments for
    shapes from 0.3 to 5 000, a Kolmogorov-Smirnov test against the Gamma law, and an element's draw does not depend on
    what is drawn beside it (what lets a chain be dealt over ranks without changing)"""
    import scipy.stats as st
    from desi_mcmc_amd.celeste_mcmc import gamma_by_stream
    for a in (0.3, 1.0, 5.0, 37.5, 5000.0):
        x = gamma_by_stream(np.full(200000, a), 11, np.arange(200000))
        assert abs(x.mean() / a - 1) < 4 / np.sqrt(200000 * a) + 1e-3 and abs(x.var() / a - 1) < 0.02
        assert st.kstest(x[:20000], "gamma", args=(a,)).pvalue > 1e-4
    a = np.random.RandomState(1).uniform(0.5, 300.0, size=1000)
    full = gamma_by_stream(a, 7, np.arange(1000))
    pick = np.array([999, 3, 500, 17])
    assert np.array_equal(gamma_by_stream(a[pick], 7, pick), full[pick])
    assert not np.array_equal(gamma_by_stream(a, 8, np.arange(1000)), full)
