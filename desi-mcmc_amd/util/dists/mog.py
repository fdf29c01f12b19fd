"""mog_loglike and MixtureOfGaussians with the names, arguments and component order of
CelestePy/util/dists/mog.py:5-21, 38-112, for callers that use them directly
(`mog_funs.mog_loglike(...)`, `cmix.evaluate_grid(...)`, `FitsImage.psf_mog`).

The evaluation (mog_loglike, and through it logpdf / pdf / evaluate_grid) runs on the device
behind cel_mog_loglike; the component algebra (convex_combine, apply_affine, convolve) is a few
K x 2 x 2 numpy operations on the host, in the reference's order (convolve is self-major:
index j * K_other + k).  There is no CPU evaluator: without the HIP library and a
GPU the evaluating methods raise.  Sampling (mog_samples, rvs) is outside the render path.
"""
import ctypes as C

import numpy as np

from ... import _lib as L
from ... import field as _field


def mog_loglike(x, means, icovs, dets, pis, device=0):
    """log sum_k pis[k] N(x; means[k], inv(icovs[k])) at x (N, 2) or (2,)  -- mog.py:5-21.
    pis must be positive: log(pis) is taken (a negative weight gives NaN)."""
    x = np.asarray(x, dtype=np.float64)
    xx = np.ascontiguousarray(np.atleast_2d(x))
    means, icovs = L.f64(means), L.f64(icovs)
    K = means.shape[0]
    if xx.shape[1] != 2 or means.shape != (K, 2) or icovs.shape != (K, 2, 2):
        raise ValueError("mog_loglike: this path evaluates 2-D mixtures: x (N,2), means (K,2), icovs (K,2,2)")
    with np.errstate(divide="ignore", invalid="ignore"):
        logw = L.f64(-np.log(2 * np.pi) - 0.5 * np.log(np.asarray(dets, dtype=np.float64))
                     + np.log(np.asarray(pis, dtype=np.float64)))
    if logw.shape != (K,):
        raise ValueError("mog_loglike: dets and pis must have one entry per component")
    out = np.zeros(xx.shape[0])
    ctx = _field.default_context(device)
    L.check(L.lib().cel_mog_loglike(ctx._h, xx.ctypes.data, xx.shape[0], L.dptr(means), L.dptr(icovs), L.dptr(logw),
                                    int(K), out.ctypes.data, L.CEL_HOST))
    if x.ndim == 1:
        return out[0]
    return out


def discrete(p, shape, rng=None):
    """component labels drawn with probabilities p  -- mog.py:34-37: label = K - #{cumulative weights above the
    uniform} = #{cumulative weights at or below it}, found by bisection instead of an (N, K) comparison.  Weights
    that sum to less than one (a PSF's, 0.998) would send the last sliver to label K: it is given to K - 1.
    Host-side draws from `rng` (numpy's global generator by default, as the reference)."""
    rng = np.random if rng is None else rng
    cum = np.cumsum(np.asarray(p, dtype=np.float64))
    k = np.searchsorted(cum, rng.rand(int(np.prod(shape))), side="right")
    return np.minimum(k, cum.shape[0] - 1).reshape(shape)


def mog_samples(N, means, chols, pis, rng=None):
    """N draws from the mixture: a label per draw, then mean + chol . white noise  -- mog.py:25-32"""
    rng = np.random if rng is None else rng
    means, chols = np.asarray(means, dtype=np.float64), np.asarray(chols, dtype=np.float64)
    labels = discrete(pis, (N,), rng=rng)
    white = rng.randn(N, means.shape[1])
    return np.matmul(chols[labels], white[:, :, None])[:, :, 0] + means[labels]


class MixtureOfGaussians(object):
    """Evaluate the (log) density of a 2-D mixture of Gaussians  -- mog.py:38-112"""

    def __init__(self, means, covs, pis):
        means = np.asarray(means, dtype=np.float64)
        self.K, self.D = means.shape
        self.update_params(means, np.asarray(covs, dtype=np.float64), np.asarray(pis, dtype=np.float64))

    def update_params(self, means, covs, pis):
        assert covs.shape[1] == covs.shape[2] == self.D
        assert self.K == covs.shape[0] == len(pis), "%d != %d != %d" % (self.K, covs.shape[0], len(pis))
        self.means, self.covs, self.pis = means, covs, pis
        self.dets = np.array([np.linalg.det(c) for c in covs])          # mog.py:54-56
        self.icovs = np.array([np.linalg.inv(c) for c in covs])
        self.chols = np.array([np.linalg.cholesky(c) for c in covs])

    def logpdf(self, x):
        return mog_loglike(x, means=self.means, icovs=self.icovs, dets=self.dets, pis=self.pis)

    def pdf(self, x):
        return np.exp(self.logpdf(x))

    def mean(self, x=None):
        return np.dot(self.pis, self.means)

    def var(self, x=None):
        """the weighted sum of the component covariances  -- mog.py:69-70 (which reads a global `pis`; the
        mixture's own weights are what it means)"""
        return np.sum(self.covs * self.pis[:, None, None], axis=0)

    def rvs(self, size=1, rng=None):
        """mog.py:72-73"""
        return mog_samples(size, self.means, self.chols, self.pis, rng=rng)

    def convolve(self, mog):
        """all pairs, this mixture's component index major  -- mog.py:75-81"""
        means = np.reshape(self.means[:, None] + mog.means[None, :], (-1, 2))
        weights = np.reshape(self.pis[:, None] * mog.pis[None, :], (-1,))
        covs = np.reshape(self.covs[:, None] + mog.covs[None, :], (-1, 2, 2))
        return MixtureOfGaussians(means, covs, weights)

    def apply_affine(self, A, b):
        """distribution of A x + b  -- mog.py:84-92"""
        A = np.asarray(A, dtype=np.float64)
        return MixtureOfGaussians(means=np.dot(self.means, A.T) + b,
                                  covs=np.array([np.dot(np.dot(A, c), A.T) for c in self.covs]),
                                  pis=self.pis)

    @staticmethod
    def convex_combine(mogs, mixing_weights):
        """mog.py:94-100"""
        return MixtureOfGaussians(means=np.vstack([m.means for m in mogs]),
                                  covs=np.vstack([m.covs for m in mogs]),
                                  pis=np.concatenate([w * m.pis for w, m in zip(mixing_weights, mogs)]))

    def evaluate_grid(self, xlim, ylim, pts=None):
        """density on the integer pixel grid [xlim) x [ylim), y-outer  -- mog.py:102-112"""
        assert (ylim[1] > ylim[0]) and (xlim[1] > xlim[0]), "bad limits."
        y_grid = np.arange(ylim[0], ylim[1], dtype=np.float64)
        x_grid = np.arange(xlim[0], xlim[1], dtype=np.float64)
        xx, yy = np.meshgrid(x_grid, y_grid, indexing='xy')
        if pts is None:
            pts = np.column_stack((xx.ravel(order='C'), yy.ravel(order='C')))
        lls = mog_loglike(pts, self.means, self.icovs, self.dets, self.pis)
        return np.reshape(np.exp(lls), xx.shape)
