"""the split's binomial sampler, mean and variance to parts in 10^5 (2e7 draws per case; the parity test's 2e5 see parts in 10^3)"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np
import desi_mcmc_amd as cel
from desi_mcmc_amd import _lib
from scipy import stats
ctx = cel.default_context(0)
N = 20000000
for i, (n, p) in enumerate([(220, 0.05), (220, 0.12), (260, 0.2), (300, 0.3), (400, 0.45), (400, 0.55), (600, 0.7), (1000, 0.8), (250, 0.13), (2000, 0.9), (210, 0.02), (205, 0.004)]):
    out = np.zeros(N, dtype=np.int64)
    _lib.check(_lib.lib().cel_debug_binomial(ctx._h, n, p, 99 + i, N, out.ctypes.data_as(_lib.c_int64_p)))
    mean, var = n * p, n * p * (1 - p)
    zm = (out.mean() - mean) / np.sqrt(var / N)
    m4 = var * (1 + 3 * (n - 2) * p * (1 - p))                       # fourth central moment
    zv = (out.var() - var) / np.sqrt((m4 - var * var) / N)
    ks = np.arange(0, n + 1)
    exp = stats.binom.pmf(ks, n, p) * N
    obs = np.bincount(out, minlength=n + 1).astype(float)
    keep = exp >= 20
    chi2 = np.sum((obs[keep] - exp[keep]) ** 2 / exp[keep])
    print("n = %5d p = %.3f (%s): mean off by %+.2f se (%.1e relative), variance %+.2f se; chi2 %.1f on %d cells (p = %.3g)" % (
        n, p, "BTPE" if n * min(p, 1 - p) > 30 else "inversion", zm, (out.mean() - mean) / mean, zv, chi2, keep.sum(), stats.chi2.sf(chi2, keep.sum() - 1)), flush=True)
