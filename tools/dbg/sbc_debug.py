"""which sources' flux ranks are off in tests/test_calibration.py (one replicate, device engine)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import desi_mcmc_amd as cel
import test_calibration as tc
from desi_mcmc_amd import celeste_mcmc
ctx = cel.default_context(0)
rep = int(sys.argv[1]) if len(sys.argv) > 1 else 0
sc = tc.make_scene(cel, ctx, rep)
gf = celeste_mcmc.GibbsField(sc["iset"], list(range(5)), sc["bands"][:, 2], sc["bands"][:, 1], sc["H"] * sc["W"], a_0=tc.EPS_A, b_0=tc.EPS_B)
g = celeste_mcmc.ModelGibbs([gf], sc["typ"], sc["radec"], sc["flux"], sc["shape"], seed=rep, flux_a_0=tc.FLUX_A, flux_b_0=tc.FLUX_B, engine="device")
F = []
E = []
for k in range(60):
    g.sweep()
    F.append(g.fluxes.copy()); E.append(gf.epsilon.copy())
F = np.array(F); E = np.array(E)
truth = sc["flux"]
print("eps truth", sc["bands"][:, 0], "chain mean", E[10:].mean(axis=0), "sd", E[10:].std(axis=0))
z = (F[10:].mean(axis=0) - truth) / F[10:].std(axis=0)
print("flux z-score mean by type: stars %.3f galaxies %.3f" % (z[sc["typ"] == 0].mean(), z[sc["typ"] == 1].mean()))
r = (F[10::5][:7] < truth[None]).sum(axis=0)
for t in (0, 1):
    m = sc["typ"] == t
    print("type", t, "rank hist", np.bincount(r[m].ravel(), minlength=8))
# by flux level
lo = truth < np.percentile(truth, 20)
print("faint fifth rank hist", np.bincount(r[lo], minlength=8), "bright rest", np.bincount(r[~lo], minlength=8))
# ratio chain/truth for faint
print("median chain/truth: faint %.3f bright %.3f" % (np.median((F[10:].mean(axis=0) / truth)[lo]), np.median((F[10:].mean(axis=0) / truth)[~lo])))
# acf of flux lag 1,5
Fc = F[10:] - F[10:].mean(axis=0)
for lag in (1, 5):
    ac = (Fc[lag:] * Fc[:-lag]).mean(axis=0) / Fc.var(axis=0)
    print("flux acf lag", lag, np.median(ac))
U = None
