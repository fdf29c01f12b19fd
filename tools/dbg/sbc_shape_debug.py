"""which galaxies' sigma ranks are off (tests/test_calibration.py, shape step)"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import desi_mcmc_amd as cel
import test_calibration as tc
from desi_mcmc_amd import celeste_mcmc
ctx = cel.default_context(0)
rows = []
for rep in range(4):
    sc = tc.make_scene(cel, ctx, rep, 8, True)
    gf = celeste_mcmc.GibbsField(sc["iset"], list(range(5)), sc["bands"][:, 2], sc["bands"][:, 1], sc["H"] * sc["W"], a_0=tc.EPS_A, b_0=tc.EPS_B)
    g = celeste_mcmc.ModelGibbs([gf], sc["typ"], sc["radec"], sc["flux"], sc["shape"], seed=rep, flux_a_0=tc.FLUX_A, flux_b_0=tc.FLUX_B, engine=sys.argv[1] if len(sys.argv) > 1 else "device")
    D = []
    for k in range(21):
        g.sweep(shapes=True); g.log_likelihood()
        if k % 3 == 2: D.append(g.shape.copy())
    D = np.array(D)
    gal = sc["typ"] == 1
    r = (D[:, gal, 1] < sc["shape"][None, gal, 1]).sum(axis=0)
    for i, s in enumerate(np.nonzero(gal)[0]):
        rows.append((r[i], sc["shape"][s, 1], sc["shape"][s, 0], sc["shape"][s, 3], sc["flux"][s].sum(), D[-1, s, 1]))
rows = np.array(rows)
print("forward-only ranks of sigma:", np.bincount(rows[:, 0].astype(int), minlength=8))
top = rows[rows[:, 0] == 7]
rest = rows[rows[:, 0] < 7]
for name, col in (("sigma*", 1), ("theta*", 2), ("rho*", 3), ("flux sum", 4)):
    print("%-9s rank-7 median %.3f (q10 %.3f q90 %.3f) | others median %.3f (q10 %.3f q90 %.3f)" % (name, np.median(top[:, col]), np.percentile(top[:, col], 10), np.percentile(top[:, col], 90),
          np.median(rest[:, col]), np.percentile(rest[:, col], 10), np.percentile(rest[:, col], 90)))
print("ratio last draw / truth for rank-7:", np.round(np.sort(top[:, 5] / top[:, 1])[:20], 3))
big = rows[:, 1] > 2.0
print("sigma* > 2: ranks", np.bincount(rows[big, 0].astype(int), minlength=8), " sigma* < 0.6:", np.bincount(rows[rows[:, 1] < 0.6, 0].astype(int), minlength=8))
