"""unit-stamp mass of galaxies on their own boxes (the reference's bounding radius, bounding_box.py:9-31, error 1e-5)"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import desi_mcmc_amd as cel
import test_calibration as tc
ctx = cel.default_context(0)
sc = tc.make_scene(cel, ctx, 0, 8, True)
S, B = sc["S"], 5
counts = sc["flux"] / sc["bands"][None, :, 2] * sc["bands"][None, :, 1]
sset = cel.SourceSet(ctx, S, B).set(sc["typ"], sc["radec"], counts, sc["shape"])
m = sc["iset"].stamp_mass(sset)
gal = sc["typ"] == 1
print("stars  mass: min %.6f median %.6f" % (m[~gal].min(), np.median(m[~gal])))
print("galaxy mass: min %.4f q05 %.4f q25 %.4f median %.4f q75 %.4f max %.4f" % ((m[gal].min(),) + tuple(np.percentile(m[gal], [5, 25, 50, 75])) + (m[gal].max(),)))
bx, st = sc["iset"].source_boxes(sset)
hw = (bx[2, :, 1] - bx[2, :, 0]) / 2.0
sig = sc["shape"][:, 1] / 0.396
for s in np.nonzero(gal)[0][:12]:
    print("sigma %.2f px  theta %.2f rho %.2f  box half-height %.1f px  mass %.4f" % (sig[s], sc["shape"][s, 0], sc["shape"][s, 3], hw[s], m[s, 2]))
