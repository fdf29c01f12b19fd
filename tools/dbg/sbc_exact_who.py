"""conditional="exact": which galaxies hold the excess in the top rank of sigma / rho (exact random-position ranks)
    python tools/dbg/sbc_exact_who.py ROUNDS"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import desi_mcmc_amd as cel
import test_calibration as tc
ctx = cel.default_context(0)
reps = 8 * int(sys.argv[1] if len(sys.argv) > 1 else 8)
rows = []
for rep in range(reps):
    sc = tc.make_scene(cel, ctx, rep, 8, True)
    gal = np.nonzero(sc["typ"] == 1)[0]
    pix = sc["pix"]
    d = np.sqrt(((pix[:, None, :] - pix[None, :, :]) ** 2).sum(axis=2)) + 1e9 * np.eye(pix.shape[0])
    bx, st = sc["iset"].source_boxes(cel.SourceSet(ctx, sc["S"], 5).set(sc["typ"], sc["radec"], sc["flux"] / sc["bands"][None, :, 2] * sc["bands"][None, :, 1], sc["shape"]))
    rad = (bx[2, :, 1] - bx[2, :, 0]) / 2.0
    ru, rf, rs_ = tc.run_replicate(cel, ctx, rep, "host", chain_seed=rep, J=rep % 8, ncell=8, shapes=True, conditional="exact")
    for i, s in enumerate(gal):
        rows.append((rs_[i, 1], rs_[i, 3], sc["shape"][s, 1], sc["shape"][s, 0], sc["shape"][s, 3], sc["flux"][s].sum(), d[s].min(), rad[s], rs_[i, 0]))
rows = np.array(rows)
np.save(os.path.join(R, "gpurun_out", "sbc_exact_who.npy"), rows)
def table(name, col, rk, edges):
    print("-- rank of %s by %s" % (("sigma", "rho")[rk], name))
    for lo, hi in zip(edges[:-1], edges[1:]):
        m = (rows[:, col] >= lo) & (rows[:, col] < hi)
        c = np.bincount(rows[m, rk].astype(int), minlength=8)
        print("   [%7.2f, %7.2f): n = %5d  top share %.3f (1/8 = 0.125, +- %.3f)  bottom %.3f  %s" % (lo, hi, m.sum(), c[7] / max(m.sum(), 1), np.sqrt(0.125 * 0.875 / max(m.sum(), 1)), c[0] / max(m.sum(), 1), c.tolist()))
for rk in (0, 1):
    table("sigma*", 2, rk, [0, 0.5, 0.7, 0.9, 1.2, 1.8, 3, 100])
    table("theta*", 3, rk, [0, 0.2, 0.4, 0.6, 0.8, 1.01])
    table("rho*", 4, rk, [0, 0.2, 0.4, 0.6, 0.8, 1.01])
    table("flux sum", 5, rk, [0, 60, 100, 140, 200, 1000])
    table("nearest neighbour px", 6, rk, [0, 4, 8, 16, 40, 1e10])
    table("box radius px (r band)", 7, rk, [0, 10, 15, 20, 30, 50, 1000])
