"""the field render of the calibration scenes against sky + the sum of the per-source stamps (cel_render_stamps), and the stamp mass
(cel_stamp_mass) against the stamps' sums: per source, the worst disagreement in photons"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import desi_mcmc_amd as cel
import test_calibration as tc
ctx = cel.default_context(0)
worst = []
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 16):
    sc = tc.make_scene(cel, ctx, rep, 8, True)
    S, B, H, W = sc["S"], sc["B"], sc["H"], sc["W"]
    counts = sc["flux"] / sc["bands"][None, :, 2] * sc["bands"][None, :, 1]
    sset = cel.SourceSet(ctx, S, B).set(sc["typ"], sc["radec"], counts, sc["shape"])
    sc["iset"].render(sset, loglik=False)
    lam = sc["iset"].model_images()
    mass = sc["iset"].stamp_mass(sset)
    for b in range(B):
        st, bx = sc["iset"].stamps(sset, b, scaled=True)
        img = np.full((H, W), sc["bands"][b, 0])
        for s in range(S):
            if st[s] is None: continue
            y0, y1, x0, x1 = bx[s]
            img[y0:y1, x0:x1] += st[s]
            dm = abs(mass[s, b] * counts[s, b] - st[s].sum())
            if dm > 1e-6 * max(counts[s, b], 1):
                print("rep %d source %d band %d: stamp mass %.6f x counts vs sum of the stamp %.6f" % (rep, s, b, mass[s, b] * counts[s, b], st[s].sum()))
        d = lam[b] - img
        k = np.unravel_index(np.argmax(np.abs(d)), d.shape)
        worst.append((np.abs(d).max(), rep, b, k, np.abs(d).sum()))
worst.sort(reverse=True)
for w in worst[:8]:
    print("max |field render - (sky + stamps)| = %.3e photons (rep %d band %d at pixel %s), sum |d| = %.3e" % w)
