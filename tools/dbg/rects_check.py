"""cel_samples_photon_rects against the fetched patches, split after split"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import desi_mcmc_amd as cel
import test_calibration as tc
ctx = cel.default_context(0)
bad = n = 0
for rep in range(4):
    sc = tc.make_scene(cel, ctx, rep, 8, True)
    S, B = sc["S"], sc["B"]
    counts = sc["flux"] / sc["bands"][None, :, 2] * sc["bands"][None, :, 1]
    sset = cel.SourceSet(ctx, S, B).set(sc["typ"], sc["radec"], counts, sc["shape"])
    for seed in range(10):
        sc["iset"].photon_split_resident(sset, 77 + seed)
        rects = sc["iset"].photon_rects()
        boxes, offs, data = sc["iset"].fetch_samples()
        for s in range(S):
            for b in range(B):
                y0, y1, x0, x1 = boxes[s, b]
                p = data[offs[s * B + b]:offs[s * B + b + 1]].reshape(max(y1 - y0, 0), max(x1 - x0, 0))
                ys, xs = np.nonzero(p)
                want = [0, 0, 0, 0] if ys.size == 0 else [y0 + ys.min(), y0 + ys.max() + 1, x0 + xs.min(), x0 + xs.max() + 1]
                n += 1
                if rects[s, b].tolist() != want:
                    bad += 1
                    if bad < 10:
                        print("rep %d seed %d source %d band %d: library %s, patches %s (box %s)" % (rep, seed, s, b, rects[s, b].tolist(), want, boxes[s, b].tolist()))
print("%d of %d rectangles differ" % (bad, n))
