#!/usr/bin/env python3
"""How sparse are the sample patches of the benchmark field?  (diagnostic for k_patch_ll_hw's crop)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import desi_mcmc_amd as cel
from desi_mcmc_amd import synth
ctx = cel.Context(0)
f = synth.SyntheticField.from_config(ctx, "mixed10k_2048")
f.images.photon_split_resident(f.sources, seed=1)
boxes, offs, data = f.images.fetch_samples()
S, B = boxes.shape[:2]
area = nzarea = chunkrows = nzchunkrows = chunks = nzchunks = 0
tot_ph = []
for s in range(0, S, 7):
    for b in range(B):
        y0, y1, x0, x1 = boxes[s, b]
        if y1 <= y0 or x1 <= x0:
            continue
        z = data[offs[s * B + b]:offs[s * B + b + 1]].reshape(y1 - y0, x1 - x0)
        area += z.size
        tot_ph.append(z.sum())
        ys, xs = np.nonzero(z)
        if len(ys) == 0:
            continue
        zz = z[ys.min():ys.max() + 1, xs.min():xs.max() + 1]
        nzarea += zz.size
        for cy in range(0, zz.shape[0], 64):
            for cx in range(0, zz.shape[1], 32):
                c = zz[cy:cy + 64, cx:cx + 32]
                rows = np.nonzero(c.any(axis=1))[0]
                chunks += 1
                chunkrows += c.shape[0]
                if len(rows):
                    nzchunks += 1
                    nzchunkrows += rows.max() - rows.min() + 1
print("box area %.3e, nz-rectangle area %.3e (%.2f)" % (area, nzarea, nzarea / area))
print("chunks %d, with photons %d; chunk rows %d, rows inside per-chunk photon range %d (%.2f)" % (chunks, nzchunks, chunkrows, nzchunkrows, nzchunkrows / chunkrows))
print("photons per patch: median %.0f, p10 %.0f, p90 %.0f" % tuple(np.percentile(tot_ph, [50, 10, 90])))
