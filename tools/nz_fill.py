#!/usr/bin/env python3
"""Fill of the photon rectangles of the benchmark field's sample patches: how many pixels of the rectangle a
conditional likelihood walks actually hold a photon (diagnostic for a sparse evaluation path of k_patch_ll_hw)."""
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
rows = []
for s in range(0, S, 5):
    for b in range(B):
        y0, y1, x0, x1 = boxes[s, b]
        if y1 <= y0 or x1 <= x0:
            continue
        z = data[offs[s * B + b]:offs[s * B + b + 1]].reshape(y1 - y0, x1 - x0)
        ys, xs = np.nonzero(z)
        if len(ys) == 0:
            continue
        nza = (ys.max() - ys.min() + 1) * (xs.max() - xs.min() + 1)
        nchunk = -(-(ys.max() - ys.min() + 1) // 64) * -(-(xs.max() - xs.min() + 1) // 32)
        rows.append((f.src["type"][s], len(ys), nza, nchunk, z.sum()))
r = np.array(rows, dtype=np.float64)
for t, name in ((0, "stars"), (1, "galaxies")):
    q = r[r[:, 0] == t]
    fill = q[:, 1] / q[:, 2]
    K = 3 if t == 0 else 42
    dense = q[:, 3] * 2048 * K * 4.4 / 0.6          # ~instr per comp-pixel of a chunk, lanes 60 % useful
    direct = q[:, 1] * K * 14.0
    print("%s: %d patches; nnz median %.0f; fill p10/50/90 %.2f %.2f %.2f; mean fill %.2f (area-weighted %.2f)"
          % (name, len(q), np.median(q[:, 1]), *np.percentile(fill, [10, 50, 90]), fill.mean(), q[:, 1].sum() / q[:, 2].sum()))
    print("   modelled cost: dense %.3e, direct-at-photons %.3e, best of the two per patch %.3e"
          % (dense.sum(), direct.sum(), np.minimum(dense, direct).sum()))
