#!/usr/bin/env python3
"""How many (component, photon) pairs of the galaxies' photon lists are negligible (below e^-32 of the pixel's own value)?
And how many would a per-trip test skip if the lists were sorted by radius (trips of 256)?  (diagnostic)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import desi_mcmc_amd as cel
from desi_mcmc_amd import synth
from oracle import oracle as orc
ctx = cel.Context(0)
f = synth.SyntheticField.from_config(ctx, "mixed10k_2048")
f.images.photon_split_resident(f.sources, seed=1)
boxes, offs, data = f.images.fetch_samples()
S, B = boxes.shape[:2]
bands = f.bands.copy()
for b in range(B):
    bands[b, 36] = f.images.band(b)[36]
tot = negl = trip_pairs = trip_skipped = 0
gal = np.nonzero(f.src["type"] == 1)[0][::25]
for s in gal:
    for b in (2,):
        y0, y1, x0, x1 = boxes[s, b]
        if y1 <= y0: continue
        z = data[offs[s * B + b]:offs[s * B + b + 1]].reshape(y1 - y0, x1 - x0)
        ys, xs = np.nonzero(z)
        if len(ys) == 0: continue
        pis, means, covs, pxy, tinv = orc.galaxy_table(bands[b], f.src["shape"][s], f.src["radec"][s])
        X = np.column_stack([xs + x0, ys + y0]).astype(float)
        d = X[:, None, :] - means[None, :, :]
        ic = np.linalg.inv(covs)
        q = np.einsum("nki,kij,nkj->nk", d, ic, d)
        lg = np.log(pis)[None, :] - 0.5 * np.log(np.linalg.det(covs))[None, :] - 0.5 * q
        m = np.log(np.sum(np.exp(lg - lg.max(axis=1, keepdims=True)), axis=1)) + lg.max(axis=1)
        small = lg < (m[:, None] - 32.0)
        tot += small.size; negl += small.sum()
        r = np.hypot(X[:, 0] - pxy[0], X[:, 1] - pxy[1])
        order = np.argsort(r)
        for i0 in range(0, len(order), 256):
            idx = order[i0:i0 + 256]
            sk = np.all(small[idx], axis=0)       # a component negligible for every photon of the trip
            trip_pairs += len(idx) * small.shape[1]
            trip_skipped += len(idx) * sk.sum()
print("galaxy (component, photon) pairs: %d; negligible (< e^-32 of the pixel's value): %.3f; skippable per radius-sorted trip of 256: %.3f"
      % (tot, negl / tot, trip_skipped / trip_pairs))
