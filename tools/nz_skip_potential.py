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
rule = {256: [0, 0], 128: [0, 0], 64: [0, 0]}      # the rule a kernel can apply: row-major trips, bounds over the trip's rectangle


def qmin_rect(ic, dx1, dx2, dy1, dy2):
    """min over the rectangle of the positive definite forms ic[k] (offsets relative to each component's centre)"""
    a, b, c = ic[:, 0, 0], ic[:, 0, 1], ic[:, 1, 1]
    inside = (dx1 <= 0) & (dx2 >= 0) & (dy1 <= 0) & (dy2 >= 0)
    best = np.full(len(a), np.inf)
    for xe in (dx1, dx2):          # vertical edges: minimise over y
        y = np.clip(-b * xe / c, dy1, dy2)
        best = np.minimum(best, a * xe * xe + 2 * b * xe * y + c * y * y)
    for ye in (dy1, dy2):
        x = np.clip(-b * ye / a, dx1, dx2)
        best = np.minimum(best, a * x * x + 2 * b * x * ye + c * ye * ye)
    return np.where(inside, 0.0, best)


def qmax_rect(ic, dx1, dx2, dy1, dy2):
    a, b, c = ic[:, 0, 0], ic[:, 0, 1], ic[:, 1, 1]
    return np.max([a * x * x + 2 * b * x * y + c * y * y for x in (dx1, dx2) for y in (dy1, dy2)], axis=0)

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
        la = np.log(pis) - 0.5 * np.log(np.linalg.det(covs))
        xa, xb = X[:, 0].min(), X[:, 0].max()
        for n in rule:
            for i0 in range(0, len(X), n):          # np.nonzero order = row-major = the list order
                ya, yb = X[i0, 1], X[min(i0 + n, len(X)) - 1, 1]
                U = la - 0.5 * qmin_rect(ic, xa - means[:, 0], xb - means[:, 0], ya - means[:, 1], yb - means[:, 1])
                Lo = np.max(la - 0.5 * qmax_rect(ic, xa - means[:, 0], xb - means[:, 0], ya - means[:, 1], yb - means[:, 1]))
                cnt = min(n, len(X) - i0)
                rule[n][0] += cnt * len(la)
                rule[n][1] += cnt * np.sum(U < Lo - 32.0)
print("galaxy (component, photon) pairs: %d; negligible (< e^-32 of the pixel's value): %.3f; skippable per radius-sorted trip of 256: %.3f"
      % (tot, negl / tot, trip_skipped / trip_pairs))
for n in rule:
    print("row-major trips of %d, rectangle bounds (component max < e^-32 x the largest component minimum): skipped %.3f" % (n, rule[n][1] / max(rule[n][0], 1)))
