#!/usr/bin/env python3
"""Digest of what THIS build of the library renders (CEL_HIP_LIBRARY selects it): sha256 of the model images and the per-band
log-likelihoods of the BASELINE-size field and of N random fields, one line each -- two builds that must agree bit for bit
(e.g. -DHW_SYNC_BARRIER against the shipped wave-barrier form of k_render_hw) print the same lines.

    CEL_HIP_LIBRARY=$PWD/tools/bin/x.so python tools/dbg/render_digest.py [N] > a.txt ; python tools/dbg/render_digest.py [N] > b.txt ; cmp a.txt b.txt
"""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import desi_mcmc_amd as cel  # noqa: E402
from desi_mcmc_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
ctx = cel.Context(0)


def digest(f, tag):
    ll, llb = f.images.render(f.sources, loglik=True)
    lam = f.images.model_images()
    print(tag, hashlib.sha256(lam.tobytes()).hexdigest()[:32], hashlib.sha256(np.asarray(llb, dtype=np.float64).tobytes()).hexdigest()[:16], repr(ll))


digest(synth.SyntheticField.from_config(ctx, "mixed10k_2048"), "mixed10k_2048")
rs = np.random.RandomState(7)
for i in range(n):
    S = int(rs.randint(50, 3000))
    H, W = int(rs.randint(60, 700)), int(rs.randint(60, 700))
    B = int(rs.randint(1, 5))
    f = synth.SyntheticField(ctx, S, B, H, W, frac_gal=float(rs.uniform(0.1, 0.9)), seed=int(rs.randint(1 << 30)))
    for parts in (0, 1, 2, 4):
        ctx.set_option(cel._lib.CEL_OPT_TILE_PARTS, parts)
        digest(f, "field %d S %d B %d %dx%d parts %d" % (i, S, B, H, W, parts))
    ctx.set_option(cel._lib.CEL_OPT_TILE_PARTS, 0)
