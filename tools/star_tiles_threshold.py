#!/usr/bin/env python3
"""Where k_render_stars starts to pay: star-only fields of growing frame size, CEL_OPT_STAR_TILES = 0 against 2.  (diagnostic)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import desi_mcmc_amd as cel
from desi_mcmc_amd import _lib, synth
ctx = cel.Context(0)
for H, W, S in ((512, 512, 1000), (768, 768, 1500), (1024, 1024, 2500), (1280, 1280, 4000), (1536, 1536, 5600), (2048, 2048, 10000)):
    f = synth.SyntheticField(ctx, S, 5, H, W, frac_gal=0.0, seed=3)
    tiles = 5 * ((W + 31) // 32) * ((H + 63) // 64)
    out = []
    for mode in (0, 2):
        ctx.set_option(_lib.CEL_OPT_STAR_TILES, mode)
        for _ in range(20):
            f.images.render(f.sources, loglik=True)
        ctx.profile(True)
        for _ in range(100):
            f.images.render(f.sources, loglik=True)
        ms, n, name = ctx.profile_render()
        ctx.profile(False)
        out.append("%s %.4f ms" % (name, ms))
    print("%4d x %4d, %5d stars, %5d tiles: %s | %s" % (H, W, S, tiles, out[0], out[1]))
ctx.set_option(_lib.CEL_OPT_STAR_TILES, 1)
