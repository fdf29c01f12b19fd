"""stress of the tile-parts hand-off: the same strip rendered 20 000 times per setting, every launch's per-band sums compared bit for bit"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np
import desi_mcmc_amd as cel
from desi_mcmc_amd import synth
ctx = cel.Context(0)
f = synth.SyntheticField.from_config(ctx, "mixed10k_2048", seed=42)
for parts in (2, 4):
    ctx.set_option(cel._lib.CEL_OPT_TILE_PARTS, parts)
    strip = cel.ImageSet(ctx, f.bands, 256, f.W, nelec=np.ascontiguousarray(f.nelec[:, 768:1024]))
    strip.set_window(768, f.H)
    ll0, llb0 = strip.render(f.sources, loglik=True)
    lam0 = strip.model_images()
    bad = 0
    for k in range(20000):
        ll, llb = strip.render(f.sources, loglik=True)
        if not np.array_equal(llb, llb0):
            bad += 1
    same_img = np.array_equal(lam0, strip.model_images())
    print("parts", parts, "launches 20000, differing", bad, "image identical", same_img, flush=True)
