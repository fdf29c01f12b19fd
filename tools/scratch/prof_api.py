import os, sys, time, cProfile, pstats
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import desi_mcmc_amd as cel
from desi_mcmc_amd import synth, celeste
ctx = cel.Context(0)
field = synth.SyntheticField.from_config(ctx, "mixed10k_2048")
src = field.src
imgs = synth.fits_images(field)
cat = cel.SrcCatalog((src["type"] == 1).astype(np.int64), src["radec"], field.flux5(), src["shape"])
for _ in range(5):
    celeste.celeste_likelihood_multi_image(cat, imgs)
t0 = time.perf_counter()
for _ in range(100):
    celeste.celeste_likelihood_multi_image(cat, imgs)
print("api ms", (time.perf_counter() - t0) / 100 * 1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(200):
    celeste.celeste_likelihood_multi_image(cat, imgs)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
