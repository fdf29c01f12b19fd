"""where the host engine's exact location step spends its time (cProfile, configs[4] size)"""
import sys, os, cProfile, pstats
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np
import desi_mcmc_amd as cel
from desi_mcmc_amd import celeste_mcmc, synth
ctx = cel.Context(0)
f = synth.SyntheticField(ctx, 10000, 5, 2048, 2048, frac_gal=0.5, seed=3)
gf = celeste_mcmc.GibbsField(f.images, list(range(5)), f.bands[:, 2], f.bands[:, 1], 2048 * 2048)
g = celeste_mcmc.ModelGibbs([gf], f.src["type"], f.src["radec"], f.flux5(), f.src["shape"], seed=1, conditional=sys.argv[1] if len(sys.argv) > 1 else "exact", engine="host")
g.sweep()
pr = cProfile.Profile()
pr.enable()
for _ in range(2):
    g.sweep()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(22)
