#!/usr/bin/env python3
"""cProfile of ModelGibbs.sweep on the benchmark field: where the host's share of a sweep goes.  (diagnostic)"""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import desi_mcmc_amd as cel
from desi_mcmc_amd import celeste_mcmc, synth
ctx = cel.Context(0)
f = synth.SyntheticField.from_config(ctx, "mixed10k_2048")
gf = celeste_mcmc.GibbsField(f.images, list(range(f.B)), f.bands[:, 2], f.bands[:, 1], f.H * f.W)
g = celeste_mcmc.ModelGibbs([gf], f.src["type"], f.src["radec"], f.flux5(), f.src["shape"], seed=1)
for _ in range(3):
    g.sweep(); g.log_likelihood()
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    g.sweep(); g.log_likelihood()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)
