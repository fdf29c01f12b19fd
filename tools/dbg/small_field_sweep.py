"""what a Gibbs sweep costs on small fields (latency-bound): 51 x 51 with 3 sources, 512^2 with 300, 1024^2 with 2 000"""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np
import desi_mcmc_amd as cel
from desi_mcmc_amd import celeste_mcmc, synth
ctx = cel.Context(0)
for S, H in ((3, 51), (300, 512), (2000, 1024)):
    f = synth.SyntheticField(ctx, S, 5, H, H, frac_gal=0.5, seed=3)
    gf = celeste_mcmc.GibbsField(f.images, list(range(5)), f.bands[:, 2], f.bands[:, 1], H * H)
    g = celeste_mcmc.ModelGibbs([gf], f.src["type"], f.src["radec"], f.flux5(), f.src["shape"], seed=1)
    for _ in range(5):
        g.sweep(); g.log_likelihood()
    for k in g.timing: g.timing[k] = 0
    n = 30
    t0 = time.perf_counter()
    for _ in range(n):
        g.sweep(); g.log_likelihood()
    dt = (time.perf_counter() - t0) / n * 1e3
    print("S = %5d, %4d^2: %.2f ms per sweep; split + sky %.2f, flux %.2f, location %.2f (%.0f rounds, %.0f evaluations per source)" % (
        S, H, dt, g.timing["split"] / n * 1e3, g.timing["flux"] / n * 1e3, g.timing["location"] / n * 1e3, g.timing["rounds"] / n, g.timing["evals"] / n / S))
