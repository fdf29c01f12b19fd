"""conditional="exact" (host engine) at the size of BASELINE configs[4]: time per sweep against the default device sweep"""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np
import desi_mcmc_amd as cel
from desi_mcmc_amd import celeste_mcmc, synth
ctx = cel.Context(0)
S, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (10000, 2048)
f = synth.SyntheticField(ctx, S, 5, H, H, frac_gal=0.5, seed=3)
for kw in (dict(), dict(engine="host"), dict(conditional="exact", engine="host")):
    gf = celeste_mcmc.GibbsField(f.images, list(range(5)), f.bands[:, 2], f.bands[:, 1], H * H)
    g = celeste_mcmc.ModelGibbs([gf], f.src["type"], f.src["radec"], f.flux5(), f.src["shape"], seed=1, **kw)
    g.sweep(); g.log_likelihood()
    for k in g.timing: g.timing[k] = 0
    n = 3
    t0 = time.perf_counter()
    for _ in range(n):
        g.sweep(); ll = g.log_likelihood()
    dt = (time.perf_counter() - t0) / n * 1e3
    print("%-45s %.1f ms per sweep (split + sky %.1f, flux %.1f, location %.1f: %d rounds); log-lik %.6e" % (
        kw or "default (device engine)", dt, g.timing["split"] / n * 1e3, g.timing["flux"] / n * 1e3, g.timing["location"] / n * 1e3, g.timing["rounds"] / n, ll), flush=True)
