#!/usr/bin/env python3
"""Per-sweep times of the flux step and of its host half (gamma_by_stream) on the benchmark field.  (diagnostic)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import desi_mcmc_amd as cel
from desi_mcmc_amd import celeste_mcmc, synth
ctx = cel.Context(0)
f = synth.SyntheticField.from_config(ctx, "mixed10k_2048")
gf = celeste_mcmc.GibbsField(f.images, list(range(f.B)), f.bands[:, 2], f.bands[:, 1], f.H * f.W)
g = celeste_mcmc.ModelGibbs([gf], f.src["type"], f.src["radec"], f.flux5(), f.src["shape"], seed=1)
for _ in range(3):
    g.sweep()
ts = []
for _ in range(30):
    g.resample_photons()
    t0 = time.perf_counter(); g.resample_fluxes(); ts.append((time.perf_counter() - t0) * 1e3)
    g.resample_locations(); g.sweeps += 1
print("flux step ms: min %.2f median %.2f max %.2f" % (min(ts), np.median(ts), max(ts)))
a = 5.0 + np.random.RandomState(0).poisson(800, size=50000)
tg = []
for k in range(30):
    t0 = time.perf_counter(); celeste_mcmc.gamma_by_stream(a, k, np.arange(50000)); tg.append((time.perf_counter() - t0) * 1e3)
print("gamma_by_stream alone ms: min %.2f median %.2f max %.2f" % (min(tg), np.median(tg), max(tg)))
tm = []
for k in range(30):
    t0 = time.perf_counter(); gf.iset.stamp_mass(gf.sset); tm.append((time.perf_counter() - t0) * 1e3)
print("stamp_mass call alone ms: min %.2f median %.2f max %.2f" % (min(tm), np.median(tm), max(tm)))
