#!/usr/bin/env python3
"""Diagnostic: 30 Gibbs sweeps of the benchmark field with the per-phase host timings (no trace render).  For A/B
runs on ONE box: build two libraries, copy each over desi-mcmc_amd/libceleste_hip.so in turn, run this after each."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import desi_mcmc_amd as cel
from desi_mcmc_amd import synth, celeste_mcmc
ctx = cel.Context(0)
field = synth.SyntheticField.from_config(ctx, "mixed10k_2048", seed=42)
S, B, H, W, fg = synth.CONFIGS["mixed10k_2048"]
gf = celeste_mcmc.GibbsField(field.images, list(range(B)), field.bands[:, 2], field.bands[:, 1], H * W)
g = celeste_mcmc.ModelGibbs([gf], field.src["type"], field.src["radec"], field.flux5(), field.src["shape"], seed=1,
                            slice_args=dict(step_out=False, sigma=0.001))
for _ in range(3):
    g.sweep()
for k in g.timing: g.timing[k] = 0
t0 = time.perf_counter()
n = 30
for _ in range(n):
    g.sweep()
dt = time.perf_counter() - t0
print("sweep %.2f ms  split %.2f flux %.2f location %.2f  ll %.6f" % (dt / n * 1e3, g.timing["split"] / n * 1e3, g.timing["flux"] / n * 1e3, g.timing["location"] / n * 1e3, g.log_likelihood()), flush=True)
