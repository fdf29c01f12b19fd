#!/usr/bin/env python3
"""Gibbs sweeps of CelestePy's runnable sampler on a synthetic field, entirely on the HIP path.

What CelesteBase.resample_model does (CelestePy/models.py:75-83):
    for every field:   Field.resample_photons(srcs)    -> device photon split + the sky levels' Gamma draws
    for every source:  Source.resample()               -> flux Gamma conditionals (sources.py:327-345) and the
                                                          location by slice sampling (sources.py:308-319)
Two ways to run it, both shown here:
    model.resample_model(n_sweeps)     all sources in lock-step on the device (celeste_mcmc.ModelGibbs)
    model.resample_sources()           one Source.resample() per object, as the reference loops

    python examples/gibbs_sweep.py [--sources 200] [--size 512] [--sweeps 5]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import desi_mcmc_amd as cel  # noqa: E402
from desi_mcmc_amd import models, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--sources", type=int, default=200)
ap.add_argument("--size", type=int, default=512)
ap.add_argument("--sweeps", type=int, default=5)
args = ap.parse_args()

BANDS = ["u", "g", "r", "i", "z"]
H = W = args.size
ctx = cel.default_context(0)
f = synth.SyntheticField(ctx, args.sources, 5, H, W, frac_gal=0.5, seed=1)
rec = {k: None for k in ()}
imgs = []
for b, name in enumerate(BANDS):
    band = f.bands[b]
    imgs.append(cel.FitsImage(name, f.nelec[b], epsilon=band[0], kappa=band[1], calib=band[2], weights=band[3:6],
                              means=band[6:12].reshape(3, 2), covars=band[12:24].reshape(3, 2, 2),
                              rho_n=band[24:26], phi_n=band[26:28], Ups_n=band[28:32].reshape(2, 2)))
params = [cel.SrcParams(u=f.src["radec"][s], a=int(f.src["type"][s]), fluxes=f.src["flux"][s].copy(),
                        theta=f.src["shape"][s, 0], sigma=f.src["shape"][s, 1], phi=f.src["shape"][s, 2],
                        rho=f.src["shape"][s, 3]) for s in range(args.sources)]
model = models.Celeste()
model.initialize_sources(init_src_params=params)
eps_true = [im.epsilon for im in imgs]
model.add_field(dict(zip(BANDS, imgs)))
for im, e in zip(imgs, eps_true):
    im.epsilon = e

t0 = time.perf_counter()
ll0 = model.log_likelihood()
t1 = time.perf_counter()
print("field log-likelihood %.6e   (%.1f ms incl. the first upload)" % (ll0, (t1 - t0) * 1e3))

# the reference's call passes step=0.001 degrees, which its slicesample ignores (sigma stays 1.0 degree):
# give the intent explicitly
slice_args = dict(step_out=False, sigma=0.001)
model.resample_model(1, seed=3, slice_args=slice_args)            # builds the device sampler, one sweep
t2 = time.perf_counter()
model.resample_model(args.sweeps, slice_args=slice_args)
t3 = time.perf_counter()
g = model.gibbs()
print("%d sweeps of %d sources x 5 bands x %dx%d: %.1f ms per sweep (photon split %.1f, fluxes %.1f, locations %.1f; "
      "%.0f slice rounds, %.1f likelihood evaluations per source and sweep)"
      % (args.sweeps, args.sources, H, W, (t3 - t2) / args.sweeps * 1e3, g.timing["split"] / g.sweeps * 1e3,
         g.timing["flux"] / g.sweeps * 1e3, g.timing["location"] / g.sweeps * 1e3, g.timing["rounds"] / g.sweeps,
         g.timing["evals"] / g.sweeps / max(args.sources, 1)))
print("field log-likelihood after the sweeps %.6e" % model.log_likelihood())

# the per-object form (Field.resample_photons hands every Source its sample patches, then one
# Source.resample() per object): the same conditionals, a Python loop over the sources
t4 = time.perf_counter()
for field in model.field_list:
    field.resample_photons(model.srcs, seed=11, rng=np.random.RandomState(0))
model.resample_sources(rng=np.random.RandomState(1))
t5 = time.perf_counter()
print("one per-object sweep (Source.resample x %d): %.1f ms" % (args.sources, (t5 - t4) * 1e3))
print("field log-likelihood %.6e" % model.log_likelihood())
