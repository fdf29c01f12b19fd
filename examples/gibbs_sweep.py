#!/usr/bin/env python3
"""One Gibbs sweep of CelestePy's runnable sampler on a synthetic field, entirely on the HIP path.

What CelesteBase.resample_model does (CelestePy/models.py:75-83):
    for every field:   Field.resample_photons(srcs)           -> device photon split
    for every source:  Source.resample()                       -> here: a flux Gibbs step and a
                                                                  grid "slice" over the location,
                                                                  scored by log_likelihood_batch
The samplers themselves (slice sampling with step-out, HMC) are host control flow outside the
build's scope; this script only shows the device-side calls they make, with timings.

    python examples/gibbs_sweep.py [--sources 200] [--size 512]
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
ap.add_argument("--proposals", type=int, default=32)
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
noise = model.field_list[0].resample_photons(model.srcs, seed=11, rng=np.random.RandomState(0))
t2 = time.perf_counter()
print("field log-likelihood %.6e   (%.1f ms incl. first upload)" % (ll0, (t1 - t0) * 1e3))
print("photon split of %d sources x 5 bands x %dx%d: %.1f ms; sky photons per band %s"
      % (args.sources, H, W, (t2 - t1) * 1e3, {k: int(v) for k, v in noise.items()}))

rs = np.random.RandomState(2)
n_eval, t_ll = 0, 0.0
for src in model.srcs:
    # flux step: conjugate Gamma given the attributed photons (sources.py:327-345)
    counts = {b: 0.0 for b in BANDS}
    for samp, im, _ in src.sample_image_list:
        counts[im.band] += samp.data.sum()
    src.params.fluxes = np.array([rs.gamma(1.0 + counts[b], 1.0 / (1e-3 + im.kappa / im.calib))
                                  for b, im in zip(BANDS, imgs)])
    # location step: score a cloud of proposals around the current position in ONE launch
    us = src.params.u[None, :] + rs.normal(0.0, 2e-5, size=(args.proposals, 2))
    us[0] = src.params.u
    ta = time.perf_counter()
    ll = src.log_likelihood_batch(us=us)
    t_ll += time.perf_counter() - ta
    n_eval += len(us)
    p = np.exp(ll - ll.max())
    src.params.u = us[rs.choice(len(us), p=p / p.sum())]
t3 = time.perf_counter()
print("per-source updates: %d conditional log-likelihood evaluations in %.1f ms of device calls "
      "(%.1f us each); whole source loop %.1f ms" % (n_eval, t_ll * 1e3, t_ll / n_eval * 1e6, (t3 - t2) * 1e3))
print("field log-likelihood after the sweep %.6e" % model.log_likelihood())
