#!/usr/bin/env python3
"""Device-resident Gibbs sweep on the benchmark field (BASELINE configs[4] flavour, one GPU).

One sweep = what CelesteBase.resample_model does (CelestePy/models.py:75-83):
  1. Field.resample_photons: photon split of all bands (cel_photon_split, resident form);
  2. per source: a flux step from the attributed photon counts (cel_samples_fetch sums) and a
     location step that scores P proposals per source -- ALL sources in one launch
     (cel_patch_loglik_multi, resident form) -- and picks one by its conditional posterior.
The host does the O(S*P) selection arithmetic only; patches (3.2 GB at config 3) never leave HBM.
Samplers' control flow (slice sampling with step-out, HMC) is out of scope: this measures the
device calls a sweep makes.  Not part of bench.py's contract.

    python tools/bench_gibbs.py [--workload mixed10k_2048] [--sweeps 3] [--proposals 16]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import desi_mcmc_amd as cel  # noqa: E402
from desi_mcmc_amd import synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="mixed10k_2048")
ap.add_argument("--sweeps", type=int, default=3)
ap.add_argument("--proposals", type=int, default=16)
args = ap.parse_args()

ctx = cel.Context(0)
f = synth.SyntheticField.from_config(ctx, args.workload)
S, B, P = f.S, f.B, args.proposals
rs = np.random.RandomState(0)
radec = f.src["radec"].copy()
flux = f.src["flux"].copy()
prop = cel.SourceSet(ctx, S * P, B)
owner = np.repeat(np.arange(S, dtype=np.int32), P)
typ_p = np.repeat(f.src["type"], P)
shape_p = np.repeat(f.src["shape"], P, axis=0)
t_split = t_ll = t_sums = t_host = 0.0
ll0, _ = f.images.render(f.sources, loglik=True)
for sweep in range(args.sweeps):
    t0 = time.perf_counter()
    f.sources.set(f.src["type"], radec, flux / f.bands[None, :, 2] * f.bands[None, :, 1], f.src["shape"])
    noise = f.images.photon_split_resident(f.sources, seed=100 + sweep)
    t1 = time.perf_counter()
    sums = f.images.sample_sums()
    t2 = time.perf_counter()
    # flux step: Gamma(a0 + photons, 1 / (b0 + kappa/calib))  (sources.py:327-345, unit stamp mass)
    flux = rs.gamma(1.0 + sums, 1.0 / (1e-3 + (f.bands[:, 1] / f.bands[:, 2])[None, :]))
    us = np.repeat(radec, P, axis=0) + rs.normal(0.0, 2e-5, size=(S * P, 2))
    us[::P] = radec
    counts_p = np.repeat(flux / f.bands[None, :, 2] * f.bands[None, :, 1], P, axis=0)
    t3 = time.perf_counter()
    prop.set(typ_p, us, counts_p, shape_p)
    ll = f.images.patch_loglik_resident(prop, owner).reshape(S, P)
    t4 = time.perf_counter()
    w = np.exp(ll - ll.max(axis=1, keepdims=True))
    w /= w.sum(axis=1, keepdims=True)
    pick = (w.cumsum(axis=1) > rs.rand(S, 1)).argmax(axis=1)
    radec = us.reshape(S, P, 2)[np.arange(S), pick]
    t5 = time.perf_counter()
    t_split += t1 - t0
    t_sums += t2 - t1
    t_ll += t4 - t3
    t_host += (t3 - t2) + (t5 - t4)
f.sources.set(f.src["type"], radec, flux / f.bands[None, :, 2] * f.bands[None, :, 1], f.src["shape"])
ll1, _ = f.images.render(f.sources, loglik=True)
n = args.sweeps
tot = t_split + t_sums + t_ll + t_host
print(json.dumps({
    "workload": args.workload, "sources": S, "bands": B, "proposals_per_source": P, "sweeps": n,
    "ms_per_sweep": tot / n * 1e3,
    "ms_photon_split": t_split / n * 1e3, "ms_patch_sums": t_sums / n * 1e3,
    "ms_conditional_loglik": t_ll / n * 1e3, "ms_host_selection": t_host / n * 1e3,
    "conditional_loglik_evals_per_s": S * P * n / t_ll,
    "source_updates_per_s": S * n / tot,
    "field_loglik_before": ll0, "field_loglik_after": ll1}))
