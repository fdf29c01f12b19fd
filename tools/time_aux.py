#!/usr/bin/env python3
"""Timings of the auxiliary device calls (photon split, E-step reductions, stamps) on the
benchmark field.  Diagnostic; not part of bench.py's contract."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import desi_mcmc_amd as cel  # noqa: E402
from desi_mcmc_amd import synth  # noqa: E402

ctx = cel.Context(0)
name = sys.argv[1] if len(sys.argv) > 1 else "mixed10k_2048"
f = synth.SyntheticField.from_config(ctx, name)


def timed(label, fn, n=3):
    fn()
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    dt = (time.perf_counter() - t0) / n
    print("%-28s %9.2f ms" % (label, dt * 1e3))
    return out


timed("render + loglik", lambda: f.images.render(f.sources, loglik=True), 10)
timed("photon split (5 bands)", lambda: f.images.photon_split(f.sources, seed=1), 2)
timed("E-step statistics", lambda: f.images.estep_stats(f.sources), 2)
timed("stamps, band r, all sources", lambda: f.images.stamps(f.sources, 2, scaled=True), 2)

ctx.profile(True)
f.images.photon_split(f.sources, seed=1)
print("k_photon_split kernel alone: %.2f ms" % ctx.profile_get("split")[0])
ctx.profile(True)
f.images.estep_stats(f.sources)
print("k_estep_src kernel alone: %.2f ms" % ctx.profile_get("estep")[0])

# conditional log-likelihoods of 16 proposals per source against the resident split
P = 16
prop = cel.SourceSet(ctx, f.S * P, f.B)
owner = np.repeat(np.arange(f.S, dtype=np.int32), P)
rs = np.random.RandomState(0)
us = np.repeat(f.src["radec"], P, axis=0) + rs.normal(0.0, 2e-5, size=(f.S * P, 2))
prop.set(np.repeat(f.src["type"], P), us, np.repeat(f.src["counts"], P, axis=0), np.repeat(f.src["shape"], P, axis=0))
f.images.photon_split_resident(f.sources, seed=1)
timed("resident split (5 bands)", lambda: f.images.photon_split_resident(f.sources, seed=1), 3)
ctx.profile(True)
f.images.photon_split_resident(f.sources, seed=1)
print("  split kernel alone: %.2f ms, render kernels: %.2f ms" % (ctx.profile_get("split")[0], ctx.profile_get("render")[0]))
timed("conditional ll, %d proposals" % (f.S * P), lambda: f.images.patch_loglik_resident(prop, owner), 3)
ctx.profile(True)
f.images.patch_loglik_resident(prop, owner)
print("  k_prep: %.2f ms, k_patch_ll: %.2f ms" % (ctx.profile_get("prep")[0], ctx.profile_get("patch_ll")[0]))
ctx.profile(False)
