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
print("k_photon_split kernel alone: %.2f ms" % ctx.profile_get("stamps")[0])
ctx.profile(True)
f.images.estep_stats(f.sources)
print("k_estep_src kernel alone: %.2f ms" % ctx.profile_get("stamps")[0])
