#!/usr/bin/env python3
"""kernel time of the resident photon split on the benchmark field (diagnostic)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import desi_mcmc_amd as cel
from desi_mcmc_amd import synth
ctx = cel.Context(0)
f = synth.SyntheticField.from_config(ctx, sys.argv[1] if len(sys.argv) > 1 else "mixed10k_2048")
f.images.photon_split_resident(f.sources, seed=1)
ctx.profile(True)
for i in range(3):
    f.images.photon_split_resident(f.sources, seed=2 + i)
ms, n = ctx.profile_get("split")
print("split kernel: %.3f ms (mean of %d launches)" % (ms, n))
