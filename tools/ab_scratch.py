#!/usr/bin/env python3
"""Device time of the three kernels that carry spilled registers (k_photon_split_hw, k_estep_tiles, k_patch_ll_hw<3>) on the
benchmark field, with the library named by CEL_HIP_LIBRARY: tools/ab_scratch.sh runs it for the shipped library and for the
zero-scratch build (make -C desi-mcmc_amd/csrc noscratch)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import desi_mcmc_amd as cel  # noqa: E402
from desi_mcmc_amd import _lib, synth  # noqa: E402

ctx = cel.Context(0)
f = synth.SyntheticField.from_config(ctx, "mixed10k_2048")
out = {}
for rep in range(3):
    f.images.render(f.sources, loglik=True)
    ctx.profile(True)
    for _ in range(5):
        f.images.render(f.sources, loglik=True)            # (a trace render before each split: the short way to the totals)
        f.images.photon_split_resident(f.sources, seed=3)
    out.setdefault("k_photon_split_hw", []).append(ctx.profile_get("split")[0])
    ctx.profile(True)
    for _ in range(5):
        f.images.estep_stats(f.sources)
    out.setdefault("k_estep_tiles (+ gather)", []).append(ctx.profile_get("estep")[0])
    ctx.set_option(_lib.CEL_OPT_SPLIT_REUSE, 0)           # the mass kernel proper, not the split's sums
    ctx.profile(True)
    for _ in range(5):
        f.images.stamp_mass(f.sources)
    out.setdefault("k_patch_ll_hw<3>", []).append(ctx.profile_get("mass")[0])
    ctx.set_option(_lib.CEL_OPT_SPLIT_REUSE, 2)
    ctx.profile(False)
print(os.path.basename(_lib.LIB_PATH), {k: [round(x, 4) for x in v] for k, v in out.items()})
