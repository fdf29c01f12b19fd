#!/usr/bin/env python3
"""Timing-only ablations of k_photon_split_hw (CEL_OPT_DEBUG bits 1 = no draws, 2 = every draw 1, 4 = no stamp walk):
where the photon split's kernel time goes on the benchmark field.
    make -C desi-mcmc_amd/csrc ablate && python tools/ablate_split.py
The switches exist only in the -DCEL_ABLATE build (tools/bin/libceleste_hip_ablate.so); results are WRONG when one is set."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import desi_mcmc_amd as cel  # noqa: E402
from desi_mcmc_amd import _lib, synth  # noqa: E402

subprocess.check_call(["make", "-C", os.path.join(ROOT, "desi-mcmc_amd", "csrc"), "-s", "ablate"])
_lib.LIB_PATH = os.path.join(ROOT, "tools", "bin", "libceleste_hip_ablate.so")
ctx = cel.Context(0)
f = synth.SyntheticField.from_config(ctx, "mixed10k_2048")
for name, bits in (("full", 0), ("no draws (stamps, LDS bookkeeping, patch stores)", 1), ("every queued draw = 1 (uniform + queue, no sampler)", 2),
                   ("no stamp walk", 4), ("no stamp walk, no draws", 5)):
    ctx.set_option(_lib.CEL_OPT_DEBUG, bits)
    f.images.photon_split_resident(f.sources, seed=1)
    ctx.profile(True)
    for k in range(5):
        f.images.photon_split_resident(f.sources, seed=2 + k)
    ms, n = ctx.profile_get("split")
    ctx.profile(False)
    print("%-58s k_photon_split_hw %.3f ms" % (name, ms))
ctx.set_option(_lib.CEL_OPT_DEBUG, 0)
