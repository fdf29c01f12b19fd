#!/usr/bin/env python3
"""Timing-only ablations of k_render_hw (CEL_OPT_DEBUG): where a star field's kernel time goes.
    make -C desi-mcmc_amd/csrc ablate && python tools/ablate_render.py [--workload stars10k_2048]
The switches exist only in the -DCEL_ABLATE build of the library (tools/bin/libceleste_hip_ablate.so, built
here when missing); the shipped library refuses them."""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import desi_mcmc_amd as cel  # noqa: E402
from desi_mcmc_amd import _lib, synth  # noqa: E402

subprocess.check_call(["make", "-C", os.path.join(ROOT, "desi-mcmc_amd", "csrc"), "-s", "ablate"])
_lib.LIB_PATH = os.path.join(ROOT, "tools", "bin", "libceleste_hip_ablate.so")      # before the first call loads it

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="stars10k_2048")
ap.add_argument("--steps", type=int, default=30)
args = ap.parse_args()
ctx = cel.Context(0)
f = synth.SyntheticField.from_config(ctx, args.workload)
for name, bits in (("full", 0), ("no star walk", 1), ("no star seeds + walk", 2), ("no log", 4),
                   ("no seeds/walk/log", 6), ("no star pass at all", 32 + 4), ("... and no global epilogue traffic", 32 + 16 + 4),
                   ("only epilogue stores skipped", 16), ("launch + header only", 8)):
    ctx.set_option(_lib.CEL_OPT_DEBUG, bits)
    for _ in range(3):
        f.images.render(f.sources, loglik=True)
    ctx.profile(True)
    for _ in range(args.steps):
        f.images.render(f.sources, loglik=True)
    ms, n = ctx.profile_get("render")
    ctx.profile(False)
    print("%-24s k_render %.4f ms" % (name, ms))
ctx.set_option(_lib.CEL_OPT_DEBUG, 0)
