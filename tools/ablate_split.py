#!/usr/bin/env python3
"""Timing-only ablations of k_photon_split_hw (CEL_OPT_DEBUG in the -DCEL_ABLATE build): where the split's time goes.
    python tools/ablate_split.py [--workload mixed10k_2048]
Results are WRONG when a switch is set; the shipped library refuses them."""
import argparse, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import desi_mcmc_amd as cel  # noqa: E402
from desi_mcmc_amd import _lib, synth  # noqa: E402

subprocess.check_call(["make", "-C", os.path.join(ROOT, "desi-mcmc_amd", "csrc"), "-s", "ablate"])
_lib.LIB_PATH = os.path.join(ROOT, "tools", "bin", "libceleste_hip_ablate.so")
ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="mixed10k_2048")
ap.add_argument("--steps", type=int, default=5)
args = ap.parse_args()
ctx = cel.Context(0)
f = synth.SyntheticField.from_config(ctx, args.workload)
for name, bits in (("full", 0), ("full, stars through the component table (the round-5 path)", 1024), ("no draws at all (every pixel fast, no uniform)", 1), ("sampler replaced by z = 1", 2), ("no stamp walk", 4),
                   ("no walk, no draws", 5)):
    ctx.set_option(_lib.CEL_OPT_DEBUG, bits)
    for _ in range(2):
        f.images.photon_split_resident(f.sources, seed=3)
    ctx.profile(True)
    for _ in range(args.steps):
        f.images.photon_split_resident(f.sources, seed=3)
    ms, n = ctx.profile_get("split")
    ctx.profile(False)
    print("%-60s k_photon_split_hw %.3f ms" % (name, ms))
# work counters of the same build (returned in place of the noise sums)
for name, bits in (("queued draws (pixels that failed the first test)", 8), ("sampler trips of 64 lanes", 16), ("(source, half-tile) pairs walked", 32),
                   ("pairs with a non-empty queue", 512), ("draws by BTPE", 64), ("queued draws that left a photon", 128), ("first-pass steps (row pairs)", 256)):
    ctx.set_option(_lib.CEL_OPT_DEBUG, bits)
    tot = f.images.photon_split_resident(f.sources, seed=3).sum()
    print("%-52s %.4e" % (name, tot))
ctx.set_option(_lib.CEL_OPT_DEBUG, 0)
