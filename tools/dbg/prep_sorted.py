"""k_prep on the benchmark catalogue as it comes (types mixed at random) and sorted by type (galaxies first): what all-galaxy
waves would buy."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import desi_mcmc_amd as cel
from desi_mcmc_amd import synth
ctx = cel.default_context(0)
f = synth.SyntheticField.from_config(ctx, "mixed10k_2048")
for name, order in (("as it comes", np.arange(f.S)), ("galaxies first", np.argsort(-f.src["type"], kind="stable"))):
    f.sources.set(f.src["type"][order], f.src["radec"][order], f.src["counts"][order], f.src["shape"][order])
    for _ in range(20):
        f.images.render(f.sources, loglik=True)
    ctx.profile(1)
    for _ in range(100):
        ll, _ = f.images.render(f.sources, loglik=True)
    tp, n = ctx.profile_get("prep")
    tr = ctx.profile_render()[0]
    tb, _ = ctx.profile_get("bin")
    ctx.profile(False)
    print("%-15s k_prep %.4f ms  k_bin %.4f  k_render %.4f   ll %.6f" % (name, tp, tb, tr, ll))
