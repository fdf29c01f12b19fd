import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import desi_mcmc_amd as cel
from desi_mcmc_amd import synth, celeste_mcmc, field as F
ctx = cel.Context(0)
fld = synth.SyntheticField.from_config(ctx, "mixed10k_2048", seed=42)
S, B, H, W, fg = synth.CONFIGS["mixed10k_2048"]
gf = celeste_mcmc.GibbsField(fld.images, list(range(B)), fld.bands[:, 2], fld.bands[:, 1], H * W)
g = celeste_mcmc.ModelGibbs([gf], fld.src["type"], fld.src["radec"], fld.flux5(), fld.src["shape"], seed=1, slice_args=dict(step_out=False, sigma=0.001))
for _ in range(3):
    g.sweep(); g.log_likelihood()
import collections
acc = collections.Counter()
def wrap(obj, name):
    fn = getattr(obj, name)
    def w(*a, **k):
        t = time.perf_counter(); r = fn(*a, **k); acc[name] += time.perf_counter() - t; return r
    setattr(obj, name, w)
for nm in ("sample_sums", "sample_box_areas", "stamp_mass_begin", "stamp_mass_end", "photon_split_resident", "slice_locations", "render", "set_epsilon"):
    wrap(gf.iset, nm)
wrap(gf.iset.ctx, "gamma_streams")
wrap(g, "_sources"); wrap(g, "_resample_sky"); wrap(g, "resample_fluxes"); wrap(g, "resample_photons"); wrap(g, "resample_locations"); wrap(g, "log_likelihood")
n = 30
t0 = time.perf_counter()
for _ in range(n):
    g.sweep(); g.log_likelihood()
dt = time.perf_counter() - t0
print("sweep+trace %.3f ms" % (dt / n * 1e3))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print("  %-24s %8.3f ms" % (k, v / n * 1e3))
