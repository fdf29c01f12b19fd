#!/usr/bin/env python3
"""A/B of two builds of the library on ONE box: N Gibbs sweeps of the benchmark field, the likelihood kernel's and the split
kernel's mean time per sweep (HIP events) and the phases' host times.   python tools/ab_nz.py [path/to/libceleste_hip.so]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from desi_mcmc_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
import desi_mcmc_amd as cel
from desi_mcmc_amd import synth, celeste_mcmc
ctx = cel.Context(0)
field = synth.SyntheticField.from_config(ctx, "mixed10k_2048", seed=42)
S, B, H, W, fg = synth.CONFIGS["mixed10k_2048"]
gf = celeste_mcmc.GibbsField(field.images, list(range(B)), field.bands[:, 2], field.bands[:, 1], H * W)
g = celeste_mcmc.ModelGibbs([gf], field.src["type"], field.src["radec"], field.flux5(), field.src["shape"], seed=1,
                            slice_args=dict(step_out=False, sigma=0.001))
for _ in range(3):
    g.sweep()
for k in g.timing: g.timing[k] = 0
ctx.profile(2)
n = int(os.environ.get("AB_SWEEPS", "30"))
t0 = time.perf_counter()
for _ in range(n):
    g.sweep()
dt = time.perf_counter() - t0
t_ll, n_ll = ctx.profile_get("patch_ll")
t_sp, n_sp = ctx.profile_get("split")
print("%s: sweep %.2f ms  split %.2f flux %.2f location %.2f | k_patch_ll %.3f ms/sweep (%d launches, %d evals/sweep)  k_photon_split_hw %.3f ms"
      % (os.path.basename(_lib.LIB_PATH), dt / n * 1e3, g.timing["split"] / n * 1e3, g.timing["flux"] / n * 1e3, g.timing["location"] / n * 1e3,
         t_ll * n_ll / n, n_ll, g.timing["evals"] / n, t_sp), flush=True)
