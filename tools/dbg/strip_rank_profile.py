"""where a strip rank's sweep goes on the host: cProfile of 6 sweeps of rank 3 of 8 (solo)"""
import sys, os, cProfile, pstats
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np
import desi_mcmc_amd as cel
from desi_mcmc_amd import celeste_mcmc, dist, synth
ctx = cel.Context(0)
field = synth.SyntheticField.from_config(ctx, "mixed10k_2048", seed=42)
S, B, H, W, fg = synth.CONFIGS["mixed10k_2048"]
boxes, status = field.images.source_boxes(field.sources)
edges = dist.strip_edges(H, 8, align=64)
deal, gf = celeste_mcmc.strip_gibbs_field(ctx, field.bands, field.nelec, field.src["pix"][:, 1], boxes, status, 8, 3, edges=edges, solo=True)
g = celeste_mcmc.ModelGibbs([gf], field.src["type"], field.src["radec"], field.flux5(), field.src["shape"], seed=1, slice_args=dict(step_out=False, sigma=0.001), deal=deal)
for _ in range(3):
    g.sweep(); g.log_likelihood()
pr = cProfile.Profile()
pr.enable()
for _ in range(6):
    g.sweep(); g.log_likelihood()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(28)
ctx.profile(True)
for _ in range(4):
    g.sweep(); g.log_likelihood()
for k in ("totals", "render", "split", "mass", "patch_ll", "prep", "bin"):
    t, n = ctx.profile_get(k)
    print("%-10s %8.3f ms x %d per 4 sweeps" % (k, t, n))
ctx.profile(False)
import time
ctx.profile(True)
for _ in range(4):
    g.log_likelihood(); g._split_photons()
print("split alone, back to back with the trace render: %.3f ms" % ctx.profile_get("split")[0])
ctx.profile(True)
for _ in range(4):
    time.sleep(0.005)
    g.log_likelihood(); g._split_photons()
print("... with 5 ms of idle GPU before each: %.3f ms" % ctx.profile_get("split")[0])
ctx.profile(False)
