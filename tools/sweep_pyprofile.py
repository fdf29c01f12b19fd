"""Host-side profile of ModelGibbs.sweep on the benchmark field: cProfile over N sweeps, top functions by own time.
    python tools/sweep_pyprofile.py [sweeps]"""
import cProfile, pstats, os, sys, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
import desi_mcmc_amd as cel
from desi_mcmc_amd import synth, celeste_mcmc
ctx = cel.default_context(0)
f = synth.SyntheticField.from_config(ctx, "mixed10k_2048")
gf = celeste_mcmc.GibbsField(f.images, list(range(5)), f.bands[:, 2], f.bands[:, 1], f.H * f.W)
g = celeste_mcmc.ModelGibbs([gf], f.src["type"], f.src["radec"], f.flux5(), f.src["shape"], seed=3)
for _ in range(3):
    g.sweep(); g.log_likelihood()
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    g.sweep(); g.log_likelihood()
pr.disable()
out = io.StringIO()
st = pstats.Stats(pr, stream=out).sort_stats("tottime")
st.print_stats(28)
txt = out.getvalue()
print("(times are totals over %d sweeps: divide by %d)" % (n, n))
print(txt[txt.index("ncalls"):])
