import os, sys, time, cProfile, pstats
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import desi_mcmc_amd as cel
from desi_mcmc_amd import synth, celeste_mcmc
ctx = cel.Context(0)
fld = synth.SyntheticField.from_config(ctx, "mixed10k_2048", seed=42)
S, B, H, W, fg = synth.CONFIGS["mixed10k_2048"]
gf = celeste_mcmc.GibbsField(fld.images, list(range(B)), fld.bands[:, 2], fld.bands[:, 1], H * W)
g = celeste_mcmc.ModelGibbs([gf], fld.src["type"], fld.src["radec"], fld.flux5(), fld.src["shape"], seed=1, slice_args=dict(step_out=False, sigma=0.001))
for _ in range(5):
    g.sweep(); g.log_likelihood()
n = 40
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for _ in range(n):
    g.sweep(); g.log_likelihood()
pr.disable()
dt = time.perf_counter() - t0
print("sweep+trace %.3f ms (under cProfile)" % (dt / n * 1e3))
st = pstats.Stats(pr); st.sort_stats("tottime")
import io
buf = io.StringIO(); st.stream = buf; st.print_stats(28); print(buf.getvalue()[:6000])
