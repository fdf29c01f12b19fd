import time, numpy as np, sys
sys.path.insert(0,'/root/repo')
import desi_mcmc_amd as cel
from desi_mcmc_amd import celeste_mcmc, synth
ctx = cel.default_context(0)
f = synth.SyntheticField.from_config(ctx, "mixed10k_2048")
gf = celeste_mcmc.GibbsField(f.images, list(range(5)), f.bands[:, 2], f.bands[:, 1], f.H*f.W)
g = celeste_mcmc.ModelGibbs([gf], f.src["type"], f.src["radec"], f.flux5(), f.src["shape"], seed=1)
for it in range(4):
    t0=time.perf_counter(); g.sweep(shapes=True); dt=time.perf_counter()-t0
    print(it, "sweep %.1f ms"%(dt*1e3), {k:(round(v*1e3,1) if isinstance(v,float) else v) for k,v in g.timing.items()}, flush=True)
    for k in g.timing: g.timing[k]=0
print("shape drift", np.abs(g.shape-f.src["shape"])[f.src["type"]==1].mean(axis=0))
