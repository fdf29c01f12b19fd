"""device memory after many create / use / destroy cycles of contexts, image sets and source sets (every kind of call in between)"""
import sys, os, gc
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np
import ctypes
import desi_mcmc_amd as cel
from desi_mcmc_amd import synth, celeste_mcmc
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rs = np.random.RandomState(0)
hip = ctypes.CDLL("libamdhip64.so")


def free_mb():
    hip.hipDeviceSynchronize()
    a, b = ctypes.c_size_t(0), ctypes.c_size_t(0)
    assert hip.hipMemGetInfo(ctypes.byref(a), ctypes.byref(b)) == 0
    return a.value / 2 ** 20


def cycle(k):
    if os.environ.get("LEAK_TRACE"): print("cycle", k, flush=True)
    ctx = cel.Context(0)
    H, W = int(rs.randint(100, 700)), int(rs.randint(100, 700))
    S = int(rs.randint(5, 400))
    tr = (lambda m: print("   ", m, flush=True)) if os.environ.get("LEAK_TRACE") else (lambda m: None)
    Bk = int(rs.randint(1, 6)); fg = rs.rand()
    tr("field S=%d B=%d %dx%d frac_gal=%.2f" % (S, Bk, H, W, fg))
    f = synth.SyntheticField(ctx, S, Bk, H, W, frac_gal=fg, seed=k)
    f.images.render(f.sources, loglik=True); tr("render")
    f.images.estep_stats(f.sources); tr("estep")
    f.images.photon_split_resident(f.sources, k); tr("split")
    f.images.stamp_mass(f.sources); tr("mass")
    f.images.slice_locations(f.sources, 1e-3, k); tr("slice")
    f.images.stamps(f.sources, 0); tr("stamps")
    if f.B == 5:
        gf = celeste_mcmc.GibbsField(f.images, list(range(5)), f.bands[:, 2], f.bands[:, 1], H * W)
        g = celeste_mcmc.ModelGibbs([gf], f.src["type"], f.src["radec"], f.flux5(), f.src["shape"], seed=k)
        g.sweep(shapes=True); g.log_likelihood(); tr("gibbs")
        del g, gf
    f.images.close()
    del f                   # (source sets and contexts are released when collected)
    tr("images closed")
    del ctx
    tr("context released")


for k in range(1000, 1010):          # (the first cycles load code objects and grow the runtime's pools: the baseline comes after them)
    cycle(k)
gc.collect()
base = free_mb()
for k in range(1, N + 1):
    cycle(k)
    if os.environ.get("LEAK_SYNC_EACH"):
        hip.hipDeviceSynchronize(); print("   device synchronised", flush=True)
    if k % 10 == 0:
        gc.collect()
        print("after %3d cycles: free device memory %+.1f MB against the baseline" % (k, free_mb() - base), flush=True)
gc.collect()
d = free_mb() - base
print("ok: %d cycles, %+.1f MB" % (N, d) if d > -16 else "LEAK: %+.1f MB after %d cycles" % (d, N))
