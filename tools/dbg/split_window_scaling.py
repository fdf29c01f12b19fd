"""k_photon_split_hw on row windows of the benchmark field: does its time follow the window's share of the frame?"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np
import desi_mcmc_amd as cel
from desi_mcmc_amd import synth
ctx = cel.Context(0)
f = synth.SyntheticField.from_config(ctx, "mixed10k_2048", seed=42)
H, W = f.H, f.W
for (y0, y1) in ((0, 2048), (0, 1024), (576, 1216), (704, 1088), (768, 1024), (768, 896)):
    iset = f.images if (y0, y1) == (0, H) else cel.ImageSet(ctx, f.bands, y1 - y0, W, nelec=np.ascontiguousarray(f.nelec[:, y0:y1]))
    if (y0, y1) != (0, H):
        iset.set_window(y0, H)
    for reuse in (0, 2):
        ctx.set_option(cel._lib.CEL_OPT_SPLIT_REUSE, reuse)
        iset.render(f.sources, loglik=True)
        iset.photon_split_resident(f.sources, seed=3)
        ctx.profile(True)
        for _ in range(3):
            iset.render(f.sources, loglik=True)
            iset.photon_split_resident(f.sources, seed=3)
        t, n = ctx.profile_get("split")
        tr = ctx.profile_render()[0]
        ctx.profile(False)
        print("rows [%4d, %4d) = %.3f of the frame, reuse %d: split kernel %.3f ms (%.2f of the whole frame's), render %.3f ms" % (y0, y1, (y1 - y0) / H, reuse, t, t / 6.5, tr))
ctx.set_option(cel._lib.CEL_OPT_SPLIT_REUSE, 2)
# the chain's window: the same rows with the owned rows restricted, and with a moved catalogue
iset = cel.ImageSet(ctx, f.bands, 640, W, nelec=np.ascontiguousarray(f.nelec[:, 576:1216]))
iset.set_window(576, H)
iset.set_noise_rows(192, 448)
for moved in (False, True):
    src = f.sources
    if moved:
        rs = np.random.RandomState(0)
        src = cel.SourceSet(ctx, f.S, f.B).set(f.src["type"], f.src["radec"] + rs.normal(0, 3e-5, (f.S, 2)), f.src["counts"] * np.exp(rs.normal(0, 0.3, (f.S, f.B))), f.src["shape"])
    iset.render(src, loglik=True); iset.photon_split_resident(src, seed=3)
    ctx.profile(True)
    for _ in range(3):
        iset.render(src, loglik=True); iset.photon_split_resident(src, seed=3)
    print("owned rows [192, 448) of the 640-row window, catalogue moved %s: split kernel %.3f ms" % (moved, ctx.profile_get("split")[0]))
    ctx.profile(False)
