"""How far the masses read off the split's sums (CEL_OPT_SPLIT_REUSE = 2) are from the mass kernel's, as a function of
counts / eps: a -DMASS_VOUCH_ALL build of the library (every source vouched for) on a field whose sources are all set to
the same counts / eps ratio."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import desi_mcmc_amd as cel
from desi_mcmc_amd import _lib, synth
ctx = cel.default_context(0)
f = synth.SyntheticField(ctx, 2000, 5, 1024, 1024, frac_gal=0.5, seed=5)
eps = f.bands[:, 0]
gal = f.src["type"] == 1
for ratio in (100.0, 10.0, 1.0, 0.25, 0.05, 0.01, 1e-3):
    counts = np.tile(eps[None, :] * ratio, (2000, 1))
    f.sources.set(f.src["type"], f.src["radec"], counts, f.src["shape"])
    ctx.set_option(_lib.CEL_OPT_SPLIT_REUSE, 1)
    f.images.render(f.sources, loglik=True)
    f.images.photon_split_resident(f.sources, seed=5)
    exact = f.images.stamp_mass(f.sources)
    ctx.set_option(_lib.CEL_OPT_SPLIT_REUSE, 2)
    f.images.render(f.sources, loglik=True)
    f.images.photon_split_resident(f.sources, seed=5)
    quick = f.images.stamp_mass(f.sources)
    ok = exact > 0
    rel = np.abs(quick - exact)[ok] / exact[ok]
    rg = (np.abs(quick - exact) / np.where(ok, exact, 1))[gal][ok[gal]]
    rs = (np.abs(quick - exact) / np.where(ok, exact, 1))[~gal][ok[~gal]]
    print("counts / eps = %-7g  relative difference: stars median %.1e max %.1e   galaxies median %.1e max %.1e   identical %.2f"
          % (ratio, np.median(rs), rs.max(), np.median(rg), rg.max(), (quick == exact).mean()))
