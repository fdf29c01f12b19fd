"""the calibration ranks of sweep(shapes=True) under conditional="exact", with the shape step's sampler options varied and
with blocks of the sweep switched off (their numbers stay at the truth):
    python tools/dbg/sbc_exact.py ROUNDS [accept=neal] [compwise] [nostepout] [mass_only] [blocks=sky,flux,loc,shape]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
os.environ["CEL_SBC_ROUNDS"] = sys.argv[1] if len(sys.argv) > 1 else "8"
import numpy as np
import desi_mcmc_amd as cel
from desi_mcmc_amd import celeste_mcmc
import test_calibration as tc
args = {}
if "accept=neal" in sys.argv: args["accept"] = "neal"
if "compwise" in sys.argv: args["compwise"] = True
if "nostepout" in sys.argv: args.update(step_out=False)
blocks = [a[7:].split(",") for a in sys.argv if a.startswith("blocks=")]
if blocks:
    on = set(blocks[0])
    M = celeste_mcmc.ModelGibbs

    def sweep(self, shapes=False):
        self._split_photons()
        if "sky" in on: self._resample_sky()
        if "flux" in on: self.resample_fluxes()
        if "loc" in on: self.resample_locations()
        if "shape" in on: self.resample_shapes()
        self.sweeps += 1

    def sweep_reversed(self, shapes=False):
        if "shape" in on: self.resample_shapes()
        if "loc" in on: self.resample_locations()
        if "flux" in on: self.resample_fluxes()
        if "sky" in on: self._resample_sky()
        self._split_photons()
        self.sweeps += 1
    M.sweep, M.sweep_reversed = sweep, sweep_reversed
    M.log_likelihood = lambda self: 0.0
    tc.MOVE_CHECK = "loc" in on
ctx = cel.default_context(0)
_real = tc.run_replicate


def _progress(cel_, ctx_, rep, *a, **k):          # (a line every few replicates: a silent GPU job is taken for hung)
    if rep % 8 == 0:
        print("replicate %d" % rep, flush=True)
    return _real(cel_, ctx_, rep, *a, **k)


tc.run_replicate = _progress
kw = dict(shape_mass="exact") if "mass_only" in sys.argv else dict(conditional="exact")
if "reference" in sys.argv: kw = {}
ru, rf, rs_ = tc.pooled_ranks(cel, ctx, "host", 8, shapes=True, shape_args=args or None, **kw)
print(sys.argv[1:], {k: (v[0], round(v[1], 4), v[2]) for k, v in tc.shape_rank_table(ru, rf, rs_).items()})
