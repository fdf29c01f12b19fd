"""Where a small field's step goes: per-kernel device time (dispatch-attached events, CEL_OPT_PROFILE = 1), the step with the
render kernel timed only (level 2: what bench.py's timed region runs) and unprofiled.
    python tools/step_breakdown.py [workload] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import desi_mcmc_amd as cel
from desi_mcmc_amd import synth, _lib

name = sys.argv[1] if len(sys.argv) > 1 else "stars1k_512"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 500
ctx = cel.default_context(0)
for k, v in (("CEL_STAR_TILES", _lib.CEL_OPT_STAR_TILES),):
    if os.environ.get(k):
        ctx.set_option(v, float(os.environ[k]))
f = synth.SyntheticField.from_config(ctx, name)
for _ in range(100):
    f.images.render(f.sources, loglik=True)
out = {}
for level in (0, 3, 2, 1):
    ctx.profile(level if level else False)
    for _ in range(20):
        f.images.render(f.sources, loglik=True)
    if level:
        ctx.profile(level)
    t0 = time.perf_counter()
    for _ in range(steps):
        ll, llb = f.images.render(f.sources, loglik=True)
    out[level] = (time.perf_counter() - t0) / steps * 1e3
    if level == 1:
        tr, nr, kname = ctx.profile_render()
        parts = {k: ctx.profile_get(k) for k in ("prep", "bin", "reduce")}
        print("kernels (mean ms per launch): render[%s] %.4f" % (kname, tr), " ".join("%s %.4f" % (k, t) for k, (t, n) in parts.items()))
    ctx.profile(False)
print("%s: step unprofiled %.4f ms, render events on every 4th launch %.4f ms, on every launch %.4f ms, all kernels' events %.4f ms; ll %.6f"
      % (name, out[0], out[3], out[2], out[1], ll))
