import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import desi_mcmc_amd as cel
from desi_mcmc_amd import _lib, synth
ctx = cel.Context(0)
wl = sys.argv[1] if len(sys.argv) > 1 else "mixed10k_2048"
bits = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 256]
f = synth.SyntheticField.from_config(ctx, wl)
for rep in range(3):
    for b in bits:
        ctx.set_option(_lib.CEL_OPT_DEBUG, b)
        for _ in range(5):
            f.images.render(f.sources, loglik=True)
        ctx.profile(True)
        for _ in range(30):
            f.images.render(f.sources, loglik=True)
        ms, n = ctx.profile_get("render")
        ctx.profile(False)
        print("debug=%d  k_render %.4f ms" % (b, ms), flush=True)
