import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import desi_mcmc_amd as cel
from desi_mcmc_amd import _lib, synth
ctx = cel.Context(0)
f = synth.SyntheticField.from_config(ctx, "mixed10k_2048")
f.images.render(f.sources, loglik=True)
for rep in range(2):
    for b in [int(x) for x in sys.argv[1].split(",")]:
        ctx.set_option(_lib.CEL_OPT_DEBUG, b)
        for _ in range(2):
            f.images.photon_split_resident(f.sources, seed=3)
        t0 = time.perf_counter()
        for i in range(5):
            f.images.photon_split_resident(f.sources, seed=4 + i)
        print("debug=%d split call %.3f ms" % (b, (time.perf_counter() - t0) / 5 * 1e3), flush=True)
