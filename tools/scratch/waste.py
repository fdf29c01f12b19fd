import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import desi_mcmc_amd as cel
from desi_mcmc_amd import _lib, synth
ctx = cel.Context(0)
f = synth.SyntheticField.from_config(ctx, "mixed10k_2048")
for _ in range(3):
    f.images.render(f.sources, loglik=True)
def run(dbg):
    ctx.set_option(8, float(dbg))
    ctx.set_option(6, 1.0)
    f.images.render(f.sources, loglik=True)
    n = C.c_int64(0)
    _lib.check(_lib.lib().cel_debug_tile_timing(f.images._h, None, C.byref(n)))
    buf = np.zeros(3 * n.value, dtype=np.uint64)
    _lib.check(_lib.lib().cel_debug_tile_timing(f.images._h, buf.ctypes.data, C.byref(n)))
    return buf.reshape(-1, 3)[:, 2]
t = run(0)
comprows = (t >> np.uint64(32)).astype(np.int64).sum()
t = run(128)
ideal_rows = (t & np.uint64(0xffffffff)).astype(np.int64).sum()
ideal_area = (t >> np.uint64(32)).astype(np.int64).sum()
print("walked component-rows %.3e ; sum of individual row ranges %.3e (%.2f) ; column-clipped area/32 %.3e (%.2f)"
      % (comprows, ideal_rows, ideal_rows / comprows, ideal_area, ideal_area / comprows))
