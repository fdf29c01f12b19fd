"""where the incremental render's time goes: config 3, one source changed per evaluation"""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np
import desi_mcmc_amd as cel
from desi_mcmc_amd import synth
ctx = cel.Context(0)
f = synth.SyntheticField.from_config(ctx, "mixed10k_2048", seed=42)
src = f.src
f.images.render(f.sources, loglik=True)
rs = np.random.RandomState(1)
pick = rs.choice(f.S, 200, replace=False).astype(np.int32)
def loop(n, prof):
    ctx.profile(prof)
    t0 = time.perf_counter()
    tiles = []
    for k in range(n):
        row = pick[k:k + 1]
        f.sources.set_rows(row, src["type"][row], src["radec"][row] + 1e-5 * (1 + k % 3), src["counts"][row] * 1.01, src["shape"][row])
        f.images.render(f.sources, loglik=True)
        if prof:
            tiles.append(f.images.last_render_dirty_tiles())
    dt = (time.perf_counter() - t0) / n * 1e3
    return dt, tiles
loop(20, 0)
dt, _ = loop(100, 0)
print("unprofiled: %.3f ms per evaluation" % dt)
dt, tiles = loop(100, 1)
print("kernels (ms): prep %.4f bin %.4f render %.4f reduce %.4f; tiles rendered: median %d max %d" % (ctx.profile_get("prep")[0], ctx.profile_get("bin")[0],
      ctx.profile_render()[0], ctx.profile_get("reduce")[0], np.median(tiles), max(tiles)))
t0 = time.perf_counter()
for k in range(100):
    row = pick[k:k + 1]
    f.sources.set_rows(row, src["type"][row], src["radec"][row], src["counts"][row], src["shape"][row])
print("set_rows alone: %.3f ms" % ((time.perf_counter() - t0) / 100 * 1e3))
