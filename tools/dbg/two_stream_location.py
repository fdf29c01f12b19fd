"""Feasibility: would the location step gain from two halves of the chains on two HIP streams?  Two contexts (a stream each), each
with its own copy of the field and the same photon split; the location step over (a) all chains on one context, (b) the even
chains on context 0 and the odd chains on context 1 at the same time (two host threads).   python tools/dbg/two_stream_location.py"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import desi_mcmc_amd as cel
from desi_mcmc_amd import synth
S = 10000
ctxs = [cel.Context(0), cel.Context(0)]
flds = [synth.SyntheticField.from_config(c, "mixed10k_2048", seed=42) for c in ctxs]
u0 = flds[0].src["radec"].copy()
ids = [np.where(np.arange(S) % 2 == k, np.arange(S), -1).astype(np.int32) for k in (0, 1)]

def prepare(f):
    f.sources.set(f.src["type"], u0, f.src["counts"], f.src["shape"])
    f.images.render(f.sources, loglik=True)
    f.images.photon_split_resident(f.sources, seed=11)

def run(f, chain_ids, out, k):
    t = time.perf_counter()
    out[k] = f.images.slice_locations(f.sources, 1e-3, seed=5, chain_ids=chain_ids)
    out[k + 2] = time.perf_counter() - t

for rep in range(3):
    for f in flds:
        prepare(f)
    res = [None] * 4
    t0 = time.perf_counter()
    run(flds[0], None, res, 0)
    t_all = time.perf_counter() - t0
    u_all = res[0][0]
    for f in flds:
        prepare(f)
    # one half alone
    t0 = time.perf_counter()
    run(flds[0], ids[0], res, 0)
    t_half = time.perf_counter() - t0
    for f in flds:
        prepare(f)
    th = [threading.Thread(target=run, args=(flds[k], ids[k], res, k)) for k in (0, 1)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    t_two = time.perf_counter() - t0
    ua, ub = res[0][0], res[1][0]
    merged = np.where((np.arange(S) % 2 == 0)[:, None], ua, ub)
    print("all chains, one stream %.3f ms | even chains alone %.3f ms | halves on two streams at once %.3f ms (%.3f / %.3f)   same chains: %s"
          % (t_all * 1e3, t_half * 1e3, t_two * 1e3, res[2] * 1e3, res[3] * 1e3, np.array_equal(merged, u_all)), flush=True)
