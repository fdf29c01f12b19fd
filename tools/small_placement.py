"""Where the blocks of k_small_stars run (diagnostic; -DSMALL_STAMPS build, see tools/small_stamps.py): XCC, SE, CU and SIMD of
every block from the stamps dump, and which block indices share a CU."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
path = os.path.join(ROOT, "gpurun_out", "small_stamps.bin")
os.makedirs(os.path.dirname(path), exist_ok=True)
os.environ["CEL_SMALL_STAMPS"] = path
import numpy as np
import desi_mcmc_amd as cel
from desi_mcmc_amd import synth
ctx = cel.default_context(0)
f = synth.SyntheticField.from_config(ctx, sys.argv[1] if len(sys.argv) > 1 else "stars1k_512")
maps = []
for it in range(4):
    for _ in range(3):
        f.images.render(f.sources, loglik=True)
    st = np.fromfile(path, dtype=np.uint64).reshape(-1, 8)
    w = st[:, 7]
    xcc = (w & 0xF).astype(int); hw = (w >> 8).astype(np.int64)
    cu = ((hw >> 8) & 0xF).astype(int); sh = ((hw >> 12) & 1).astype(int); se = ((hw >> 13) & 7).astype(int)
    place = xcc * 1000 + se * 100 + sh * 50 + cu
    maps.append(place)
    if it == 0:
        print("blocks", len(st), "distinct (xcc,se,sh,cu):", len(set(place.tolist())))
        per = collections.Counter(place.tolist())
        print("blocks per CU histogram:", sorted(collections.Counter(per.values()).items()))
        print("block -> xcc of the first 24:", xcc[:24].tolist())
        print("block -> se/sh/cu of blocks 0,8,16,...,120 (xcc 0):", [(int(se[i]), int(sh[i]), int(cu[i])) for i in range(0, 128, 8)])
        groups = collections.defaultdict(list)
        for i, p in enumerate(place.tolist()):
            groups[p].append(i)
        three = [g for g in groups.values() if len(g) >= 3]
        print("some CUs with three blocks:", three[:8])
        two = [g for g in groups.values() if len(g) == 2]
        print("some CUs with two blocks:", two[:8])
        dur = (st[:, 5] - st[:, 0]).astype(np.int64) / 100.0
        n3 = np.array([len(groups[p]) for p in place.tolist()])
        for k in sorted(set(n3.tolist())):
            print("blocks on a CU with %d blocks: mean duration %.2f us, max %.2f (n=%d)" % (k, dur[n3 == k].mean(), dur[n3 == k].max(), (n3 == k).sum()))
print("same placement in every call:", all((m == maps[0]).all() for m in maps[1:]), "; blocks that moved between call 0 and 1:", int((maps[0] != maps[1]).sum()))
