"""two (or more) host threads, a context each, the same GPU: every call's result must be what the thread gets alone (renders and
likelihoods bit for bit, splits photon for photon, location steps coordinate for coordinate)"""
import sys, os, threading
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np
import desi_mcmc_amd as cel
from desi_mcmc_amd import synth
NT = int(sys.argv[1]) if len(sys.argv) > 1 else 3
N = int(sys.argv[2]) if len(sys.argv) > 2 else 60


def work(t, out):
    ctx = cel.Context(0)
    f = synth.SyntheticField(ctx, 400 + 150 * t, 2 + t % 3, 300 + 40 * t, 260 + 30 * t, frac_gal=0.5, seed=100 + t)
    rs = np.random.RandomState(t)
    res = []
    cur = {k: np.array(f.src[k], copy=True) for k in ("type", "radec", "counts", "shape")}
    for k in range(N):
        rows = rs.choice(f.S, 5, replace=False).astype(np.int32)
        cur["counts"][rows] *= 1.01
        f.sources.set_rows(rows, cur["type"][rows], cur["radec"][rows], cur["counts"][rows], cur["shape"][rows])
        ll, llb = f.images.render(f.sources, loglik=True)
        noise = f.images.photon_split_resident(f.sources, 7 * k + t)
        sums = f.images.sample_sums()
        mass = f.images.stamp_mass(f.sources)
        u, llh, st = f.images.slice_locations(f.sources, 1e-3, 11 * k + t)
        cur["radec"] = u.copy()
        res.append((llb.copy(), noise.copy(), sums.copy(), mass.copy(), u.copy(), llh.copy()))
    out[t] = res


alone = {}
for t in range(NT):
    work(t, alone)
together = {}
th = [threading.Thread(target=work, args=(t, together)) for t in range(NT)]
for x in th: x.start()
for x in th: x.join()
bad = 0
for t in range(NT):
    if t not in together:
        print("thread %d died" % t); bad += 1; continue
    for k, (a, b) in enumerate(zip(alone[t], together[t])):
        for name, x, y in zip(("ll", "noise", "sums", "mass", "u", "llh"), a, b):
            if not np.array_equal(x, y):
                bad += 1
                if bad < 6: print("thread %d step %d: %s differs (max %g)" % (t, k, name, np.abs(x - y).max()))
print("ok: %d threads x %d steps equal to the threads run alone" % (NT, N) if not bad else "MISMATCH: %d" % bad)
sys.exit(1 if bad else 0)
