"""ONE Gibbs chain dealt over 2-4 ranks (gloo; the ranks share this GPU), random field sizes: the replicated deal must be the
single-rank chain bit for bit, the strip partition photon for photon in the first split and to rounding afterwards, every rank
holding the same merged state (what tests/test_gibbs.py checks for two ranks on two fixed fields)"""
import os, sys, socket, subprocess, tempfile
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from _dealt_chain_rank import run_chain
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
tmp = tempfile.mkdtemp()
for it in range(N):
    size = int(rs.choice([512, 640, 768, 1024]))
    S = int(rs.randint(100, 2500))
    world = int(rs.choice([2, 3, 4]))
    split = str(rs.choice(["strips", "replicated"]))
    shapes = int(split == "replicated" and rs.rand() < 0.5)
    sweeps = 2
    one = run_chain(S, size, sweeps, "device", shapes=bool(shapes))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        args = [sys.executable, os.path.join(R, "tests", "_dealt_chain_rank.py"), os.path.join(tmp, "r%d.npz" % r), str(S), str(size), str(sweeps), "device", str(shapes)]
        if split == "strips":
            args.append("strips")
        procs.append(subprocess.Popen(args, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    tag = "S=%d %dx%d world=%d %s shapes=%d" % (S, size, size, world, split, shapes)
    if any(p.returncode != 0 for p in procs):
        msg = [o.strip().splitlines()[-1] for o in outs if o.strip()]
        known = any("window" in m or "box" in m for m in msg)
        print("%s: ranks failed (%s): %s" % (tag, "a box reaches beyond a window: refused by design" if known else "UNEXPECTED", msg[:2]), flush=True)
        bad += not known
        continue
    got = [np.load(os.path.join(tmp, "r%d.npz" % r)) for r in range(world)]
    ok = True
    for k in ("u", "fluxes", "eps", "ll", "noise", "shape"):
        ok &= all(np.array_equal(got[0][k], got[r][k]) for r in range(1, world))
    if split == "replicated":
        for k in ("u", "fluxes", "eps", "ll", "shape"):
            ok &= np.array_equal(got[0][k], one[k])
    else:
        own = sum(g["sums"] for g in got)
        ok &= np.array_equal(own[0], one["sums"][0])
        ok &= np.array_equal(own[-1].sum(axis=0) + got[0]["noise"], got[0]["nelec_sum"])
        ok &= np.allclose(got[0]["u"][0], one["u"][0], rtol=1e-9, atol=0) and np.allclose(got[0]["fluxes"][0], one["fluxes"][0], rtol=1e-9)
        ok &= np.allclose(got[0]["ll"], one["ll"], rtol=1e-12)
    print("%s: %s" % (tag, "ok" if ok else "MISMATCH"), flush=True)
    bad += not ok
print("ok: %d configurations" % N if not bad else "MISMATCH in %d" % bad)
sys.exit(1 if bad else 0)
