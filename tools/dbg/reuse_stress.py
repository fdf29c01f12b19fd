"""random sequences of the calls a Gibbs chain makes on ONE image set -- catalogue edits (rows / whole), trace renders, photon splits,
stamp masses (plain and begin / end), conditional likelihoods, location slices, new sky levels, E-step sums -- with the re-use short
cuts on (CEL_OPT_SPLIT_REUSE = 2: the split's totals from the model image on the device, the masses from the split's own sums)
against a second context on the same GPU with CEL_OPT_SPLIT_REUSE = 0 that is told the same things.  Photons per (source, band)
and sky sums: a draw compares a uniform with a ratio of rates that the two routes form to ~1e-10, so a few photons may differ;
masses at 1e-9 (the short cut's documented bound), likelihoods at 1e-10.   python tools/dbg/reuse_stress.py [STEPS] [seed]"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np
import desi_mcmc_amd as cel
from desi_mcmc_amd import synth
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
L = cel._lib
rs = np.random.RandomState(seed)
ctxs = [cel.Context(0), cel.Context(0)]
ctxs[1].set_option(L.CEL_OPT_SPLIT_REUSE, 0)
ctxs[1].set_option(L.CEL_OPT_INCREMENTAL, 0)
f = synth.SyntheticField(ctxs[0], 900, 3, 640, 512, frac_gal=0.5, seed=23)
S, B = f.S, f.B
cur = {k: np.array(f.src[k], copy=True) for k in ("type", "radec", "counts", "shape")}
img = [cel.ImageSet(c, f.bands, f.H, f.W, nelec=f.nelec) for c in ctxs]
sset = [cel.SourceSet(c, S, B).set(cur["type"], cur["radec"], cur["counts"], cur["shape"]) for c in ctxs]
eps = f.bands[:, 0].copy()
log = []
have_split = False
pending = False


def fail(msg):
    print("MISMATCH at step %d: %s; last ops %s" % (len(log), msg, log[-14:]))
    sys.exit(1)


n = dict(split=0, split_after_trace=0, mass=0, mass_from_split=0, render=0, pll=0, slice=0)
for step in range(STEPS):
    op = rs.choice(["rows", "rows", "set", "render", "render", "split", "split", "mass", "mass", "mass_be", "pll", "slice", "eps", "estep", "render_noll"])
    if pending and op not in ("rows", "set"):         # (no other call on the context between begin and end)
        out = [i.stamp_mass_end() for i in img]
        pending = False
        if not np.allclose(out[0], out[1], rtol=1e-9, atol=1e-15): fail("stamp_mass_end %g" % np.abs(out[0] / np.maximum(out[1], 1e-300) - 1).max())
        log.append("mass_end")
    if op == "rows":
        if pending: continue
        k = int(rs.choice([1, 3, 20, 70]))
        rows = rs.choice(S, k, replace=False).astype(np.int32)
        cur["radec"][rows] += rs.normal(0, 2e-5, (k, 2))
        cur["counts"][rows] *= np.exp(rs.normal(0, 0.3, (k, B)))
        g = rows[cur["type"][rows] == 1]
        cur["shape"][g, 1] *= np.exp(rs.normal(0, 0.2, g.size))
        for s_ in sset: s_.set_rows(rows, cur["type"][rows], cur["radec"][rows], cur["counts"][rows], cur["shape"][rows])
        log.append("rows(%d)" % k)
    elif op == "set":
        if pending: continue
        cur["counts"] *= np.exp(rs.normal(0, 0.05, cur["counts"].shape))
        for s_ in sset: s_.set(cur["type"], cur["radec"], cur["counts"], cur["shape"])
        log.append("set")
    elif op in ("render", "render_noll"):
        out = [i.render(s_, loglik=(op == "render")) for i, s_ in zip(img, sset)]
        if op == "render" and not np.allclose(out[0][1], out[1][1], rtol=1e-12): fail("trace render ll %r %r" % (out[0][1], out[1][1]))
        n["render"] += 1; log.append(op)
    elif op == "split":
        sd = int(rs.randint(1 << 30))
        noise = [i.photon_split_resident(s_, sd) for i, s_ in zip(img, sset)]
        sums = [i.sample_sums() for i in img]
        d = np.abs(sums[0] - sums[1])
        if d.max() > 3 or d.sum() > 12 or np.abs(noise[0] - noise[1]).max() > 12: fail("split sums differ by up to %g photons (%g in all), sky %s" % (d.max(), d.sum(), noise[0] - noise[1]))
        if sums[0].sum() + noise[0].sum() != f.nelec.sum(): fail("conservation")
        have_split = True; n["split"] += 1; n["split_after_trace"] += bool(log) and log[-1] == "render"; log.append("split")
    elif op == "mass":
        ready = img[0].stamp_mass_ready(sset[0]); n["mass_from_split"] += ready
        out = [i.stamp_mass(s_) for i, s_ in zip(img, sset)]
        if not np.allclose(out[0], out[1], rtol=1e-9, atol=1e-15): fail("stamp_mass off by %g" % np.abs(out[0] / np.maximum(out[1], 1e-300) - 1).max())
        n["mass"] += 1; log.append("mass(ready=%d)" % ready)
    elif op == "mass_be":
        for i, s_ in zip(img, sset): i.stamp_mass_begin(s_)
        pending = True; log.append("mass_begin")
    elif op == "pll" and have_split:
        P = 16
        own = rs.choice(S, P, replace=False).astype(np.int32)
        U = cur["radec"][own] + rs.normal(0, 1e-5, (P, 2))
        out = []
        for c, i in zip(ctxs, img):
            prop = cel.SourceSet(c, P, B).set(cur["type"][own], U, cur["counts"][own], cur["shape"][own])
            out.append(i.patch_loglik_resident(prop, own))
        # (the two splits may differ by a photon: compare where they are alike is not possible cheaply -- loose)
        if not np.allclose(out[0], out[1], rtol=1e-3, atol=50.0): fail("pll %r %r" % (out[0], out[1]))
        n["pll"] += 1; log.append("pll")
    elif op == "slice" and have_split:
        sd = int(rs.randint(1 << 30))
        out = [i.slice_locations(s_, 1e-3, sd) for i, s_ in zip(img, sset)]
        # chains on splits that differ by a photon take other paths: the library's catalogues are told ONE result
        cur["radec"] = out[1][0].copy()
        for s_ in sset: s_.set(cur["type"], cur["radec"], cur["counts"], cur["shape"])
        n["slice"] += 1; log.append("slice")
    elif op == "eps":
        b = int(rs.randint(B)); e = eps[b] * (1 + 1e-2 * rs.rand())
        for i in img: i.set_epsilon(b, e)
        log.append("eps")
    elif op == "estep":
        out = []
        for i, s_ in zip(img, sset):
            i.render(s_, loglik=False); out.append(i.estep_stats(s_))
        for a, b_ in zip(out[0], out[1]):
            if not np.allclose(a, b_, rtol=1e-10, atol=1e-9): fail("estep")
        log.append("estep")
    if step % 250 == 249:
        print("step %d: %s" % (step + 1, n), flush=True)
print("ok: %d steps, %s" % (STEPS, n))
