"""random sequences of catalogue edits, renders and the Gibbs / E-step / stamp calls that share an image set's buffers: after
every render the model image and the log-likelihood must be, bit for bit, those of a fresh image set rendering the whole
catalogue.    python tools/dbg/incremental_stress.py [STEPS] [seed] [big]"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np
import desi_mcmc_amd as cel
from desi_mcmc_amd import synth
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
big = "big" in sys.argv
ctx = cel.default_context(0)
L = cel._lib
f = synth.SyntheticField(ctx, 10000, 5, 2048, 2048, frac_gal=0.5, seed=7) if big else synth.SyntheticField(ctx, 700, 3, 400, 300, frac_gal=0.5, seed=19)
S, B = f.S, f.B
rs = np.random.RandomState(seed)
if not big:
    ctx.set_option(L.CEL_OPT_TILE_PARTS, 1)          # (the dirty-tile path is the one-wave-per-tile kernel's)
cur = {k: np.array(f.src[k], copy=True) for k in ("type", "radec", "counts", "shape")}
sset = cel.SourceSet(ctx, S, B).set(cur["type"], cur["radec"], cur["counts"], cur["shape"])
other = cel.SourceSet(ctx, S, B).set(cur["type"], cur["radec"] + 1e-5, cur["counts"], cur["shape"])
ref_set = cel.SourceSet(ctx, S, B)
img = cel.ImageSet(ctx, f.bands, f.H, f.W, nelec=f.nelec)
ref_img = cel.ImageSet(ctx, f.bands, f.H, f.W, nelec=f.nelec)
eps0 = f.bands[:, 0].copy()
counts = dict(incremental=0, full=0)
have_split = False
log = []
since = {}


def edit():
    k = int(rs.choice([1, 1, 1, 1, 1, 1, 2, 2, 5, 5, 30, 64, 65, 120]))
    rows = rs.choice(S, k, replace=False)
    for s in rows:
        what = rs.randint(7)
        if what == 0:
            cur["radec"][s] += rs.normal(0, 2e-5, 2)
        elif what == 1:
            cur["radec"][s] = synth.pixel2equa(f.bands[0], np.array([[rs.uniform(-30, f.W + 30), rs.uniform(-30, f.H + 30)]]))[0]
        elif what == 2:
            cur["radec"][s] = synth.pixel2equa(f.bands[0], np.array([[-500.0 - rs.uniform(0, 100), rs.uniform(0, f.H)]]))[0]
        elif what == 3:
            cur["type"][s] = 1 - cur["type"][s]
            cur["shape"][s] = [rs.uniform(0.05, 0.95), np.exp(rs.uniform(np.log(0.3), np.log(3.0))), rs.uniform(0, 180), rs.uniform(0.2, 1.0)] if cur["type"][s] == 1 else [0, 0, 0, 0]
        elif what == 4:
            cur["counts"][s] *= np.exp(rs.normal(0, 0.5, B))
        elif what == 5 and cur["type"][s] == 1:
            cur["shape"][s, 1] *= np.exp(rs.normal(0, 0.3))
        else:
            cur["counts"][s] *= 1.0 + 1e-3 * rs.rand()
    rows = rows.astype(np.int32)
    sset.set_rows(rows, cur["type"][rows], cur["radec"][rows], cur["counts"][rows], cur["shape"][rows])
    return "set_rows(%d)" % k


def render(loglik):
    out = img.render(sset, loglik=loglik)
    d = img.last_render_dirty_tiles()
    counts["incremental" if d >= 0 else "full"] += 1
    # what happened since the last render, and which path this one took
    k = len(log) - 1
    while k >= 0 and not log[k].startswith("render("):
        k -= 1
    key = ",".join(sorted(set(x.split("(")[0].split("=")[0] for x in log[k + 1:]))) or "nothing"
    since.setdefault(key, [0, 0])[0 if d >= 0 else 1] += 1
    want = ref_img.render(ref_set.set(cur["type"], cur["radec"], cur["counts"], cur["shape"]), loglik=True)
    ok = np.array_equal(img.model_images(), ref_img.model_images())
    if loglik:
        ok = ok and np.array_equal(out[1], want[1]) and out[0] == want[0]
    if not ok:
        print("MISMATCH at step %d (loglik=%s) after:" % (len(log), loglik), log[-12:], "dirty", d)
        a, b_ = img.model_images(), ref_img.model_images()
        bad = np.argwhere(a != b_)
        print("   pixels that differ: %d; bands %s; rows %d..%d, cols %d..%d; max |d| %.3e (relative %.3e)" % (
            bad.shape[0], sorted(set(bad[:, 0].tolist())), bad[:, 1].min(), bad[:, 1].max(), bad[:, 2].min(), bad[:, 2].max(),
            np.abs(a - b_).max(), np.abs(a / b_ - 1).max()) if bad.size else "   model images equal")
        if bad.size:
            tiles = sorted(set((int(t[0]), int(t[1]) // 64, int(t[2]) // 32) for t in bad))
            print("   tiles (band, ty, tx):", tiles[:20], "of", len(tiles))
        if loglik:
            print("   ll", out[1], want[1])
        # is a second render of the same state right?
        out2 = img.render(sset, loglik=True)
        print("   rendered again (dirty %d): model images equal now: %s" % (img.last_render_dirty_tiles(), np.array_equal(img.model_images(), b_)))
        sys.exit(1)
    return "render(ll=%d)->%d" % (loglik, d)


for step in range(STEPS):
    op = rs.choice(["edit"] * 10 + ["render"] * 10 + ["render_noll", "mass", "mass", "boxes", "boxes", "estep", "split", "pll", "pll", "stamps", "stamps", "other", "eps", "nelec",
                    "full_set", "parts", "option"])
    if op == "edit":
        log.append(edit())
    elif op == "render":
        log.append(render(True))
    elif op == "render_noll":
        log.append(render(False))
    elif op == "mass":
        img.stamp_mass(sset); log.append("mass")
    elif op == "boxes":
        img.source_boxes(sset); log.append("boxes")
    elif op == "estep":
        img.render(sset, loglik=False); img.estep_stats(sset); log.append("render+estep")
    elif op == "split" and not big:
        img.photon_split_resident(sset, int(rs.randint(1 << 30))); have_split = True; log.append("split")
    elif op == "pll" and have_split and not big:
        P = 8
        own = rs.choice(S, P, replace=False).astype(np.int32)
        prop = cel.SourceSet(ctx, P, B).set(cur["type"][own], cur["radec"][own] + rs.normal(0, 1e-5, (P, 2)), cur["counts"][own], cur["shape"][own])
        img.patch_loglik_resident(prop, own); log.append("pll")
    elif op == "stamps":
        img.stamps(cel.SourceSet(ctx, 4, B).set(cur["type"][:4], cur["radec"][:4], cur["counts"][:4], cur["shape"][:4]), int(rs.randint(B))); log.append("stamps")
    elif op == "other":
        img.render(other, loglik=bool(rs.randint(2))); log.append("other")
    elif op == "eps":
        b = int(rs.randint(B)); e = eps0[b] * (1 + 1e-3 * rs.rand())
        img.set_epsilon(b, e); ref_img.set_epsilon(b, e); log.append("eps")
    elif op == "nelec" and rs.rand() < 0.3:
        img.set_nelec(f.nelec); have_split = False; log.append("nelec")
    elif op == "full_set":
        sset.set(cur["type"], cur["radec"], cur["counts"], cur["shape"]); log.append("full_set")
    elif op == "parts" and not big:
        p = int(rs.choice([1, 1, 1, 1, 1, 0, 2, 4])); ctx.set_option(L.CEL_OPT_TILE_PARTS, p)
        # (the reference image set renders with the same number of parts: the parts' sums differ in rounding)
        log.append("parts=%d" % p)
    elif op == "option":
        v = int(rs.rand() < 0.85); ctx.set_option(L.CEL_OPT_INCREMENTAL, v); log.append("incremental=%d" % v)
    if step % 200 == 199:
        print("step %d: %s" % (step + 1, counts), flush=True)
ctx.set_option(L.CEL_OPT_TILE_PARTS, 0); ctx.set_option(L.CEL_OPT_INCREMENTAL, 1)
print("ok: %d steps, renders %s" % (STEPS, counts))
for k, v in sorted(since.items(), key=lambda kv: -sum(kv[1]))[:25]:
    print("   since the last render: %-40s incremental %4d, every tile %4d" % (k, v[0], v[1]))
