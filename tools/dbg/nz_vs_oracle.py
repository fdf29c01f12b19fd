"""debug: which proposals of test_photon_list_route_vs_oracle disagree with the oracle, and how (per band, both routes)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import desi_mcmc_amd as cel
from desi_mcmc_amd import _lib, synth
from oracle import oracle as orc
ctx = cel.default_context(0)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rs = np.random.RandomState(300 + seed)
S, H, W = 48, int(rs.choice([192, 256])), int(rs.choice([224, 320]))
f = synth.SyntheticField(ctx, S, 5, H, W, frac_gal=0.6, seed=400 + seed, with_nelec=False)
src = f.src
src["counts"] = src["counts"] * np.exp(rs.uniform(np.log(0.02), np.log(60.0), size=(S, 1)))
src["shape"][:, 1] = np.exp(rs.uniform(np.log(0.05), np.log(6.0), S))
edge = rs.rand(S) < 0.2
src["radec"][edge] = synth.pixel2equa(f.bands[0], np.column_stack([rs.choice([-3.0, 1.5, W - 2.0, W + 2.5], edge.sum()), rs.uniform(0, H, edge.sum())]))
gal = np.nonzero(src["type"] == 1)[0][:6]
if seed >= 1:
    src["type"][gal] = 2
    src["shape"][gal[:3], 1:] = [[9.0, 2.0, 4.0], [2.5, -1.0, 6.0], [30.0, 12.0, 8.0]]
    src["shape"][gal[3:], 1:] = [[9.0, 6.0, 4.0], [1.0, 1.0, 1.0], [16.0, -8.0, 4.0]]
f.sources.set(src["type"], src["radec"], src["counts"], src["shape"])
f.images.render(f.sources)
f.images.set_nelec(rs.poisson(f.images.model_images()).astype(np.float64))
P = 4
own = np.repeat(np.arange(S, dtype=np.int32), P)
jit = rs.normal(0, 1.0, size=(S * P, 2)) * np.repeat(rs.choice([3e-6, 3e-5, 4e-4, 2e-2, 6e-2], S), P)[:, None]
typ, U = np.repeat(src["type"], P), np.repeat(src["radec"], P, axis=0) + jit
cts, shp = np.repeat(src["counts"], P, axis=0), np.repeat(src["shape"], P, axis=0)
prop = cel.SourceSet(ctx, S * P, 5).set(typ, U, cts, shp)
got = {}
for route in (1, 2):
    ctx.set_option(_lib.CEL_OPT_PHOTON_LISTS, route)
    f.images.photon_split_resident(f.sources, seed=seed)
    got[route] = f.images.patch_loglik_resident(prop, own)
ctx.set_option(_lib.CEL_OPT_PHOTON_LISTS, 0)
boxes, offs, data = f.images.fetch_samples()
bands = [f.bands[b].copy() for b in range(5)]
for b in range(5):
    bands[b][36] = f.images.band(b)[36]
for p in range(S * P):
    o = own[p]
    t = np.zeros(4); per = []
    for b in range(5):
        z = data[offs[o * 5 + b]:offs[o * 5 + b + 1]]
        tb = orc.patch_loglik_terms(bands[b], H, W, typ[p], U[p], shp[p], cts[p, b], boxes[o, b], z)
        per.append(tb); t += tb
    want = t[0] - t[2]
    e1, e2 = got[1][p] - want, got[2][p] - want
    if abs(e1) > 1e-11 * t[1] + 1e-12 * t[2] + t[3] or abs(e2) > 1e-11 * t[1] + 1e-12 * t[2] + t[3]:
        pix = orc.equa2pixel(bands[2], U[p]); pix0 = orc.equa2pixel(bands[2], src["radec"][o])
        print("p", p, "src", o, "type", typ[p], "shape", shp[p], "counts", cts[p], "pix", pix, "from", pix0, "\n   want", want, "nz err", e1, "dense err", e2,
              "apt", t[1], "mass", t[2])
        for b in range(5):
            print("    band", b, "box", boxes[o, b], "terms", per[b], "nph", data[offs[o * 5 + b]:offs[o * 5 + b + 1]].sum())
