"""the exact conditional of a big galaxy's sigma (ModelGibbs._exact_terms + cel_patch_loglik_multi) against numpy on the library's own
per-source stamps (cel_render_stamps) and the fetched photons: value by value along sigma"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np
import desi_mcmc_amd as cel
from desi_mcmc_amd import celeste_mcmc, synth
ctx = cel.default_context(0)
rs = np.random.RandomState(11)
H = W = 256
B = 5
bands = synth.make_bands(H, W, B)
bands[:, 0] = 200.0
th, sg, ph, rh, fl = (0.1, 3.5, 100., float(sys.argv[1]) if len(sys.argv) > 1 else 0.8, 200.)
S = 3
typ = np.array([1, 0, 1], np.int32)
pix = np.array([[128.3, 127.6], [140.2, 131.0], [168.0, 120.0]])
shape = np.array([[th, sg, ph, rh], [0, 0, 0, 0], [0.5, 1.0, 20., 0.5]])
flux = np.array([[fl / 5] * 5, [30.] * 5, [25.] * 5])
radec = synth.pixel2equa(bands[0], pix)
counts = flux / bands[None, :, 2] * bands[None, :, 1]
iset = cel.ImageSet(ctx, bands, H, W)
sset = cel.SourceSet(ctx, S, B).set(typ, radec, counts, shape)
iset.render(sset, loglik=False)
nelec = rs.poisson(iset.model_images()).astype(np.float64)
iset.set_nelec(nelec)
gf = celeste_mcmc.GibbsField(iset, list(range(B)), bands[:, 2], bands[:, 1], H * W, a_0=400., b_0=2.)
g = celeste_mcmc.ModelGibbs([gf], typ, radec, flux, shape, seed=3, flux_a_0=3., flux_b_0=.1, engine="host",
                            shape_logprior=lambda TH: np.zeros(TH.shape[0]), conditional="exact")
g._split_photons()
f = g.fields[0]
f._counts = g.counts(f)
boxes, offs, data = iset.fetch_samples()
grid = sg * np.linspace(0.97, 1.03, 13)
TH = np.tile(shape[0], (grid.size, 1)); TH[:, 1] = grid
lp = g.shape_logprob(np.zeros(grid.size, dtype=np.int64), TH)
print("photons of the galaxy per band:", f.sums[0], " patch boxes", boxes[0].tolist())
ref = []
for k, s_ in enumerate(grid):
    sh = shape.copy(); sh[0, 1] = s_
    ps = cel.SourceSet(ctx, S, B).set(typ, radec, counts, sh)
    tot = 0.0
    parts = []
    for b in range(B):
        st, bx = iset.stamps(ps, b, scaled=True)
        y0, y1, x0, x1 = bx[0]
        stamp = np.zeros((H, W)); stamp[y0:y1, x0:x1] = st[0]
        py0, py1, px0, px1 = boxes[0, b]
        n = data[offs[b]:offs[b + 1]].reshape(py1 - py0, px1 - px0)          # source 0: index s * B + b = b
        img = np.zeros((H, W)); img[py0:py1, px0:px1] = n
        m = img > 0
        if (stamp[m] <= 0).any():
            tot = -np.inf
            break
        a, c = (img[m] * np.log(stamp[m])).sum(), stamp.sum()
        parts.append((a, c))
        tot += a - c
    ref.append(tot)
ref = np.array(ref)
d = lp - ref
print("sigma grid:", np.round(grid, 3))
print("library - numpy (constant wanted):", np.round(d - d[grid.size // 2], 6))
print("numpy conditional relative to centre:", np.round(ref - ref[grid.size // 2], 3))
obs = []
for s_ in grid:
    sh = shape.copy(); sh[0, 1] = s_
    obs.append(iset.render(cel.SourceSet(ctx, S, B).set(typ, radec, counts, sh), loglik=True)[0])
obs = np.array(obs)
print("observed-data log-lik relative to centre:", np.round(obs - obs[grid.size // 2], 3))
bxs = [iset.source_boxes(cel.SourceSet(ctx, S, B).set(typ, radec, counts, np.vstack([[th, s_, ph, rh], shape[1:]])))[0][:, 0] for s_ in grid]
print("boxes of band 1 along the grid:", [b[1].tolist() for b in bxs])
print("photon rects:", f.photon_rects[0].tolist())

# ---- the field render of the galaxy alone against its stamp (cel_render_stamps), band by band
one = cel.SourceSet(ctx, 1, B).set(typ[:1], radec[:1], counts[:1], shape[:1])
iset.render(one, loglik=False)
lam = iset.model_images()
for b in range(B):
    st, bx = iset.stamps(one, b, scaled=True)
    y0, y1, x0, x1 = bx[0]
    img = np.full((H, W), bands[b, 0]); img[y0:y1, x0:x1] += st[0]
    d = lam[b] - img
    edge = st[0][0].sum() + st[0][-1].sum() + st[0][:, 0].sum() + st[0][:, -1].sum()
    print("band %d: box %s, stamp sum %.3f, edge rows/cols hold %.4f photons; field render - (sky + stamp): max |d| %.3e, sum d %.3e, outside the box %.3e" % (
        b, bx[0].tolist(), st[0].sum(), edge, np.abs(d).max(), d.sum(), np.abs(np.where(img == bands[b, 0], d, 0)).max()))
