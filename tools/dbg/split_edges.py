"""the photon split at the edges of a big galaxy's box: photons drawn in the outermost rows / columns (and in 8-px frames further in)
against their expectation nelec * rate / lambda, over many splits"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np
import desi_mcmc_amd as cel
from desi_mcmc_amd import synth
rho = float(sys.argv[1]) if len(sys.argv) > 1 else 0.5
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 400
ctx = cel.default_context(0)
if "full" in sys.argv:
    ctx.set_option(cel._lib.CEL_OPT_SPLIT_FULL_BOX, 1)
rs = np.random.RandomState(11)
H = W = 256
B = 5
bands = synth.make_bands(H, W, B)
bands[:, 0] = 200.0
th, sg, ph, rh, fl = (0.1, 3.5, 100., rho, 200.)
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
lam = iset.model_images()
nelec = rs.poisson(lam).astype(np.float64)
iset.set_nelec(nelec)
acc = None
for k in range(NS):
    iset.photon_split_resident(sset, 1000 + k)
    boxes, offs, data = iset.fetch_samples()
    if acc is None:
        acc = np.zeros_like(data)
    acc += data
acc /= NS
for b in range(B):
    st, bx = iset.stamps(sset, b, scaled=True)
    y0, y1, x0, x1 = boxes[0, b]
    assert bx[0].tolist() == [y0, y1, x0, x1], (bx[0], boxes[0, b])
    got = acc[offs[b]:offs[b + 1]].reshape(y1 - y0, x1 - x0)
    want = nelec[b, y0:y1, x0:x1] * st[0] / lam[b, y0:y1, x0:x1]
    line = []
    for d in (0, 1, 2, 4, 8, 16, 32):
        m = np.zeros_like(got, dtype=bool)
        m[d, d:got.shape[1] - d] = m[got.shape[0] - 1 - d, d:got.shape[1] - d] = True
        m[d:got.shape[0] - d, d] = m[d:got.shape[0] - d, got.shape[1] - 1 - d] = True
        se = np.sqrt(want[m].sum() / NS)
        line.append("ring %2d: %.3f / %.3f (%+.1f se)" % (d, got[m].sum(), want[m].sum(), (got[m].sum() - want[m].sum()) / max(se, 1e-9)))
    print("band %d box %s total %.1f / %.1f | " % (b, boxes[0, b].tolist(), got.sum(), want.sum()) + "; ".join(line))
