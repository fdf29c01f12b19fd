"""One galaxy's sigma by griddy Gibbs (photon split, then sigma drawn from the shape step's conditional evaluated on a grid; every
other number at the truth) against the observed-data posterior on the same grid (the image log-likelihood, rendered): the two are
the same distribution when the conditional is the model's.    python tools/dbg/griddy_sigma.py [reference|mass|exact] [NSWEEP]"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np
import desi_mcmc_amd as cel
from desi_mcmc_amd import celeste_mcmc, synth
mode = sys.argv[1] if len(sys.argv) > 1 else "exact"
NSW = int(sys.argv[2]) if len(sys.argv) > 2 else 400
ctx = cel.default_context(0)
rs = np.random.RandomState(11)
H = W = 256
B = 5
bands = synth.make_bands(H, W, B)
bands[:, 0] = 200.0
cases = [(0.3, 2.5, 40., 0.6, 150.), (0.1, 3.5, 100., 0.8, 200.), (0.8, 1.5, 10., 0.4, 120.), (0.5, 0.7, 70., 0.5, 100.), (0.2, 5.0, 130., 0.7, 300.)]
cases += [(0.5, 3.5, 100., 0.8, 200.), (0.1, 2.0, 100., 0.8, 200.), (0.1, 3.5, 100., 0.5, 200.), (0.1, 3.5, 100., 0.8, 80.), (0.1, 3.5, 10., 0.8, 200.),
          (0.1, 1.2, 100., 0.8, 200.), (0.9, 3.5, 100., 0.8, 200.), (0.1, 3.5, 100., 0.8, 400.)]
pick = [int(a) for a in sys.argv[3].split(",")] if len(sys.argv) > 3 else range(len(cases))
for ci, (th, sg, ph, rh, fl) in enumerate(cases):
    if ci not in pick:
        continue
    S = 3                                                       # the galaxy, a star on top of its wing, a galaxy 40 px away
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
    # the observed-data posterior of sigma on a grid around the truth (flat prior on the grid)
    G = 81
    ll = np.zeros(G)
    grid = None
    span = 0.12
    for it in range(2):                                         # second pass: the grid centred on the mode, +- 5 sd
        grid = sg * np.linspace(1 - span, 1 + span, G) if it == 0 else grid
        for k, s_ in enumerate(grid):
            sh = shape.copy(); sh[0, 1] = s_
            ll[k] = iset.render(cel.SourceSet(ctx, S, B).set(typ, radec, counts, sh), loglik=True)[0]
        w = np.exp(ll - ll.max()); w /= w.sum()
        pm = (w * grid).sum(); psd = np.sqrt((w * (grid - pm) ** 2).sum())
        if it == 0:
            grid = np.linspace(pm - 5 * psd, pm + 5 * psd, G)
    kw = dict(reference={}, mass=dict(shape_mass="exact"), exact=dict(conditional="exact"))[mode]
    gf = celeste_mcmc.GibbsField(iset, list(range(B)), bands[:, 2], bands[:, 1], H * W, a_0=400., b_0=2.)
    g = celeste_mcmc.ModelGibbs([gf], typ, radec, flux, shape, seed=3 + ci + 1000 * int(os.environ.get("GRIDDY_SEED", "0")), flux_a_0=3., flux_b_0=.1, engine="host",
                                shape_logprior=lambda TH: np.zeros(TH.shape[0]), **kw)
    draws = []
    r2 = np.random.RandomState(100 + ci + 1000 * int(os.environ.get("GRIDDY_SEED", "0")))
    cur = sg
    idx = np.zeros(G, dtype=np.int64)
    for sw in range(NSW):
        g.shape[0, 1] = cur
        g._split_photons()
        for f in g.fields:
            f._counts = g.counts(f)
        TH = np.tile(shape[0], (G, 1)); TH[:, 1] = grid
        lp = g.shape_logprob(idx, TH)
        for f in g.fields:
            f._counts = None
        p = np.exp(lp - lp.max()); p /= p.sum()
        k0 = int(np.argmin(np.abs(grid - cur)))
        cur = grid[r2.choice(G, p=p)]
        if os.environ.get("GRIDDY_TRACE") and sw < 60:
            fin = np.isfinite(lp)
            print("sweep %3d: at grid %2d, allowed grid range %2d..%2d, conditional mode at %2d, lp[k0-2..k0+2] - lp[k0] = %s -> drew %2d" % (
                sw, k0, np.nonzero(fin)[0].min(), np.nonzero(fin)[0].max(), int(np.argmax(lp)), np.round(lp[max(k0 - 2, 0):k0 + 3] - lp[k0], 2).tolist(), int(np.argmin(np.abs(grid - cur)))))
        g.sweeps += 1
        draws.append(cur)
    d = np.array(draws[20:])
    nb = 20
    bm = d[:len(d) // nb * nb].reshape(nb, -1).mean(axis=1)
    se = bm.std(ddof=1) / np.sqrt(nb)
    print("case %d (theta %.1f sigma %.2f rho %.1f flux %.0f) %-9s: observed-data posterior mean %.4f sd %.4f | griddy chain mean %.4f sd %.4f  -> (chain - exact) / se = %+.1f   (%+.2f posterior sd)"
          % (ci, th, sg, rh, fl, mode, pm, psd, d.mean(), d.std(), (d.mean() - pm) / se, (d.mean() - pm) / psd), flush=True)
