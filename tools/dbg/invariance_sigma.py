"""Does (photon split, then sigma from the exact conditional) leave the observed-data posterior invariant?  On a grid of G values
of one galaxy's sigma: pi_k from the rendered log-likelihood; P[k, k'] = the average over M splits at sigma_k of the
conditional's probability of sigma_k' (Rao-Blackwellised); then pi P against pi.  Exact whatever the chain's mixing.
    python tools/dbg/invariance_sigma.py [rho] [M] [mode]"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np
import desi_mcmc_amd as cel
from desi_mcmc_amd import celeste_mcmc, synth
rho = float(sys.argv[1]) if len(sys.argv) > 1 else 0.5
M = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
mode = sys.argv[3] if len(sys.argv) > 3 else "exact"
ctx = cel.default_context(0)
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
nelec = rs.poisson(iset.model_images()).astype(np.float64)
iset.set_nelec(nelec)
G = 13
grid = sg * np.linspace(0.97, 1.03, G)
ll = np.array([iset.render(cel.SourceSet(ctx, S, B).set(typ, radec, counts, np.vstack([[th, s_, ph, rh], shape[1:]])), loglik=True)[0] for s_ in grid])
pi = np.exp(ll - ll.max()); pi /= pi.sum()
kw = dict(reference={}, mass=dict(shape_mass="exact"), exact=dict(conditional="exact"))[mode]
gf = celeste_mcmc.GibbsField(iset, list(range(B)), bands[:, 2], bands[:, 1], H * W, a_0=400., b_0=2.)
g = celeste_mcmc.ModelGibbs([gf], typ, radec, flux, shape, seed=5, flux_a_0=3., flux_b_0=.1, engine="host",
                            shape_logprior=lambda TH: np.zeros(TH.shape[0]), **kw)
P = np.zeros((G, G)); P2 = np.zeros((G, G))
idx = np.zeros(G, dtype=np.int64)
TH = np.tile(shape[0], (G, 1)); TH[:, 1] = grid
for k in range(G):
    for m in range(M):
        g.shape[0, 1] = grid[k]
        g._split_photons()
        for f in g.fields: f._counts = g.counts(f)
        lp = g.shape_logprob(idx, TH)
        for f in g.fields: f._counts = None
        g.sweeps += 1
        p = np.exp(lp - lp.max()); p /= p.sum()
        P[k] += p; P2[k] += p * p
    P[k] /= M; P2[k] = np.sqrt(np.maximum(P2[k] / M - P[k] ** 2, 0) / M)
out = pi @ P
se = np.sqrt(((pi[:, None] * P2) ** 2).sum(axis=0))
print("mode %s, rho %.2f, %d splits per state" % (mode, rho, M))
print("sigma      ", np.round(grid, 3).tolist())
print("pi         ", np.round(pi, 4).tolist())
print("pi P       ", np.round(out, 4).tolist())
print("(pi P - pi) / se", np.round((out - pi) / np.maximum(se, 1e-12), 1).tolist())
print("mean under pi %.5f, under pi P %.5f (posterior sd %.5f)" % ((pi * grid).sum(), (out * grid).sum(), np.sqrt((pi * (grid - (pi * grid).sum()) ** 2).sum())))
print("P diagonal (staying put):", np.round(np.diag(P), 3).tolist())
