"""isolated galaxies: is the shape step's stationary sigma the posterior's?  z = (chain mean - truth) / chain sd per galaxy, and for
a few galaxies the exact observed-data log-likelihood along sigma (everything else at the truth) against the chain's mean"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import desi_mcmc_amd as cel
from desi_mcmc_amd import celeste_mcmc, synth
import test_calibration as tc
ctx = cel.default_context(0)
rs = np.random.RandomState(5)
N = 8
H = W = 64 * N
B = 5
bands = synth.make_bands(H, W, B)
bands[:, 0] = 200.0
cy, cx = np.meshgrid(np.arange(N), np.arange(N), indexing="ij")
pix = np.column_stack([cx.ravel() * 64 + 32.0, cy.ravel() * 64 + 32.0]) + rs.uniform(-3, 3, (N * N, 2))
S = N * N
typ = np.ones(S, np.int32)
shape = np.column_stack([rs.uniform(0, 1, S), 1.0 / np.sqrt(rs.gamma(1.5, 1.0, S)), rs.uniform(0, 180, S), rs.uniform(0, 1, S)])
flux = rs.gamma(3.0, 10.0, (S, 5))
radec = synth.pixel2equa(bands[0], pix)
counts = flux / bands[None, :, 2] * bands[None, :, 1]
iset = cel.ImageSet(ctx, bands, H, W)
sset = cel.SourceSet(ctx, S, B).set(typ, radec, counts, shape)
iset.render(sset, loglik=False)
nelec = rs.poisson(iset.model_images()).astype(np.float64)
iset.set_nelec(nelec)
only = sys.argv[1] if len(sys.argv) > 1 else "all"
gf = celeste_mcmc.GibbsField(iset, list(range(B)), bands[:, 2], bands[:, 1], H * W, a_0=400., b_0=2.)
g = celeste_mcmc.ModelGibbs([gf], typ, radec, flux, shape, seed=3, flux_a_0=3., flux_b_0=.1)
D = []
for k in range(150):
    if only == "shape":           # the shape step alone: photons, then shapes (fluxes, locations, sky stay at the truth)
        g._split_photons(); g.resample_shapes(); g.sweeps += 1
    else:
        g.sweep(shapes=True)
    D.append(g.shape.copy())
D = np.array(D)[30:]
z = (D[:, :, 1].mean(axis=0) - shape[:, 1]) / D[:, :, 1].std(axis=0)
print("mode", only, "z of sigma: mean %.2f sd %.2f; fraction z < -2: %.2f, > 2: %.2f" % (z.mean(), z.std(), (z < -2).mean(), (z > 2).mean()))
zz = (D[:, :, 3].mean(axis=0) - shape[:, 3]) / D[:, :, 3].std(axis=0)
print("z of rho: mean %.2f sd %.2f" % (zz.mean(), zz.std()))
zt = (D[:, :, 0].mean(axis=0) - shape[:, 0]) / D[:, :, 0].std(axis=0)
print("z of theta: mean %.2f sd %.2f" % (zt.mean(), zt.std()))
# exact observed-data log-likelihood along sigma for the 4 worst galaxies
worst = np.argsort(z)[:4]
for s in worst:
    grid = shape[s, 1] * np.linspace(0.8, 1.2, 41)
    ll = []
    for sg in grid:
        sh = shape.copy(); sh[s, 1] = sg
        ll.append(iset.render(cel.SourceSet(ctx, S, B).set(typ, radec, counts, sh), loglik=True)[0])
    ll = np.array(ll) + (-4 * np.log(grid) - grid ** -2.0)
    w = np.exp(ll - ll.max()); w /= w.sum()
    pm = (w * grid).sum(); psd = np.sqrt((w * (grid - pm) ** 2).sum())
    print("galaxy %d: truth %.4f  exact conditional posterior (others at truth) mean %.4f sd %.4f | chain mean %.4f sd %.4f  (z = %.1f)  theta %.2f rho %.2f flux %.0f"
          % (s, shape[s, 1], pm, psd, D[:, s, 1].mean(), D[:, s, 1].std(), z[s], shape[s, 0], shape[s, 3], flux[s].sum()))
