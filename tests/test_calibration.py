"""Simulation-based calibration of the COMPOSED Gibbs sweep ("only statistical parity (moment / Geweke tests)
applies" to the random-number rows; the reference keeps experiments/rjump/geweke_test.py for the same purpose).

The pieces of ModelGibbs are pinned one by one elsewhere (slicesample bit for bit, the binomial sampler's pmf, the Gamma
streams, photon conservation).  This file checks what none of them can: that a whole sweep -- photon split under the strict
boxes, sky levels, flux Gamma conditionals on the split's own mass sums, location slices on the photon lists, at the
LIBRARY DEFAULTS (CEL_OPT_SPLIT_REUSE = 2, photon lists by the layout pass, the shipping drop thresholds) -- leaves the
posterior invariant.

Method (Talts et al. 2018; Cook, Gelman & Rubin 2006), made exact for a chain that mixes slowly.  Draw theta* from the prior
and data from the model at theta*; then theta* is a draw from the posterior, and with one photon split at theta* the augmented
state (theta*, sky, photons) is a draw from the augmented posterior.  Run the chain FORWARD from it (ModelGibbs.sweep, the
production sweep, with the trace render that lets the next split re-use the model image) for K - J thinned draws and BACKWARD
(ModelGibbs.sweep_reversed: the sweep's blocks in reverse order, each reversible with respect to its conditional) for J: the
K + 1 states are a stationary stretch of the sweep's chain with theta* at position J.  With J taking each of 0..K equally
often over the replicates, the rank of theta* among the K + 1 states is EXACTLY uniform on 0..K whatever the autocorrelation
(in any sequence exactly one position holds each rank) -- if every conditional the sweep samples is the model's.  (Ranks of
theta* against thinned draws of a chain started at theta* are NOT uniform when sources overlap: two stars a pixel apart trade
flux for hundreds of sweeps, the walk stays on one side of its start, the histogram turns U-shaped.  The first version of
this test found that, not a bug.)  One replicate is ONE field of well-separated scenes of 1-3 overlapping sources (the
scenes share the bands' sky levels and nothing else), so the chain under test is the catalogue-wide device-resident sweep
itself, not a toy; K + 1 = 8 replicates (chain seeds 0..7: seed 0 is where round 3's shared-stream bug lived).

Priors (the sampler's own conjugate ones, with hyper-parameters that put the scenes in the synthetic benchmark's range):
    flux[s, band] ~ Gamma(a = 3, rate = 0.1) nmgy       (ModelGibbs(flux_a_0, flux_b_0); Source.resample_fluxes, sources.py:321-349)
    eps[band]     ~ Gamma(a = 400, rate = 2) counts/px  (GibbsField(a_0, b_0); Field.resample_photons, models.py:155-160)
    location      ~ uniform on the scene's 12 x 12 px core (the sampler's prior is flat: with posteriors a few tenths of a
                    pixel wide the box's edges matter to a percent of the scenes)
    galaxy shape  ~ the shape step's own log-prior, galaxy_shape_prior_constrained (celeste_galaxy_conditionals.py:268-275):
                    theta, rho ~ U(0, 1), phi ~ U(0, 180), sigma with density sigma^-4 exp(-sigma^-2), i.e. sigma^-2 ~ Gamma(3/2)
                    -- test_sweep_with_the_shape_step_..., sweep(shapes=True)
Deliberately broken sweeps -- the stamp masses off by 3 %, every sweep re-using the streams of the first -- must FAIL it.

With the shape step the test found three rules of the reference's sweep that are not the model's conditionals (DESIGN Q20);
ModelGibbs(conditional="exact") makes the three corrections and passes -- test_sweep_with_the_shape_step_..., and, free of any
sampler, test_split_then_sigma_leaves_the_observed_data_posterior_invariant (pi P = pi on a grid of sigma).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

K_DRAWS, THIN = 7, 3
FLUX_A, FLUX_B = 3.0, 0.1
EPS_A, EPS_B = 400.0, 2.0
CELL = 64
MOVE_CHECK = True          # (tools/dbg/sbc_exact.py switches blocks of the sweep off)


def chi2_pvalue(ranks, K):
    from scipy.stats import chi2
    counts = np.bincount(np.asarray(ranks).ravel(), minlength=K + 1).astype(float)
    expect = counts.sum() / (K + 1)
    stat = float(((counts - expect) ** 2 / expect).sum())
    return stat, float(chi2.sf(stat, K)), counts


def make_scene(cel, ctx, rep, NCELL=12, shapes=False):
    """theta* ~ prior and data ~ model(theta*) for one replicate -> everything ModelGibbs needs"""
    from desi_mcmc_amd import synth
    rs = np.random.RandomState(9000 + rep)
    H = W = max(CELL * NCELL, 320)
    B = 5
    bands = synth.make_bands(H, W, B)
    eps = rs.gamma(EPS_A, 1.0 / EPS_B, B)
    bands[:, 0] = eps
    cy, cx = np.meshgrid(np.arange(NCELL), np.arange(NCELL), indexing="ij")
    centres = np.column_stack([cx.ravel() * CELL + CELL / 2.0, cy.ravel() * CELL + CELL / 2.0])
    nper = rs.choice([1, 2, 3], centres.shape[0], p=[0.4, 0.4, 0.2])
    pix = np.concatenate([c[None, :] + rs.uniform(-6.0, 6.0, (n, 2)) for c, n in zip(centres, nper)])
    S = pix.shape[0]
    typ = (rs.rand(S) < 0.4).astype(np.int32)
    shape = np.column_stack([rs.uniform(0.1, 0.9, S), np.exp(rs.uniform(np.log(0.4), np.log(1.2), S)), rs.uniform(0, 180, S),
                             rs.uniform(0.3, 0.95, S)])
    if shapes:          # the shape step's prior (module docstring)
        shape = np.column_stack([rs.uniform(0, 1, S), 1.0 / np.sqrt(rs.gamma(1.5, 1.0, S)), rs.uniform(0, 180, S), rs.uniform(0, 1, S)])
    shape[typ == 0] = 0.0
    flux = rs.gamma(FLUX_A, 1.0 / FLUX_B, (S, 5))
    radec = synth.pixel2equa(bands[0], pix)
    counts = flux / bands[None, :, 2] * bands[None, :, 1]
    iset = cel.ImageSet(ctx, bands, H, W)
    sset = cel.SourceSet(ctx, S, B).set(typ, radec, counts, shape)
    iset.render(sset, loglik=False)
    nelec = rs.poisson(iset.model_images()).astype(np.float64)
    iset.set_nelec(nelec)
    return dict(bands=bands, iset=iset, typ=typ, radec=radec, flux=flux, shape=shape, H=H, W=W, B=B, S=S, pix=pix)


def run_replicate(cel, ctx, rep, engine, chain_seed, J, ncell=12, shapes=False, shape_args=None, shape_mass="reference", conditional="reference"):
    """theta* at position J of a stationary stretch of K + 1 states -> its ranks: (S, 2) for the location, (S, 5) for the fluxes"""
    from desi_mcmc_amd import celeste_mcmc
    sc = make_scene(cel, ctx, rep, ncell, shapes)

    def chain():
        gf = celeste_mcmc.GibbsField(sc["iset"], list(range(sc["B"])), sc["bands"][:, 2], sc["bands"][:, 1], sc["H"] * sc["W"], a_0=EPS_A, b_0=EPS_B)
        for b in range(sc["B"]):
            sc["iset"].set_epsilon(b, sc["bands"][b, 0])                 # (the forward chain moved the sky levels)
        return celeste_mcmc.ModelGibbs([gf], sc["typ"], sc["radec"], sc["flux"], sc["shape"], seed=chain_seed, flux_a_0=FLUX_A,
                                       flux_b_0=FLUX_B, engine=engine, shape_args=shape_args, shape_mass=shape_mass, conditional=conditional)
    du, df, ds = [], [], []
    g = chain()                                                          # forward: the production sweep + its trace render
    for k in range((K_DRAWS - J) * THIN):
        g.sweep(shapes=shapes)
        g.log_likelihood()
        assert g.active.all()                                            # every source keeps a sample patch
        if k % THIN == THIN - 1:
            du.append(g.u.copy())
            df.append(g.fluxes.copy())
            ds.append(g.shape.copy())
    if K_DRAWS - J > 0 and MOVE_CHECK:
        assert (np.abs(du[-1] - sc["radec"]).max(axis=1) > 0).mean() > 0.99     # the chains do move
    g = chain()                                                          # backward: from (theta*, a split at theta*)
    g.seed = chain_seed + 7919                                           # (its own streams)
    g._split_photons()
    for k in range(J * THIN):
        g.sweep_reversed(shapes=shapes)
        if k % THIN == THIN - 1:
            du.append(g.u.copy())
            df.append(g.fluxes.copy())
            ds.append(g.shape.copy())
    du, df, ds = np.array(du), np.array(df), np.array(ds)
    assert du.shape[0] == K_DRAWS
    ranks = (du < sc["radec"][None]).sum(axis=0), (df < sc["flux"][None]).sum(axis=0)
    if shapes:
        gal = sc["typ"] == 1
        return ranks + ((ds[:, gal] < sc["shape"][None, gal]).sum(axis=0),)
    return ranks


def pooled_ranks(cel, ctx, engine, ncell, shapes=False, shape_args=None, shape_mass="reference", conditional="reference"):
    import os
    reps = (K_DRAWS + 1) * max(1, int(os.environ.get("CEL_SBC_ROUNDS", "1")))       # every position J equally often (CEL_SBC_ROUNDS=8: 64 replicates)
    parts = zip(*[run_replicate(cel, ctx, rep, engine, chain_seed=rep, J=rep % (K_DRAWS + 1), ncell=ncell, shapes=shapes, shape_args=shape_args,
                                shape_mass=shape_mass, conditional=conditional)
                  for rep in range(reps)])
    return tuple(np.concatenate(p) for p in parts)


def shape_rank_table(ru, rf, rs_):
    out = {}
    for name, r in (("location", ru), ("flux", rf), ("theta", rs_[:, 0]), ("sigma", rs_[:, 1]), ("phi", rs_[:, 2]), ("rho", rs_[:, 3])):
        stat, p, counts = chi2_pvalue(r, K_DRAWS)
        out[name] = (round(stat, 2), p, counts.astype(int).tolist())
    return out


def test_sweep_with_the_shape_step_leaves_the_posterior_invariant():
    """sweep(shapes=True) with conditional="exact": the galaxies' (theta, sigma, phi, rho) by slice sampling along random
    directions with stepping out by doubling, under the shape step's own prior -- the photons split on whole boxes
    (CEL_OPT_SPLIT_FULL_BOX), and in the location AND the shape step every proposal charged counts * (its unit stamp summed over
    its own box) and refused when a photon of the source lies outside that box: the Gibbs conditionals of the model the
    renderer draws from.  Ranks of all four, with the locations and fluxes again.  (CEL_SBC_ROUNDS=24, 192 replicates, 8 829
    galaxies: p = 0.91 / 0.02 / 0.42 / 0.33 / 0.22 / 0.38, profiles/r05_calibration_exact_192_replicates.txt; with the mass term
    alone -- shape_mass="exact" -- sigma and rho fail at 192 replicates, p = 0.001 / 0.01: DESIGN Q20.)"""
    import desi_mcmc_amd as cel
    ctx = cel.default_context(0)
    ru, rf, rs_ = pooled_ranks(cel, ctx, "host", 8, shapes=True, conditional="exact")
    out = shape_rank_table(ru, rf, rs_)
    print("SBC ranks with the shape step, exact conditionals (%d sources, %d galaxies): %s" % (ru.shape[0], rs_.shape[0], out))
    for name, (stat, p, counts) in out.items():
        assert p > 1e-3 / 6, (name, stat, p, counts)


def test_the_reference_conditional_leaves_sigma_low():
    """What the calibration test FOUND (DESIGN Q20).  The shape step on the reference's conditional likelihood
    (Source.log_likelihood(shape=), sources.py:134-183: the source charged band_flux * sum(psf weights) whatever its shape,
    "model_outside ... should be small") is not the Gibbs conditional of the model the renderer draws from: the photons were
    split on the source's box, and the share of a proposal's stamp that lies on its box falls as sigma grows -- by parts in
    10^3 for an extended de Vaucouleurs-dominated galaxy, which at 10^5 photons is tens of nats per posterior standard
    deviation.  The chain's sigma sits 5-20 % low for a tenth of the galaxies: the top rank (theta* above every draw) holds
    twice its share.  theta, phi, rho, the locations and the fluxes stay calibrated.  The device engine runs this conditional
    (the default, parity with the reference); conditional="exact" (above) is the corrected sweep."""
    import desi_mcmc_amd as cel
    ctx = cel.default_context(0)
    ru, rf, rs_ = pooled_ranks(cel, ctx, "device", 8, shapes=True)
    out = shape_rank_table(ru, rf, rs_)
    print("SBC ranks with the shape step, the reference's mass term (%d sources, %d galaxies): %s" % (ru.shape[0], rs_.shape[0], out))
    stat, p, counts = out["sigma"]
    assert p < 1e-4 and counts[-1] > 1.5 * sum(counts) / len(counts), out["sigma"]
    for name in ("location", "flux", "phi", "rho"):
        assert out[name][1] > 1e-3 / 6, (name, out[name])


@pytest.mark.parametrize("engine", ["device", "host"])
def test_sweep_leaves_the_posterior_invariant(engine):
    import desi_mcmc_amd as cel
    ctx = cel.default_context(0)
    assert ctx.get_option(cel._lib.CEL_OPT_SPLIT_REUSE) == 2 and ctx.get_option(cel._lib.CEL_OPT_PHOTON_LISTS) == 0
    assert ctx.get_option(cel._lib.CEL_OPT_TAIL_LOG) == cel._lib.TAIL_LOG_DEFAULT     # the shipping configuration
    ru, rf = pooled_ranks(cel, ctx, engine, 12 if engine == "device" else 5)          # (the numpy engine: smaller fields)
    out = {}
    for name, r in (("location x", ru[:, 0]), ("location y", ru[:, 1]), ("flux", rf)):
        stat, p, counts = chi2_pvalue(r, K_DRAWS)
        out[name] = (round(stat, 2), p, counts.astype(int).tolist())
    print("SBC ranks (%s engine, %d sources in %d replicates): %s" % (engine, ru.shape[0], K_DRAWS + 1, out))
    for name, (stat, p, counts) in out.items():
        assert p > 1e-3 / 3, (name, stat, p, counts)                     # three tests: Bonferroni at 1e-3 overall
    # (the flux ranks of one source's five bands, and of the sources of one scene, are not independent: the chi-square's
    # nominal level is a guide, its threshold is what a broken sweep must miss by orders of magnitude -- below)


def test_a_broken_sweep_fails_the_calibration(monkeypatch):
    """the same test on sweeps whose flux conditional is off by 3 % in its rate (the stamp masses scaled: what a wrong mass
    short-cut would do), and on sweeps whose every step re-uses the streams of the chain's FIRST sweep (step_seed ignoring the
    sweep counter: the kind of bug round 3's review found by reading)"""
    import desi_mcmc_amd as cel
    from desi_mcmc_amd import celeste_mcmc, field
    ctx = cel.default_context(0)
    real_end = field.ImageSet.stamp_mass_end
    monkeypatch.setattr(field.ImageSet, "stamp_mass_end", lambda self: real_end(self) * 1.03)
    monkeypatch.setenv("CEL_HOST_FLUX", "1")          # (the flux step's host form is where the masses can be tampered with from here;
    ru, rf = pooled_ranks(cel, ctx, "device", 12)     #  the device form is pinned to it bit for bit, tests/test_gibbs.py)
    p_flux = chi2_pvalue(rf, K_DRAWS)[1]
    print("masses off by 3 %%: p(flux) = %.3g" % p_flux)
    assert p_flux < 1e-8
    monkeypatch.setattr(field.ImageSet, "stamp_mass_end", real_end)
    monkeypatch.delenv("CEL_HOST_FLUX")
    real = celeste_mcmc.step_seed
    monkeypatch.setattr(celeste_mcmc, "step_seed", lambda seed, step, sweep, k=0: real(seed, step, 0, k))
    ru, rf = pooled_ranks(cel, ctx, "device", 12)
    p_loc = min(chi2_pvalue(ru[:, 0], K_DRAWS)[1], chi2_pvalue(ru[:, 1], K_DRAWS)[1])
    p_flux = chi2_pvalue(rf, K_DRAWS)[1]
    print("streams re-used by every sweep: p(location) = %.3g, p(flux) = %.3g" % (p_loc, p_flux))
    assert min(p_loc, p_flux) < 1e-8


def _one_big_galaxy(cel, ctx, rho=0.5):
    """an extended de Vaucouleurs-dominated galaxy (sigma 3.5", axis ratio rho, 1.2e5 photons) with a star on its wing and a
    galaxy 40 px away, on a 256 x 256 frame with data drawn from the model"""
    from desi_mcmc_amd import synth
    rs = np.random.RandomState(11)
    H = W = 256
    bands = synth.make_bands(H, W, 5)
    bands[:, 0] = 200.0
    typ = np.array([1, 0, 1], np.int32)
    pix = np.array([[128.3, 127.6], [140.2, 131.0], [168.0, 120.0]])
    shape = np.array([[0.1, 3.5, 100., rho], [0, 0, 0, 0], [0.5, 1.0, 20., 0.5]])
    flux = np.array([[40.] * 5, [30.] * 5, [25.] * 5])
    radec = synth.pixel2equa(bands[0], pix)
    counts = flux / bands[None, :, 2] * bands[None, :, 1]
    iset = cel.ImageSet(ctx, bands, H, W)
    iset.render(cel.SourceSet(ctx, 3, 5).set(typ, radec, counts, shape), loglik=False)
    lam = iset.model_images()
    nelec = rs.poisson(lam).astype(np.float64)
    iset.set_nelec(nelec)
    return dict(iset=iset, bands=bands, typ=typ, radec=radec, shape=shape, flux=flux, counts=counts, lam=lam, nelec=nelec, H=H, W=W)


@pytest.mark.parametrize("conditional", ["exact", "reference"])
def test_split_then_sigma_leaves_the_observed_data_posterior_invariant(conditional):
    """The sharpest form of the calibration question, free of any sampler: on a grid of values of ONE galaxy's sigma (everything
    else at the truth), pi_k is the observed-data posterior (the rendered log-likelihood) and P[k, k'] the probability that a
    photon split at sigma_k followed by a draw of sigma from the shape step's conditional on the grid lands on sigma_k'
    (Rao-Blackwellised over M splits).  A Gibbs sweep whose conditionals are the model's has pi P = pi -- whatever its mixing;
    the test compares the two in total variation and in the mean.
    conditional="exact" (the proposal's stamp mass on ITS box, no photon outside that box, the split on WHOLE boxes:
    CEL_OPT_SPLIT_FULL_BOX) passes to the Monte-Carlo error.  The reference's three rules together (sources.py:166-170 the
    constant mass term; sources.py:134-183 the fixed data patch scored whatever the proposal's box;
    celeste_sample_sources.pyx:50-51 no photons in a box's first row and column) do not: for this galaxy, whose box the
    reference's bounding radius cuts where each edge row still holds two photons, probability flows towards smaller sigma
    across every change of the integer box (DESIGN Q20)."""
    import desi_mcmc_amd as cel
    from desi_mcmc_amd import celeste_mcmc
    ctx = cel.default_context(0)
    sc = _one_big_galaxy(cel, ctx)
    iset, typ, radec, counts, shape = sc["iset"], sc["typ"], sc["radec"], sc["counts"], sc["shape"]
    G, M = 9, 300
    grid = shape[0, 1] * np.linspace(0.98, 1.02, G)
    ll = np.array([iset.render(cel.SourceSet(ctx, 3, 5).set(typ, radec, counts, np.vstack([[0.1, s_, 100., 0.5], shape[1:]])), loglik=True)[0]
                   for s_ in grid])
    pi = np.exp(ll - ll.max())
    pi /= pi.sum()
    gf = celeste_mcmc.GibbsField(iset, list(range(5)), sc["bands"][:, 2], sc["bands"][:, 1], sc["H"] * sc["W"], a_0=EPS_A, b_0=EPS_B)
    g = celeste_mcmc.ModelGibbs([gf], typ, radec, sc["flux"], shape, seed=5, flux_a_0=FLUX_A, flux_b_0=FLUX_B, engine="host",
                                shape_logprior=lambda TH: np.zeros(TH.shape[0]), conditional=conditional)
    P, V = np.zeros((G, G)), np.zeros((G, G))
    TH = np.tile(shape[0], (G, 1))
    TH[:, 1] = grid
    idx = np.zeros(G, dtype=np.int64)
    for k in range(G):
        for m in range(M):
            g.shape[0, 1] = grid[k]
            g._split_photons()
            for f in g.fields:
                f._counts = g.counts(f)
            lp = g.shape_logprob(idx, TH)
            for f in g.fields:
                f._counts = None
            g.sweeps += 1
            p = np.exp(lp - lp.max())
            p /= p.sum()
            P[k] += p
            V[k] += p * p
        P[k] /= M
        V[k] = np.maximum(V[k] / M - P[k] ** 2, 0) / M
    out = pi @ P
    se = np.sqrt((pi[:, None] ** 2 * V).sum(axis=0))
    tv = 0.5 * np.abs(out - pi).sum()                                 # total variation between pi P and pi
    tv_noise = 0.5 * np.sqrt(2 / np.pi) * se.sum()                    # ... expected from the Monte-Carlo error alone
    shift = ((out - pi) * grid).sum() / np.sqrt((pi * (grid - (pi * grid).sum()) ** 2).sum())
    print("conditional=%s: pi %s\n    pi P %s\n    |pi P - pi| = %.4f (Monte-Carlo error %.4f); one sweep moves the mean by %+.4f posterior sd" % (
        conditional, np.round(pi, 4).tolist(), np.round(out, 4).tolist(), tv, tv_noise, shift))
    if conditional == "exact":
        assert tv < 0.004 + 3 * tv_noise and abs(shift) < 0.01, (tv, tv_noise, shift)
    else:
        assert tv > 0.03 and shift < -0.02, (tv, tv_noise, shift)


def test_full_box_split_reaches_the_first_row_and_column():
    """CEL_OPT_SPLIT_FULL_BOX.  0 (the reference, celeste_sample_sources.pyx:50-51): the first row and column of every sample patch
    stay empty; 1: a source takes part on its whole box -- its photons there match their expectation nelec * rate / lambda, every
    pixel's photons are conserved, pixels strictly inside draw the same numbers either way, and cel_samples_photon_rects is the
    patches' own rectangle."""
    import desi_mcmc_amd as cel
    L = cel._lib
    ctx = cel.default_context(0)
    sc = _one_big_galaxy(cel, ctx)
    iset = sc["iset"]
    sset = cel.SourceSet(ctx, 3, 5).set(sc["typ"], sc["radec"], sc["counts"], sc["shape"])
    assert ctx.get_option(L.CEL_OPT_SPLIT_FULL_BOX) == 0
    with pytest.raises(Exception):
        ctx.set_option(L.CEL_OPT_SPLIT_FULL_BOX, 2)
    NS = 120
    acc = {}
    try:
        for full in (0, 1):
            ctx.set_option(L.CEL_OPT_SPLIT_FULL_BOX, full)
            assert ctx.get_option(L.CEL_OPT_SPLIT_FULL_BOX) == full
            for k in range(NS):
                noise = iset.photon_split_resident(sset, 4000 + k)
                boxes, offs, data = iset.fetch_samples()
                if k == 0:
                    acc[full] = np.zeros_like(data)
                    first = data.copy()
                    rects = iset.photon_rects()
                    for s in range(3):
                        for b in range(5):
                            y0, y1, x0, x1 = boxes[s, b]
                            p = data[offs[s * 5 + b]:offs[s * 5 + b + 1]].reshape(y1 - y0, x1 - x0)
                            ys, xs = np.nonzero(p)
                            assert rects[s, b].tolist() == [y0 + ys.min(), y0 + ys.max() + 1, x0 + xs.min(), x0 + xs.max() + 1]
                    assert data.sum() + noise.sum() == sc["nelec"].sum()                     # conservation
                acc[full] += data
            acc[full] /= NS
            if full == 0:
                first0 = first
            else:                                                                            # strictly inside: the same draws
                for s in range(3):
                    for b in range(5):
                        y0, y1, x0, x1 = boxes[s, b]
                        a0 = first0[offs[s * 5 + b]:offs[s * 5 + b + 1]].reshape(y1 - y0, x1 - x0)
                        a1 = first[offs[s * 5 + b]:offs[s * 5 + b + 1]].reshape(y1 - y0, x1 - x0)
                        # (a pixel in ANOTHER source's first row or column has a different total, hence other draws: compare where
                        # no box edge passes)
                        edge = np.zeros((sc["H"], sc["W"]), dtype=bool)
                        for t in range(3):
                            ty0, ty1, tx0, tx1 = boxes[t, b]
                            edge[ty0, tx0:tx1] = True
                            edge[ty0:ty1, tx0] = True
                        m = ~edge[y0:y1, x0:x1]
                        assert np.array_equal(a0[m], a1[m])
    finally:
        ctx.set_option(L.CEL_OPT_SPLIT_FULL_BOX, 0)
    for b in range(5):
        st, bx = iset.stamps(sset, b, scaled=True)
        y0, y1, x0, x1 = boxes[0, b]
        want = sc["nelec"][b, y0:y1, x0:x1] * st[0] / sc["lam"][b, y0:y1, x0:x1]
        for full in (0, 1):
            got = acc[full][offs[b]:offs[b + 1]].reshape(y1 - y0, x1 - x0)
            edge_got = got[0].sum() + got[1:, 0].sum()
            edge_want = want[0].sum() + want[1:, 0].sum()
            if full == 0:
                assert edge_got == 0.0
            else:
                assert abs(edge_got - edge_want) < 5 * np.sqrt(edge_want / NS) + 1e-9, (b, edge_got, edge_want)
            inner_got, inner_want = got[1:, 1:].sum(), want[1:, 1:].sum()
            assert abs(inner_got - inner_want) < 5 * np.sqrt(inner_want / NS), (b, full, inner_got, inner_want)


def test_two_exposures_of_the_same_sources():
    """CelesteBase holds a LIST of fields (models.py:60-83: several exposures of the same sky); every source's flux and location
    conditionals add over them.  The same calibration test with two image sets -- all five bands, and a second exposure of g, r,
    i with other sky levels, calibrations and seeing -- sharing one catalogue (the host engine: the device engines run one field):
    a field counted twice, or left out of a conditional, fails it by orders of magnitude (asserted with the second field's masses
    dropped from the flux step)."""
    import desi_mcmc_amd as cel
    from desi_mcmc_amd import celeste_mcmc, synth
    ctx = cel.default_context(0)

    def ranks(break_it=False):
        allu, allf = [], []
        for rep in range(K_DRAWS + 1):
            sc = make_scene(cel, ctx, rep + 500, 5)
            rs = np.random.RandomState(77 + rep)
            b2 = sc["bands"][[1, 2, 3]].copy()
            b2[:, 0] = rs.gamma(EPS_A, 1.0 / EPS_B, 3)                 # another night: sky,
            b2[:, 2] *= rs.uniform(0.7, 1.4, 3)                        # calibration,
            b2[:, 12:24] *= 1.3                                        # seeing
            counts2 = sc["flux"][:, [1, 2, 3]] / b2[None, :, 2] * b2[None, :, 1]
            i2 = cel.ImageSet(ctx, b2, sc["H"], sc["W"])
            i2.render(cel.SourceSet(ctx, sc["S"], 3).set(sc["typ"], sc["radec"], counts2, sc["shape"]), loglik=False)
            i2.set_nelec(rs.poisson(i2.model_images()).astype(np.float64))
            J = rep % (K_DRAWS + 1)

            def chain(seed):
                for b in range(5):
                    sc["iset"].set_epsilon(b, sc["bands"][b, 0])
                for b in range(3):
                    i2.set_epsilon(b, b2[b, 0])
                g1 = celeste_mcmc.GibbsField(sc["iset"], [0, 1, 2, 3, 4], sc["bands"][:, 2], sc["bands"][:, 1], sc["H"] * sc["W"], a_0=EPS_A, b_0=EPS_B)
                g2 = celeste_mcmc.GibbsField(i2, [1, 2, 3], b2[:, 2], b2[:, 1], sc["H"] * sc["W"], a_0=EPS_A, b_0=EPS_B)
                g = celeste_mcmc.ModelGibbs([g1, g2], sc["typ"], sc["radec"], sc["flux"], sc["shape"], seed=seed, flux_a_0=FLUX_A, flux_b_0=FLUX_B)
                if break_it:
                    real = g2.iset.stamp_mass_end
                    g2.iset.stamp_mass_end = lambda: real() * 0.0
                return g
            du, df = [], []
            g = chain(rep)
            for k in range((K_DRAWS - J) * THIN):
                g.sweep()
                if k % THIN == THIN - 1:
                    du.append(g.u.copy()); df.append(g.fluxes.copy())
            g = chain(rep + 7919)
            g._split_photons()
            for k in range(J * THIN):
                g.sweep_reversed()
                if k % THIN == THIN - 1:
                    du.append(g.u.copy()); df.append(g.fluxes.copy())
            allu.append((np.array(du) < sc["radec"][None]).sum(axis=0))
            allf.append((np.array(df) < sc["flux"][None]).sum(axis=0))
        return np.concatenate(allu), np.concatenate(allf)
    ru, rf = ranks()
    out = {}
    for name, r in (("location", ru), ("flux g r i (both exposures)", rf[:, 1:4]), ("flux u z (one)", rf[:, [0, 4]])):
        stat, p, counts = chi2_pvalue(r, K_DRAWS)
        out[name] = (round(stat, 2), p, counts.astype(int).tolist())
    print("SBC ranks, two fields (%d sources): %s" % (ru.shape[0], out))
    for name, (stat, p, counts) in out.items():
        assert p > 1e-3 / 3, (name, stat, p, counts)
    ru, rf = ranks(break_it=True)
    p_bad = chi2_pvalue(rf[:, 1:4], K_DRAWS)[1]
    print("second field's masses dropped from the flux step: p(flux g r i) = %.3g" % p_bad)
    assert p_bad < 1e-8
