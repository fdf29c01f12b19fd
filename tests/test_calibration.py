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
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

K_DRAWS, THIN = 7, 3
FLUX_A, FLUX_B = 3.0, 0.1
EPS_A, EPS_B = 400.0, 2.0
CELL = 64


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


def run_replicate(cel, ctx, rep, engine, chain_seed, J, ncell=12, shapes=False, shape_args=None, shape_mass="reference"):
    """theta* at position J of a stationary stretch of K + 1 states -> its ranks: (S, 2) for the location, (S, 5) for the fluxes"""
    from desi_mcmc_amd import celeste_mcmc
    sc = make_scene(cel, ctx, rep, ncell, shapes)

    def chain():
        gf = celeste_mcmc.GibbsField(sc["iset"], list(range(sc["B"])), sc["bands"][:, 2], sc["bands"][:, 1], sc["H"] * sc["W"], a_0=EPS_A, b_0=EPS_B)
        for b in range(sc["B"]):
            sc["iset"].set_epsilon(b, sc["bands"][b, 0])                 # (the forward chain moved the sky levels)
        return celeste_mcmc.ModelGibbs([gf], sc["typ"], sc["radec"], sc["flux"], sc["shape"], seed=chain_seed, flux_a_0=FLUX_A,
                                       flux_b_0=FLUX_B, engine=engine, shape_args=shape_args, shape_mass=shape_mass)
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
    if K_DRAWS - J > 0:
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


def pooled_ranks(cel, ctx, engine, ncell, shapes=False, shape_args=None, shape_mass="reference"):
    import os
    reps = (K_DRAWS + 1) * max(1, int(os.environ.get("CEL_SBC_ROUNDS", "1")))       # every position J equally often (CEL_SBC_ROUNDS=8: 64 replicates)
    parts = zip(*[run_replicate(cel, ctx, rep, engine, chain_seed=rep, J=rep % (K_DRAWS + 1), ncell=ncell, shapes=shapes, shape_args=shape_args,
                                shape_mass=shape_mass)
                  for rep in range(reps)])
    return tuple(np.concatenate(p) for p in parts)


def shape_rank_table(ru, rf, rs_):
    out = {}
    for name, r in (("location", ru), ("flux", rf), ("theta", rs_[:, 0]), ("sigma", rs_[:, 1]), ("phi", rs_[:, 2]), ("rho", rs_[:, 3])):
        stat, p, counts = chi2_pvalue(r, K_DRAWS)
        out[name] = (round(stat, 2), p, counts.astype(int).tolist())
    return out


def test_sweep_with_the_shape_step_leaves_the_posterior_invariant():
    """sweep(shapes=True) with shape_mass="exact": the galaxies' (theta, sigma, phi, rho) by slice sampling along random
    directions with stepping out by doubling, under the shape step's own prior, every proposal charged counts * (its unit stamp
    summed over its own box) -- ranks of all four, with the locations and fluxes again"""
    import desi_mcmc_amd as cel
    ctx = cel.default_context(0)
    ru, rf, rs_ = pooled_ranks(cel, ctx, "host", 8, shapes=True, shape_mass="exact")
    out = shape_rank_table(ru, rf, rs_)
    print("SBC ranks with the shape step, exact mass term (%d sources, %d galaxies): %s" % (ru.shape[0], rs_.shape[0], out))
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
    (the default, parity with the reference); shape_mass="exact" (above) is the corrected step."""
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
    ru, rf = pooled_ranks(cel, ctx, "device", 12)
    p_flux = chi2_pvalue(rf, K_DRAWS)[1]
    print("masses off by 3 %%: p(flux) = %.3g" % p_flux)
    assert p_flux < 1e-8
    monkeypatch.setattr(field.ImageSet, "stamp_mass_end", real_end)
    real = celeste_mcmc.step_seed
    monkeypatch.setattr(celeste_mcmc, "step_seed", lambda seed, step, sweep, k=0: real(seed, step, 0, k))
    ru, rf = pooled_ranks(cel, ctx, "device", 12)
    p_loc = min(chi2_pvalue(ru[:, 0], K_DRAWS)[1], chi2_pvalue(ru[:, 1], K_DRAWS)[1])
    p_flux = chi2_pvalue(rf, K_DRAWS)[1]
    print("streams re-used by every sweep: p(location) = %.3g, p(flux) = %.3g" % (p_loc, p_flux))
    assert min(p_loc, p_flux) < 1e-8
