"""GPU tests of the Gibbs sweep (BASELINE configs[4] flavour): the catalogue-wide device sampler
ModelGibbs (CelesteBase.resample_model, models.py:75-83) and the per-object mirrors
Source.resample / resample_fluxes / resample_location (sources.py:242-349).

RNG parity with the reference (randomkit + numpy's global MT19937) is impossible; what is checked:
exact photon conservation, the Gamma conditionals' parameters, agreement of the batched device
likelihood with the per-object one, and the posterior a short chain reaches for a bright star."""
import os

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu
BANDS = ["u", "g", "r", "i", "z"]


@pytest.fixture(scope="module")
def cel():
    import desi_mcmc_amd as m
    return m


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    return oracle


def small_scene(cel, H=96, W=112, seed=3):
    """5 band images of one synthetic frame: a bright isolated star, a fainter star, two galaxies"""
    from test_hip_parity import frame_images
    rec = load_golden("bands_253.npz")
    imgs0 = frame_images(cel, rec, H, W)
    pix = np.array([[30.3, 40.6], [80.2, 25.7], [60.5, 70.1], [20.9, 78.4]])
    typ = np.array([0, 0, 1, 1])
    flux = np.array([[400., 500., 600., 550., 450.], [30., 40., 50., 45., 35.], [150., 200., 260., 240., 180.],
                     [60., 80., 100., 90., 70.]])
    shape = np.array([[0, 0, 0, 0], [0, 0, 0, 0], [0.4, 1.2, 35.0, 0.6], [0.7, 0.8, 120.0, 0.8]], dtype=float)
    params = []
    for s in range(4):
        u = imgs0[2].pixel2equa(pix[s])
        if typ[s]:
            params.append(cel.SrcParams(u=u, a=1, fluxes=flux[s].copy(), theta=shape[s, 0], sigma=shape[s, 1],
                                        phi=shape[s, 2], rho=shape[s, 3]))
        else:
            params.append(cel.SrcParams(u=u, a=0, fluxes=flux[s].copy()))
    from desi_mcmc_amd import models
    m = models.Celeste()
    m.initialize_sources(init_src_params=params)
    lam = np.stack([m.render_model_image(im) for im in imgs0])
    nelec = np.random.RandomState(seed).poisson(lam).astype(np.float64)
    imgs = frame_images(cel, rec, H, W, nelec=nelec)
    return imgs, params, pix, flux, nelec


def fisher_sigma_pix(cel, img, params, s):
    """Cramer-Rao error (pixels) of source s's x and y position in one image, all else fixed"""
    from desi_mcmc_amd import models
    m = models.Celeste()
    m.initialize_sources(init_src_params=params)
    lam = m.render_model_image(img)
    h = 1e-3
    u0 = np.array(params[s].u, copy=True)
    px = img.equa2pixel(u0)
    d = []
    for ax in range(2):
        dp = np.zeros(2)
        dp[ax] = h
        params[s].u = img.pixel2equa(px + dp)
        lp = m.render_model_image(img)
        params[s].u = img.pixel2equa(px - dp)
        lm = m.render_model_image(img)
        d.append((lp - lm) / (2 * h))
    params[s].u = u0
    info = np.array([[np.sum(d[i] * d[j] / lam) for j in range(2)] for i in range(2)])
    return info


def test_stamp_mass_vs_oracle(cel, orc):
    imgs, params, pix, flux, nelec = small_scene(cel)
    from desi_mcmc_amd import celeste_mcmc
    g = celeste_mcmc.ModelGibbs.from_images([dict(zip(BANDS, imgs))], params, seed=1)
    f = g.fields[0]
    mass = f.iset.stamp_mass(g._sources(f))
    rec = load_golden("bands_253.npz")
    B = orc.pack_bands(rec)
    for b in range(5):
        band = B[b].copy()
        band[24:26] = [112 / 2.0, 96 / 2.0]
        for s in range(4):
            p, _, _ = orc.source_patch(band, 96, 112, g.typ[s], g.u[s], g.shape[s])
            np.testing.assert_allclose(mass[s, b], p.sum(), rtol=1e-10)


def test_lockstep_device_loglik_equals_per_object_loglik(cel):
    """ModelGibbs.location_loglik (resident patches, one launch for all sources) == Source.log_likelihood
    on the same photons fetched to the host"""
    from desi_mcmc_amd import celeste_mcmc, sources
    imgs, params, pix, flux, nelec = small_scene(cel)
    g = celeste_mcmc.ModelGibbs.from_images([dict(zip(BANDS, imgs))], params, seed=5)
    g.resample_photons()
    f = g.fields[0]
    boxes, offs, data = f.iset.fetch_samples()
    rs = np.random.RandomState(0)
    idx = np.array([0, 1, 2, 3, 0, 2])
    U = g.u[idx] + rs.normal(0, 3e-5, size=(6, 2))
    U[4] = g.u[0] + 0.5                      # half a degree off: the star fails the overlap test -> -counts * sum(w)
    got = g.location_loglik(idx, U)
    for i, (s, u) in enumerate(zip(idx, U)):
        src = sources.Source(params[s])
        for b in range(5):
            k = s * 5 + b
            y0, y1, x0, x1 = boxes[s, b]
            if y1 > y0:
                src.sample_image_list.append((sources.SamplePatch(data[offs[k]:offs[k + 1]].reshape(y1 - y0, x1 - x0),
                                                                  (y0, y1), (x0, x1)), imgs[b], None))
        np.testing.assert_allclose(got[i], src.log_likelihood(u=u), rtol=1e-11)
    wsum = np.array([im.weights.sum() for im in imgs])
    np.testing.assert_allclose(got[4], -np.sum(g.counts(f)[0] * wsum), rtol=1e-13)


def test_model_gibbs_conserves_photons_and_draws_the_gamma_conditionals(cel):
    from desi_mcmc_amd import celeste_mcmc
    imgs, params, pix, flux, nelec = small_scene(cel)
    g = celeste_mcmc.ModelGibbs.from_images([dict(zip(BANDS, imgs))], params, seed=11, slice_args=dict(sigma=1e-3))
    f = g.fields[0]
    for sweep in range(3):
        noise = g.resample_photons()[0]
        # every photon of every image goes to exactly one source or to the sky
        np.testing.assert_array_equal(f.sums.sum(axis=0) + noise, nelec.reshape(5, -1).sum(axis=1))
        # the sky level's Gamma conditional (models.py:155-160): mean (a_0 + noise) / (b_0 + npix)
        assert np.all(np.abs(f.epsilon / ((5 + noise) / (.005 + 96 * 112)) - 1.0) < 0.01)
        assert np.all([abs(im.epsilon - e) == 0 for im, e in zip(imgs, f.epsilon)])
        # the flux Gamma conditional (sources.py:341-345): shape 5 + photons, rate .005 + mass kappa / calib, every
        # (source, band) drawn from its own stream -- re-drawn here from the same streams
        mass = f.iset.stamp_mass(f.sset)
        fl = g.resample_fluxes().copy()
        std = celeste_mcmc.gamma_by_stream((5. + f.sums).ravel(), g.step_seed("flux"), np.arange(20)).reshape(4, 5)
        np.testing.assert_allclose(fl, std / (.005 + mass * (f.kappa / f.calib)[None, :]), rtol=1e-12)
        g.resample_locations()
        g.sweeps += 1
    assert g.timing["rounds"] > 0 and g.timing["evals"] >= 4 * g.timing["rounds"] // 4


@pytest.mark.parametrize("letters", [[0, 1, 2, 3, 4], [2, 2, 4]])
def test_device_flux_step_is_the_host_flux_step(cel, letters):
    """cel_flux_conditionals (round 6: sums, stamp masses, Gamma variates and the new expected counts never leave the device)
    against the host form of Source.resample_fluxes (sources.py:321-349: sums and masses read back, fluxes formed in numpy):
    two chains from the same seed, one per form, stay equal BIT FOR BIT through whole sweeps -- fluxes, locations, the trace --
    with very faint sources (the mass short cut's leftovers take the mass kernel proper), a source off the frame (no patch:
    left alone) and, in the second case, two images of one band letter and letters without an image (drawn from the prior)."""
    from desi_mcmc_amd import celeste_mcmc, synth
    ctx = cel.default_context(0)
    B = len(letters)
    S = 260
    f0 = synth.SyntheticField(ctx, S, B, 288, 320, frac_gal=0.5, seed=17)
    flux5 = np.zeros((S, 5))
    for b, L in enumerate(letters):
        flux5[:, L] = f0.src["flux"][:, b]
    flux5[flux5 == 0] = 3.0
    flux5[5] *= 1e-4                                     # far below eps / 1024: not vouched for by the split's mass sums
    flux5[6] *= 1e-3
    u = f0.src["radec"].copy()
    u[9] = synth.pixel2equa(f0.bands[0], np.array([[5000.0, 40.0]]))[0]         # off the frame: no patch anywhere
    chains = []
    for device_flux in (True, False):
        imgs = cel.ImageSet(ctx, f0.bands, f0.H, f0.W, nelec=f0.nelec)
        gf = celeste_mcmc.GibbsField(imgs, letters, f0.bands[:, 2], f0.bands[:, 1], f0.H * f0.W)
        g = celeste_mcmc.ModelGibbs([gf], f0.src["type"], u, flux5, f0.src["shape"], seed=5, slice_args=dict(step_out=False, sigma=0.001))
        g.device_flux = device_flux
        chains.append((g, gf))
    (ga, fa), (gb, fb) = chains
    assert ga.device_flux and not gb.device_flux
    for sweep in range(3):
        for g in (ga, gb):
            g.resample_photons()
        assert ga._device_flux_applies() and not gb._device_flux_applies()
        assert np.array_equal(fa.sums, fb.sums) and np.array_equal(ga.active, gb.active) and not ga.active[9]
        fl_a, fl_b = ga.resample_fluxes().copy(), gb.resample_fluxes().copy()
        assert np.array_equal(fl_a, fl_b), (sweep, np.nonzero(fl_a != fl_b))
        assert np.array_equal(fl_a[9], flux5[9])                                   # the source without a patch kept its fluxes
        # the device's catalogue holds the new counts: the location step that follows needs no upload and gives the same chains
        ua, ub = ga.resample_locations().copy(), gb.resample_locations().copy()
        assert np.array_equal(ua, ub)
        for g in (ga, gb):
            g.merge_ranks(); g.sweeps += 1
        assert ga.log_likelihood() == gb.log_likelihood()
    unused = [L for L in range(5) if L not in letters]
    if unused:                                          # a letter without an image: Gamma(a0) / b0 from its own stream, in both forms
        assert np.all(fl_a[ga.active][:, unused] > 0)


def test_flux_conditionals_error_paths(cel):
    """cel_flux_conditionals refuses what it cannot do: no resident split, another catalogue than the split's, a band letter outside
    ugriz, non-positive priors or calibration"""
    from desi_mcmc_amd import synth
    ctx = cel.default_context(0)
    f = synth.SyntheticField(ctx, 30, 5, 128, 160, frac_gal=0.5, seed=4)
    letters, cal, kap = [0, 1, 2, 3, 4], f.bands[:, 2], f.bands[:, 1]
    with pytest.raises(ValueError, match="resident photon split"):
        f.images.flux_conditionals(f.sources, 1, 5.0, 0.005, letters, cal, kap)
    f.images.photon_split_resident(f.sources, seed=3)
    other = cel.SourceSet(ctx, 7, 5).set(f.src["type"][:7], f.src["radec"][:7], f.src["counts"][:7], f.src["shape"][:7])
    with pytest.raises(ValueError, match="resident photon split"):
        f.images.flux_conditionals(other, 1, 5.0, 0.005, letters, cal, kap)
    for bad in (dict(letters=[0, 1, 2, 3, 5]), dict(a0=0.0), dict(b0=-1.0), dict(cal=cal * 0.0)):
        with pytest.raises(ValueError):
            f.images.flux_conditionals(f.sources, 1, bad.get("a0", 5.0), bad.get("b0", 0.005), bad.get("letters", letters), bad.get("cal", cal), kap)
    with pytest.raises(ValueError):
        f.images.flux_conditionals(f.sources, 1, 5.0, 0.005, letters[:3], cal, kap)
    new, act = f.images.flux_conditionals(f.sources, 1, 5.0, 0.005, letters, cal, kap)
    assert new.shape == (30, 5) and act.shape == (30,) and act.dtype == bool and np.all(new > 0) and act.any()
    # same seed, same split: the same Gamma draws (the masses now come from the mass kernel proper -- the catalogue's generation
    # moved with its counts, so the split's own sums no longer vouch -- which agrees with them to 1e-10)
    again, _ = f.images.flux_conditionals(f.sources, 1, 5.0, 0.005, letters, cal, kap)
    np.testing.assert_allclose(again, new, rtol=1e-9)


def test_short_chain_recovers_a_bright_star(cel):
    """the posterior of the bright star's position: mean within 4 Fisher sigma of the truth, spread
    of the order of the Fisher sigma; fluxes within 5 sigma of their Poisson error"""
    from desi_mcmc_amd import celeste_mcmc
    imgs, params, pix, flux, nelec = small_scene(cel)
    info = sum(fisher_sigma_pix(cel, im, params, 0) for im in imgs)
    sig = np.sqrt(np.diag(np.linalg.inv(info)))                 # pixels
    assert np.all(sig < 0.05)
    g = celeste_mcmc.ModelGibbs.from_images([dict(zip(BANDS, imgs))], params, seed=2, slice_args=dict(sigma=1e-3))
    # start the star a fifth of a pixel off
    g.u[0] = imgs[2].pixel2equa(pix[0] + np.array([0.2, -0.15]))
    burn, keep = 15, 60
    locs, fls = [], []
    for it in range(burn + keep):
        g.sweep()
        if it >= burn:
            locs.append(imgs[2].equa2pixel(g.u[0]))
            fls.append(g.fluxes[0].copy())
    locs, fls = np.array(locs), np.array(fls)
    err = locs.mean(axis=0) - pix[0]
    assert np.all(np.abs(err) < 4 * sig), (err, sig)
    spread = locs.std(axis=0)
    assert np.all(spread < 4 * sig) and np.all(spread > 0.2 * sig), (spread, sig)
    counts = flux[0] / np.array([im.calib for im in imgs]) * np.array([im.kappa for im in imgs])
    rel = (fls.mean(axis=0) - flux[0]) / flux[0]
    assert np.all(np.abs(rel) < 5.0 / np.sqrt(counts) + 0.01), rel
    # the galaxies stay where their photons are, too
    for s in (2, 3):
        assert np.all(np.abs(imgs[2].equa2pixel(g.u[s]) - pix[s]) < 0.5)


def test_reference_effective_call_sigma_one_degree(cel):
    """Source.resample_location's call as the reference makes it (sigma = 1 degree, no stepping out):
    the first shrink steps land hundreds of pixels away, where the model underflows to exactly 0
    (the kernel's far-proposal shortcut); the chain must still end next to the photons"""
    from desi_mcmc_amd import celeste_mcmc
    imgs, params, pix, flux, nelec = small_scene(cel)
    g = celeste_mcmc.ModelGibbs.from_images([dict(zip(BANDS, imgs))], params, seed=4, slice_args="literal")
    assert g.slice_args == dict(step_out=False)
    assert celeste_mcmc.ModelGibbs.from_images([dict(zip(BANDS, imgs))], params).slice_args == dict(step_out=False, sigma=1e-3)
    for it in range(3):
        g.sweep()
    assert g.timing["rounds"] / 3 > 20                         # ~log2(1 deg / posterior width) shrinks per axis
    assert np.all(np.abs(imgs[2].equa2pixel(g.u[0]) - pix[0]) < 0.3)


def test_celeste_base_resample_model_and_per_object_resample(cel):
    from desi_mcmc_amd import models
    imgs, params, pix, flux, nelec = small_scene(cel)
    m = models.Celeste()
    m.add_field(dict(zip(BANDS, imgs)))
    m.initialize_sources(init_src_params=params)
    u0 = np.array([s.params.u for s in m.srcs])
    m.resample_model(n_sweeps=2, seed=9, slice_args=dict(sigma=1e-3))
    u1 = np.array([s.params.u for s in m.srcs])
    assert np.all(u1 != u0) and np.all(np.abs(u1 - u0) < 2e-4)
    assert all(np.asarray(s.params.fluxes).shape == (5,) and np.all(np.asarray(s.params.fluxes) > 0) for s in m.srcs)
    # the sky levels were redrawn and written to the image objects (models.py:156-160)
    assert all(abs(im.epsilon / np.median(im.nelec) - 1) < 0.2 for im in imgs)
    # the reference's per-object loop: photons to the host, then one Source.resample per source
    rng = np.random.RandomState(3)
    m.field_list[0].resample_photons(m.srcs, seed=77, rng=rng)
    assert all(len(s.sample_image_list) == 5 for s in m.srcs)
    before = np.array([s.params.u for s in m.srcs])
    m.srcs[0].resample_fluxes(rng=rng)
    z = np.array([np.sum(sp.data) for sp, _, _ in m.srcs[0].sample_image_list])
    cnt = np.asarray(m.srcs[0].params.fluxes) / np.array([i.calib for i in imgs]) * np.array([i.kappa for i in imgs])
    assert np.all(np.abs(cnt / (5 + z) - 1) < 6 / np.sqrt(z))            # Gamma(5 + z, ~1 / (kappa / calib))
    m.srcs[0].resample_location(rng=rng, sigma=1e-3)
    after = np.array(m.srcs[0].params.u)
    assert np.all(after != before[0]) and np.all(np.abs(imgs[2].equa2pixel(after) - pix[0]) < 0.3)
    m.srcs[1].resample_location(rng=rng, sigma=1.0)                         # the call as the reference executes it (sigma = 1 deg)
    assert np.all(np.abs(imgs[2].equa2pixel(m.srcs[1].params.u) - pix[1]) < 1.0)
    m.resample_sources(rng=rng)                                             # every source, the intended 0.001-degree interval
    assert np.all(np.abs(imgs[2].equa2pixel(m.srcs[0].params.u) - pix[0]) < 0.3)
    m.srcs[0].store_sample()
    m.srcs[0].store_loglike()
    assert m.srcs[0].location_samples.shape == (1, 2) and np.isfinite(m.srcs[0].loglike_samples[0])


def test_background_patch_and_image_like(cel, orc):
    """generate_background_patch / get_active_sources / make_bbox_dict (sources.py:434-483) and the
    star <-> galaxy move's image_like (:277-291)"""
    from desi_mcmc_amd import models, sources
    imgs, params, pix, flux, nelec = small_scene(cel)
    m = models.Celeste()
    m.add_field(dict(zip(BANDS, imgs)))
    m.initialize_sources(init_src_params=params)
    img = imgs[2]
    for s in m.srcs:
        s.bounding_boxes = sources.make_bbox_dict(s.params, imgs, pixel_radius=45)
    # the boxes: floor / ceil of centre -+ radius, cut to the frame (sources.py:441-455), scalar restatement
    for s in m.srcs:
        for im in imgs:
            px, py = im.equa2pixel(s.params.u)
            want = ((max(0, int(np.floor(px - 45))), min(112, int(np.ceil(px + 45)))),
                    (max(0, int(np.floor(py - 45))), min(96, int(np.ceil(py + 45)))))
            assert s.bounding_boxes[im] == want
    with pytest.raises(NotImplementedError):
        sources.make_bbox_dict(params[0], imgs)
    src = m.srcs[2]
    act = sources.get_active_sources(src, m.srcs, img)
    assert src not in act and len(act) >= 1
    # the reference's pairwise test (sources.py:459-471), written out per pair
    for other in m.srcs:
        if other is src:
            continue
        (ax, ay), (bx, by) = src.bounding_boxes[img], other.bounding_boxes[img]
        hit = (abs(ax[0] - bx[0]) * 2 < (ax[1] - ax[0]) + (bx[1] - bx[0])) and (abs(ay[0] - by[0]) * 2 < (ay[1] - ay[0]) + (by[1] - by[0]))
        assert hit == (other in act)
    assert sources.get_active_sources(src, [src], img) == []
    bg = sources.generate_background_patch(src, m.srcs, img)
    xlim, ylim = src.bounding_boxes[img]
    box = [ylim[0], ylim[1], xlim[0], xlim[1]]
    band = orc.pack_bands(load_golden("bands_253.npz"))[2].copy()
    band[24:26] = [112 / 2.0, 96 / 2.0]
    want = np.zeros((ylim[1] - ylim[0], xlim[1] - xlim[0])) + img.epsilon
    for a in act:
        p, _, _ = a.compute_model_patch(img, xlim=xlim, ylim=ylim)
        want += p
    np.testing.assert_allclose(bg, want, rtol=1e-12)
    src.background_image_dict = {img: bg}
    ll = src.image_like(src, img)
    cts = src.flux_in_image(img)
    obs = img.nelec[ylim[0]:ylim[1], xlim[0]:xlim[1]]
    o_ll = orc.patch_loglik(band, 96, 112, 1, src.params.u, src.params.shape, cts, box, np.stack([obs, bg]).ravel(), mode=4)
    np.testing.assert_allclose(ll, o_ll, rtol=1e-11)
    model, _, _ = src.compute_model_patch(img, xlim=xlim, ylim=ylim)
    np.testing.assert_allclose(ll, orc.poisson_loglike(obs, bg + model), rtol=1e-11)
    # masked pixels (invvar == 0) drop out of both sums
    img.invvar = np.ones_like(img.nelec)
    img.invvar[ylim[0] + 3:ylim[0] + 20, xlim[0] + 5:xlim[0] + 30] = 0.0
    nel0 = img.nelec
    try:
        mask = img.invvar[ylim[0]:ylim[1], xlim[0]:xlim[1]]
        np.testing.assert_allclose(src.image_like(src, img), orc.poisson_loglike(obs, bg + model, mask), rtol=1e-11)
        # a NEGATIVE observed count (sky-subtracted data) is data, not a mask: poisson_loglike keeps it (sources.py:9)
        neg = nel0.copy()
        neg[ylim[0] + 25:ylim[0] + 28, xlim[0] + 2:xlim[0] + 9] = -3.0
        neg[ylim[0] + 5, xlim[0] + 7] = -1.0                        # masked AND negative
        img.nelec = neg                                             # (the move reads the observed box from the image object)
        obs2 = neg[ylim[0]:ylim[1], xlim[0]:xlim[1]]
        want = orc.poisson_loglike(obs2, bg + model, mask)
        assert abs(want - orc.poisson_loglike(obs, bg + model, mask)) > 1.0
        np.testing.assert_allclose(src.image_like(src, img), want, rtol=1e-11)
    finally:
        del img.invvar
        img.nelec = nel0


class _FlatPrior(object):
    """stand-in for the model's priors (out of scope): what Source.resample_type needs of its model"""
    bands = BANDS

    def __init__(self, cel, imgs, gal_shape):
        import types
        self.cel = cel
        self.field_list = [types.SimpleNamespace(img_dict=dict(zip(BANDS, imgs)))]
        self.gal_shape = gal_shape
        self.lp = {0: -3.0, 1: -4.5}

    def logprior(self, params):
        return self.lp[params.a]

    def prior_sample(self, kind, u):
        if kind == 'galaxy':
            th, sg, ph, rh = self.gal_shape
            return self.cel.SrcParams(u=u, a=1, fluxes=self.fluxes.copy(), theta=th, sigma=sg, phi=ph, rho=rh), -2.0
        return self.cel.SrcParams(u=u, a=0, fluxes=self.fluxes.copy()), -1.0


def test_type_move_scores_both_types_in_one_call(cel, orc):
    """Source.resample_type / calculate_acceptance_logprob / propose_other_type_prior (sources.py:247-306): the
    acceptance ratio's likelihood terms against the oracle (mode 4 of orc_patch_loglik: poisson_loglike of the
    observed box on background + model, all five bands), its prior / proposal terms, and the move itself: a
    galaxy mis-typed as a star is flipped, a true star is not turned into an extended galaxy."""
    from desi_mcmc_amd import models, sources
    imgs, params, pix, flux, nelec = small_scene(cel)
    m = models.Celeste()
    m.add_field(dict(zip(BANDS, imgs)))
    m.initialize_sources(init_src_params=params)
    for s in m.srcs:
        s.bounding_boxes = sources.make_bbox_dict(s.params, imgs, pixel_radius=40)
    bands = orc.pack_bands(load_golden("bands_253.npz"))
    for who, truth_is_galaxy in ((2, True), (0, False)):
        src = m.srcs[who]
        true_params = src.params
        src.background_image_dict = {im: sources.generate_background_patch(src, m.srcs, im) for im in imgs}
        prior = _FlatPrior(cel, imgs, gal_shape=(0.4, 1.2, 35.0, 0.6))
        prior.fluxes = np.array([true_params.flux_dict[b] for b in BANDS])
        src.model = prior
        if truth_is_galaxy:                       # start from the wrong type: a star with the galaxy's fluxes
            src.params = cel.SrcParams(u=true_params.u, a=0, fluxes=prior.fluxes.copy())
        proposal, logq, logrev, logdet = src.propose_other_type_prior()
        assert proposal.a == (1 if src.is_star() else 0) and logdet == 0. and logrev == prior.logprior(src.params)
        assert logq == (-2.0 if proposal.a == 1 else -1.0)
        got = src.calculate_acceptance_logprob(proposal, logq, logrev, logdet, imgs)
        like = {}
        for name, q in (("cur", src.params), ("prop", proposal)):
            tot = 0.0
            for b, im in enumerate(imgs):
                band = bands[b].copy()
                band[24:26] = [112 / 2.0, 96 / 2.0]
                xlim, ylim = src.bounding_boxes[im]
                obs = im.nelec[ylim[0]:ylim[1], xlim[0]:xlim[1]]
                data = np.stack([obs, src.background_image_dict[im]]).ravel()
                shape = q.shape if q.a == 1 else np.zeros(4)
                tot += orc.patch_loglik(band, 96, 112, q.a, q.u, shape, q.flux_dict[im.band] / im.calib * im.kappa,
                                        [ylim[0], ylim[1], xlim[0], xlim[1]], data, mode=4)
            like[name] = tot
        want = (like["prop"] + prior.logprior(proposal)) - (like["cur"] + prior.logprior(src.params)) + (logrev - logq) + logdet
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-10 * abs(like["cur"]))      # a difference of two sums of ~|like| each
        np.testing.assert_allclose(src.image_like_batch([src.params, proposal], imgs), [like["cur"], like["prop"]], rtol=1e-11)
        accepted = src.resample_type(rng=np.random.RandomState(1))
        if truth_is_galaxy:
            assert got > 50.0 and accepted and src.is_galaxy()            # the galaxy's photons do not fit a point source
        else:
            assert got < -50.0 and not accepted and src.is_star()
        src.params = true_params


def test_device_slice_sampler_follows_the_host_engine_chain_by_chain(cel):
    """cel_slice_locations (state machine on the device) and the numpy engine draw the same per-chain
    streams and do the same arithmetic: after a sweep's location update every source sits at the same
    place, to the last bit, whichever engine ran -- also with the reference's 1-degree interval."""
    from desi_mcmc_amd import celeste_mcmc
    imgs, params, pix, flux, nelec = small_scene(cel)
    eps0 = [im.epsilon for im in imgs]
    for sa in (None, "literal"):
        gs = {}
        for eng in ("host", "device"):
            for im, e in zip(imgs, eps0):          # a sweep redraws the images' sky levels: both engines start from the same
                im.epsilon = e
            g = celeste_mcmc.ModelGibbs.from_images([dict(zip(BANDS, imgs))], params, seed=21, slice_args=sa, engine=eng)
            for _ in range(3):
                g.sweep()
            gs[eng] = g
        assert gs["host"].timing["rounds"] == gs["device"].timing["rounds"]
        assert gs["host"].timing["evals"] == gs["device"].timing["evals"]
        assert np.array_equal(gs["host"].u, gs["device"].u)
        assert np.array_equal(gs["host"].fluxes, gs["device"].fluxes)
    # a source without any patch (off the frame) is left alone by both
    far = cel.SrcParams(u=imgs[2].pixel2equa(np.array([5000.0, 40.0])), a=0, fluxes=np.full(5, 50.0))
    for eng in ("host", "device"):
        for im, e in zip(imgs, eps0):
            im.epsilon = e
        g = celeste_mcmc.ModelGibbs.from_images([dict(zip(BANDS, imgs))], params + [far], seed=3, slice_args=dict(sigma=1e-3), engine=eng)
        g.sweep()
        assert not g.active[4] and np.array_equal(g.u[4], np.asarray(far.u)) and np.array_equal(g.fluxes[4], np.full(5, 50.0))
    # options the device engine does not run fall back to the host engine under "auto" and are refused under "device"
    g = celeste_mcmc.ModelGibbs.from_images([dict(zip(BANDS, imgs))], params, seed=1, slice_args=dict(sigma=1e-4, step_out=True))
    assert not g._device_engine_applies()
    g.sweep()
    g2 = celeste_mcmc.ModelGibbs.from_images([dict(zip(BANDS, imgs))], params, seed=1, slice_args=dict(sigma=1e-4, step_out=True), engine="device")
    g2.resample_photons()
    g2.resample_fluxes()
    with pytest.raises(ValueError):
        g2.resample_locations()


def test_celeste_em_driver_increases_the_likelihood(cel):
    """celeste_em (celeste_em.py:17-180) on the device reductions, with a stand-in for the black-body
    photometry (planck.py is outside the path): EM must not decrease the marginal likelihood, and
    it recovers the brightness of stars whose photons it is given."""
    from desi_mcmc_amd import celeste, celeste_em
    from test_hip_parity import frame_images

    class Planck(object):          # a smooth, band-dependent photons-per-joule law (the reference's needs filter curves)
        lens_area, exposure_duration, sun_wattage, m_per_ly = 3.68, 54.0, 3.846e26, 9.4607e15
        centre = dict(u=3500., g=4800., r=6200., i=7600., z=9000.)

        @classmethod
        def photons_per_joule(cls, t, band):
            x = cls.centre[band] / 1e4
            return 1e18 * x * np.exp(-1.4388 / (x * (t / 1e4))) / (1.0 + x)
    rec = load_golden("bands_253.npz")
    H, W = 72, 80
    imgs0 = frame_images(cel, rec, H, W)
    pix = np.array([[20.5, 30.2], [55.1, 44.8], [38.0, 12.3]])
    truth = [(5200.0, 2.0e-9), (7800.0, 6.0e-10), (3900.0, 9.0e-9)]
    srcs = [cel.SrcParams(u=imgs0[2].pixel2equa(p), a=0, t=t, b=b) for p, (t, b) in zip(pix, truth)]
    celeste.photons_expected_brightness = celeste_em._expected_brightness(Planck)
    try:
        lam = np.stack([celeste.gen_model_image(srcs, im) for im in imgs0])
    finally:
        celeste.photons_expected_brightness = None
    assert lam.max() > 5 * lam.min()
    nelec = np.random.RandomState(8).poisson(lam).astype(np.float64)
    imgs = frame_images(cel, rec, H, W, nelec=nelec)
    for s in srcs:                                     # start away from the truth
        s.t, s.b = 6000.0, s.b * 1.6
    trace, converged = celeste_em.celeste_em(srcs, imgs, maxiter=12, verbose=False, planck=Planck)
    assert len(trace) >= 3 and all(b >= a - 1e-6 * abs(a) for a, b in zip(trace, trace[1:]))
    assert trace[-1] > trace[0]
    for s, (t, b) in zip(srcs, truth):
        counts = sum(celeste_em._expected_brightness(Planck)(s.t, s.b, band) for band in BANDS)
        want = sum(celeste_em._expected_brightness(Planck)(t, b, band) for band in BANDS)
        assert abs(counts / want - 1.0) < 0.05          # total photons recovered to a few Poisson sigmas
    assert celeste.photons_expected_brightness is None  # the hook is restored
    with pytest.raises(NotImplementedError):
        celeste_em.celeste_em(srcs, imgs, maxiter=1, verbose=False)


def test_device_and_host_engines_agree_on_a_crowded_field(cel):
    """500 mixed sources on 5 x 512^2 (thousands of one-wave jobs of very unequal length per round, retired
    slots, the heaviest-first job order): the device state machine and the numpy engine leave every
    source at the same place, bit for bit, after two full sweeps"""
    from desi_mcmc_amd import celeste_mcmc, synth
    ctx = cel.default_context(0)
    out = {}
    for eng in ("host", "device"):
        f = synth.SyntheticField(ctx, 500, 5, 512, 512, frac_gal=0.5, seed=9)
        gf = celeste_mcmc.GibbsField(f.images, list(range(5)), f.bands[:, 2], f.bands[:, 1], 512 * 512)
        g = celeste_mcmc.ModelGibbs([gf], f.src["type"], f.src["radec"], f.flux5(), f.src["shape"], seed=4,
                                    slice_args=dict(step_out=False, sigma=1e-3), engine=eng)
        for _ in range(2):
            g.sweep()
        out[eng] = g
    h, d = out["host"], out["device"]
    assert h.timing["rounds"] == d.timing["rounds"] and h.timing["evals"] == d.timing["evals"]
    assert np.array_equal(h.u, d.u) and np.array_equal(h.fluxes, d.fluxes)
    moved = np.abs(h.u - f.src["radec"]).max(axis=1)
    assert np.all(moved[h.active] > 0) and np.all(moved < 5e-4)          # every sampled source moved, none ran away


def test_slice_locations_error_paths(cel):
    from desi_mcmc_amd import synth
    ctx = cel.default_context(0)
    f = synth.SyntheticField(ctx, 20, 5, 128, 128, frac_gal=0.5, seed=2)
    with pytest.raises(ValueError, match="resident photon split"):
        f.images.slice_locations(f.sources, 1e-3, seed=1)                 # no split yet
    f.images.photon_split_resident(f.sources, seed=3)
    with pytest.raises(ValueError):
        f.images.slice_locations(f.sources, 0.0, seed=1)                  # sigma must be positive
    other = cel.SourceSet(ctx, 7, 5).set(f.src["type"][:7], f.src["radec"][:7], f.src["counts"][:7], f.src["shape"][:7])
    with pytest.raises(ValueError, match="resident photon split"):
        f.images.slice_locations(other, 1e-3, seed=1)                     # not the split's sources
    with pytest.raises(ValueError, match="rounds"):
        f.images.slice_locations(f.sources, 1e-3, seed=1, max_rounds=1)
    radec, llh, st = f.images.slice_locations(f.sources, 1e-3, seed=1)
    assert st["rounds"] >= 4 and st["evals"] >= 4 * 20 and np.all(np.isfinite(llh))


@pytest.mark.parametrize("S,H,W", [(60, 192, 224), (700, 512, 640)])
def test_fused_slice_rounds_are_the_three_launch_rounds_bit_for_bit(cel, S, H, W):
    """CEL_OPT_SLICE_FUSE (round 6; an option, off by default: measured without gain): a round of cel_slice_locations as ONE launch -- the likelihood block that finishes a chain's
    last job of the round steps the chain -- against the round of three launches (likelihoods, k_slice_step): every chain ends at
    the same place with the same log-likelihood, to the last bit, after the same number of rounds and evaluations; repeated runs
    of the fused form agree with themselves (whichever block draws a chain's last ticket); chains of another rank (negative ids)
    stay put; a call with densely scored patches (CEL_OPT_PHOTON_LISTS = 2) is not fused and still agrees.  The larger field
    has rounds of full lists, of live lists and of live lists with every job dealt."""
    from desi_mcmc_amd import synth, _lib
    ctx = cel.default_context(0)
    f = synth.SyntheticField(ctx, S, 5, H, W, frac_gal=0.5, seed=6)
    u0 = f.src["radec"].copy()

    def run(fuse, ids=None, lists=0):
        ctx.set_option(_lib.CEL_OPT_SLICE_FUSE, fuse)
        ctx.set_option(_lib.CEL_OPT_PHOTON_LISTS, lists)
        try:
            f.sources.set(f.src["type"], u0, f.src["counts"], f.src["shape"])
            f.images.render(f.sources, loglik=True)
            f.images.photon_split_resident(f.sources, seed=11)
            return f.images.slice_locations(f.sources, 1e-3, seed=5, chain_ids=ids)
        finally:
            ctx.set_option(_lib.CEL_OPT_SLICE_FUSE, default)
            ctx.set_option(_lib.CEL_OPT_PHOTON_LISTS, 0)
    default = ctx.get_option(_lib.CEL_OPT_SLICE_FUSE)      # N: rounds of at most N likelihood blocks are fused (1: every round; 0, the default: none)
    assert default == 0
    u_ref, l_ref, st_ref = run(0)
    for fuse in (1, 1, 1, 4096, 64):                      # every round fused (three times); fused from a middle / a late round on
        u, l, st = run(fuse)
        assert np.array_equal(u, u_ref) and np.array_equal(l, l_ref, equal_nan=True)
        assert st["rounds"] == st_ref["rounds"] and st["evals"] == st_ref["evals"] and st["evals"] >= 4 * S
    assert not np.array_equal(u_ref, u0)
    ids = np.where(np.arange(S) % 3 == 1, -1, np.arange(S)).astype(np.int32)           # a third of the chains are another rank's
    u_a, l_a, st_a = run(0, ids)
    u_b, l_b, st_b = run(1, ids)
    assert np.array_equal(u_a, u_b) and np.array_equal(l_a, l_b, equal_nan=True) and st_a["evals"] == st_b["evals"]
    assert np.array_equal(u_b[ids < 0], u0[ids < 0]) and np.array_equal(u_b[ids >= 0], u_ref[ids >= 0])
    u_d, l_d, st_d = run(1, None, 2)                      # every patch densely: the fused form does not apply, the call still runs
    assert st_d["rounds"] == st_ref["rounds"]
    np.testing.assert_allclose(l_d, l_ref, rtol=1e-9)


def test_conditional_loglik_does_not_depend_on_how_its_jobs_are_dealt(cel):
    """a proposal's value is bit for bit the same in a small call (every (proposal, band) job dealt to four
    blocks by chunk) and inside a call of 2 000 proposals (one block per job): the kernel sums a job's chunks in
    four classes either way and the parts are added in one order.  The host and the device slice engines split
    at different moments and must stay on the same trajectory."""
    from desi_mcmc_amd import celeste_mcmc
    imgs, params, pix, flux, nelec = small_scene(cel)
    g = celeste_mcmc.ModelGibbs.from_images([dict(zip(BANDS, imgs))], params, seed=9)
    g.resample_photons()
    rs = np.random.RandomState(1)
    idx = rs.randint(0, 4, size=2000)
    U = g.u[idx] + rs.normal(0, 4e-5, size=(2000, 2))
    big = g.location_loglik(idx, U)                     # 10 000 jobs: not split
    for lo in (0, 7, 1500):
        small = g.location_loglik(idx[lo:lo + 6], U[lo:lo + 6])     # 30 jobs: split
        assert np.array_equal(small, big[lo:lo + 6])


def test_config5_full_size_sweeps(cel, orc):
    """BASELINE configs[4] at its full size -- 10 000 mixed sources x 5 bands x 2048^2, the field of
    `bench.py --workload gibbs10k` -- through ModelGibbs.sweep (CelesteBase.resample_model, models.py:75-83;
    Source.resample*, sources.py:242-349; slicesample, util/infer/slicesample.py:89-227): two sweeps with the
    0.001-degree interval the reference's call intends and one with the literal 1 degree it effectively runs.
      * every photon of every band goes to exactly one source or to the sky, in every sweep;
      * the device state machine (cel_slice_locations) and the numpy engine (pinned to the reference's own
        run, tests/test_slicesample.py) leave all 10 000 chains at the same place and flux, bit for bit;
      * with the 0.001-degree interval every sampled source moves, none by as much as the interval's width
        and 99.9 % by less than 5e-4 degrees;
      * cel_stamp_mass (the rate term of resample_fluxes) against the oracle on a 200-source sample."""
    from desi_mcmc_amd import celeste_mcmc, synth
    ctx = cel.default_context(0)
    f = synth.SyntheticField.from_config(ctx, "mixed10k_2048")
    S, B = f.S, f.B
    nel = f.nelec.reshape(B, -1).sum(axis=1)
    eps0 = f.bands[:, 0].copy()
    out = {}
    for eng in ("host", "device"):
        for b in range(B):
            f.images.set_epsilon(b, eps0[b])                 # a sweep redraws the sky levels: same start for both engines
        gf = celeste_mcmc.GibbsField(f.images, list(range(B)), f.bands[:, 2], f.bands[:, 1], f.H * f.W)
        g = celeste_mcmc.ModelGibbs([gf], f.src["type"], f.src["radec"], f.flux5(), f.src["shape"], seed=12,
                                    slice_args=dict(step_out=False, sigma=1e-3), engine=eng)
        trace = []
        for sweep in range(3):
            if sweep == 2:
                g.slice_args = g.slice_preset("literal")     # the call as written: sigma stays 1.0 degree (Q13)
            noise = g.resample_photons()[0]
            assert np.array_equal(gf.sums.sum(axis=0) + noise, nel), (eng, sweep)
            assert gf.sums.shape == (S, B) and np.all(gf.sums >= 0) and np.all(gf.sums == np.rint(gf.sums))
            g.resample_fluxes()
            before = g.u.copy()
            g.resample_locations()
            g.sweeps += 1
            trace.append((g.u.copy(), g.fluxes.copy(), g.active.copy(), before))
        out[eng] = (g, trace)
    (gh, th), (gd, td) = out["host"], out["device"]
    for sweep in range(3):
        assert np.array_equal(th[sweep][0], td[sweep][0]), "positions differ after sweep %d" % sweep
        assert np.array_equal(th[sweep][1], td[sweep][1]), "fluxes differ after sweep %d" % sweep
        assert np.array_equal(th[sweep][2], td[sweep][2])
    assert gh.timing["rounds"] == gd.timing["rounds"] and gh.timing["evals"] == gd.timing["evals"]
    assert gd.timing["evals"] >= 3 * 4 * S
    for sweep in range(2):
        u, _, active, before = td[sweep]
        step = np.abs(u - before).max(axis=1)
        assert active.sum() >= 0.99 * S
        assert np.all(step[active] > 0), "a sampled source did not move"
        assert np.all(step[~active] == 0)
        # without stepping out a coordinate moves by less than the interval's width (1e-3 deg) -- a hard bound;
        # all but a few faint, flat-likelihood sources stay within half of it
        assert step.max() < 1e-3
        assert np.mean(step < 5e-4) > 0.999 and np.median(step[active]) < 2e-5
    assert np.abs(td[1][0] - f.src["radec"]).max() < 2e-3
    # the rate term of the flux conditional at the chain's current state, against the oracle
    sset = gd._sources(gd.fields[0])
    mass = f.images.stamp_mass(sset)
    pick = np.random.RandomState(3).choice(S, 200, replace=False)
    for b in range(B):
        band = f.bands[b].copy()
        band[36] = orc.checked_radius(band, f.images.band(b)[36])                      # R as the library computed it (pinned by test_fitsimage_radius_matches_reference)
        for s in pick:
            p, _, _ = orc.source_patch(band, f.H, f.W, gd.typ[s], gd.u[s], gd.shape[s])
            want = 0.0 if p is None else p.sum()
            np.testing.assert_allclose(mass[s, b], want, rtol=1e-10, atol=1e-300)
    # the location step's likelihood at the chain's state against the oracle, on BOTH routes: every patch at its photons
    # (k_patch_ll_nz) and every patch densely (k_patch_ll_hw<0>) -- the same split (same seed: same photons) laid out twice
    from desi_mcmc_amd import _lib
    rs = np.random.RandomState(8)
    idx = np.sort(rs.choice(np.nonzero(gd.active)[0], 200, replace=False))
    U = gd.u[idx] + rs.normal(0, 3e-5, size=(200, 2))
    gfd = gd.fields[0]
    gfd._counts = None
    terms = None
    for route in (1, 2):
        ctx.set_option(_lib.CEL_OPT_PHOTON_LISTS, route)
        try:
            f.images.photon_split_resident(sset, seed=77)
            got = gd.location_loglik(idx, U)
        finally:
            ctx.set_option(_lib.CEL_OPT_PHOTON_LISTS, 0)
        if terms is None:
            boxes, offs, data = f.images.fetch_samples()
            terms, _ = _oracle_terms(orc, f, gd.typ[idx], U, gd.counts(gfd, idx=idx), gd.shape[idx], idx, boxes, offs, data)
            del data
        _assert_terms(got, terms, "route %d" % route)


@pytest.mark.parametrize("engine", ["device", "host"])
def test_one_chain_on_two_ranks_is_the_single_rank_chain(cel, tmp_path, engine):
    """ONE Gibbs chain partitioned over the GPUs.  Two fresh child processes (sharing GPU 0,
    gloo for the exchange -- the arithmetic of bench.py --workload gibbs10k --scaling strong, which runs the same
    dist.SourceDeal over RCCL) each run the replicated photon split and update the fluxes and locations of THEIR
    sources only (the host-engine variant: the galaxies' shapes too); one all-gather per sweep.  After every one of
    3 sweeps both ranks hold the state the single-rank chain (run here, in this process) holds: locations, fluxes,
    shapes, sky levels and the field log-likelihood, bit for bit."""
    import socket
    import subprocess
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _dealt_chain_rank import run_chain
    S, size, sweeps = 600, 512, 3
    shapes = engine == "host"                   # the host-engine variant also runs the galaxies' shape step
    one = run_chain(S, size, sweeps, engine, shapes=shapes)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "_dealt_chain_rank.py"),
                                       str(tmp_path / ("rank%d.npz" % r)), str(S), str(size), str(sweeps), engine, "1" if shapes else "0"],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    evals = 0
    for r in range(2):
        got = np.load(str(tmp_path / ("rank%d.npz" % r)))
        for k in ("u", "fluxes", "eps", "ll", "active", "shape"):
            assert np.array_equal(got[k], one[k]), (r, k)
        evals += int(got["evals"])
        assert (int(got["shape_evals"]) > 0) == shapes
    assert evals == int(one["evals"])                      # the two ranks shared the single chain's evaluations ...
    assert 0.3 < int(got["evals"]) / evals < 0.7           # ... about evenly
    assert np.all(np.abs(one["u"][-1] - one["u"][0]).max(axis=1)[one["active"]] > 0)


def test_shape_logprob_vs_oracle_and_prior(cel, orc):
    """ModelGibbs.shape_logprob = skew_likelihood (celeste_mcmc.py:209-222): log-prior + the conditional likelihood
    as a function of (theta, sigma, phi, rho), against the oracle on the split's photons fetched to the host;
    a shape outside the prior's support scores -inf and is never rendered"""
    from desi_mcmc_amd import celeste_mcmc
    from desi_mcmc_amd.celeste_galaxy_conditionals import galaxy_shape_prior_constrained as prior
    imgs, params, pix, flux, nelec = small_scene(cel)
    g = celeste_mcmc.ModelGibbs.from_images([dict(zip(BANDS, imgs))], params, seed=5)
    g.resample_photons()
    f = g.fields[0]
    boxes, offs, data = f.iset.fetch_samples()
    rs = np.random.RandomState(2)
    idx = np.array([2, 3, 2, 3, 2, 3, 2])
    TH = g.shape[idx] * rs.uniform(0.7, 1.4, size=(7, 4))
    TH[:, 0] = np.clip(TH[:, 0], 0.05, 0.95)
    TH[:, 3] = np.clip(TH[:, 3], 0.1, 0.95)
    TH[4] = [1.2, 1.0, 30.0, 0.5]                      # theta outside (0, 1)
    TH[5] = [0.5, -0.3, 30.0, 0.5]                     # negative radius
    TH[6] = [0.5, 1.0, 30.0, 1.0]                      # rho on the boundary
    got = g.shape_logprob(idx, TH)
    assert np.all(got[4:] == -np.inf) and np.all(np.isfinite(got[:4]))
    bands = orc.pack_bands(load_golden("bands_253.npz"))
    for i in range(4):
        s = idx[i]
        want = prior(*TH[i])
        for b in range(5):
            band = bands[b].copy()
            band[24:26] = [112 / 2.0, 96 / 2.0]
            k = s * 5 + b
            want += orc.patch_loglik(band, 96, 112, 1, g.u[s], TH[i], g.counts(f)[s, b], boxes[s, b], data[offs[k]:offs[k + 1]], mode=0)
        np.testing.assert_allclose(got[i], want, rtol=1e-11)


def _shape_fisher(cel, imgs, params, s):
    """Cramer-Rao matrix of galaxy s's (sigma, rho) over all images, everything else fixed"""
    from desi_mcmc_amd import models
    m = models.Celeste()
    m.initialize_sources(init_src_params=params)
    info = np.zeros((2, 2))
    p = params[s]
    base = (p.sigma, p.rho)
    for im in imgs:
        lam = m.render_model_image(im)
        d = []
        for name, h in (("sigma", 1e-4), ("rho", 1e-4)):
            v0 = getattr(p, name)
            setattr(p, name, v0 + h)
            lp = m.render_model_image(im)
            setattr(p, name, v0 - h)
            lm = m.render_model_image(im)
            setattr(p, name, v0)
            d.append((lp - lm) / (2 * h))
        info += np.array([[np.sum(d[i] * d[j] / lam) for j in range(2)] for i in range(2)])
    p.sigma, p.rho = base
    return info


def test_short_chain_recovers_a_bright_galaxys_shape(cel):
    """sweeps with the shape step (slice_sample_skew, celeste_mcmc.py:224-243) on a scene whose brightest galaxy is
    made 10x brighter: its (sigma, rho) posterior sits within a few Cramer-Rao errors of the truth with a spread of
    that order; theta and phi stay in their supports; the stars' shape rows are never touched"""
    from desi_mcmc_amd import celeste_mcmc
    imgs, params, pix, flux, nelec = small_scene(cel)
    params[2].fluxes = np.asarray(params[2].fluxes) * 10.0
    from desi_mcmc_amd import models
    from test_hip_parity import frame_images
    m = models.Celeste()
    m.initialize_sources(init_src_params=params)
    rec = load_golden("bands_253.npz")
    imgs0 = frame_images(cel, rec, 96, 112)
    lam = np.stack([m.render_model_image(im) for im in imgs0])
    imgs = frame_images(cel, rec, 96, 112, nelec=np.random.RandomState(13).poisson(lam).astype(np.float64))
    cov = np.linalg.inv(_shape_fisher(cel, imgs, params, 2))
    err = np.sqrt(np.diag(cov))
    g = celeste_mcmc.ModelGibbs.from_images([dict(zip(BANDS, imgs))], params, seed=6)
    truth = g.shape[2].copy()
    keep = []
    for it in range(60):
        g.sweep(shapes=True)
        if it >= 15:
            keep.append(g.shape[2].copy())
    keep = np.array(keep)
    assert g.timing["shape_evals"] > 60 * 2 * 4 and g.timing["shape_rounds"] > 0
    assert np.all(g.shape[:2] == 0.0)                                   # stars
    assert np.all((keep[:, 0] > 0) & (keep[:, 0] < 1) & (keep[:, 2] >= 0) & (keep[:, 2] < 180))
    mean, sd = keep[:, [1, 3]].mean(axis=0), keep[:, [1, 3]].std(axis=0)
    assert np.all(np.abs(mean - truth[[1, 3]]) < 5 * err + 0.02), (mean, truth, err)
    assert np.all(sd < 6 * err + 0.02) and np.all(sd > 0.15 * err), (sd, err)
    assert abs(((keep[:, 2].mean() - truth[2] + 90) % 180) - 90) < 15   # the position angle, degrees


@pytest.mark.parametrize("shape_args", [None, dict(compwise=True), dict(step_out=False, sigma=0.05, numdir=3, compwise=False)])
def test_shape_step_device_engine_follows_the_host_engine(cel, shape_args):
    """the general device slice sampler (cel_slice_sample: random directions from the chain's normal stream, stepping out by
    doubling, the `acceptable` test, the built-in shape prior) and the numpy engine leave every galaxy with the same
    (theta, sigma, phi, rho), bit for bit, after two sweeps with the shape step -- 300 mixed sources on 5 x 384^2 -- for
    slice_sample_skew's own call (celeste_mcmc.py:229-239), its component-wise form and a no-step-out form; the stars'
    rows are untouched; a custom log-prior sends the step to the host engine"""
    from desi_mcmc_amd import celeste_mcmc, synth
    ctx = cel.default_context(0)
    out = {}
    for eng in ("host", "device"):
        f = synth.SyntheticField(ctx, 300, 5, 384, 384, frac_gal=0.5, seed=11)
        gf = celeste_mcmc.GibbsField(f.images, list(range(5)), f.bands[:, 2], f.bands[:, 1], 384 * 384)
        g = celeste_mcmc.ModelGibbs([gf], f.src["type"], f.src["radec"], f.flux5(), f.src["shape"], seed=8, engine=eng,
                                    shape_args=shape_args)
        assert g._shape_engine_on_device() == (eng == "device")
        for _ in range(2):
            g.sweep(shapes=True)
        out[eng] = g
    h, d = out["host"], out["device"]
    assert np.array_equal(h.u, d.u) and np.array_equal(h.fluxes, d.fluxes)
    assert np.array_equal(h.shape, d.shape), np.abs(h.shape - d.shape).max()
    assert h.timing["shape_rounds"] == d.timing["shape_rounds"] and h.timing["shape_evals"] == d.timing["shape_evals"]
    gal = (f.src["type"] == 1) & h.active
    assert np.all(np.any(h.shape[gal] != f.src["shape"][gal], axis=1))               # every sampled galaxy moved
    assert np.array_equal(h.shape[f.src["type"] == 0], f.src["shape"][f.src["type"] == 0])
    assert np.all((h.shape[gal, 0] > 0) & (h.shape[gal, 0] < 1) & (h.shape[gal, 1] > 0) & (h.shape[gal, 3] > 0) & (h.shape[gal, 3] < 1))
    assert np.all((h.shape[gal, 2] >= 0) & (h.shape[gal, 2] < 180))
    custom = celeste_mcmc.ModelGibbs([gf], f.src["type"], f.src["radec"], f.flux5(), f.src["shape"], seed=8,
                                     shape_logprior=lambda TH: np.where((TH[:, 0] > 0) & (TH[:, 0] < 1) & (TH[:, 1] > 0) & (TH[:, 3] > 0) & (TH[:, 3] < 1), 0.0, -np.inf))
    assert not custom._shape_engine_on_device()


@pytest.mark.parametrize("S,size,sweeps", [(600, 512, 3), (10000, 2048, 2)], ids=["600x512", "config5_full_size"])
def test_one_chain_partitioned_by_row_strips(cel, tmp_path, S, size, sweeps):
    """SURVEY 8e, config 5, the spatial partition: two child ranks (sharing GPU 0, gloo) each hold the images on their row
    strip plus a halo, split only those rows' photons, own the sources whose row lies in their strip, count their strip's
    sky photons (the sums are added over the ranks) and add their strip's log-likelihood to the trace (dist.StripDeal).
      * every photon of the frame is accounted for once: the ranks' own sources' photons + the summed sky photons;
      * after the first sweep the own sources' photon sums are the single-rank split's, photon for photon (a pixel's draws
        are keyed by its full-frame index and the source: whoever splits it draws the same), and the locations, fluxes, sky
        levels and the field log-likelihood agree with the single-rank chain to rounding (the window's row origin enters
        the pixel arithmetic: 1e-9 here, not bit for bit);
      * both ranks hold the same merged state after every sweep, bit for bit."""
    import socket
    import subprocess
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _dealt_chain_rank import run_chain
    one = run_chain(S, size, sweeps, "device")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "_dealt_chain_rank.py"),
                                       str(tmp_path / ("strip%d.npz" % r)), str(S), str(size), str(sweeps), "device", "0", "strips"],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    got = [np.load(str(tmp_path / ("strip%d.npz" % r))) for r in range(2)]
    for k in ("u", "fluxes", "eps", "ll", "noise"):
        assert np.array_equal(got[0][k], got[1][k]), k                   # one chain, the same on both ranks
    own_sums = got[0]["sums"] + got[1]["sums"]                            # each rank reports its own sources' rows
    assert np.all((got[0]["sums"] == 0) | (got[1]["sums"] == 0))
    assert np.array_equal(own_sums[-1].sum(axis=0) + got[0]["noise"], got[0]["nelec_sum"])     # conservation over the ranks
    assert np.array_equal(own_sums[0], one["sums"][0])                    # the first split: photon for photon
    np.testing.assert_allclose(got[0]["u"][0], one["u"][0], rtol=1e-9, atol=0)
    np.testing.assert_allclose(got[0]["fluxes"][0], one["fluxes"][0], rtol=1e-9)
    np.testing.assert_allclose(got[0]["eps"][0], one["eps"][0], rtol=1e-12)
    np.testing.assert_allclose(got[0]["ll"][0], one["ll"][0], rtol=1e-10)
    # the field's log-likelihood trace over all sweeps: 1e-12 (the window's row origin enters the arithmetic: rounding)
    np.testing.assert_allclose(got[0]["ll"], one["ll"], rtol=1e-12)
    for k in range(sweeps):                                               # conservation over the ranks in EVERY sweep
        assert own_sums[k].sum() <= got[0]["nelec_sum"].sum()
    assert 0.25 < (got[0]["sums"][0].sum() / own_sums[0].sum()) < 0.75    # the ranks shared the sources about evenly



def test_photon_lists_agree_with_the_dense_form_fuzz(cel):
    """The conditional likelihood read at the photons (k_patch_ll_nz: direct exponentials at the pixels that hold a photon) against
    the dense form (k_patch_ll_hw<0>: the column recurrence over the photon rectangle, CEL_OPT_PHOTON_LISTS = 2) on random
    fields -- faint and very bright sources (lists from a handful to tens of thousands of photons: whole jobs and jobs dealt
    to four blocks), tiny and large galaxies, sources on the frame's edge, proposals a fraction of a pixel to hundreds of
    pixels away (the far shortcut) -- to 1e-12; and a value does not depend on the size of the call it is part of."""
    from desi_mcmc_amd import _lib, synth
    ctx = cel.default_context(0)
    longest, shortest = 0, 1 << 30
    for seed in range(6):
        rs = np.random.RandomState(100 + seed)
        S, H, W = 60, int(rs.choice([160, 256, 384])), int(rs.choice([192, 256, 320]))
        f = synth.SyntheticField(ctx, S, 5, H, W, frac_gal=0.6, seed=200 + seed, with_nelec=False)
        src = f.src
        src["counts"] = src["counts"] * np.exp(rs.uniform(np.log(0.02), np.log(60.0), size=(S, 1)))     # 20 ... 5e6 photons
        src["shape"][:, 1] = np.exp(rs.uniform(np.log(0.05), np.log(6.0), S))                            # r_e 0.05" ... 6"
        edge = rs.rand(S) < 0.2
        src["radec"][edge] = synth.pixel2equa(f.bands[0], np.column_stack([rs.choice([-3.0, 1.5, W - 2.0, W + 2.5], edge.sum()),
                                                                           rs.uniform(0, H, edge.sum())]))
        if seed == 5:
            # the older per-profile route (type 2: shape = theta, W00, W01, W11): a positive definite W takes the kernel's
            # rotated form like any galaxy, a rank-1 W (no Cholesky factor) its general form
            gal = np.nonzero(src["type"] == 1)[0][:6]
            src["type"][gal] = 2
            src["shape"][gal[:3], 1:] = [[9.0, 2.0, 4.0], [2.5, -1.0, 6.0], [30.0, 12.0, 8.0]]
            src["shape"][gal[3:], 1:] = [[9.0, 6.0, 4.0], [1.0, 1.0, 1.0], [16.0, -8.0, 4.0]]
        f.sources.set(src["type"], src["radec"], src["counts"], src["shape"])
        f.images.render(f.sources)
        f.images.set_nelec(rs.poisson(f.images.model_images()).astype(np.float64))
        P = 5
        own = np.repeat(np.arange(S, dtype=np.int32), P)
        jit = rs.normal(0, 1.0, size=(S * P, 2)) * np.repeat(rs.choice([3e-6, 3e-5, 4e-4, 2e-2], S), P)[:, None]
        prop = cel.SourceSet(ctx, S * P, 5).set(np.repeat(src["type"], P), np.repeat(src["radec"], P, axis=0) + jit,
                                                np.repeat(src["counts"], P, axis=0), np.repeat(src["shape"], P, axis=0))
        out = {}
        for mode in (2, 1, 0):
            ctx.set_option(_lib.CEL_OPT_PHOTON_LISTS, mode)
            try:
                f.images.photon_split_resident(f.sources, seed=seed)
                out[mode] = f.images.patch_loglik_resident(prop, own)
                if mode == 1:
                    few = cel.SourceSet(ctx, 7, 5).set(np.repeat(src["type"], P)[11:18], (np.repeat(src["radec"], P, axis=0) + jit)[11:18],
                                                       np.repeat(src["counts"], P, axis=0)[11:18], np.repeat(src["shape"], P, axis=0)[11:18])
                    assert np.array_equal(f.images.patch_loglik_resident(few, own[11:18]), out[1][11:18])
            finally:
                ctx.set_option(_lib.CEL_OPT_PHOTON_LISTS, 0)
        assert np.all(np.isfinite(out[2]))
        # 1e-12 of the value -- or 1e-9 absolute where the photon term and the mass term nearly cancel (a value of -1 from terms
        # of 1e3: the photon kernel's exponential is a cubic on a 256-entry table, good to 1.4e-13 of a pixel's value)
        np.testing.assert_allclose(out[1], out[2], rtol=1e-12, atol=1e-9)
        np.testing.assert_allclose(out[0], out[2], rtol=1e-12, atol=1e-9)
        boxes, offs, data = f.images.fetch_samples()
        nnz = np.array([np.count_nonzero(data[offs[i]:offs[i + 1]]) for i in range(S * 5)])
        longest = max(longest, int(nnz.max()))
        shortest = min(shortest, int(nnz[nnz > 0].min()))
    assert longest > 3 * 2048 and shortest < 64                   # lists dealt to four blocks, and lists shorter than one step


def _oracle_terms(orc, f, typ, U, counts, shape, own, boxes, offs, data, threads=8):
    """orc_patch_loglik_terms of proposal p on its owner's photon patches, summed over the bands -> (P, 3), (P,) photons"""
    from concurrent.futures import ThreadPoolExecutor
    B = f.B
    bands = [f.bands[b].copy() for b in range(B)]
    for b in range(B):
        bands[b][36] = orc.checked_radius(bands[b], f.images.band(b)[36])              # R as the library computed it (pinned to the reference's elsewhere)

    def one(p):
        t, nph = np.zeros(4), 0.0
        o = int(own[p])
        for b in range(B):
            bx = boxes[o, b]
            z = data[offs[o * B + b]:offs[o * B + b + 1]]
            t += orc.patch_loglik_terms(bands[b], f.H, f.W, typ[p], U[p], shape[p], counts[p, b], bx, z)
            nph += z.sum()
        return t, nph
    with ThreadPoolExecutor(threads) as ex:
        res = list(ex.map(one, range(len(own))))
    return np.array([r[0] for r in res]), np.array([r[1] for r in res])


def _assert_terms(got, terms, what=""):
    """got = photon term - mass term of the HIP path against the oracle's terms (P, 4) = (photon, |photon|, mass, quantum): the
    photon term to 1e-11 of the scale its rounding is relative to (sum |z log m|; the mass term, counts * sum w, is formed
    alike on both sides), and the value itself to rtol = 1e-11 -- no atol -- wherever the two terms do not cancel.
    `quantum`: where the unit stamp is a SUBNORMAL number (a proposal hundreds of pixels from its photons, exponents between
    -708 and -750) every evaluator -- the reference's two included -- holds it to a few of the range's quanta only; the
    oracle prices that (orc_patch_loglik_terms: 64 quanta through log(), the whole term within 64 quanta of zero)."""
    pt, apt, mass, quantum = terms[:, 0], terms[:, 1], terms[:, 2], terms[:, 3]
    err = np.abs((got + mass) - pt)
    bound = 1e-11 * apt + 8 * np.finfo(float).eps * mass + quantum
    bad = np.nonzero(err > bound)[0]
    assert bad.size == 0, "%s photon term off at %s: err %s, bound %s" % (what, bad[:5], err[bad[:5]], bound[bad[:5]])
    want = pt - mass
    clean = (np.abs(want) >= 0.01 * (apt + mass)) & (quantum == 0)
    assert clean.sum() >= 0.7 * clean.size
    np.testing.assert_allclose(got[clean], want[clean], rtol=1e-11, atol=0)


def test_photon_list_route_vs_oracle(cel, orc):
    """The conditional likelihood read AT THE PHOTONS (k_patch_ll_nz, CEL_OPT_PHOTON_LISTS = 1: every patch takes that route)
    directly against the oracle's Source.log_likelihood (sources.py:134-183) on the split's photon patches fetched to the
    host -- stars (general form), galaxies (rotated basis, cubic table exponential), per-profile sources with a positive
    definite and with a rank-1 W (rotated / general form), sources on the frame's edge and off it, proposals from a fraction
    of a pixel to hundreds of pixels away (the far shortcut: -counts * sum w), photon lists from under 64 to over 8 192
    entries (whole jobs and jobs dealt to four blocks, in calls of 7 and of hundreds of proposals)."""
    from desi_mcmc_amd import _lib, synth
    ctx = cel.default_context(0)
    from conftest import fuzz_seeds
    shortest, longest, far = 1 << 30, 0, 0
    for seed in fuzz_seeds(3):
        rs = np.random.RandomState(300 + seed)
        S, H, W = 48, int(rs.choice([192, 256])), int(rs.choice([224, 320]))
        f = synth.SyntheticField(ctx, S, 5, H, W, frac_gal=0.6, seed=400 + seed, with_nelec=False)
        src = f.src
        src["counts"] = src["counts"] * np.exp(rs.uniform(np.log(0.02), np.log(60.0), size=(S, 1)))     # 20 ... 5e6 photons
        src["shape"][:, 1] = np.exp(rs.uniform(np.log(0.05), np.log(6.0), S))                            # r_e 0.05" ... 6"
        edge = rs.rand(S) < 0.2
        src["radec"][edge] = synth.pixel2equa(f.bands[0], np.column_stack([rs.choice([-3.0, 1.5, W - 2.0, W + 2.5], edge.sum()),
                                                                           rs.uniform(0, H, edge.sum())]))
        gal = np.nonzero(src["type"] == 1)[0][:6]
        if seed >= 1:           # type 2: shape = theta, W00, W01, W11 -- three positive definite, three rank-1 (no Cholesky factor)
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
        ctx.set_option(_lib.CEL_OPT_PHOTON_LISTS, 1)
        try:
            f.images.photon_split_resident(f.sources, seed=seed)
            got = f.images.patch_loglik_resident(prop, own)
            few = cel.SourceSet(ctx, 7, 5).set(typ[11:18], U[11:18], cts[11:18], shp[11:18])       # 35 jobs: every job dealt
            assert np.array_equal(f.images.patch_loglik_resident(few, own[11:18]), got[11:18])
        finally:
            ctx.set_option(_lib.CEL_OPT_PHOTON_LISTS, 0)
        boxes, offs, data = f.images.fetch_samples()
        for i in range(S * 5):
            n = int(np.count_nonzero(data[offs[i]:offs[i + 1]]))
            if offs[i + 1] > offs[i]:
                shortest, longest = min(shortest, n), max(longest, n)
        terms, nph = _oracle_terms(orc, f, typ, U, cts, shp, own, boxes, offs, data)
        assert np.all(np.isfinite(got))
        _assert_terms(got, terms, "seed %d" % seed)
        far += int(np.sum((terms[:, 1] == 0) & (nph > 0)))       # the model is exactly 0 on every photon: only the mass term
    assert shortest < 64 and longest > 8192, (shortest, longest)
    assert far >= 3


def test_slice_sample_and_planes_error_paths(cel):
    """bad arguments of the round-3 entry points come back as ValueError (CEL_ERR_INVALID), never as a crash"""
    from desi_mcmc_amd import synth
    ctx = cel.default_context(0)
    f = synth.SyntheticField(ctx, 20, 5, 128, 128, frac_gal=0.5, seed=2)
    dirs = np.tile(np.eye(4)[None, :2, :], (20, 1, 1))
    with pytest.raises(ValueError, match="resident photon split"):
        f.images.slice_sample(f.sources, 1, 1.0, seed=1, dirs=dirs)                  # no split yet
    f.images.photon_split_resident(f.sources, seed=3)
    with pytest.raises(ValueError):
        f.images.slice_sample(f.sources, 1, 0.0, seed=1, dirs=dirs)                  # sigma must be positive
    with pytest.raises(ValueError, match="dirs"):
        f.images.slice_sample(f.sources, 1, 1.0, seed=1, dirs=dirs[:, :, :2])        # shape directions are 4-vectors
    with pytest.raises(ValueError, match="rounds"):
        f.images.slice_sample(f.sources, 1, 1.0, seed=1, dirs=dirs, max_rounds=1)
    other = cel.SourceSet(ctx, 7, 5).set(f.src["type"][:7], f.src["radec"][:7], f.src["counts"][:7], f.src["shape"][:7])
    with pytest.raises(ValueError, match="resident photon split"):
        f.images.slice_sample(other, 0, 1e-3, seed=1)                                # not the split's sources
    x, llh, st = f.images.slice_sample(f.sources, 0, 1e-3, seed=1, step_out=False)   # component-wise locations through the general engine
    assert x.shape == (20, 2) and st["evals"] >= 4 * 20 and np.all(np.abs(x - f.src["radec"]) < 1e-3)
    th, llh, st = f.images.slice_sample(f.sources, 1, 1.0, seed=1, dirs=dirs)
    gal = f.src["type"] == 1
    assert np.array_equal(th[~gal], f.src["shape"][~gal]) and np.all(np.isfinite(llh[gal])) and np.all(np.isnan(llh[~gal]))
    # mode 4: two planes per band
    one = cel.SourceSet(ctx, 1, 5).set(f.src["type"][:1], f.src["radec"][:1], f.src["counts"][:1], f.src["shape"][:1])
    boxes = np.zeros((5, 4), dtype=np.int32)
    boxes[2] = [10, 30, 20, 50]
    with pytest.raises(ValueError, match="planes"):
        f.images.patch_loglik_planes(one, boxes, [None, None, np.zeros((20, 30)), None, None])
    ll = f.images.patch_loglik_planes(one, boxes, [None, None, np.stack([np.full((20, 30), 3.0), np.full((20, 30), 400.0)]), None, None])
    assert np.isfinite(ll[0])


def test_the_two_device_samplers_in_either_order(cel):
    """cel_slice_sample, then cel_slice_locations, then cel_slice_sample again on ONE image set: each sampler owns its state
    (round 3 freed the general sampler's state inside the first cel_slice_locations call: a use-after-free on the next
    cel_slice_sample and a double free at destroy).  The second shape call must reproduce a fresh image set's."""
    from desi_mcmc_amd import synth
    ctx = cel.default_context(0)
    dirs = np.tile(np.eye(4)[None, :2, :], (60, 1, 1))

    def run(order):
        f = synth.SyntheticField(ctx, 60, 5, 192, 192, frac_gal=0.5, seed=21)
        f.images.photon_split_resident(f.sources, seed=5)
        out = []
        for what in order:
            if what == "shape":
                th, llh, _ = f.images.slice_sample(f.sources, 1, 1.0, seed=9, dirs=dirs)
                out.append(th)
            else:
                u, _, _ = f.images.slice_locations(f.sources, 1e-3, seed=4)
                out.append(u)
            f.sources.set(f.src["type"], f.src["radec"], f.src["counts"], f.src["shape"])     # back to the start (both samplers update the catalogue in place)
        del f
        return out
    a = run(["shape", "loc", "shape", "loc"])
    b = run(["loc", "shape"])
    assert np.array_equal(a[0], a[2]) and np.array_equal(a[0], b[1])
    assert np.array_equal(a[1], a[3]) and np.array_equal(a[1], b[0])
    # a catalogue that outgrows the location sampler's proposal set after the general sampler ran
    f = synth.SyntheticField(ctx, 40, 5, 160, 160, frac_gal=0.5, seed=22)
    f.images.photon_split_resident(f.sources, seed=5)
    f.images.slice_sample(f.sources, 0, 1e-3, seed=1, step_out=False)
    f.images.slice_locations(f.sources, 1e-3, seed=4)
    big = synth.SyntheticField(ctx, 400, 5, 160, 160, frac_gal=0.5, seed=23)
    f.images.photon_split_resident(big.sources, seed=5)
    f.images.slice_sample(big.sources, 0, 1e-3, seed=1, step_out=False)
    f.images.slice_locations(big.sources, 1e-3, seed=4)
    x, _, st = f.images.slice_sample(big.sources, 0, 1e-3, seed=1, step_out=False)
    assert x.shape == (400, 2) and st["evals"] >= 4 * 400



def test_stamp_masses_read_off_the_split(cel, orc):
    """CEL_OPT_SPLIT_REUSE = 2: the split that follows a trace render (k_strict_totals + k_photon_split_hw) adds up every unit
    stamp it evaluates -- the boxes' first rows and columns in the one kernel, everything strictly inside in the other -- in
    integer units of 2^-60, and a stamp_mass of the same catalogue right after reads the masses (sources.py:338-339: the unit
    stamp summed over the source's box) off those sums.  Against the mass kernel: 1e-11 eps / counts (the sums carry the split's
    drop rule); a galaxy fainter than a sixteenth of a sky pixel (a star: 1/1024) or a source without counts in a band is not vouched for and takes the mass kernel
    (the very same numbers); any change of the catalogue sends everything back to the mass kernel; against the oracle."""
    from desi_mcmc_amd import _lib, synth
    ctx = cel.default_context(0)
    f = synth.SyntheticField(ctx, 300, 5, 320, 352, frac_gal=0.5, seed=23)
    counts = f.src["counts"].copy()
    eps = f.bands[:, 0]
    faint = np.arange(0, 300, 7)
    counts[np.arange(3, 300, 11)] = eps[None, :] * 0.1      # faint, and vouched for still
    counts[faint] = eps[None, :] * 5e-4                     # fainter than a star's 1/1024 (a galaxy's 1/16) of a sky pixel: the mass kernel's
    counts[5, 2] = 0.0                                      # no counts in one band: nothing to divide by
    f.sources.set(f.src["type"], f.src["radec"], counts, f.src["shape"])
    try:
        ctx.set_option(_lib.CEL_OPT_SPLIT_REUSE, 1)
        f.images.render(f.sources, loglik=True)
        f.images.photon_split_resident(f.sources, seed=5)
        exact = f.images.stamp_mass(f.sources)
        ctx.set_option(_lib.CEL_OPT_SPLIT_REUSE, 2)
        f.images.render(f.sources, loglik=True)
        f.images.photon_split_resident(f.sources, seed=5)
        quick = f.images.stamp_mass(f.sources)
        assert exact.shape == quick.shape == (300, 5)
        np.testing.assert_allclose(quick, exact, rtol=3e-10, atol=1e-13)      # 1e-11 eps / counts at most, counts >= eps / 16
        bright = np.ones(300, bool); bright[faint] = False; bright[5] = False; bright[np.arange(3, 300, 11)] = False
        np.testing.assert_allclose(quick[bright], exact[bright], rtol=1e-11)
        assert np.any(quick[bright] != exact[bright])        # ... and it WAS the short cut (another order of summation)
        assert np.array_equal(quick[faint], exact[faint])    # not vouched for: the mass kernel's own numbers
        assert quick[5, 2] == exact[5, 2] and exact[5, 2] > 0.5
        assert np.array_equal(f.images.stamp_mass(f.sources), quick)     # the sums stay valid while nothing changes
        # a split that had to render its totals (the model image on the device is of another sky level) sums nothing: the mass kernel
        f.images.set_epsilon(2, f.bands[2, 0] * 1.01)
        f.images.photon_split_resident(f.sources, seed=6)
        assert np.array_equal(f.images.stamp_mass(f.sources), exact)
        # a changed catalogue after a short-cut split: the sums are of another generation
        f.images.render(f.sources, loglik=True)
        f.images.photon_split_resident(f.sources, seed=5)
        f.sources.set(f.src["type"], f.src["radec"], counts, f.src["shape"])
        assert np.array_equal(f.images.stamp_mass(f.sources), exact)
    finally:
        ctx.set_option(_lib.CEL_OPT_SPLIT_REUSE, 2)
    # the oracle: unit stamps summed over their boxes
    for s in (0, 1, 2, 7, 150, 151, 299):
        for b in (0, 3):
            p, yl, xl = orc.source_patch(f.images.band(b), f.H, f.W, f.src["type"][s], f.src["radec"][s], f.src["shape"][s])
            want = 0.0 if p is None else p.sum()
            assert abs(quick[s, b] - want) <= 1e-10 * max(want, 1e-3), (s, b, quick[s, b], want)


@pytest.mark.parametrize("y0,hw", [(37, 150), (64, 128), (91, 165)])
def test_split_draws_do_not_depend_on_the_window(cel, y0, hw):
    """A pixel's draws are keyed by its FULL-FRAME coordinates -- the first test's shared Philox half-word by (column, the row with
    bits 1, 2 and 3 cleared), the sampler's stream by the pixel -- so an image set that holds rows [y0, y0 + hw) of the frame
    (cel_images_set_window: a rank's strip) splits its pixels exactly as the full frame's set does, wherever the window starts
    (odd rows, rows that are no multiple of a tile: the lanes' eight-row groups then straddle the kernel's steps; a star's
    stride-two recurrence starts on another row)."""
    from desi_mcmc_amd import synth
    import desi_mcmc_amd as celmod
    ctx = cel.default_context(0)
    H, W, S, B = 256, 192, 220, 5
    f = synth.SyntheticField(ctx, S, B, H, W, frac_gal=0.5, seed=31)
    f.images.photon_split_resident(f.sources, seed=12)
    bf, of, df = f.images.fetch_samples()
    win = celmod.ImageSet(ctx, f.bands, hw, W, nelec=np.ascontiguousarray(f.nelec[:, y0:y0 + hw]))
    win.set_window(y0, H)
    win.photon_split_resident(f.sources, seed=12)
    bw, ow, dw = win.fetch_samples()
    compared = photons = 0
    for s in range(S):
        for b in range(B):
            fy0, fy1, fx0, fx1 = bf[s, b]
            wy0, wy1, wx0, wx1 = bw[s, b]
            if fy1 <= fy0 or wy1 <= wy0:
                continue
            pf = df[of[s * B + b]:of[s * B + b + 1]].reshape(fy1 - fy0, fx1 - fx0)
            pw = dw[ow[s * B + b]:ow[s * B + b + 1]].reshape(wy1 - wy0, wx1 - wx0)
            assert (wx0, wx1) == (fx0, fx1)
            # the window's box is the frame's cut to the window's rows (window-relative)
            lo, hi = max(fy0, y0), min(fy1, y0 + hw)
            assert (wy0 + y0, wy1 + y0) == (lo, hi)
            # the box's first row takes no photons (celeste_sample_sources.pyx:50-51): where the window cuts a box its first
            # row in the window is not the box's first row in the frame -- compare the rows strictly inside both
            a = pf[lo - fy0 + 1:hi - fy0]
            c = pw[1:]
            assert np.array_equal(a, c), (s, b)
            compared += a.size
            photons += int(a.sum())
    assert compared > 2e5 and photons > 1e4

def test_split_totals_from_the_trace_image(cel):
    """CEL_OPT_SPLIT_REUSE: when the model image of exactly these sources and sky levels is on the device (a chain's trace
    render came last), the photon split forms its totals image -- every pixel's rate under the strict-box rule of
    celeste_sample_sources.pyx:50-51 -- from that image by subtracting each source's first box row and column
    (k_strict_totals) instead of rendering it again.  The two images agree to 1e-9 on every pixel (exactly off the border
    pixels); both splits conserve every photon; a changed sky level, a changed catalogue or the option send the split back
    to the render; a whole sweep of ModelGibbs takes the short way from its second sweep on."""
    from desi_mcmc_amd import _lib, celeste_mcmc, synth
    ctx = cel.default_context(0)
    f = synth.SyntheticField(ctx, 400, 5, 384, 416, frac_gal=0.5, seed=17)
    nel = f.nelec.reshape(5, -1).sum(axis=1)

    def split(expect_short):
        ctx.profile(True)
        noise = f.images.photon_split_resident(f.sources, seed=7)
        n_short, n_render = ctx.profile_get("totals")[1], ctx.profile_get("render")[1]
        ctx.profile(False)
        assert (n_short, n_render) == ((1, 0) if expect_short else (0, 1)), (n_short, n_render)
        sums = f.images.sample_sums()
        assert np.array_equal(sums.sum(axis=0) + noise, nel)
        return sums, noise, f.images.split_rates()
    try:
        f.images.render(f.sources, loglik=True)                 # the trace render: full boxes, model image stored
        s_short, n_short, r_short = split(True)
        ctx.set_option(_lib.CEL_OPT_SPLIT_REUSE, 0)
        f.images.render(f.sources, loglik=True)
        s_full, n_full, r_full = split(False)
        ctx.set_option(_lib.CEL_OPT_SPLIT_REUSE, 2)
        np.testing.assert_allclose(r_short, r_full, rtol=1e-9)
        same = r_short == r_full
        assert same.mean() > 0.5                                 # off the border pixels: the very same numbers
        lam = f.images.model_images()
        # strict boxes: something was taken away (the two renders drop their components on different rectangles: at the
        # shipping threshold a total may exceed the full-box image by what the rule allows, a few e^-24 of the sky)
        assert np.all(r_full <= lam * (1 + 1e-9)) and np.any(r_full < lam * (1 - 1e-6))
        assert np.mean(s_short == s_full) > 0.999 and abs(s_short - s_full).sum() <= 8        # the same draws but for a rare flip
        split(True)                                              # the render of the totals left the model image alone: still valid
    finally:
        ctx.set_option(_lib.CEL_OPT_SPLIT_REUSE, 2)
        ctx.profile(False)
    f.images.render(f.sources, loglik=True)
    f.images.set_epsilon(2, f.bands[2, 0] * 1.01)                # a new sky level: the image on the device is stale
    split(False)
    f.images.render(f.sources, loglik=True)
    f.sources.set(f.src["type"], f.src["radec"], f.src["counts"], f.src["shape"])      # a new catalogue (same numbers, nobody knows)
    split(False)
    f.images.render(f.sources, loglik=False)                     # a render without the log-likelihood stores the image too
    split(True)
    # a chain: from the second sweep on the split follows the trace render of the sweep before
    gf = celeste_mcmc.GibbsField(f.images, list(range(5)), f.bands[:, 2], f.bands[:, 1], f.H * f.W)
    g = celeste_mcmc.ModelGibbs([gf], f.src["type"], f.src["radec"], f.flux5(), f.src["shape"], seed=3)
    ctx.profile(True)
    for k in range(3):
        g.sweep()
        assert np.array_equal(gf.sums.sum(axis=0) + g.noise_sums[0], nel)
        g.log_likelihood()
    assert ctx.profile_get("totals")[1] == 2 and ctx.profile_get("render")[1] == 1 + 3
    ctx.profile(False)


def test_device_gamma_streams_are_the_host_sampler(cel):
    """cel_gamma_streams (k_gamma_streams: the flux conditionals' Gamma draws of Source.resample_fluxes, sources.py:341-345, for a
    whole catalogue on the device) against celeste_mcmc.gamma_by_stream, the numpy form of the same sampler -- the same
    per-element SplitMix64 streams, the same Marsaglia-Tsang decisions: values equal to rounding (cos and log are the
    device's), shapes from 0.05 (the boosted branch) to 1e7, order-free; bad shapes are refused."""
    from desi_mcmc_amd.celeste_mcmc import gamma_by_stream
    ctx = cel.default_context(0)
    rs = np.random.RandomState(0)
    a = np.concatenate([np.exp(rs.uniform(np.log(0.05), np.log(1e7), 60000)), [1.0, 0.999999, 1.0 / 3.0 + 1e-9, 5.0, 1e-3]])
    for seed in (0, 11, 2 ** 63 + 12345):
        dev = ctx.gamma_streams(a, seed)
        host = gamma_by_stream(a, seed, np.arange(a.shape[0]))
        assert np.all(dev[a >= 0.05] > 0) and np.all(np.isfinite(dev)) and np.all(dev >= 0)      # (a = 1e-3: u^1000 may underflow, on the host too)
        close = np.abs(dev - host) <= 1e-12 * np.abs(host)
        assert close.mean() > 0.99999, close.mean()          # (an accept / reject decision within rounding of its boundary may differ)
    assert np.array_equal(ctx.gamma_streams(a, 11), ctx.gamma_streams(a, 11))
    assert np.array_equal(ctx.gamma_streams(a[:100], 11), ctx.gamma_streams(a, 11)[:100])     # element i depends on (seed, i) only
    big = ctx.gamma_streams(np.full(200000, 2.5), 3)
    assert abs(big.mean() - 2.5) < 0.02 and abs(big.var() - 2.5) < 0.06
    with pytest.raises(ValueError):
        ctx.gamma_streams(np.array([1.0, -2.0]), 1)
    with pytest.raises(ValueError):
        ctx.gamma_streams(np.array([1.0, np.nan]), 1)
    assert ctx.gamma_streams(np.zeros(0), 1).shape == (0,)


def test_gibbs_sweeps_on_a_real_sdss_field(cel):
    """BASELINE configs[4] on real data: ModelGibbs over the five SDSS images of data/stamps 253.1147-11.6072 (the reference's
    own FitsImage fields and catalogue, tests/golden/real_fields.npz), the catalogue's sources as stars.  Every observed
    photon of every band goes to exactly one source or to the sky in every sweep; the device and the host engine follow the
    same trajectory bit for bit; the sky levels stay near their catalogued values; the chain's log-likelihood settles above
    the reference's own celeste_likelihood_multi_image value for the raw catalogue."""
    from conftest import real_fields
    from desi_mcmc_amd import celeste, celeste_mcmc
    _, fields = real_fields(dirs=("stamps",))
    f = [q for q in fields if q["name"].endswith("253.1147-11.6072")][0]
    imgs = [cel.FitsImage.from_record(BANDS[b], f["rec"], b, f["nelec"][b]) for b in range(5)]
    params = [cel.SrcParams(u=f["radec"][s].copy(), a=0, fluxes=dict(zip(BANDS, np.maximum(f["flux"][s], 1e-3)))) for s in range(len(f["radec"]))]
    nel = f["nelec"].reshape(5, -1).sum(axis=1)
    out = {}
    for eng in ("host", "device"):
        for im, e in zip(imgs, f["rec"]["eps"]):
            im.epsilon = float(e)
        g = celeste_mcmc.ModelGibbs.from_images([dict(zip(BANDS, imgs))], params, seed=6, engine=eng)
        trace = []
        for sweep in range(4):
            g.sweep()
            gf = g.fields[0]
            assert np.array_equal(gf.sums.sum(axis=0) + g.noise_sums[0], nel), (eng, sweep)
            trace.append((g.u.copy(), g.fluxes.copy(), gf.epsilon.copy(), g.log_likelihood()))
        out[eng] = trace
    for a, b in zip(out["host"], out["device"]):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and a[3] == b[3]
    u, fl, eps, ll = out["device"][-1]
    assert np.all(np.abs(eps / f["rec"]["eps"] - 1.0) < 0.05)                 # sky levels: SKY * GAIN of the headers
    # the chain fits fluxes and sky to the pixels: it ends above the reference's log-likelihood of the raw catalogue (whose
    # untyped rows take the kappa * flux convention, celeste.py:52-55) and has settled (sweep to sweep within 1e-3)
    lls = np.array([t[3] for t in out["device"]])
    assert ll > f["ll"] and np.all(np.abs(np.diff(lls)) < 1e-3 * abs(ll))
    assert np.all(np.abs(u - f["radec"]) < 2e-3) and np.all(fl > 0)
