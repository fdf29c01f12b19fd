"""GPU parity tests of the older per-profile galaxy route, galaxy_source_like
(row (f)1's other half) and the named mixture API (mog_loglike, MixtureOfGaussians), all through
the C ABI, against the reference-run goldens and the CPU oracle.

Tolerances: stamps / tables 1e-10, log-likelihoods 1e-11, boxes bit-exact."""
import numpy as np
import pytest

from conftest import load_golden, unpack_ragged

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


@pytest.fixture(scope="module")
def imgs(cel):
    rec = load_golden("bands_253.npz")
    return rec, [cel.FitsImage.from_record(BANDS[b], rec, b, rec["nelec"][b]) for b in range(5)]


def _perm():
    return np.array([(c % 14) * 3 + c // 14 for c in range(42)])     # galaxy-major -> PSF-major


def test_a16_named_entries_match_reference_tables_by_permutation(cel, orc, imgs):
    """gen_galaxy_psf_mixture_params / gen_galaxy_prof_psf_mixture_params (celeste_fast.pyx:29-140):
    device-built, PSF-major; the reference's own MixtureOfGaussians tables (goldens, galaxy-major)
    are the same numbers in another order."""
    from desi_mcmc_amd import celeste_fast, mixture_profiles as mp
    rec, im = imgs
    g = load_golden("galaxy_stamps.npz")
    perm = _perm()
    for tag in ("s", "b"):
        for i in range(len(g[tag + "_th"])):
            bi = g[tag + "_band"][i]
            th, tinv = g[tag + "_th"][i], g[tag + "_tinv"][i]
            w, m, c = celeste_fast.gen_galaxy_psf_mixture_params(
                np.array([th[0], 1. - th[0]]), tinv @ tinv.T, g[tag + "_pix"][i], im[bi].weights, im[bi].means,
                im[bi].covars, mp.exp_amp, mp.exp_var, mp.dev_amp, mp.dev_var)
            assert w.shape == (42,) and m.shape == (42, 2) and c.shape == (42, 2, 2)
            np.testing.assert_allclose(w, g[tag + "_cw"][i][perm], rtol=1e-13)
            np.testing.assert_allclose(m, g[tag + "_cm"][i][perm], rtol=1e-13)
            np.testing.assert_allclose(c, g[tag + "_cc"][i][perm], rtol=1e-9, atol=1e-18)
    # one profile, against the oracle's restatement of celeste_fast.pyx:100-140 (the device fuses
    # cov + var * W into one fma: last-bit differences only)
    W = np.array([[2.3, 0.4], [0.4, 1.1]])
    for amp, var in ((mp.exp_amp, mp.exp_var), (mp.dev_amp, mp.dev_var)):
        w, m, c = celeste_fast.gen_galaxy_prof_psf_mixture_params(W, [20.25, 30.5], im[2].weights, im[2].means,
                                                                  im[2].covars, amp, var)
        ow, om, oc = orc.galaxy_prof_psf_mixture_params(W, [20.25, 30.5], im[2].weights, im[2].means, im[2].covars, amp, var)
        assert np.array_equal(w, ow) and np.array_equal(m, om)
        np.testing.assert_allclose(c, oc, rtol=1e-14)
    # batch form: N sources in one device call
    rs = np.random.RandomState(0)
    A = rs.randn(5, 2, 2)
    Ws = A @ A.transpose(0, 2, 1)
    vs = rs.rand(5, 2) * 50
    bw, bm, bc = celeste_fast.gen_galaxy_prof_psf_mixture_params_batch(Ws, vs, im[1].weights, im[1].means, im[1].covars,
                                                                       mp.dev_amp, mp.dev_var)
    for n in range(5):
        ow, om, oc = orc.galaxy_prof_psf_mixture_params(Ws[n], vs[n], im[1].weights, im[1].means, im[1].covars,
                                                        mp.dev_amp, mp.dev_var)
        assert np.array_equal(bw[n], ow) and np.array_equal(bm[n], om)
        np.testing.assert_allclose(bc[n], oc, rtol=1e-14)
    with pytest.raises(ValueError):
        celeste_fast.gen_galaxy_prof_psf_mixture_params(W, [1., 2.], im[2].weights, im[2].means, im[2].covars,
                                                        mp.exp_amp, mp.dev_var)


def test_a18_profile_images_vs_oracle(cel, orc, imgs):
    """gen_galaxy_prof_psf_image (celeste_galaxy_conditionals.py:134-182): own int() box bit-exact,
    patch values, caller limits, return_patch=False, and the bound helper."""
    from desi_mcmc_amd import celeste_galaxy_conditionals as gal
    rec, im = imgs
    B = orc.pack_bands(rec)
    g = load_golden("galaxy_stamps.npz")
    n_checked = 0
    for i in range(0, len(g["s_th"]), 2):
        bi = g["s_band"][i]
        th, u = g["s_th"][i], g["s_u"][i]
        R = gal.gen_galaxy_transformation(th[1], th[3], th[2], im[bi].Ups_n)
        np.testing.assert_allclose(R, orc.galaxy_tinv(th[1], th[3], th[2], rec["ups"][bi]), rtol=1e-12)
        for prof in ("exp", "dev"):
            want, wy, wx = orc.galaxy_prof_psf_image(B[bi], 51, 51, prof, R, u)
            got, gy, gx = gal.gen_galaxy_prof_psf_image(prof, R, u, im[bi])
            assert (tuple(gy), tuple(gx)) == (wy, wx)
            np.testing.assert_allclose(got, want, rtol=1e-10, atol=1e-300)
            # caller-imposed limits (:162-164)
            lim = (3, 40, 7, 51)
            want, _, _ = orc.galaxy_prof_psf_image(B[bi], 51, 51, prof, R, u, lims=lim)
            got, gy, gx = gal.gen_galaxy_prof_psf_image(prof, R, u, im[bi], xlim=(7, 51), ylim=(3, 40))
            assert (gy, gx) == ((3, 40), (7, 51))
            np.testing.assert_allclose(got, want, rtol=1e-10, atol=1e-300)
            n_checked += 1
        # return_patch=False is honoured on this route (:178-182)
        full, fy, fx = gal.gen_galaxy_prof_psf_image("dev", R, u, im[bi], return_patch=False)
        want, wy, wx = orc.galaxy_prof_psf_image(B[bi], 51, 51, "dev", R, u)
        assert full.shape == (51, 51) and fy == (0, 51) and fx == (0, 51)
        emb = np.zeros((51, 51))
        emb[wy[0]:wy[1], wx[0]:wx[1]] = want
        np.testing.assert_allclose(full, emb, rtol=1e-10, atol=1e-300)
        # the bound helper (:235-256)
        from desi_mcmc_amd import mixture_profiles as mp
        w, m, c = orc.galaxy_prof_psf_mixture_params(R @ R.T, g["s_pix"][i], rec["weights"][bi], rec["means"][bi],
                                                     rec["covars"][bi], mp.exp_amp, mp.exp_var)
        np.testing.assert_allclose(gal.gen_galaxy_prof_psf_image_bound("exp", R, u, im[bi], ERROR=1e-5),
                                   orc.bounding_radius(w, m, c, 1e-5, center=g["s_pix"][i]), rtol=1e-12)
    assert n_checked >= 60
    with pytest.raises(AssertionError):
        gal.gen_galaxy_prof_psf_image("sersic", np.eye(2), g["s_u"][0], im[0])


def test_q9_old_and_current_galaxy_routes_differ_across_a_big_frame(cel, orc):
    """the current renderer (A17, cd_at_pixel) and the older route (A18, constant Ups_n)
    agree on the frame's reference declination and drift apart by ~1e-4 across a 2048-px frame."""
    from desi_mcmc_amd import celeste_galaxy_conditionals as gal, synth
    from test_hip_parity import frame_images
    rec = load_golden("bands_253.npz")
    H = W = 2048
    im = frame_images(cel, rec, H, W)[2]
    th = np.array([0.4, 1.5, 30.0, 0.6])
    R = gal.gen_galaxy_transformation(th[1], th[3], th[2], im.Ups_n)
    diffs = []
    for py in (H / 2.0, 40.0, 2000.0):
        u = im.pixel2equa(np.array([W / 2.0 + 0.3, py]))
        p17, yl, xl = gal.gen_galaxy_psf_image(th, u, im)
        yl, xl = (int(yl[0]), int(yl[1])), (int(xl[0]), int(xl[1]))
        fe, _, _ = gal.gen_galaxy_prof_psf_image("exp", R, u, im, xlim=xl, ylim=yl)
        fd, _, _ = gal.gen_galaxy_prof_psf_image("dev", R, u, im, xlim=xl, ylim=yl)
        f = th[0] * fe + (1. - th[0]) * fd
        core = p17 > 1e-3 * p17.max()
        diffs.append(np.max(np.abs(f[core] / p17[core] - 1.0)))
    assert diffs[0] < 1e-9                      # on the reference declination
    assert 1e-6 < diffs[1] < 5e-3 and 1e-6 < diffs[2] < 5e-3


def test_galaxy_source_like_and_grad_vs_oracle(cel, orc, imgs):
    from desi_mcmc_amd import celeste_galaxy_conditionals as gal
    from desi_mcmc_amd.sources import SamplePatch
    rec, im = imgs
    B = orc.pack_bands(rec)
    rs = np.random.RandomState(11)
    th = np.array([0.35, 1.2, 40.0, 0.7, 0.0, 0.0, 12.0, 25.0, 33.0, 41.0, 18.0])
    th[4:6] = im[2].pixel2equa(np.array([24.6, 26.2]))
    use = [0, 2, 3]
    images = [im[b] for b in use]
    boxes = [(4, 47, 6, 45), (0, 51, 0, 51), (10, 40, 12, 44)]
    # photons: Poisson around the model itself so that the gradient is small but not zero
    Zs, want = [], 0.0
    for b, box in zip(use, boxes):
        flux = th[6 + b] / rec["calib"][b] * rec["kappa"][b]
        R = orc.galaxy_tinv(th[1], th[3], th[2], rec["ups"][b])
        fe, _, _ = orc.galaxy_prof_psf_image(B[b], 51, 51, "exp", R, th[4:6], lims=box)
        fd, _, _ = orc.galaxy_prof_psf_image(B[b], 51, 51, "dev", R, th[4:6], lims=box)
        Zs.append(rs.poisson(flux * (th[0] * fe + (1 - th[0]) * fd)).astype(float))
        want += orc.galaxy_source_like(B[b], 51, 51, th[0:4], th[4:6], flux, box, Zs[-1])
    lims = [((b[0], b[1]), (b[2], b[3])) for b in boxes]
    got = gal.galaxy_source_like(th, Zs, images, limits=lims)
    np.testing.assert_allclose(got, want, rtol=1e-11)
    # the same photons handed over as sample patches, and (band 2) as a full frame
    sp = [SamplePatch(z, (b[0], b[1]), (b[2], b[3])) for z, b in zip(Zs, boxes)]
    np.testing.assert_allclose(gal.galaxy_source_like(th, sp, images), want, rtol=1e-11)
    np.testing.assert_allclose(gal.galaxy_source_like(th, [sp[0], Zs[1], sp[2]], images), want, rtol=1e-11)
    with pytest.raises(ValueError):
        gal.galaxy_source_like(th, Zs, images)            # a sub-frame patch without limits

    # gradient: the reference's formulas (celeste_galaxy_conditionals.py:44-88) restated with the oracle
    def like(t):
        tot = 0.0
        for b, box, Z in zip(use, boxes, Zs):
            tot += orc.galaxy_source_like(B[b], 51, 51, t[0:4], t[4:6], t[6 + b] / rec["calib"][b] * rec["kappa"][b], box, Z)
        return tot
    gth, gbs = 0.0, np.zeros(5)
    for b, box, Z in zip(use, boxes, Zs):
        R = orc.galaxy_tinv(th[1], th[3], th[2], rec["ups"][b])
        fe, _, _ = orc.galaxy_prof_psf_image(B[b], 51, 51, "exp", R, th[4:6], lims=box)
        fd, _, _ = orc.galaxy_prof_psf_image(B[b], 51, 51, "dev", R, th[4:6], lims=box)
        f = th[0] * fe + (1 - th[0]) * fd
        gth += np.sum((Z / f - th[6 + b]) * (fe - fd))
        gbs[b] += 1. / th[6 + b] * np.sum(Z) - np.sum(f)
    gru = np.zeros(5)
    for i, k in enumerate([1, 2, 3, 4, 5]):
        de = np.zeros(11)
        de[k] = 1e-5
        gru[i] = (like(th + de) - like(th - de)) / 2e-5
    want_grad = np.concatenate([[gth], gru, gbs])
    got_grad = gal.galaxy_source_like_grad(th, Zs, images, limits=lims)
    assert got_grad.shape == (11,)
    np.testing.assert_allclose(got_grad[[0, 6, 7, 8, 9, 10]], want_grad[[0, 6, 7, 8, 9, 10]], rtol=1e-9)
    # central differences of numbers ~1e6 with step 1e-5: ll agrees to 1e-11 relative -> 1e-5 * 1e6 / 2e-5
    scale = abs(want) * 1e-11 / 2e-5
    np.testing.assert_allclose(got_grad[1:6], want_grad[1:6], atol=10 * scale + 1e-6, rtol=1e-6)


def test_field_render_accepts_profile_route_sources(cel, orc, imgs):
    """source type 2 (shape = theta, W) through cel_render_field: lambda = eps + sum counts * A18 stamp."""
    from desi_mcmc_amd import celeste_galaxy_conditionals as gal
    rec, im = imgs
    B = orc.pack_bands(rec)
    g = load_golden("galaxy_stamps.npz")
    ctx = cel.default_context(0)
    iset = cel.ImageSet(ctx, np.stack([i.band_record() for i in im]), 51, 51, nelec=rec["nelec"])
    idx = [0, 9, 17, 30, 44, 60]
    typ = np.full(len(idx), 2, dtype=np.int32)
    radec = g["s_u"][idx]
    shapes, Rs, profs = [], [], []
    for n, i in enumerate(idx):
        th = g["s_th"][i]
        R = gal.gen_galaxy_transformation(th[1], th[3], th[2], im[0].Ups_n)     # all five stamps share the WCS
        Wm = R @ R.T
        theta = 1.0 if n % 2 == 0 else 0.0
        shapes.append([theta, Wm[0, 0], Wm[0, 1], Wm[1, 1]])
        Rs.append(R)
        profs.append("exp" if theta == 1.0 else "dev")
    counts = np.random.RandomState(2).uniform(500, 5000, size=(len(idx), 5))
    sset = cel.SourceSet(ctx, len(idx), 5).set(typ, radec, counts, np.array(shapes))
    ll, llb = iset.render(sset, loglik=True)
    lam = iset.model_images()
    for b in range(5):
        want = np.full((51, 51), rec["eps"][b])
        for n in range(len(idx)):
            p, yl, xl = orc.galaxy_prof_psf_image(B[b], 51, 51, profs[n], Rs[n], radec[n])
            if p is not None:
                want[yl[0]:yl[1], xl[0]:xl[1]] += counts[n, b] * p
        np.testing.assert_allclose(lam[b], want, rtol=1e-9)        # (the shipping drop threshold of the field render: T = 24)
        np.testing.assert_allclose(llb[b], orc.poisson_loglike(rec["nelec"][b], want), rtol=1e-11)


def test_mog_loglike_and_mixture_class(cel, orc, imgs):
    """util.dists.mog: mog_loglike against the reference's own values (evaluator.npz), the class's
    algebra against the reference's convolved tables, evaluate_grid against the reference's galaxy
    patches, and the weight <= 0 semantics."""
    from desi_mcmc_amd.util.dists import mog
    from desi_mcmc_amd import celeste_galaxy_conditionals as gal
    rec, im = imgs
    e = load_golden("evaluator.npz")
    got = mog.mog_loglike(e["X"], e["means"], e["invcovs"], np.exp(e["logdets"]), e["ws"])
    np.testing.assert_allclose(got, e["mog_loglike"], rtol=1e-10, atol=1e-10)
    one = mog.mog_loglike(e["X"][17], e["means"], e["invcovs"], np.exp(e["logdets"]), e["ws"])
    assert np.ndim(one) == 0 and abs(one - e["mog_loglike"][17]) <= 1e-10 * abs(e["mog_loglike"][17])
    # far from every component: finite log-density (logsumexp), where the direct sum underflows
    far = np.array([[4000.0, -3000.0]])
    v = mog.mog_loglike(far, e["means"], e["invcovs"], np.exp(e["logdets"]), e["ws"])
    np.testing.assert_allclose(v, orc.np_mog_loglike(far, e["means"], e["invcovs"], np.exp(e["logdets"]), e["ws"]), rtol=1e-12)
    assert np.isfinite(v[0]) and v[0] < -1e4
    # Q8: a negative weight is NaN on this route, a zero weight just drops out
    ws = e["ws"].copy()
    ws[3] = -ws[3]
    with np.errstate(all="ignore"):
        assert np.all(np.isnan(mog.mog_loglike(e["X"][:5], e["means"], e["invcovs"], np.exp(e["logdets"]), ws)))
        ws[3] = 0.0
        z = mog.mog_loglike(e["X"][:50], e["means"], e["invcovs"], np.exp(e["logdets"]), ws)
        keep = np.arange(42) != 3
        np.testing.assert_allclose(z, orc.np_mog_loglike(e["X"][:50], e["means"][keep], e["invcovs"][keep],
                                                         np.exp(e["logdets"])[keep], e["ws"][keep]), rtol=1e-11)
    # the class: convex_combine -> apply_affine -> convolve, as gen_galaxy_psf_image builds cmix
    g = load_golden("galaxy_stamps.npz")
    patches = unpack_ragged(g["s_flat"], g["s_offs"], g["s_shapes"])
    stride = int(g["s_stride"])
    for i in (0, 13, 29, 52):
        bi = g["s_band"][i]
        th = g["s_th"][i]
        galmix = mog.MixtureOfGaussians.convex_combine([gal.galaxy_prof_dict['exp'], gal.galaxy_prof_dict['dev']],
                                                       [th[0], 1. - th[0]])
        amix = galmix.apply_affine(g["s_tinv"][i], g["s_pix"][i])
        cmix = amix.convolve(im[bi].psf)
        np.testing.assert_allclose(cmix.pis, g["s_cw"][i], rtol=1e-13)
        np.testing.assert_allclose(cmix.means, g["s_cm"][i], rtol=1e-13)
        np.testing.assert_allclose(cmix.covs, g["s_cc"][i], rtol=1e-9, atol=1e-18)
        y0, y1, x0, x1 = g["s_box"][i]
        grid = cmix.evaluate_grid((x0, x1), (y0, y1))
        np.testing.assert_allclose(grid[::stride, ::stride], patches[i], rtol=1e-10, atol=1e-300)
        np.testing.assert_allclose(cmix.pdf(np.array([[x0 + 1., y0 + 2.]]))[0], grid[2, 1], rtol=1e-12)
    assert im[2].psf_mog.K == 3 and im[2].psf is im[2].psf_mog
    with pytest.raises(AssertionError):
        cmix.evaluate_grid((5, 5), (0, 3))


def test_off_image_star_with_imposed_limits_is_none_not_stale(cel, imgs):
    """A star failing the reference's overlap test stays (None) under caller limits, also in a batch,
    and its slot never carries data of an earlier call (celeste.py:130-135)."""
    rec, im = imgs
    ctx = cel.default_context(0)
    iset = cel.ImageSet(ctx, np.stack([i.band_record() for i in im]), 51, 51, nelec=rec["nelec"])
    u_in = im[2].pixel2equa(np.array([25.0, 25.0]))
    u_off = im[2].pixel2equa(np.array([400.0, 25.0]))            # x > 2 * rows: overlap test fails
    typ = np.zeros(3, dtype=np.int32)
    radec = np.array([u_in, u_off, u_in])
    counts = np.full((3, 5), 1000.0)
    sset = cel.SourceSet(ctx, 3, 5).set(typ, radec, counts)
    boxes = np.tile(np.array([[5, 45, 5, 45]], dtype=np.int32), (3, 1))
    first, _ = iset.stamps(sset, 2, scaled=True, boxes_in=boxes)              # fills the scratch buffer
    assert first[1] is None and first[0] is not None and np.array_equal(first[0], first[2])
    # raw ABI with a non-empty slot for the missing star: zero-filled, not stale
    import ctypes as C
    from desi_mcmc_amd import _lib as L
    offs = np.arange(4, dtype=np.int64) * 1600
    flat = np.full(4800, 7.0)
    L.check(L.lib().cel_render_stamps(iset._h, sset._h, 2, 1, boxes.ctypes.data_as(L.c_int32_p),
                                      offs.ctypes.data_as(L.c_int64_p), flat.ctypes.data, L.CEL_HOST))
    assert np.all(flat[1600:3200] == 0.0)
    np.testing.assert_array_equal(flat[:1600].reshape(40, 40), first[0])
    # the model-class route that sums patches on imposed limits
    from desi_mcmc_amd import models
    m = models.Celeste()
    m.initialize_sources(init_src_params=[cel.SrcParams(u=u, a=0, fluxes=np.full(5, 10.0)) for u in radec])
    mod = m.render_model_image(im[2], xlim=(5, 45), ylim=(5, 45))
    one = models.Celeste()
    one.initialize_sources(init_src_params=[cel.SrcParams(u=u_in, a=0, fluxes=np.full(5, 10.0))])
    single = one.render_model_image(im[2], xlim=(5, 45), ylim=(5, 45))
    np.testing.assert_allclose(mod - im[2].epsilon, 2 * (single - im[2].epsilon), rtol=1e-12)
