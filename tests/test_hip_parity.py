"""GPU parity tests: the HIP path (through the C ABI) against the goldens and the CPU oracle.

Tolerances.  north_star: 1e-6 relative for fp64 model pixels and log-lik.  The tests assert
tighter bounds that the design guarantees:
  RT_STAMP 1e-10  unit-flux stamps (direct evaluator) against goldens / oracle
  RT_LAM   1e-9   model pixels lambda at the library's SHIPPING drop threshold (T = 24 for the field render: a dropped
                  component is < eps*e^-24 = 3.8e-11 eps on its tile; a pixel under a dozen galaxies' tails collects a few e-10)
  RT_LAM_STRICT 1e-10  the same under tail_log(ctx, "strict") (T = 32), with the direct evaluator, and wherever nothing is
                  dropped: star-only fields (the star passes keep all three components)
  RT_LL    1e-11  log-likelihoods
The suite runs at the shipping defaults (tests/conftest.py sets no threshold); one strict variant per kernel family:
test_mixed_field_vs_oracle[recurrence-1-strict] and test_config3_full_vs_oracle (k_render_hw),
test_crowded_field_...[strict], test_mini_field_golden (tails 32 / 0 / 25 by name), test_patch_loglik_adversarial_...
(per-source kernels: their default IS 32), tests/test_gibbs.py::test_photon_list_route_vs_oracle.
Boxes (integer work) are compared bit-exact.
"""
import numpy as np
import pytest

from conftest import fuzz_seeds as _fuzz_seeds, load_golden, tail_log, unpack_ragged

pytestmark = pytest.mark.gpu

RT_STAMP, RT_LAM, RT_LL = 1e-10, 1e-9, 1e-11
RT_LAM_STRICT = 1e-10
RT_LAM_DEFAULT = RT_LAM
BANDS = ["u", "g", "r", "i", "z"]


@pytest.fixture(scope="module")
def cel():
    import desi_mcmc_amd as m
    return m


@pytest.fixture(scope="module")
def ctx(cel):
    return cel.default_context(0)


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    return oracle


@pytest.fixture(scope="module")
def stamp_images(cel):
    rec = load_golden("bands_253.npz")
    return rec, [cel.FitsImage.from_record(BANDS[b], rec, b, rec["nelec"][b]) for b in range(5)]


def frame_images(cel, rec, H, W, nelec=None):
    """FitsImage objects of a synthetic H x W frame whose band constants come from `rec`."""
    out = []
    for b in range(len(rec["eps"])):
        r = {k: np.array(v, copy=True) for k, v in rec.items() if k in
             ("eps", "kappa", "calib", "weights", "means", "covars", "rho", "phi", "ups")}
        r["rho"][b] = [W / 2.0, H / 2.0]
        out.append(cel.FitsImage.from_record(BANDS[b % 5], r, b, np.zeros((H, W)) if nelec is None else nelec[b]))
    return out


# ------------------------------------------------------------------------------------------
def test_fitsimage_radius_matches_reference(stamp_images):
    rec, imgs = stamp_images
    for b, im in enumerate(imgs):
        np.testing.assert_allclose(im.R, rec["R"][b], rtol=1e-13)
        np.testing.assert_allclose(im.invcovars, rec["invcovars"][b], rtol=1e-12)
        np.testing.assert_allclose(im.logdets, rec["logdets"][b], rtol=1e-12)


def test_gmm_like_2d_reference_seed41(cel):
    """the reference's own test (CelestePy/test/test_gmm.py:63-105) against the HIP evaluator"""
    from desi_mcmc_amd.util.like import gmm_like_2d
    g = load_golden("evaluator.npz")
    got = gmm_like_2d(g["X"], g["ws"], g["means"], g["covs"])
    np.testing.assert_allclose(got, g["gmm_prob"], rtol=1e-10, atol=1e-300)
    assert np.allclose(got, g["gmm_prob"])            # the reference's own criterion
    buf = np.zeros(g["X"].shape[0])
    out = gmm_like_2d(g["X"], g["ws"], g["means"], g["covs"], probs=buf)
    assert out is buf and np.array_equal(buf, got)    # caller buffer is filled in place


def test_gmm_like_2d_edge_cases(cel, orc):
    from desi_mcmc_amd.util.like import gmm_like_2d
    rs = np.random.RandomState(3)
    for N, K in ((1, 1), (63, 3), (257, 64), (1000, 65), (4097, 130)):
        x = rs.randn(N, 2) * 3
        ws = rs.rand(K) - 0.2          # negative weights are legal on this route (Q8)
        mus = rs.randn(K, 2)
        A = rs.randn(K, 2, 2)
        sigs = A @ A.transpose(0, 2, 1) + 0.05 * np.eye(2)
        got = gmm_like_2d(x, ws, mus, sigs)
        np.testing.assert_allclose(got, orc.gmm_like_2d(x, ws, mus, sigs), rtol=1e-10, atol=1e-280)
    assert gmm_like_2d(np.zeros((0, 2)), np.ones(1), np.zeros((1, 2)), np.eye(2)[None]).shape == (0,)
    with pytest.raises(ValueError):
        gmm_like_2d(np.zeros((4, 2)), np.ones(3), np.zeros((2, 2)), np.zeros((3, 2, 2)))
    with pytest.raises(ValueError):
        gmm_like_2d(np.zeros((4, 2)), np.ones(3), np.zeros((3, 2)), np.zeros((3, 2, 3)))


def test_star_stamps_golden_boxes_edges_q1(cel, stamp_images):
    from desi_mcmc_amd import celeste
    rec, imgs = stamp_images
    g = load_golden("star_stamps.npz")
    patches = unpack_ragged(g["flat"], g["offs"], g["shapes"])
    for i, (bi, u, box, none) in enumerate(zip(g["band"], g["u"], g["box"], g["is_none"])):
        patch, yl, xl = celeste.gen_point_source_psf_image(u, imgs[bi])
        if none:
            assert patch is None and yl is None and xl is None        # Q1: (None, None, None)
            continue
        if patches[i].size == 0:
            assert patch is None
            continue
        assert (yl[0], yl[1], xl[0], xl[1]) == tuple(int(t) for t in box)
        np.testing.assert_allclose(patch, patches[i], rtol=RT_STAMP, atol=1e-300)
    # caller limits and full-frame embedding (celeste.py:145-152,169-176)
    p, yl, xl = celeste.gen_point_source_psf_image(g["lim_u"], imgs[2], xlim=tuple(g["lim_xlim"]),
                                                   ylim=tuple(g["lim_ylim"]))
    np.testing.assert_allclose(p, g["lim_patch"], rtol=RT_STAMP)
    full, yl, xl = celeste.gen_point_source_psf_image(g["lim_u"], imgs[2], return_patch=False)
    assert yl == (0, 51) and xl == (0, 51)
    np.testing.assert_allclose(full, g["full_image"], rtol=RT_STAMP)
    grid = np.ones((51, 51))
    out, _, _ = celeste.gen_point_source_psf_image(g["lim_u"], imgs[2], return_patch=False, psf_grid=grid)
    assert out is grid                                                # written in place
    # pixel_grid route (generic points)
    xx, yy = np.meshgrid(np.arange(5., 40.), np.arange(12., 51.), indexing="xy")
    pg = np.column_stack((xx.ravel(), yy.ravel()))
    p2, _, _ = celeste.gen_point_source_psf_image(g["lim_u"], imgs[2], xlim=(5, 40), ylim=(12, 51), pixel_grid=pg)
    np.testing.assert_allclose(p2, g["lim_patch"], rtol=RT_STAMP)


@pytest.mark.parametrize("tag", ["s", "b"])
def test_galaxy_stamps_golden(cel, stamp_images, tag):
    from desi_mcmc_amd import celeste_galaxy_conditionals as gal
    rec, imgs = stamp_images
    g = load_golden("galaxy_stamps.npz")
    if tag == "b":
        imgs = frame_images(cel, rec, int(g["big_H"]), int(g["big_W"]))
    stride = int(g[tag + "_stride"])
    patches = unpack_ragged(g[tag + "_flat"], g[tag + "_offs"], g[tag + "_shapes"])
    for i in range(len(patches)):
        img = imgs[g[tag + "_band"][i]]
        th, u = g[tag + "_th"][i], g[tag + "_u"][i]
        patch, yl, xl = gal.gen_galaxy_psf_image(th, u, img)
        assert (yl[0], yl[1], xl[0], xl[1]) == tuple(g[tag + "_box"][i])
        assert isinstance(yl[0], float)                               # Q5: float limits
        np.testing.assert_allclose(patch.sum(), g[tag + "_sum"][i], rtol=1e-10)
        np.testing.assert_allclose(patch[::stride, ::stride], patches[i], rtol=RT_STAMP, atol=1e-300)
        np.testing.assert_allclose(gal.gen_galaxy_transformation(th[1], th[3], th[2], img.cd_at_pixel(*g[tag + "_pix"][i])),
                                   g[tag + "_tinv"][i], rtol=1e-9)
        pis, means, covs, _ = gal.galaxy_mixture(th, u, img)
        np.testing.assert_allclose(pis, g[tag + "_cw"][i], rtol=1e-13)
        np.testing.assert_allclose(covs, g[tag + "_cc"][i], rtol=1e-9)


@pytest.mark.parametrize("kernel,tail", [("direct", 32.0), ("recurrence", "default"), ("recurrence", 32.0), ("recurrence", 0.0),
                                         ("recurrence", 25.0)])
def test_mini_field_golden(cel, ctx, kernel, tail):
    """mixed star/galaxy 96x80 field, 5 bands: lambda, per-band ll, per-source patches"""
    g = load_golden("mini_field.npz")
    H, W = int(g["H"]), int(g["W"])
    ctx.set_kernel(kernel)
    ctx.set_tail_log(tail)
    try:
        from desi_mcmc_amd import field
        bands = field.pack_bands(g)
        iset = cel.ImageSet(ctx, bands, H, W, nelec=g["nelec"])
        counts = g["flux"] / g["calib"][None, :] * g["kappa"][None, :]
        sset = cel.SourceSet(ctx, 12, 5).set(g["is_gal"], g["radec"], counts, g["shape"])
        ll, llb = iset.render(sset, loglik=True)
        lam = iset.model_images()
        rt = 1e-8 if tail == 25.0 else (RT_LAM if tail == "default" else RT_LAM_STRICT)
        np.testing.assert_allclose(lam, g["lam"], rtol=rt)
        np.testing.assert_allclose(llb, g["ll_band"], rtol=RT_LL)
        np.testing.assert_allclose(ll, g["ll"], rtol=RT_LL)
        st = iset.stats()
        assert st["n_srcpix"] == sum(int(np.prod(s)) for s in g["patch_shapes"])
        # scaled per-source patches (gen_src_image_with_fluxes)
        patches = unpack_ragged(g["patch_flat"], g["patch_offs"], g["patch_shapes"])
        for b in (0, 2, 4):
            got, boxes = iset.stamps(sset, b, scaled=True)
            for s in range(12):
                i = b * 12 + s
                assert tuple(boxes[s]) == tuple(g["patch_box"][i])
                np.testing.assert_allclose(got[s], patches[i], rtol=RT_STAMP, atol=1e-300)
    finally:
        ctx.set_kernel("recurrence")
        ctx.set_tail_log("default")


def test_reference_api_on_mini_field(cel, stamp_images):
    """gen_model_image / celeste_likelihood[_multi_image] / gen_src_image with SrcParams lists"""
    from desi_mcmc_amd import celeste
    rec, _ = stamp_images
    g = load_golden("mini_field.npz")
    H, W = int(g["H"]), int(g["W"])
    imgs = frame_images(cel, {k: g[k] for k in g}, H, W, nelec=g["nelec"])
    idx = g["star_idx"]
    stars = [cel.SrcParams(u=g["radec"][s], a=0, fluxes=dict(zip(BANDS, g["flux"][s]))) for s in idx]
    for b in (0, 3):
        np.testing.assert_allclose(celeste.gen_model_image(stars, imgs[b]), g["star_lam"][b], rtol=RT_LAM_STRICT)
    np.testing.assert_allclose(celeste.celeste_likelihood_multi_image(stars, imgs), g["star_ll"], rtol=RT_LL)
    # mixed list through the extended gen_model_image (Q3)
    srcs = [cel.SrcParams(u=g["radec"][s], a=int(g["is_gal"][s]), fluxes=g["flux"][s], theta=g["shape"][s, 0],
                          sigma=g["shape"][s, 1], phi=g["shape"][s, 2], rho=g["shape"][s, 3]) for s in range(12)]
    np.testing.assert_allclose(celeste.gen_model_image(srcs, imgs[2]), g["lam"][2], rtol=RT_LAM)
    np.testing.assert_allclose(celeste.celeste_likelihood(srcs, imgs[1]), g["ll_band"][1], rtol=RT_LL)
    np.testing.assert_allclose(celeste.celeste_likelihood_multi_image(srcs, imgs), g["ll"], rtol=RT_LL)
    # per-source images
    patches = unpack_ragged(g["patch_flat"], g["patch_offs"], g["patch_shapes"])
    for s in (0, 1, 4, 7):
        p, yl, xl = celeste.gen_src_image_with_fluxes(srcs[s], imgs[2])
        i = 2 * 12 + s
        assert (int(yl[0]), int(yl[1]), int(xl[0]), int(xl[1])) == tuple(g["patch_box"][i])
        np.testing.assert_allclose(p, patches[i], rtol=RT_STAMP, atol=1e-300)
    # responsibilities sum to one (celeste.py:222-234)
    layers = celeste.gen_src_prob_layers(srcs, imgs[2])
    assert layers.shape == (13, H, W)
    # (the layers are stamps at the per-source threshold, their denominator the model image at the field render's: 1e-9 at the
    # shipping defaults, 1e-12 when both run at T = 32)
    np.testing.assert_allclose(layers.sum(axis=0), 1.0, rtol=1e-9)
    with tail_log(cel.default_context(0), "strict"):
        np.testing.assert_allclose(celeste.gen_src_prob_layers(srcs, imgs[2]).sum(axis=0), 1.0, rtol=1e-12)
    # epsilon is updatable without re-upload (models.py:156-160)
    old = imgs[1].epsilon
    imgs[1].epsilon = old * 1.5
    lam = celeste.gen_model_image(srcs, imgs[1])
    np.testing.assert_allclose(lam, g["lam"][1] + 0.5 * old, rtol=RT_LAM)        # (what is dropped is measured against the sky level)
    imgs[1].epsilon = old


def test_config1_real_stamps(cel, stamp_images):
    from desi_mcmc_amd import celeste
    rec, imgs = stamp_images
    g = load_golden("config1.npz")
    srcs = [cel.SrcParams(u=row[:2], fluxes=dict(zip(BANDS, row[2:]))) for row in g["cat"]]   # a=None (Q2)
    for b in range(5):
        np.testing.assert_allclose(celeste.gen_model_image(srcs, imgs[b]), g["lam"][b], rtol=RT_LAM)
        np.testing.assert_allclose(celeste.celeste_likelihood(srcs, imgs[b]), g["ll_band"][b], rtol=RT_LL)
    np.testing.assert_allclose(celeste.celeste_likelihood_multi_image(srcs, imgs), g["ll"], rtol=RT_LL)
    star = cel.SrcParams(u=g["one_u"], a=0, fluxes=dict(zip(BANDS, g["one_flux"])))
    np.testing.assert_allclose(celeste.gen_src_image(star, imgs[2]), g["one_patch"], rtol=RT_STAMP)
    np.testing.assert_allclose(celeste.gen_model_image([star], imgs[2]), g["one_lam"], rtol=RT_LAM_STRICT)
    np.testing.assert_allclose(celeste.celeste_likelihood([star], imgs[2]), g["one_ll"], rtol=RT_LL)


# ------------------------------------------------------------------------------------------
# every real field the reference ships (tests/golden/real_fields.npz: its own run on all of them)
# ------------------------------------------------------------------------------------------
def _real_images(cel, f):
    return [cel.FitsImage.from_record(BANDS[b], f["rec"], b, f["nelec"][b]) for b in range(5)]


def _real_sources(cel, f):
    return [cel.SrcParams(u=f["radec"][s], fluxes=dict(zip(BANDS, f["flux"][s]))) for s in range(len(f["radec"]))]    # a = None (Q2)


def test_every_real_field_through_the_reference_api(cel):
    """configs[0] on ALL the real data in the reference's tree: 100 fields (data/stamps 11, data/stamp_catalog 63,
    data/galaxy_stamps 25, data/real 1: 500 SDSS images, 221 catalogue sources loaded as util/misc/init_utils.py:9-60 loads
    them) through the drop-in API -- FitsImage's derived fields, gen_model_image, celeste_likelihood[_multi_image] (celeste.py:203-252),
    every source's box from gen_point_source_psf_image, the star stamps and gen_galaxy_psf_image
    (celeste_galaxy_conditionals.py:185-214) on each field's own PSF and WCS -- against the reference's own run."""
    from conftest import real_fields
    from desi_mcmc_amd import celeste
    from desi_mcmc_amd import celeste_galaxy_conditionals as gal
    g, fields = real_fields()
    assert len(fields) == 100
    imgs_of = {}
    for f in fields:
        imgs = imgs_of[f["index"]] = _real_images(cel, f)
        srcs = _real_sources(cel, f)
        for b, im in enumerate(imgs):
            np.testing.assert_allclose(im.R, f["rec"]["R"][b], rtol=1e-13)
            np.testing.assert_allclose(im.invcovars, f["rec"]["invcovars"][b], rtol=1e-12)
            np.testing.assert_allclose(im.logdets, f["rec"]["logdets"][b], rtol=1e-12)
            lam = celeste.gen_model_image(srcs, im)
            if f["lam"] is not None:
                np.testing.assert_allclose(lam, f["lam"][b], rtol=RT_LAM_STRICT, err_msg=f["name"])
            else:
                np.testing.assert_allclose(lam[::4, ::4], f["lam_sub"][b], rtol=RT_LAM_STRICT, err_msg=f["name"])
            np.testing.assert_allclose(celeste.celeste_likelihood(srcs, im), f["ll_band"][b], rtol=RT_LL, err_msg=f["name"])
            for s, q in enumerate(srcs):
                patch, yl, xl = celeste.gen_point_source_psf_image(q.u, im)
                assert (patch is None) == bool(f["src_none"][s, b])
                if patch is not None:
                    assert (yl[0], yl[1], xl[0], xl[1]) == tuple(int(t) for t in f["src_box"][s, b]), (f["name"], s, b)
        np.testing.assert_allclose(celeste.celeste_likelihood_multi_image(srcs, imgs), f["ll"], rtol=RT_LL, err_msg=f["name"])
    by_index = {f["index"]: f for f in fields}
    for k, want in enumerate(unpack_ragged(g["st_flat"], g["st_offs"], g["st_shapes"])):
        f = by_index[int(g["st_field"][k])]
        patch, yl, xl = celeste.gen_point_source_psf_image(f["radec"][int(g["st_src"][k])], imgs_of[f["index"]][2])
        assert (yl[0], yl[1], xl[0], xl[1]) == tuple(int(t) for t in g["st_box"][k])
        np.testing.assert_allclose(patch, want, rtol=RT_STAMP, atol=1e-300)
    for k, want in enumerate(unpack_ragged(g["g_flat"], g["g_offs"], g["g_shapes"])):
        img = imgs_of[int(g["g_field"][k])][int(g["g_band"][k])]
        patch, yl, xl = gal.gen_galaxy_psf_image(g["g_th"][k], g["g_u"][k], img)
        assert (yl[0], yl[1], xl[0], xl[1]) == tuple(g["g_box"][k]) and isinstance(yl[0], float)
        np.testing.assert_allclose(patch, want, rtol=RT_STAMP, atol=1e-300)


def test_small_star_path_scans_a_catalogue_of_thousands(cel, ctx, orc):
    """k_small_stars with more stars than one scan pass holds (1 024: 16 per lane): 3 000 stars on 5 x 384 x 448 -- the second
    and third pass of the position filter, ~60 candidates per part -- against the oracle and the general kernel"""
    from desi_mcmc_amd import synth
    L = cel._lib
    f = synth.SyntheticField(ctx, 3000, 5, 384, 448, frac_gal=0.0, seed=5)
    ctx.profile(True)
    ll, llb = f.images.render(f.sources, loglik=True)
    assert ctx.profile_get("small_stars")[1] == 1 and ctx.profile_get("render")[1] == 0
    ctx.profile(False)
    lam = f.images.model_images()
    o_lam, o_ll, o_st = orc.render_field(oracle_bands(f), f.H, f.W, f.src["type"], f.src["radec"], f.src["counts"], f.src["shape"], f.nelec)
    np.testing.assert_allclose(lam, o_lam, rtol=RT_LAM_STRICT)
    np.testing.assert_allclose(llb, o_ll, rtol=RT_LL)
    st = f.images.stats()
    assert st["n_srcpix"] == o_st["n_srcpix"]
    ctx.set_option(L.CEL_OPT_STAR_TILES, 0)
    try:
        ll0, llb0 = f.images.render(f.sources, loglik=True)
        np.testing.assert_allclose(f.images.model_images(), lam, rtol=1e-13)
        np.testing.assert_allclose(llb0, llb, rtol=1e-13)
    finally:
        ctx.set_option(L.CEL_OPT_STAR_TILES, 1)


def test_real_field_set_dealt_to_the_ranks(cel, ctx):
    """BASELINE configs[3] on the only real data that exists (the Stripe-82 set is absent from the reference tree,
    .MISSING_LARGE_BLOBS:2-4): the 99 51 x 51 real fields as ONE field set dealt by dist.field_shard -- world 1 here, on the
    GPU, through the resident ImageSet / SourceSet path bench.py --workload fields8_2048 runs; world 2 on gloo in
    tests/test_dist_gloo.py -- every field's per-band log-likelihoods and the job's all-reduced sum equal the reference's."""
    from conftest import real_fields
    from desi_mcmc_amd import dist, field
    g, fields = real_fields()
    fields = [f for f in fields if (f["H"], f["W"]) == (51, 51)]
    assert len(fields) == 99
    mine = dist.field_shard(len(fields), 1, 0)
    assert mine == list(range(99))
    total, want = np.zeros(5), np.zeros(5)
    sets = []
    for k in mine:
        f = fields[k]
        iset = cel.ImageSet(ctx, field.pack_bands(f["rec"]), 51, 51, nelec=f["nelec"])
        S = len(f["radec"])
        sset = cel.SourceSet(ctx, max(S, 1), 5).set(np.zeros(S, np.int32), f["radec"].reshape(S, 2),
                                                    (f["flux"] * f["rec"]["kappa"][None, :]).reshape(S, 5), np.zeros((S, 4)))
        sets.append((iset, sset))
    for rep in range(2):                                   # resident: a second pass re-renders without any upload
        total[:] = 0.0
        for k, (iset, sset) in zip(mine, sets):
            _, llb = iset.render(sset, loglik=True)
            np.testing.assert_allclose(llb, fields[k]["ll_band"], rtol=RT_LL, err_msg=fields[k]["name"])
            total += llb
    for k in mine:
        want += fields[k]["ll_band"]
    np.testing.assert_allclose(dist.allreduce_loglik(total), want, rtol=RT_LL)
    np.testing.assert_allclose(total.sum(), sum(f["ll"] for f in fields), rtol=RT_LL)


# ------------------------------------------------------------------------------------------
# seeded synthetic fields against the CPU oracle
# ------------------------------------------------------------------------------------------
def oracle_bands(field):
    """The field's band records for the oracle.  The synthetic records leave the star radius R = 0 = "compute it": the checker
    gets the ORACLE's radius (orc_bounding_radius on the band's PSF, pinned to the reference's calc_bounding_radius by golden
    radius.npz), and orc.checked_radius first asserts that the library derived the same number (1e-13) -- in every oracle test,
    the scaled-PSF and fuzz cases included, so a wrong R cannot move the boxes of library and checker alike."""
    from oracle import oracle as orc
    b = field.bands.copy()
    for i in range(b.shape[0]):
        b[i, 36] = orc.checked_radius(b[i], field.images.band(i)[36])      # the oracle's own radius; the library's must equal it
    return b


@pytest.mark.parametrize("kernel", ["direct", "recurrence"])
def test_config2_stars_512_vs_oracle(cel, ctx, orc, kernel):
    from desi_mcmc_amd import synth
    ctx.set_kernel(kernel)
    try:
        f = synth.SyntheticField.from_config(ctx, "stars1k_512")
        ll, llb = f.images.render(f.sources, loglik=True)
        lam = f.images.model_images()
        o_lam, o_ll, o_st = orc.render_field(oracle_bands(f), f.H, f.W, f.src["type"], f.src["radec"],
                                             f.src["counts"], f.src["shape"], f.nelec)
        np.testing.assert_allclose(lam, o_lam, rtol=RT_LAM_STRICT)
        np.testing.assert_allclose(llb, o_ll, rtol=RT_LL)
        st = f.images.stats()
        assert st["n_srcpix"] == o_st["n_srcpix"] and st["n_gauss"] == o_st["n_gauss"]
    finally:
        ctx.set_kernel("recurrence")


@pytest.mark.parametrize("kernel,layout,tail", [("direct", 1, "default"), ("recurrence", 1, "default"), ("recurrence", 1, "strict"),
                                                ("recurrence", 0, "default"), ("direct", 2, "default"), ("recurrence", 2, "default")])
def test_mixed_field_vs_oracle(cel, ctx, orc, kernel, layout, tail):
    """config 3's source population at a size the oracle finishes in seconds: 400 sources, 3 bands,
    non-multiple-of-tile frame 500 x 333; every render-tile layout (CEL_OPT_TILE_LAYOUT: 64 x 32
    full-wave, 32 x 64 half-wave = default, 16 x 128 quarter-wave)"""
    from desi_mcmc_amd import synth
    ctx.set_kernel(kernel)
    ctx.set_tail_log(tail)
    ctx.set_option(cel._lib.CEL_OPT_TILE_LAYOUT, layout)            # read when the image set is created
    try:
        f = synth.SyntheticField(ctx, 400, 3, 333, 500, frac_gal=0.5, seed=7)
        ctx.set_option(cel._lib.CEL_OPT_TILE_LAYOUT, 1)
        ll, llb = f.images.render(f.sources, loglik=True)
        lam = f.images.model_images()
        o_lam, o_ll, o_st = orc.render_field(oracle_bands(f), f.H, f.W, f.src["type"], f.src["radec"],
                                             f.src["counts"], f.src["shape"], f.nelec)
        np.testing.assert_allclose(lam, o_lam, rtol=RT_LAM_STRICT if (tail == "strict" or kernel == "direct") else RT_LAM)
        np.testing.assert_allclose(llb, o_ll, rtol=RT_LL)
        assert f.images.stats()["n_srcpix"] == o_st["n_srcpix"]
    finally:
        ctx.set_kernel("recurrence")
        ctx.set_tail_log("default")
        ctx.set_option(cel._lib.CEL_OPT_TILE_LAYOUT, 1)


def test_edge_cases_empty_ragged_offimage(cel, ctx, orc):
    from desi_mcmc_amd import synth
    # no sources at all: lambda == epsilon, ll = sum(n log eps - eps)
    for H, W in ((1, 1), (31, 65), (33, 64), (64, 63)):
        bands = synth.make_bands(H, W, 2)
        iset = cel.ImageSet(ctx, bands, H, W, nelec=np.full((2, H, W), 3.0))
        sset = cel.SourceSet(ctx, 4, 2).set(np.zeros(0, np.int32), np.zeros((0, 2)), np.zeros((0, 2)), np.zeros((0, 4)))
        ll, llb = iset.render(sset, loglik=True)
        lam = iset.model_images()
        for b in range(2):
            assert np.all(lam[b] == bands[b, 0])
            np.testing.assert_allclose(llb[b], H * W * (3.0 * np.log(bands[b, 0]) - bands[b, 0]), rtol=1e-13)
    # mostly empty sky on a tile-aligned frame: the streaming path of empty tiles (16 B per lane)
    H, W = 192, 128
    bands = synth.make_bands(H, W, 2)
    nelec = np.random.RandomState(4).poisson(250.0, size=(2, H, W)).astype(float)
    iset = cel.ImageSet(ctx, bands, H, W, nelec=nelec)
    radec = synth.pixel2equa(bands[0], np.array([[20.5, 30.25]]))
    sset = cel.SourceSet(ctx, 1, 2).set(np.zeros(1, np.int32), radec, np.full((1, 2), 3e4), np.tile([0.5, 2.0, 30.0, 0.5], (1, 1)))
    ll, llb = iset.render(sset, loglik=True)
    ob = bands.copy()
    ob[:, 36] = [orc.checked_radius(ob[b], iset.band(b)[36]) for b in range(2)]
    o_lam, o_ll, _ = orc.render_field(ob, H, W, np.zeros(1, np.int32), radec, np.full((1, 2), 3e4),
                                      np.tile([0.5, 2.0, 30.0, 0.5], (1, 1)), nelec)
    lam = iset.model_images()
    np.testing.assert_allclose(lam, o_lam, rtol=RT_LAM)
    np.testing.assert_allclose(llb, o_ll, rtol=RT_LL)
    assert np.all(lam[:, 128:, :] == bands[:, 0][:, None, None])        # untouched tiles hold eps exactly
    # sources off-image / on the border / failing the Q1 test / degenerate galaxy sizes
    H, W = 70, 130
    bands = synth.make_bands(H, W, 2)
    pix = np.array([[-30., 10.], [-60., 10.], [129.9, 69.9], [0., 0.], [500., 500.], [65., -20.], [64.5, 35.2],
                    [-200., 35.]])
    typ = np.array([0, 0, 1, 1, 1, 0, 1, 1], np.int32)
    radec = synth.pixel2equa(bands[0], pix)
    shape = np.tile([0.5, 2.0, 30.0, 0.5], (8, 1))
    shape[6, 1] = 1e-4          # below the 1/30 arcsec floor
    counts = np.full((8, 2), 1000.0)
    nelec = np.random.RandomState(1).poisson(300.0, size=(2, H, W)).astype(float)
    iset = cel.ImageSet(ctx, bands, H, W, nelec=nelec)
    sset = cel.SourceSet(ctx, 8, 2).set(typ, radec, counts, shape)
    ll, llb = iset.render(sset, loglik=True)
    ob = bands.copy()
    ob[:, 36] = [orc.checked_radius(ob[b], iset.band(b)[36]) for b in range(2)]
    o_lam, o_ll, _ = orc.render_field(ob, H, W, typ, radec, counts, shape, nelec)
    np.testing.assert_allclose(iset.model_images(), o_lam, rtol=RT_LAM)
    np.testing.assert_allclose(llb, o_ll, rtol=RT_LL)
    boxes, status = iset.stamp_boxes(sset, 0)
    # status: 1 stamp, 0 empty box, -1 the reference's overlap test fails (Q1: returns None)
    assert status[1] == -1 and status[4] == 0 and status[7] == 0


def test_sharp_psf_forces_direct_fallback_and_short_segments(cel, ctx, orc):
    """very narrow components (var 0.02..0.3 px^2): the recurrence must re-seed often or fall back"""
    from desi_mcmc_amd import synth
    H, W = 96, 128
    bands = synth.make_bands(H, W, 1)
    for scale in (0.15, 0.02):
        b = bands.copy()
        b[0, 12:24] *= scale
        b[0, 36] = 0.0
        typ = np.array([0, 1, 0], np.int32)
        radec = synth.pixel2equa(b[0], np.array([[40.3, 50.7], [90.2, 30.1], [64.0, 64.0]]))
        shape = np.tile([0.4, 1.0, 20.0, 0.7], (3, 1))
        counts = np.full((3, 1), 5e4)
        nelec = np.random.RandomState(2).poisson(500.0, size=(1, H, W)).astype(float)
        iset = cel.ImageSet(ctx, b, H, W, nelec=nelec)
        sset = cel.SourceSet(ctx, 3, 1).set(typ, radec, counts, shape)
        ll, llb = iset.render(sset, loglik=True)
        b[0, 36] = orc.checked_radius(b[0], iset.band(0)[36])
        o_lam, o_ll, _ = orc.render_field(b, H, W, typ, radec, counts, shape, nelec)
        np.testing.assert_allclose(iset.model_images(), o_lam, rtol=RT_LAM)
        np.testing.assert_allclose(llb, o_ll, rtol=RT_LL)


# ------------------------------------------------------------------------------------------
# BASELINE-size field: size-independent properties
# ------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def big_field(cel, ctx):
    from desi_mcmc_amd import synth
    return synth.SyntheticField.from_config(ctx, "mixed10k_2048")


def test_full_size_properties(cel, ctx, big_field):
    f = big_field
    ll1, llb1 = f.images.render(f.sources, loglik=True)
    lam1 = f.images.model_images()
    # (1) the fused reduction equals a host recomputation from the stored model pixels
    host = np.array([np.sum(f.nelec[b] * np.log(lam1[b]) - lam1[b]) for b in range(f.B)])
    np.testing.assert_allclose(llb1, host, rtol=1e-12)
    # (2) bitwise reproducible run to run (fixed-order lists and reduction)
    ll2, llb2 = f.images.render(f.sources, loglik=True)
    assert ll1 == ll2 and np.array_equal(llb1, llb2)
    assert np.array_equal(lam1, f.images.model_images())
    # (3) direct and recurrence evaluators agree
    ctx.set_kernel("direct")
    try:
        ll3, llb3 = f.images.render(f.sources, loglik=True)
        lam3 = f.images.model_images()
    finally:
        ctx.set_kernel("recurrence")
    np.testing.assert_allclose(lam1, lam3, rtol=RT_LAM)             # (the recurrence form at the shipping threshold; the direct form drops nothing)
    np.testing.assert_allclose(llb1, llb3, rtol=1e-11)
    # (4) linearity in the source set: (lam(A) - eps) + (lam(B) - eps) == lam(A u B) - eps
    half = f.S // 2
    eps = f.bands[:, 0][:, None, None]
    parts = []
    for sl in (slice(0, half), slice(half, f.S)):
        ss = cel.SourceSet(ctx, f.S, f.B).set(f.src["type"][sl], f.src["radec"][sl], f.src["counts"][sl],
                                              f.src["shape"][sl])
        f.images.render(ss)
        parts.append(f.images.model_images() - eps)
    np.testing.assert_allclose(parts[0] + parts[1], lam1 - eps, rtol=1e-9, atol=1e-9)
    # (5) flux conservation: the total model flux is the sum of counts x stamp mass inside the frame
    st = f.images.stats()
    assert st["n_srcpix"] > 1e8 and st["n_gauss"] > st["n_srcpix"]


def test_render_work_counters_of_the_benchmark_field(cel, ctx, big_field):
    """The work k_render_hw does on BASELINE configs[2] is a function of the data and the walk's rules, not of timing: the tile
    entries, the pairs of groups and the kept component-rows its counters report (CEL_OPT_TILE_TIMING, third word) are the same in
    every run, and stay where round 6 left them -- 1.555e8 component-rows of 32 columns at the shipping threshold (1.765e8 before
    the nested walk and the exact rows; the CPU model of tools/dbg/task_model.py).  A change that makes the kernel walk more rows
    again fails here, without a stopwatch."""
    f = big_field
    L = cel._lib
    ctx.set_option(L.CEL_OPT_TILE_TIMING, 1)
    try:
        got = []
        for _ in range(2):
            f.images.render(f.sources, loglik=True)
            tt = f.images.tile_timing()
            w = tt[:, 2]
            got.append((int(np.sum(w & np.uint64(0xfff))), int(np.sum((w >> np.uint64(12)) & np.uint64(0xfffff))), int(np.sum(w >> np.uint64(32)))))
    finally:
        ctx.set_option(L.CEL_OPT_TILE_TIMING, 0)
    assert got[0] == got[1]
    entries, pairs, comprows = got[0]
    assert entries == int(f.images.stats()["n_tile_entries"]) == 412326
    assert 6.1e5 < pairs < 6.3e5
    assert comprows < 1.60e8, comprows


def test_full_size_spot_check_vs_oracle(cel, ctx, orc, big_field):
    """a 256 x 192 window of the 2048^2 field against the oracle: only sources near the window"""
    f = big_field
    f.images.render(f.sources, loglik=False)
    lam = f.images.model_images()
    ob = oracle_bands(f)
    x0, y0, w, h = 900, 1100, 256, 192
    near = np.where((np.abs(f.src["pix"][:, 0] - (x0 + w / 2)) < w / 2 + 330) &
                    (np.abs(f.src["pix"][:, 1] - (y0 + h / 2)) < h / 2 + 330))[0]
    for b in (0, 2):
        o_lam, _, _ = orc.render_field(ob[b:b + 1], f.H, f.W, f.src["type"][near], f.src["radec"][near],
                                       f.src["counts"][near][:, b:b + 1], f.src["shape"][near])
        np.testing.assert_allclose(lam[b, y0:y0 + h, x0:x0 + w], o_lam[0, y0:y0 + h, x0:x0 + w], rtol=RT_LAM)


def test_star_field_full_size_star_tile_kernel(cel, ctx, orc):
    """10 000 stars x 5 bands x 2048^2 (bench workload stars10k_2048): a catalogue without galaxies on a frame of
    10 240 tiles takes k_render_stars by default.  Every model pixel against the oracle (1e-10), the per-band
    log-likelihoods (1e-11), the general kernel on the same input (rounding), run-to-run bit identity."""
    from desi_mcmc_amd import synth
    L = cel._lib
    f = synth.SyntheticField.from_config(ctx, "stars10k_2048")
    try:
        ctx.profile(True)
        ll1, llb1 = f.images.render(f.sources, loglik=True)
        assert ctx.profile_render()[1:] == (1, "k_render_stars")
        ctx.profile(False)
        lam1 = f.images.model_images()
        ll2, llb2 = f.images.render(f.sources, loglik=True)
        assert ll1 == ll2 and np.array_equal(llb1, llb2) and np.array_equal(lam1, f.images.model_images())
        ctx.set_option(L.CEL_OPT_STAR_TILES, 0)
        ctx.profile(True)
        ll0, llb0 = f.images.render(f.sources, loglik=True)
        assert ctx.profile_render()[1:] == (1, "k_render_hw")
        ctx.profile(False)
        np.testing.assert_allclose(f.images.model_images(), lam1, rtol=1e-13)
        np.testing.assert_allclose(llb0, llb1, rtol=1e-13)
        o_lam, o_ll, o_st = orc.render_field(oracle_bands(f), f.H, f.W, f.src["type"], f.src["radec"],
                                             f.src["counts"], f.src["shape"], f.nelec)
        np.testing.assert_allclose(lam1, o_lam, rtol=RT_LAM_STRICT)
        np.testing.assert_allclose(llb1, o_ll, rtol=RT_LL)
        assert f.images.stats()["n_srcpix"] == o_st["n_srcpix"]
    finally:
        ctx.set_option(L.CEL_OPT_STAR_TILES, 1)
        ctx.profile(False)


def test_config3_full_vs_oracle(cel, ctx, orc, big_field):
    """BASELINE configs[2] at full size against the CPU oracle over the WHOLE field: all 10 000
    sources, all 5 bands, every one of the 2.1e7 model pixels at 1e-10, per-band log-likelihoods at
    1e-11, the work counters exact; then the other two tile layouts on one band.  (The oracle needs
    ~1 min on 16 threads; set CEL_SKIP_FULL_ORACLE=1 to skip.)"""
    import os
    if os.environ.get("CEL_SKIP_FULL_ORACLE") == "1":
        pytest.skip("CEL_SKIP_FULL_ORACLE=1")
    f = big_field
    # the library's SHIPPING threshold first (T = 24 for the field render: a skipped component is below eps * e^-24 =
    # eps * 3.8e-11 on its tile): every pixel of the whole field within 1e-9 of the oracle (`north_star` states 1e-6), the
    # log-likelihoods at 1e-11 still
    assert ctx.get_option(cel._lib.CEL_OPT_TAIL_LOG) == cel._lib.TAIL_LOG_DEFAULT == 24.0
    ll24, llb24 = f.images.render(f.sources, loglik=True)
    lam24 = f.images.model_images()
    st = f.images.stats()
    ob = oracle_bands(f)
    try:
        orc.set_threads(min(orc.max_threads(), len(os.sched_getaffinity(0))))
    except AttributeError:
        pass
    o_lam, o_ll, o_st = orc.render_field(ob, f.H, f.W, f.src["type"], f.src["radec"], f.src["counts"], f.src["shape"], f.nelec)
    assert st["n_srcpix"] == o_st["n_srcpix"] and st["n_gauss"] == o_st["n_gauss"]
    worst = max(float(np.max(np.abs(lam24[b] / o_lam[b] - 1.0))) for b in range(f.B))
    assert worst < RT_LAM_DEFAULT, worst
    np.testing.assert_allclose(llb24, o_ll, rtol=RT_LL)
    np.testing.assert_allclose(ll24, o_ll.sum(), rtol=RT_LL)
    print("default threshold (T = 24): worst pixel %.2e relative, log-likelihood %.17g against %.17g" % (worst, ll24, o_ll.sum()))
    del lam24
    # ... then the strict one (T = 32 for every kernel): every pixel at 1e-10
    with tail_log(ctx, "strict"):
        assert ctx.get_option(cel._lib.CEL_OPT_TAIL_LOG) == cel._lib.TAIL_LOG_STRICT
        ll, llb = f.images.render(f.sources, loglik=True)
        lam = f.images.model_images()
    for b in range(f.B):
        np.testing.assert_allclose(lam[b], o_lam[b], rtol=RT_LAM_STRICT)
    np.testing.assert_allclose(llb, o_ll, rtol=RT_LL)
    np.testing.assert_allclose(ll, o_ll.sum(), rtol=RT_LL)
    del lam
    # the 64 x 32 and 16 x 128 tile layouts on the r band of the same field
    b = 2
    for layout in (0, 2):
        c2 = cel.Context(0)
        c2.set_option(cel._lib.CEL_OPT_TILE_LAYOUT, layout)
        iset = cel.ImageSet(c2, f.bands[b:b + 1], f.H, f.W, nelec=f.nelec[b:b + 1])
        ss = cel.SourceSet(c2, f.S, 1).set(f.src["type"], f.src["radec"], f.src["counts"][:, b:b + 1], f.src["shape"])
        l2, llb2 = iset.render(ss, loglik=True)
        np.testing.assert_allclose(iset.model_images()[0], o_lam[b], rtol=RT_LAM_DEFAULT)
        np.testing.assert_allclose(llb2[0], o_ll[b], rtol=RT_LL)
        del iset, ss, c2


def test_full_size_gibbs_kernels_properties(cel, ctx, big_field):
    """BASELINE-size field (10 000 sources x 5 bands x 2048^2), the per-source kernels, through
    properties that do not need the oracle: photon conservation of the split, the E-step's
    responsibilities summing to one, and the recurrence kernels against the direct ones."""
    f = big_field
    nel = f.nelec.reshape(f.B, -1).sum(axis=1)
    # (1) split: every observed photon goes to exactly one source or to the sky (integers: exact)
    noise = f.images.photon_split_resident(f.sources, seed=9)
    sums = f.images.sample_sums()
    assert np.array_equal(sums.sum(axis=0) + noise, nel)
    # same seed, same split (counter-based generator, fixed order)
    noise2 = f.images.photon_split_resident(f.sources, seed=9)
    assert np.array_equal(noise, noise2) and np.array_equal(sums, f.images.sample_sums())
    # the expected share of each source: E[sum z] = X~ of the E-step (celeste_em.py:85) -- the
    # totals over all sources agree to sampling noise, sqrt(N) on ~1e9 photons
    xt, mass, nz = f.images.estep_stats(f.sources)
    np.testing.assert_allclose(xt.sum(axis=0) + nz, nel, rtol=1e-9)          # (2) responsibilities sum to one
    wsum = f.bands[:, 3:6].sum(axis=1)                   # a stamp's mass is at most the PSF weights' sum
    assert np.all(mass <= wsum[None, :] * (1 + 1e-9)) and np.all(mass >= 0.0)
    # the split's strict boxes lose each source's first row/column, so its share is a little below X~
    tot_split, tot_e = sums.sum(axis=0), xt.sum(axis=0)
    assert np.all(tot_split <= tot_e * (1 + 1e-3)) and np.all(tot_split >= tot_e * 0.9)
    # (3) recurrence kernels == direct kernels on a sample of proposals / all sources
    rs = np.random.RandomState(5)
    pick = rs.choice(f.S, 300, replace=False).astype(np.int32)
    prop = cel.SourceSet(ctx, 300, f.B).set(f.src["type"][pick], f.src["radec"][pick] + rs.normal(0, 2e-5, (300, 2)),
                                            f.src["counts"][pick], f.src["shape"][pick])
    ll_r = f.images.patch_loglik_resident(prop, pick)
    iso_r = f.images.patch_loglik_resident(prop, pick, isolated=True)
    ctx.set_kernel("direct")
    try:
        ll_d = f.images.patch_loglik_resident(prop, pick)
        iso_d = f.images.patch_loglik_resident(prop, pick, isolated=True)
        xt_d, mass_d, nz_d = f.images.estep_stats(f.sources)
    finally:
        ctx.set_kernel("recurrence")
    np.testing.assert_allclose(ll_r, ll_d, rtol=RT_LL)
    np.testing.assert_allclose(iso_r, iso_d, rtol=RT_LL)
    # (X~ sums x F / lambda with lambda from the field render at its shipping threshold, T = 24: the drop rule's bound on a pixel is
    #  n_skipped * e^-24 = n * 3.8e-11, measured 5e-10 at worst over the field (test_config3_full_vs_oracle); the direct evaluator
    #  drops nothing.  Since round 6 a component's rows on a tile are exactly the integer rows inside its threshold ellipse --
    #  one fewer at either end than before -- and one of the 50 000 sums moved from inside 1e-10 to 1.07e-10)
    np.testing.assert_allclose(xt, xt_d, rtol=3e-10, atol=1e-12)
    np.testing.assert_allclose(mass, mass_d, rtol=1e-10)
    np.testing.assert_allclose(nz, nz_d, rtol=1e-10)                # (the sky term is nelec * eps / lambda summed: lambda at the render's threshold)


@pytest.mark.parametrize("world,frac_gal,tail", [(2, 0.5, "default"), (2, 0.5, "strict"), (3, 0.5, "default"), (3, 0.0, "default")])
def test_row_strips_tile_the_frame(cel, ctx, world, frac_gal, tail):
    """strong-scaling partition on ONE gpu: strips rendered through cel_images_set_window must
    reproduce the whole frame's model pixels, and their ll partials must add up.  frac_gal = 0: a star-only catalogue on a
    small frame -- frame and strips take the one-launch path (k_small_stars), whose blocks apply the window themselves"""
    from desi_mcmc_amd import dist, synth
    H, W = 200, 300
    f = synth.SyntheticField(ctx, 300, 2, H, W, frac_gal=frac_gal, seed=11)
    if tail == "strict":
        with tail_log(ctx, "strict"):
            _row_strips_body(cel, ctx, f, H, W, world, frac_gal, 1e-12)
    else:
        # strips whose first row is no multiple of the 64-row tile cut the frame into other tiles: the drop rule (a component
        # below eps * e^-24 on ITS tile) then skips other components -- 1e-9 at the shipping threshold, rounding at T = 32
        _row_strips_body(cel, ctx, f, H, W, world, frac_gal, 1e-12 if frac_gal == 0.0 else RT_LAM)


def test_measured_tile_costs_land_on_their_tiles_with_the_tile_order_on(cel, ctx):
    """cel_debug_tile_timing's row i is TILE i, whatever launch position the heaviest-first order (CEL_OPT_TILE_ORDER = 1, in force
    on frames of more than 2 048 tiles) gave its block: a frame of 2 560 tiles whose tile row 3 holds 500 galaxies on top of a
    sparse star background.  The per-row cost dist.strip_cost_from_tiles forms from the measured durations -- what `bench.py
    --scaling strong --strip-cut measured` cuts the strips from -- peaks on that row, the counters name its list lengths, and the
    cut evens out the cost (round 5 recorded by LAUNCH position: the per-row cost was scrambled and every measured cut came out
    equal)."""
    from desi_mcmc_amd import _lib, dist, synth
    B, H, W = 5, 512, 2048
    bands = synth.make_bands(H, W, B)
    rs = np.random.RandomState(3)
    S_bg, S_row = 400, 500
    pix = np.vstack([np.column_stack([rs.uniform(0, W, S_bg), rs.uniform(0, H, S_bg)]),
                     np.column_stack([rs.uniform(0, W, S_row), rs.uniform(200, 248, S_row)])])      # tile row 3 = rows 192..255
    typ = np.concatenate([np.zeros(S_bg, np.int32), np.ones(S_row, np.int32)])
    S = S_bg + S_row
    shape = np.column_stack([rs.uniform(0.2, 0.8, S), rs.uniform(0.5, 1.0, S), rs.uniform(0, 180, S), rs.uniform(0.4, 0.9, S)])
    counts = np.exp(rs.uniform(np.log(5e3), np.log(5e4), size=(S, B)))
    iset = cel.ImageSet(ctx, bands, H, W)
    sset = cel.SourceSet(ctx, S, B).set(typ, synth.pixel2equa(bands[0], pix), counts, shape)
    ntx, nty = W // 32, H // 64
    assert B * ntx * nty > 2048 and ctx.get_option(_lib.CEL_OPT_TILE_ORDER) == 1
    for _ in range(3):                                   # the order of a render is the durations of the one before
        iset.render(sset, loglik=False)
    lam = iset.model_images()
    ctx.set_option(_lib.CEL_OPT_TILE_TIMING, 1)
    try:
        iset.render(sset, loglik=False)
        tt = iset.tile_timing()
    finally:
        ctx.set_option(_lib.CEL_OPT_TILE_TIMING, 0)
    # (the timing instantiation keeps the arithmetic; it runs one wave per tile where this frame's renders use two parts per
    # tile, whose slabs add in another order: rounding)
    np.testing.assert_allclose(iset.model_images(), lam, rtol=1e-12)
    assert tt.shape == (B * ntx * nty, 3)
    cnt = (tt[:, 2] & np.uint64(0xfff)).astype(np.int64).reshape(B, nty, ntx)
    # the list lengths the counters carry are those of the tiles they are filed under: row 3 holds the galaxies in every band
    per_row = cnt.sum(axis=(0, 2))
    assert per_row.argmax() == 3 and per_row[3] > 4 * np.delete(per_row, [2, 3, 4]).max(), per_row
    dur = (tt[:, 1] - tt[:, 0]).astype(np.float64)
    cost = dist.strip_cost_from_tiles(dur, B, nty, ntx, 64, 64, H=H)
    assert cost.shape == (nty,) and cost.argmax() == 3 and cost[3] > 2 * np.delete(cost, [2, 3, 4]).max(), cost
    edges = dist.strip_edges(H, 4, cost, align=64)
    assert edges != dist.strip_edges(H, 4, align=64) and 192 in edges and 256 in edges, edges       # the heavy row is a strip of its own
    iset.render(sset, loglik=False)                       # and a DIAG render vouches for nothing: this one is complete
    assert np.array_equal(iset.model_images(), lam)


def _row_strips_body(cel, ctx, f, H, W, world, frac_gal, rt):
    from desi_mcmc_amd import dist
    ctx.profile(True)
    ll, llb = f.images.render(f.sources, loglik=True)
    assert ctx.profile_get("small_stars")[1] == (1 if frac_gal == 0.0 else 0)
    ctx.profile(False)
    lam = f.images.model_images()
    parts = np.zeros(2)
    for r in range(world):
        y0, y1 = dist.strip_rows(H, world, r)
        if y1 <= y0:
            continue
        strip = cel.ImageSet(ctx, f.bands, y1 - y0, W, nelec=f.nelec[:, y0:y1])
        strip.set_window(y0, H)
        _, p = strip.render(f.sources, loglik=True)
        # same pixels, same tiles, same source order; only the row origin of the fp arithmetic moves
        np.testing.assert_allclose(strip.model_images(), lam[:, y0:y1], rtol=rt)
        parts += p
    np.testing.assert_allclose(parts, llb, rtol=1e-13 if rt <= 1e-12 else 1e-11)
    with pytest.raises(ValueError):
        cel.ImageSet(ctx, f.bands, 64, W).set_window(150, H)       # window does not fit the frame


def test_incremental_render_is_the_full_render_bit_for_bit(cel, ctx, big_field):
    """CEL_OPT_INCREMENTAL: after cel_sources_set_rows changed a few rows of the catalogue an image set already holds the model
    image of, cel_render_field renders only the tiles the changed sources' old and new boxes touch.  Every such tile is
    rendered from its complete list, so model pixels, per-band log-likelihoods and work counters equal a full render of the
    same catalogue BIT FOR BIT -- a source moved within its tile, across tiles, off the frame and back, a flux change, a
    star <-> galaxy change, a galaxy grown to a box of hundreds of pixels, many rows at once; more than 64 changed rows, a
    whole-catalogue upload, a new sky level, another catalogue in between, a render without the log-likelihood before one
    with it, and the option switched off all fall back to rendering every tile; a render with NOTHING changed renders every
    tile too (nothing is answered from a cache).  On the BASELINE-size field (10 240 tiles) and on a small frame."""
    from desi_mcmc_amd import synth
    L = cel._lib
    small = synth.SyntheticField(ctx, 700, 3, 400, 300, frac_gal=0.5, seed=19)
    for f, force_parts in ((big_field, 0), (small, 1)):
        ctx.set_option(L.CEL_OPT_TILE_PARTS, force_parts)           # (a frame of few tiles takes the several-waves-per-tile kernel: every tile)
        try:
            S, B = f.S, f.B
            cur = {k: np.array(f.src[k], copy=True) for k in ("type", "radec", "counts", "shape")}
            sset = cel.SourceSet(ctx, S, B).set(cur["type"], cur["radec"], cur["counts"], cur["shape"])
            ref_set = cel.SourceSet(ctx, S, B)
            ref_img = cel.ImageSet(ctx, f.bands, f.H, f.W, nelec=f.nelec)
            img = cel.ImageSet(ctx, f.bands, f.H, f.W, nelec=f.nelec)
            rs = np.random.RandomState(3)

            def check(rows, expect_incremental, loglik=True):
                rows = np.asarray(rows, dtype=np.int32)
                if rows.size:
                    sset.set_rows(rows, cur["type"][rows], cur["radec"][rows], cur["counts"][rows], cur["shape"][rows])
                out = img.render(sset, loglik=loglik)
                dirty = img.last_render_dirty_tiles()
                assert (dirty >= 0) == expect_incremental, (dirty, expect_incremental, rows[:8])
                want = ref_img.render(ref_set.set(cur["type"], cur["radec"], cur["counts"], cur["shape"]), loglik=True)
                assert ref_img.last_render_dirty_tiles() == -1
                assert np.array_equal(img.model_images(), ref_img.model_images())
                if loglik:
                    assert np.array_equal(out[1], want[1]) and out[0] == want[0]
                    assert img.stats() == ref_img.stats()
                return dirty
            check([], False)                                         # the first render: every tile
            check([], False)                                         # nothing changed: every tile again, not a cached answer
            s0 = int(np.nonzero(cur["type"] == 1)[0][5])
            cur["radec"][s0] += [2e-5, -1e-5]                        # a galaxy moved by a fraction of a pixel
            d1 = check([s0], True)
            assert 0 < d1 < 0.2 * B * ((f.W + 31) // 32) * ((f.H + 63) // 64)
            cur["radec"][s0] = synth.pixel2equa(f.bands[0], np.array([[f.W * 0.2, f.H * 0.8]]))[0]     # across the frame: old and new tiles
            check([s0], True)
            s1 = int(np.nonzero(cur["type"] == 0)[0][7])
            cur["counts"][s1] *= 3.0                                 # a star's flux
            cur["type"][s1] = 1                                      # ... and it turns into a galaxy
            cur["shape"][s1] = [0.4, 1.3, 30.0, 0.6]
            check([s1], True)
            cur["radec"][s1] = synth.pixel2equa(f.bands[0], np.array([[-400.0, -300.0]]))[0]            # off the frame: an empty box now
            check([s1], True)
            cur["radec"][s1] = synth.pixel2equa(f.bands[0], np.array([[f.W / 2.0, f.H / 2.0]]))[0]      # and back
            cur["shape"][s1] = [0.1, 4.0, 120.0, 0.9]                # a 4-arcsec galaxy: a box hundreds of pixels wide
            check([s1], True)
            many = rs.choice(S, 40, replace=False)
            cur["radec"][many] += rs.normal(0, 3e-5, (40, 2))
            cur["counts"][many] *= np.exp(rs.normal(0, 0.2, (40, B)))
            check(many, True)
            some = rs.choice(S, 30, replace=False)                   # two batches before one render: 60 rows in all
            cur["counts"][some] *= 1.1
            sset.set_rows(some.astype(np.int32), cur["type"][some], cur["radec"][some], cur["counts"][some], cur["shape"][some])
            more = np.setdiff1d(rs.choice(S, 40, replace=False), some)[:30]
            cur["counts"][more] *= 0.9
            check(more, True)
            lots = rs.choice(S, 100, replace=False)                  # more rows than the path takes: every tile
            cur["counts"][lots] *= 1.05
            check(lots, False)
            cur["counts"][s0] *= 1.01
            check([s0], True)
            cur["counts"][s0] *= 1.01                                # the whole catalogue uploaded: every tile
            sset.set(cur["type"], cur["radec"], cur["counts"], cur["shape"])
            check([], False)
            cur["counts"][s0] *= 1.01
            img.set_epsilon(0, f.bands[0, 0] * 1.001)                # a new sky level: the image on the device is stale
            ref_img.set_epsilon(0, f.bands[0, 0] * 1.001)
            check([s0], False)
            cur["counts"][s0] *= 1.01                                # a render WITHOUT the log-likelihood does not refresh the partials ...
            check([s0], True, loglik=False)
            cur["counts"][s0] *= 1.01                                # ... so the next one with it renders every tile
            check([s0], False)
            cur["counts"][s0] *= 1.01                                # another catalogue rendered in between: every tile
            img.render(ref_set, loglik=True)
            check([s0], False)
            cur["counts"][s0] *= 1.01                                # new observed pixels: the kept tiles' Poisson partials are stale
            img.set_nelec(f.nelec)
            check([s0], False)
            cur["counts"][s0] *= 1.01                                # another drop threshold: the kept tiles were rendered at the old one
            with tail_log(ctx, "strict"):
                check([s0], False)
                cur["counts"][s0] *= 1.01
                check([s0], True)
            cur["counts"][s0] *= 1.01
            check([s0], False)
            cur["counts"][s0] *= 1.01
            ctx.set_option(L.CEL_OPT_INCREMENTAL, 0)
            try:
                check([s0], False)
            finally:
                ctx.set_option(L.CEL_OPT_INCREMENTAL, 1)
            cur["counts"][s0] *= 1.01
            check([s0], True)
            img.set_epsilon(0, f.bands[0, 0])
            ref_img.set_epsilon(0, f.bands[0, 0])
            # (found by tools/dbg/incremental_stress.py) a new sky level, then a render WITHOUT the log-likelihood while the
            # catalogue's generation has not moved, then an edit: the partials in the buffer are the old sky level's
            check([], False)
            img.set_epsilon(B - 1, f.bands[B - 1, 0] * 1.002)
            ref_img.set_epsilon(B - 1, f.bands[B - 1, 0] * 1.002)
            check([], False, loglik=False)
            cur["counts"][s0] *= 1.01
            check([s0], False)
            cur["counts"][s0] *= 1.01
            check([s0], True)
            with tail_log(ctx, "strict"):                            # the same with the drop threshold
                check([], False, loglik=False)
                cur["counts"][s0] *= 1.01
                check([s0], False)
            img.set_epsilon(B - 1, f.bands[B - 1, 0])
            ref_img.set_epsilon(B - 1, f.bands[B - 1, 0])
        finally:
            ctx.set_option(L.CEL_OPT_TILE_PARTS, 0)
    with pytest.raises(ValueError):
        ctx.set_option(L.CEL_OPT_INCREMENTAL, 2)


def test_owned_rows_restrict_the_log_likelihood(cel, ctx):
    """cel_images_set_noise_rows: an image set that holds a halo around the rows it OWNS (a rank's window of the
    strip-partitioned chain) renders the model image on every row and adds the Poisson terms of its own rows only -- the
    log-likelihood of an image set of exactly those rows; rows that do not begin and end on render tiles are refused."""
    from desi_mcmc_amd import synth
    f = synth.SyntheticField(ctx, 500, 3, 448, 300, frac_gal=0.5, seed=12)
    ll, llb = f.images.render(f.sources, loglik=True)
    lam = f.images.model_images()
    win = cel.ImageSet(ctx, f.bands, 320, f.W, nelec=f.nelec[:, 64:384])
    win.set_window(64, f.H)
    win.set_noise_rows(64, 256)                                  # frame rows [128, 320)
    _, llw = win.render(f.sources, loglik=True)
    np.testing.assert_allclose(win.model_images(), lam[:, 64:384], rtol=1e-12)
    strip = cel.ImageSet(ctx, f.bands, 192, f.W, nelec=f.nelec[:, 128:320])
    strip.set_window(128, f.H)
    _, lls = strip.render(f.sources, loglik=True)
    np.testing.assert_allclose(llw, lls, rtol=1e-13)
    want = np.array([np.sum(f.nelec[b, 128:320] * np.log(lam[b, 128:320]) - lam[b, 128:320]) for b in range(3)])
    np.testing.assert_allclose(llw, want, rtol=1e-12)
    win.set_noise_rows(0, 320)                                   # everything again
    np.testing.assert_allclose(win.render(f.sources, loglik=True)[1],
                               [np.sum(f.nelec[b, 64:384] * np.log(lam[b, 64:384]) - lam[b, 64:384]) for b in range(3)], rtol=1e-12)
    win.set_noise_rows(32, 256)
    with pytest.raises(ValueError):
        win.render(f.sources, loglik=True)
    win.render(f.sources, loglik=False)                          # (the model image alone does not care)
    win.set_noise_rows(64, 320)                                  # up to the set's last row: fine whatever its height
    win.render(f.sources, loglik=True)


@pytest.mark.parametrize("frac_gal", [0.6, 0.05])
def test_tile_parts_agree_and_each_is_reproducible(cel, ctx, orc, frac_gal):
    """CEL_OPT_TILE_PARTS: a frame of few tiles is rendered by 2 or 4 one-wave blocks per tile (k_render_hw<, PARTS>: every
    PARTS-th entry of the tile's list each, accumulator slabs added in part order by the last block to arrive).  Against the
    oracle at the strict threshold; 1, 2 and 4 parts agree to rounding; each is the same bits run after run (50 launches:
    whoever arrives last adds the slabs in the same order); a row window, a ragged frame, crowded tiles (more stars than one
    batch of 64, more sources than one index window), tiles without any source; the model image alone (no log-likelihood)."""
    from desi_mcmc_amd import synth
    L = cel._lib
    H, W, B, S = 333, 500, 3, 2600
    bands = synth.make_bands(H, W, B)
    src = synth.make_sources(S, H, W, bands, frac_gal=frac_gal, seed=21)
    src["radec"][:400] = synth.pixel2equa(bands[0], np.column_stack([np.random.RandomState(1).uniform(40, 90, 400),
                                                                    np.random.RandomState(2).uniform(100, 160, 400)]))    # a crowded corner
    nelec = np.random.RandomState(3).poisson(900.0, size=(B, H, W)).astype(float)
    got = {}
    try:
        for parts in (1, 2, 4):
            ctx.set_option(L.CEL_OPT_TILE_PARTS, parts)
            assert ctx.get_option(L.CEL_OPT_TILE_PARTS) == parts
            with tail_log(ctx, "strict"):
                iset = cel.ImageSet(ctx, bands, H, W, nelec=nelec)
                sset = cel.SourceSet(ctx, S, B).set(src["type"], src["radec"], src["counts"], src["shape"])
                ll, llb = iset.render(sset, loglik=True)
                lam = iset.model_images()
                for _ in range(50):
                    ll2, llb2 = iset.render(sset, loglik=True)
                    assert np.array_equal(llb, llb2)
                assert np.array_equal(lam, iset.model_images())
                iset.render(sset, loglik=False)                      # gen_model_image alone
                assert np.array_equal(lam, iset.model_images())
                win = cel.ImageSet(ctx, bands, 128, W, nelec=nelec[:, 64:192])
                win.set_window(64, H)
                _, llw = win.render(sset, loglik=True)
                got[parts] = (lam, llb, win.model_images(), llw)
        with pytest.raises(ValueError):
            ctx.set_option(L.CEL_OPT_TILE_PARTS, 3)
    finally:
        ctx.set_option(L.CEL_OPT_TILE_PARTS, 0)
    ob = bands.copy()
    ob[:, 36] = [orc.checked_radius(ob[b], iset.band(b)[36]) for b in range(B)]
    o_lam, o_ll, _ = orc.render_field(ob, H, W, src["type"], src["radec"], src["counts"], src["shape"], nelec)
    for parts in (1, 2, 4):
        lam, llb, wlam, llw = got[parts]
        np.testing.assert_allclose(lam, o_lam, rtol=RT_LAM_STRICT)
        np.testing.assert_allclose(llb, o_ll, rtol=RT_LL)
        np.testing.assert_allclose(lam, got[1][0], rtol=1e-13)
        np.testing.assert_allclose(wlam, got[1][2], rtol=1e-13)
        np.testing.assert_allclose(wlam, lam[:, 64:192], rtol=1e-12)     # (64-row aligned: the same tiles, the same drops)
        np.testing.assert_allclose(llw, got[1][3], rtol=1e-13)
    assert not np.array_equal(got[4][0], got[1][0])                 # the parts do change the order of the additions


def test_estep_statistics_and_model_classes(cel, orc):
    """E-step reductions (celeste_em.py:38-91) and the CelesteBase render / likelihood surface
    (models.py:88-108) on the mini field"""
    from desi_mcmc_amd import celeste, models, sources
    g = load_golden("mini_field.npz")
    e = load_golden("estep.npz")
    H, W = int(g["H"]), int(g["W"])
    imgs = frame_images(cel, {k: g[k] for k in g}, H, W, nelec=g["nelec"])
    idx = e["star_idx"]
    stars = [cel.SrcParams(u=g["radec"][s], a=0, fluxes=dict(zip(BANDS, g["flux"][s]))) for s in idx]
    X, F, Z = celeste.estep_statistics(stars, imgs)
    np.testing.assert_allclose(X, e["xtilde"], rtol=1e-10)
    np.testing.assert_allclose(F, np.minimum(1.0, e["mass"]), rtol=1e-10)
    np.testing.assert_allclose(Z, e["noise"], rtol=1e-11)
    # mixed star/galaxy list against the oracle, and conservation of photons
    srcs = [cel.SrcParams(u=g["radec"][s], a=int(g["is_gal"][s]), fluxes=g["flux"][s], theta=g["shape"][s, 0],
                          sigma=g["shape"][s, 1], phi=g["shape"][s, 2], rho=g["shape"][s, 3]) for s in range(12)]
    X, F, Z = celeste.estep_statistics(srcs, imgs)
    counts = g["flux"] / g["calib"][None, :] * g["kappa"][None, :]
    from desi_mcmc_amd import field
    ob = field.pack_bands(g)
    ob[:, 36] = [orc.checked_radius(ob[b], im.R) for b, im in enumerate(imgs)]
    oxt, oms, onz = orc.estep_stats(ob, H, W, g["is_gal"], g["radec"], counts, g["shape"], g["nelec"])
    np.testing.assert_allclose(X, oxt, rtol=1e-10)
    np.testing.assert_allclose(F, np.minimum(1.0, oms), rtol=1e-10)
    np.testing.assert_allclose(X.sum(axis=0) + Z, g["nelec"].sum(axis=(1, 2)), rtol=1e-10)      # (stamps at T = 32 over a model image at T = 24)
    # model classes: render_model_image / img_log_likelihood / log_likelihood
    m = models.Celeste()
    m.initialize_sources(init_src_params=srcs)
    assert isinstance(m.srcs[0], sources.Source) and list(m.source_types[:2]) == ["star", "galaxy"]
    np.testing.assert_allclose(m.render_model_image(imgs[2]), g["lam"][2], rtol=RT_LAM)
    np.testing.assert_allclose(m.img_log_likelihood(imgs[1]), g["ll_band"][1], rtol=RT_LL)
    np.testing.assert_allclose(m.img_log_likelihood(imgs[1], mod_img=g["lam"][1]), g["ll_band"][1], rtol=1e-12)
    # excluding a source removes exactly its patch
    full = m.render_model_image(imgs[2])
    wo = m.render_model_image(imgs[2], exclude=m.srcs[3])
    p, yl, xl = m.srcs[3].compute_model_patch(imgs[2])
    diff = full - wo
    # (a difference of two field renders: each within RT_LAM of ITS pixels' values)
    np.testing.assert_allclose(diff[int(yl[0]):int(yl[1]), int(xl[0]):int(xl[1])], p, rtol=1e-9, atol=2 * RT_LAM * float(full.max()))
    # caller-imposed limits crop to the box (models.py:99-100)
    sub = m.render_model_image(imgs[2], xlim=(10, 60), ylim=(5, 50))
    assert sub.shape == (45, 50)
    one = models.Celeste()
    one.initialize_sources(init_src_params=[srcs[0]])
    # with imposed limits the source is evaluated on the WHOLE box (compute_model_patch gets the
    # limits, models.py:96), i.e. also beyond its own bounding box: equal inside it, >= outside
    lim = one.render_model_image(imgs[2], xlim=(10, 60), ylim=(5, 50))
    own = one.render_model_image(imgs[2])[5:50, 10:60]
    _, yl0, xl0 = one.srcs[0].compute_model_patch(imgs[2])
    ys = slice(max(int(yl0[0]), 5) - 5, min(int(yl0[1]), 50) - 5)
    xs = slice(max(int(xl0[0]), 10) - 10, min(int(xl0[1]), 60) - 10)
    np.testing.assert_allclose(lim[ys, xs], own[ys, xs], rtol=1e-9)
    assert np.all(lim >= own * (1 - 1e-12)) and np.max(lim / own) < 1 + 1e-5
    # add_field sets epsilon to the median (models.py:115-117) and log_likelihood sums the field
    old = [im.epsilon for im in imgs]
    m.add_field(dict(zip(BANDS, imgs)))
    assert imgs[0].epsilon == np.median(imgs[0].nelec)
    ll = m.log_likelihood()
    np.testing.assert_allclose(ll, sum(m.img_log_likelihood(im) for im in imgs), rtol=1e-12)
    for im, o in zip(imgs, old):
        im.epsilon = o


@pytest.mark.parametrize("tail", ["default", "strict"])
def test_crowded_field_exercises_list_chunking_and_regrowth(cel, ctx, orc, tail):
    """8 000 stars + 300 galaxies on 320 x 448 x 2 bands: > 1024 candidates per super-tile (the fine binning
    pass streams them through LDS in chunks), > 64 sources per render tile (the render kernel reloads
    its index window), and tile lists that outgrow their first allocation (overflow -> regrow -> rerun)"""
    from desi_mcmc_amd import synth
    H, W, S = 320, 448, 8300
    bands = synth.make_bands(H, W, 2)
    src = synth.make_sources(S, H, W, bands, frac_gal=0.0, seed=3)
    src["type"][:300] = 1
    rs = np.random.RandomState(4)
    nelec = rs.poisson(2000.0, size=(2, H, W)).astype(float)
    for layout in (1, 0, 2):
        c2 = cel.Context(0)
        c2.set_tail_log(tail)
        c2.set_option(cel._lib.CEL_OPT_TILE_LAYOUT, layout)
        iset = cel.ImageSet(c2, bands, H, W, nelec=nelec)
        # a 10-source render first: the list buffers get sized for it (10*2*6 + 1024 entries) ...
        small = cel.SourceSet(c2, 16, 2).set(src["type"][:10], src["radec"][:10], src["counts"][:10], src["shape"][:10])
        iset.render(small, loglik=True)
        # ... and the crowded catalogue must overflow them, be detected on the device, and rerun
        sset = cel.SourceSet(c2, S, 2).set(src["type"], src["radec"], src["counts"], src["shape"])
        ll, llb = iset.render(sset, loglik=True)
        st = iset.stats()
        assert st["n_tile_entries"] > 20 * (10 * 2 * 6 + 1024)
        lam = iset.model_images()
        ll2, llb2 = iset.render(sset, loglik=True)                # steady state: same answer, bit for bit
        assert np.array_equal(llb, llb2) and np.array_equal(lam, iset.model_images())
        if layout == 1:
            ob = bands.copy()
            ob[:, 36] = [orc.checked_radius(ob[b], iset.band(b)[36]) for b in range(2)]
            o_lam, o_ll, o_st = orc.render_field(ob, H, W, src["type"], src["radec"], src["counts"], src["shape"], nelec)
            assert st["n_srcpix"] == o_st["n_srcpix"]
            lam1, llb1 = lam, llb
        np.testing.assert_allclose(lam, o_lam, rtol=RT_LAM_STRICT if tail == "strict" else RT_LAM)
        np.testing.assert_allclose(llb, o_ll, rtol=RT_LL)
    np.testing.assert_allclose(lam, lam1, rtol=1e-11 if tail == "strict" else RT_LAM)


def test_abi_error_paths(cel, ctx):
    """bad arguments come back as ValueError (CEL_ERR_INVALID), never as a crash"""
    from desi_mcmc_amd import synth
    bands = synth.make_bands(64, 64, 2)
    iset = cel.ImageSet(ctx, bands, 64, 64)
    sset = cel.SourceSet(ctx, 4, 2).set(np.zeros(1, np.int32), np.zeros((1, 2)), np.ones((1, 2)), np.zeros((1, 4)))
    with pytest.raises(ValueError, match="set_nelec"):
        iset.render(sset, loglik=True)                          # log-lik before nelec was uploaded
    with pytest.raises(ValueError):
        iset.set_nelec(np.zeros((2, 64, 63)))
    with pytest.raises(ValueError):
        cel.SourceSet(ctx, 2, 2).set(np.zeros(3, np.int32), np.zeros((3, 2)), np.ones((3, 2)), np.zeros((3, 4)))
    with pytest.raises(ValueError, match="bands"):
        iset.render(cel.SourceSet(ctx, 4, 3).set(np.zeros(1, np.int32), np.zeros((1, 2)), np.ones((1, 3)), None))
    bad = bands.copy()
    bad[0, 12:16] = [1.0, 2.0, 2.0, 1.0]                         # PSF covariance not positive definite
    with pytest.raises(ValueError, match="positive definite"):
        cel.ImageSet(ctx, bad, 64, 64)
    with pytest.raises(ValueError):
        cel.ImageSet(ctx, bands, 0, 64)
    with pytest.raises(ValueError):
        ctx.set_option(1, 7.0)
    with pytest.raises(ValueError):
        ctx.set_tail_log(-1.0)
    from desi_mcmc_amd import _lib
    for bits in (1, 2, 4, 8, 16, 32, 6):                         # timing-only ablations: not in the shipped library
        with pytest.raises(ValueError, match="CEL_ABLATE"):
            ctx.set_option(_lib.CEL_OPT_DEBUG, bits)
    assert ctx.get_option(_lib.CEL_OPT_DEBUG) == 0
    with pytest.raises(ValueError):
        iset.patch_loglik(sset, np.array([[0, 10, 0, 10], [0, 0, 0, 0]]), [np.zeros((9, 10)), None])
    iset.set_nelec(np.ones((2, 64, 64)))
    # a caller's patch layout is checked before anything is written through it (raw ABI: the Python wrappers build it themselves)
    import ctypes as C
    Lb = _lib.lib()
    star = cel.SourceSet(ctx, 4, 2).set(np.zeros(1, np.int32), synth.pixel2equa(bands[0], np.array([[30.0, 30.0]])), np.full((1, 2), 100.0), np.zeros((1, 4)))
    bx, st = iset.source_boxes(star)
    area = [int((bx[b, 0, 1] - bx[b, 0, 0]) * (bx[b, 0, 3] - bx[b, 0, 2])) for b in range(2)]
    buf = np.zeros(sum(area) + 8)
    noise = np.zeros(2)
    good = np.array([0, area[0], area[0] + area[1]], dtype=np.int64)
    assert Lb.cel_photon_split(iset._h, star._h, C.c_uint64(1), good.ctypes.data_as(_lib.c_int64_p), buf.ctypes.data, _lib.CEL_HOST, _lib.dptr(noise)) == 0
    for wrong in (good + 4, np.array([0, area[0] - 1, area[0] + area[1] - 1]), np.array([0, area[0], area[0] + area[1] - 3])):
        w = np.ascontiguousarray(wrong, dtype=np.int64)
        assert Lb.cel_photon_split(iset._h, star._h, C.c_uint64(1), w.ctypes.data_as(_lib.c_int64_p), buf.ctypes.data, _lib.CEL_HOST,
                                   _lib.dptr(noise)) == _lib.CEL_ERR_INVALID
    boxes = np.array([0, 4, 0, 4, 0, 0, 0, 0], dtype=np.int32)
    data = np.zeros(32)
    ll = np.zeros(1)
    for offs, want in ((np.array([0, 16, 16]), 0), (np.array([3, 19, 19]), _lib.CEL_ERR_INVALID), (np.array([0, 15, 15]), _lib.CEL_ERR_INVALID)):
        o = np.ascontiguousarray(offs, dtype=np.int64)
        assert Lb.cel_patch_loglik(iset._h, star._h, boxes.ctypes.data_as(_lib.c_int32_p), o.ctypes.data_as(_lib.c_int64_p), _lib.dptr(data), _lib.CEL_HOST, 0,
                                   _lib.dptr(ll)) == want, offs
    sb = np.array([0, 4, 0, 4], dtype=np.int32)
    out = np.zeros(32)
    for offs, want in ((np.array([0, 16]), 0), (np.array([2, 18]), _lib.CEL_ERR_INVALID), (np.array([0, 12]), _lib.CEL_ERR_INVALID)):
        o = np.ascontiguousarray(offs, dtype=np.int64)
        assert Lb.cel_render_stamps(iset._h, star._h, 0, 0, sb.ctypes.data_as(_lib.c_int32_p), o.ctypes.data_as(_lib.c_int64_p), _lib.dptr(out), _lib.CEL_HOST) == want, offs
    with pytest.raises(ValueError, match="outside"):
        iset.patch_loglik(sset, np.array([[0, 70, 0, 10], [0, 0, 0, 0]]), [np.zeros((70, 10)), None])
    # NaN source parameters contribute nothing instead of poisoning the field
    nan_src = cel.SourceSet(ctx, 4, 2).set(np.array([0, 1], np.int32), np.full((2, 2), np.nan), np.ones((2, 2)),
                                           np.tile([0.5, 1.0, 0.0, 0.5], (2, 1)))
    ll, llb = iset.render(nan_src, loglik=True)
    assert np.all(iset.model_images() == bands[:, 0][:, None, None]) and np.isfinite(ll)


def test_binomial_sampler_distribution(cel, ctx):
    """the split's Binomial(n, p) sampler (inversion / BTPE) against the exact pmf: the reference's
    randomkit stream cannot be reproduced, so parity is statistical"""
    import ctypes as C
    from scipy import stats
    from desi_mcmc_amd import _lib
    N = 200000
    cases = [(10, 0.3), (1000, 0.01), (200, 0.5), (5000, 0.37), (100000, 0.9), (50, 0.999), (3000, 0.011),
             (40, 0.75), (123456, 0.5), (914, 6.29e-16)]      # the last: test_celeste_sample_sources.py:6
    for i, (n, p) in enumerate(cases):
        out = np.zeros(N, dtype=np.int64)
        _lib.check(_lib.lib().cel_debug_binomial(ctx._h, n, p, 1234 + i, N, out.ctypes.data_as(_lib.c_int64_p)))
        assert out.min() >= 0 and out.max() <= n
        mean, var = n * p, n * p * (1 - p)
        if var < 1e-9:
            assert np.all(out == round(mean))
            continue
        assert abs(out.mean() - mean) < 5 * np.sqrt(var / N), (n, p, out.mean(), mean)
        assert abs(out.var() / var - 1) < 0.03, (n, p, out.var(), var)
        # chi-square against the exact pmf on cells with expectation >= 10
        lo, hi = int(max(0, mean - 6 * np.sqrt(var))), int(min(n, mean + 6 * np.sqrt(var)))
        ks = np.arange(lo, hi + 1)
        pmf = stats.binom.pmf(ks, n, p)
        obs = np.bincount(np.clip(out, lo, hi) - lo, minlength=len(ks)).astype(float)
        exp = pmf * N
        exp[0] += stats.binom.cdf(lo - 1, n, p) * N
        exp[-1] += stats.binom.sf(hi, n, p) * N
        keep = exp >= 10
        chi2 = np.sum((obs[keep] - exp[keep]) ** 2 / exp[keep]) + (obs[~keep].sum() - exp[~keep].sum()) ** 2 / max(exp[~keep].sum(), 1)
        dof = keep.sum()
        assert stats.chi2.sf(chi2, dof) > 1e-6, (n, p, chi2, dof)
    # deterministic in the seed
    a, b = np.zeros(1000, np.int64), np.zeros(1000, np.int64)
    _lib.check(_lib.lib().cel_debug_binomial(ctx._h, 500, 0.2, 7, 1000, a.ctypes.data_as(_lib.c_int64_p)))
    _lib.check(_lib.lib().cel_debug_binomial(ctx._h, 500, 0.2, 7, 1000, b.ctypes.data_as(_lib.c_int64_p)))
    assert np.array_equal(a, b)


def test_photon_split_conservation_moments_quirks(cel, ctx, orc):
    """sample_source_counts (celeste_sample_sources.pyx:61-156): exact photon conservation, the
    strict-box quirk, first moments against nelec * F_s / lambda, seed determinism, and
    independence of the draws from the tile layout"""
    from desi_mcmc_amd import field
    g = load_golden("mini_field.npz")
    H, W = int(g["H"]), int(g["W"])
    bands = field.pack_bands(g)
    counts = g["flux"] / g["calib"][None, :] * g["kappa"][None, :] * 20.0      # bright: many photons per pixel
    S = 12
    results = {}
    for layout in (1, 0, 2):
        c2 = cel.Context(0)
        c2.set_option(cel._lib.CEL_OPT_TILE_LAYOUT, layout)
        iset = cel.ImageSet(c2, bands, H, W, nelec=g["nelec"])
        sset = cel.SourceSet(c2, S, 5).set(g["is_gal"], g["radec"], counts, g["shape"])
        patches, boxes, noise = iset.photon_split(sset, seed=99)
        results[layout] = (patches, boxes, noise)
        if layout == 1:
            p2, _, n2 = iset.photon_split(sset, seed=99)
            p3, _, n3 = iset.photon_split(sset, seed=100)
            assert all(np.array_equal(a, b) for ra, rb in zip(patches, p2) for a, b in zip(ra, rb) if a is not None)
            assert np.array_equal(noise, n2)
            assert any(not np.array_equal(a, b) for ra, rb in zip(patches, p3) for a, b in zip(ra, rb) if a is not None)
            ob = bands.copy()
            ob[:, 36] = [orc.checked_radius(ob[b], iset.band(b)[36]) for b in range(5)]
    pa, ba, na = results[1]
    pb, bb, nb = results[0]
    assert np.array_equal(ba, bb) and np.array_equal(na, nb)                     # same draws whatever the tiling
    assert all(np.array_equal(a, b) for ra, rb in zip(pa, pb) for a, b in zip(ra, rb) if a is not None)
    pc, bc, nc = results[2]
    assert np.array_equal(ba, bc) and np.array_equal(na, nc)
    assert all(np.array_equal(a, b) for ra, rb in zip(pa, pc) for a, b in zip(ra, rb) if a is not None)
    patches, boxes, noise = results[1]
    for b in range(5):
        tot = noise[b]
        lam_strict = np.full((H, W), bands[b, 0])
        F = []
        for s in range(S):
            p, yl, xl = orc.source_patch(ob[b], H, W, g["is_gal"][s], g["radec"][s], g["shape"][s])
            if p is None:
                F.append(None)
                assert patches[b][s] is None
                continue
            assert (yl[0], yl[1], xl[0], xl[1]) == tuple(boxes[b, s])
            f = p * counts[s, b]
            f[0, :] = 0.0                # strict box: y > y0
            f[:, 0] = 0.0                #             x > x0   (celeste_sample_sources.pyx:50-51)
            lam_strict[yl[0]:yl[1], xl[0]:xl[1]] += f
            F.append((f, yl, xl))
        for s in range(S):
            if F[s] is None:
                continue
            f, yl, xl = F[s]
            z = patches[b][s]
            assert z.shape == f.shape and np.all(z == np.round(z)) and z.min() >= 0
            assert np.all(z[0, :] == 0) and np.all(z[:, 0] == 0)                  # the quirk
            tot += z.sum()
            n = g["nelec"][b, yl[0]:yl[1], xl[0]:xl[1]]
            pr = f / lam_strict[yl[0]:yl[1], xl[0]:xl[1]]
            mean, var = n * pr, n * pr * (1 - pr)
            zscore = (z.sum() - mean.sum()) / np.sqrt(var.sum())
            assert abs(zscore) < 5.0, (b, s, zscore)
            # per-pixel dispersion: sum of squared standardised residuals ~ chi2(npix)
            ok = var > 5
            if ok.sum() > 50:
                r2 = ((z[ok] - mean[ok]) ** 2 / var[ok]).sum()
                assert abs(r2 - ok.sum()) / np.sqrt(2.0 * ok.sum()) < 5.0, (b, s, r2 / ok.sum())
        assert tot == g["nelec"][b].sum()                                         # every photon lands exactly once
    # The tail regime -- the draws the first test settles on a 32-bit word shared by four rows of a column (k_split.h,
    # split_first_word): among the pixels with 1e-4 < n p < 0.3 the number holding a photon against its exact expectation
    # sum 1 - (1 - p)^n, per band and over all bands (a word handed to the wrong row, or an interval of the sampler's first
    # uniform that does not join up with the test, shows here long before it moves a source's total)
    got_all = exp_all = var_all = 0.0
    for b in range(5):
        lam_strict = np.full((H, W), bands[b, 0])
        Fs = []
        for s in range(S):
            p, yl, xl = orc.source_patch(ob[b], H, W, g["is_gal"][s], g["radec"][s], g["shape"][s])
            if p is None:
                Fs.append(None)
                continue
            f = p * counts[s, b]
            f[0, :] = 0.0
            f[:, 0] = 0.0
            lam_strict[yl[0]:yl[1], xl[0]:xl[1]] += f
            Fs.append((f, yl, xl))
        # the conditional-binomial chain: source s at a pixel sees what the sources before it left -- in the tail regime
        # those took next to nothing, so n = nelec and p = F_s / (total - the earlier sources' F) to first order in p
        taken = np.zeros((H, W))
        for s in range(S):
            if Fs[s] is None:
                continue
            f, yl, xl = Fs[s]
            sl = (slice(yl[0], yl[1]), slice(xl[0], xl[1]))
            z = patches[b][s]
            n = np.floor(g["nelec"][b][sl])
            pr = f / (lam_strict[sl] - taken[sl])
            taken[sl] += f
            sel = (n * pr > 1e-4) & (n * pr < 0.3) & (f > 0)
            # photons the earlier sources took at these pixels change n by a few parts in 10^3 at most here
            q = 1.0 - (1.0 - pr[sel]) ** n[sel]
            got_all += float((z[sel] > 0).sum()); exp_all += float(q.sum()); var_all += float((q * (1 - q)).sum())
    assert exp_all > 300, exp_all
    assert abs(got_all - exp_all) < 5.0 * np.sqrt(var_all) + 0.01 * exp_all, (got_all, exp_all, np.sqrt(var_all))
    # the direct kernel (every pixel's whole draw in one go, split_draw) makes the same draws as the two-pass recurrence kernel,
    # but for a pixel whose probability differs in the last bits between the two stamp evaluators
    c3 = cel.Context(0)
    c3.set_kernel("direct")
    iset = cel.ImageSet(c3, bands, H, W, nelec=g["nelec"])
    sset = cel.SourceSet(c3, S, 5).set(g["is_gal"], g["radec"], counts, g["shape"])
    pd, bd_, nd = iset.photon_split(sset, seed=99)
    differ = total = 0
    for rb_, ra_ in zip(pd, patches):
        for x_, y_ in zip(rb_, ra_):
            if x_ is not None:
                differ += int((x_ != y_).sum()); total += x_.size
    assert differ <= 1e-4 * total, (differ, total)


def test_photon_split_image_range_instantiations(cel, ctx):
    """The split keeps its photons-left plane in 16 bits while every observed pixel holds 0 ... 65 535 photons
    (cel_images_set_nelec finds the range) and in 32 bits otherwise: the two instantiations make the same draws (a draw is
    keyed by seed, pixel and source), a 70 000-photon pixel or a negative one is conserved exactly, and handing out the
    image's device pointer (the caller may write it) switches the assumption off."""
    from desi_mcmc_amd import field
    g = load_golden("mini_field.npz")
    H, W, S = int(g["H"]), int(g["W"]), 12
    bands = field.pack_bands(g)
    counts = g["flux"] / g["calib"][None, :] * g["kappa"][None, :] * 20.0
    nelec = np.floor(g["nelec"]).astype(np.float64)
    assert nelec.min() >= 0 and nelec.max() <= 65535
    iset = cel.ImageSet(ctx, bands, H, W, nelec=nelec)
    sset = cel.SourceSet(ctx, S, 5).set(g["is_gal"], g["radec"], counts, g["shape"])
    p16, boxes, n16 = iset.photon_split(sset, seed=7)                 # 16-bit plane

    def same(pa, pb, skip=None):
        for b in range(5):
            for s in range(S):
                a_, b_ = pa[b][s], pb[b][s]
                if a_ is None:
                    assert b_ is None
                    continue
                if skip is not None and skip[0] == b:
                    y0, y1, x0, x1 = boxes[b, s]
                    yy, xx = skip[1] - y0, skip[2] - x0
                    if 0 <= yy < a_.shape[0] and 0 <= xx < a_.shape[1]:
                        a_, b_ = a_.copy(), b_.copy()
                        a_[yy, xx] = b_[yy, xx] = 0
                assert np.array_equal(a_, b_)

    iset.device_ptrs()                                                # the caller may write nelec now: 32-bit plane
    p32, _, n32 = iset.photon_split(sset, seed=7)
    same(p16, p32)
    assert np.array_equal(n16, n32)
    # one pixel far beyond 16 bits, inside source 0's box of band 2; another one negative (sky-subtracted data), uncovered or not
    y0, y1, x0, x1 = boxes[2, 0]
    py, px = (y0 + y1) // 2, (x0 + x1) // 2
    wide = nelec.copy()
    wide[2, py, px] = 70000.0
    wide[4, 3, 5] = -4.0
    iset.set_nelec(wide)
    pw, bw, nw = iset.photon_split(sset, seed=7)
    assert np.array_equal(bw, boxes)
    for b in range(5):
        tot = sum(pw[b][s].sum() for s in range(S) if pw[b][s] is not None) + nw[b]
        assert tot == wide[b].sum()
    same(p16, pw, skip=(2, py, px))       # every other pixel draws what it drew before (the negative one has no photons to split)
    got = sum(pw[2][s][py - boxes[2, s, 0], px - boxes[2, s, 2]] for s in range(S)
              if pw[2][s] is not None and boxes[2, s, 0] < py < boxes[2, s, 1] and boxes[2, s, 2] < px < boxes[2, s, 3])
    assert 0 < got <= 70000
    iset.set_nelec(nelec)                                             # back inside the range: the 16-bit plane again
    p16b, _, n16b = iset.photon_split(sset, seed=7)
    same(p16, p16b)
    assert np.array_equal(n16, n16b)


def test_resident_split_and_loglik_equal_host_buffer_forms(cel, ctx):
    """device-resident sample patches (cel_photon_split with offsets = NULL, resident
    cel_patch_loglik_multi) give exactly what the host-buffer forms give for the same seed"""
    from desi_mcmc_amd import field
    g = load_golden("mini_field.npz")
    H, W, S = int(g["H"]), int(g["W"]), 12
    bands = field.pack_bands(g)
    counts = g["flux"] / g["calib"][None, :] * g["kappa"][None, :] * 5.0
    iset = cel.ImageSet(ctx, bands, H, W, nelec=g["nelec"])
    sset = cel.SourceSet(ctx, S, 5).set(g["is_gal"], g["radec"], counts, g["shape"])
    patches, boxes, noise = iset.photon_split(sset, seed=21)
    noise_r = iset.photon_split_resident(sset, seed=21)
    assert np.array_equal(noise, noise_r)
    rb, roff, rdata = iset.fetch_samples()
    sums = iset.sample_sums()
    for s in range(S):
        for b in range(5):
            i = s * 5 + b
            if patches[b][s] is None:
                assert roff[i + 1] == roff[i] and sums[s, b] == 0
                continue
            assert tuple(rb[s, b]) == tuple(boxes[b, s])
            assert np.array_equal(rdata[roff[i]:roff[i + 1]].reshape(patches[b][s].shape), patches[b][s])
            assert sums[s, b] == patches[b][s].sum()
    # proposals of several sources against the resident patches == against host copies of them
    rs = np.random.RandomState(8)
    P = 4
    own = np.repeat(np.arange(S, dtype=np.int32), P)
    prop = cel.SourceSet(ctx, S * P, 5).set(np.repeat(g["is_gal"], P), np.repeat(g["radec"], P, axis=0) +
                                            rs.normal(0, 3e-5, size=(S * P, 2)), np.repeat(counts, P, axis=0),
                                            np.repeat(g["shape"], P, axis=0))
    ll_res = iset.patch_loglik_resident(prop, own)
    host_boxes = np.transpose(boxes, (1, 0, 2))
    host_patches = [[patches[b][s] for b in range(5)] for s in range(S)]
    for s in range(S):
        for b in range(5):
            if host_patches[s][b] is None:
                host_boxes[s, b] = 0
    ll_host = iset.patch_loglik_multi(prop, own, host_boxes, host_patches)
    # the resident form reads each patch at its photons or densely, whichever is cheaper (CEL_OPT_PHOTON_LISTS): equal to
    # the host-buffer form (always dense) to rounding, and bit for bit when the lists are switched off
    np.testing.assert_allclose(ll_res, ll_host, rtol=1e-12)
    from desi_mcmc_amd import _lib
    for mode in (1, 2, 0):
        ctx.set_option(_lib.CEL_OPT_PHOTON_LISTS, mode)
        try:
            iset.photon_split_resident(sset, seed=21)
            ll_m = iset.patch_loglik_resident(prop, own)
        finally:
            ctx.set_option(_lib.CEL_OPT_PHOTON_LISTS, 0)
        if mode == 2:
            assert np.array_equal(ll_m, ll_host)
        else:
            np.testing.assert_allclose(ll_m, ll_host, rtol=1e-12)
        if mode == 0:
            assert np.array_equal(ll_m, ll_res)
    iso_res = iset.patch_loglik_resident(prop, own, isolated=True)
    iso_host = iset.patch_loglik_multi(prop, own, host_boxes,
                                       [[None if host_patches[s][b] is None else
                                         g["nelec"][b, host_boxes[s, b, 0]:host_boxes[s, b, 1], host_boxes[s, b, 2]:host_boxes[s, b, 3]]
                                         for b in range(5)] for s in range(S)], isolated=True)
    assert np.array_equal(iso_res, iso_host)
    with pytest.raises(ValueError, match="resident"):
        cel.ImageSet(ctx, bands, H, W, nelec=g["nelec"]).patch_loglik_resident(prop, own)


def test_field_resample_photons_feeds_source_loglik(cel):
    """Field.resample_photons (models.py:123-160) -> Source.log_likelihood on the sampled patches"""
    from desi_mcmc_amd import models
    g = load_golden("mini_field.npz")
    H, W = int(g["H"]), int(g["W"])
    imgs = frame_images(cel, {k: g[k] for k in g}, H, W, nelec=g["nelec"])
    srcs = [cel.SrcParams(u=g["radec"][s], a=int(g["is_gal"][s]), fluxes=g["flux"][s], theta=g["shape"][s, 0],
                          sigma=g["shape"][s, 1], phi=g["shape"][s, 2], rho=g["shape"][s, 3]) for s in range(12)]
    m = models.Celeste()
    m.initialize_sources(init_src_params=srcs)
    old = [im.epsilon for im in imgs]
    m.add_field(dict(zip(BANDS, imgs)))
    for im, o in zip(imgs, old):
        im.epsilon = o                                   # keep the sky level the photons were drawn with
    noise = m.field_list[0].resample_photons(m.srcs, seed=5, rng=np.random.RandomState(0))
    assert set(noise) == set(BANDS)
    n_samp = [len(s.sample_image_list) for s in m.srcs]
    assert max(n_samp) == 5 and min(n_samp) >= 1
    for im, o in zip(imgs, old):
        assert im.epsilon > 0 and abs(im.epsilon / o - 1) < 0.2           # Gamma posterior sits near the truth
    # the conditional likelihood prefers the true fluxes to badly wrong ones
    src = m.srcs[0]
    ll_true = src.log_likelihood()
    ll_bad = src.log_likelihood(fluxes=np.asarray(src.params.fluxes) * 3.0)
    assert np.isfinite(ll_true) and ll_true > ll_bad
    # one launch for the proposals of every source == the per-source batches
    from desi_mcmc_amd import sources
    rs = np.random.RandomState(3)
    us = np.array([s.params.u for s in m.srcs])[:, None, :] + rs.normal(0, 3e-5, size=(12, 5, 2))
    sweep = sources.log_likelihood_sweep(m.srcs, us)
    assert sweep.shape == (12, 5)
    for i, s in enumerate(m.srcs):
        np.testing.assert_allclose(sweep[i], s.log_likelihood_batch(us=us[i]), rtol=1e-13)
    iso = sources.log_likelihood_sweep(m.srcs[:3], us[:3], isolated=True)
    np.testing.assert_allclose(iso[1], m.srcs[1].log_likelihood_batch(us=us[1], isolated=True), rtol=1e-13)
    for im, o in zip(imgs, old):
        im.epsilon = o


def test_source_conditional_loglik_golden(cel):
    """Source.log_likelihood / log_likelihood_isolated / compute_model_patch (sources.py:134-237,
    351-395) against values the reference's own Source class produced"""
    from desi_mcmc_amd import sources
    g = load_golden("source_ll.npz")
    H, W = int(g["H"]), int(g["W"])
    imgs = frame_images(cel, {k: g[k] for k in g}, H, W, nelec=g["nelec"])
    used = g["bands_used"]
    for ci in range(int(g["ncases"])):
        kind = int(g["c%d_kind" % ci])
        shape = g["c%d_shape" % ci]
        params = cel.SrcParams(u=g["c%d_u" % ci], a=kind, fluxes=g["c%d_flux" % ci], theta=shape[0],
                               sigma=shape[1], phi=shape[2], rho=shape[3])
        src = sources.Source(params)
        zs = unpack_ragged(g["c%d_z" % ci], g["c%d_zoffs" % ci], g["c%d_zshapes" % ci])
        boxes = g["c%d_boxes" % ci]
        for j, b in enumerate(used):
            # the box the reference derived (Source.get_bounding_box, sources.py:83-96)
            xlim, ylim = sources.Source.get_bounding_box(params, imgs[b])
            assert (int(ylim[0]), int(ylim[1]), int(xlim[0]), int(xlim[1])) == tuple(boxes[j])
            samp = sources.SamplePatch(zs[j], (boxes[j, 0], boxes[j, 1]), (boxes[j, 2], boxes[j, 3]))
            src.sample_image_list.append((samp, imgs[b], None))
        us, fl, sh = g["c%d_us" % ci], g["c%d_fl" % ci], g["c%d_sh" % ci]
        ll0 = src.log_likelihood_batch(us, fl, sh)
        np.testing.assert_allclose(ll0, g["c%d_ll0" % ci], rtol=1e-11)
        n1 = len(g["c%d_ll1" % ci])
        ll1 = src.log_likelihood_batch(us[:n1], fl[:n1], sh[:n1], isolated=True)
        np.testing.assert_allclose(ll1, g["c%d_ll1" % ci], rtol=1e-11)
        # scalar forms with the reference's signatures
        np.testing.assert_allclose(src.log_likelihood(), g["c%d_ll0" % ci][0], rtol=1e-11)
        np.testing.assert_allclose(src.log_likelihood(u=us[2], fluxes=fl[2], shape=sh[2]), g["c%d_ll0" % ci][2], rtol=1e-11)
        np.testing.assert_allclose(src.location_likelihood(us[1]),
                                   src.log_likelihood_batch(us[1:2])[0], rtol=1e-14)
        np.testing.assert_allclose(src.log_likelihood_isolated(), g["c%d_ll1" % ci][0], rtol=1e-11)
        # compute_model_patch reproduces the Poisson mean the golden's photons were drawn from
        p, yl, xl = src.compute_model_patch(imgs[used[0]], xlim=(boxes[0, 2], boxes[0, 3]), ylim=(boxes[0, 0], boxes[0, 1]))
        assert p.shape == zs[0].shape and p.min() >= 0


@pytest.mark.parametrize("kernel,tail", [("direct", 32.0), ("recurrence", 32.0), ("recurrence", 0.0)])
def test_patch_loglik_adversarial_patches_vs_oracle(cel, ctx, orc, kernel, tail):
    """cel_patch_loglik on patches the recurrence kernel must chunk (wider than 32, taller than
    64), shift away from the source (deep tails: every pixel carries data, so log(m) is exercised
    where m is tiny), and give up on (a patch so far out that the seeds would underflow: direct
    fallback).  Both forms, both kernels, against the oracle."""
    from desi_mcmc_amd import synth
    H, W = 300, 420
    bands = synth.make_bands(H, W, 2)
    rs = np.random.RandomState(11)
    pix = np.array([[200.3, 150.6], [201.0, 149.0], [60.2, 40.9], [199.7, 151.2]])
    typ = np.array([0, 1, 1, 0], np.int32)
    radec = synth.pixel2equa(bands[0], pix)
    shape = np.array([[0.5, 2.0, 30.0, 0.5], [0.3, 3.5, 110.0, 0.4], [0.8, 0.7, 10.0, 0.9], [0.5, 1.0, 0.0, 0.5]])
    counts = np.array([[4e4, 3e4], [9e4, 1e5], [2e3, 5e3], [0.0, 7e2]])
    nelec = rs.poisson(200.0, size=(2, H, W)).astype(float)
    iset = cel.ImageSet(ctx, bands, H, W, nelec=nelec)
    sset = cel.SourceSet(ctx, 4, 2).set(typ, radec, counts, shape)
    ob = bands.copy()
    ob[:, 36] = [orc.checked_radius(ob[b], iset.band(b)[36]) for b in range(2)]
    cases = [
        np.array([[130, 171, 181, 222], [120, 190, 170, 233]]),     # around sources 0/1/3: 41x41 and 70x63
        np.array([[60, 260, 100, 300], [149, 152, 199, 202]]),      # 200 x 200 (chunks both ways) and 3 x 3
        np.array([[150, 215, 230, 263], [0, 0, 0, 0]]),             # shifted off the centre; no patch in band 1
        np.array([[0, 64, 388, 420], [236, 300, 0, 33]]),           # far corners: below the seed-safe range
    ]
    ctx.set_kernel(kernel)
    ctx.set_tail_log(tail)          # 0: no component is ever dropped
    try:
        for boxes in cases:
            data = [rs.poisson(3.0, size=(bx[1] - bx[0], bx[3] - bx[2])).astype(float) + 1.0 if bx[1] > bx[0] else None
                    for bx in boxes]
            if data[0].shape[0] == 200:     # photons in one corner only: the conditional form crops to them
                data[0][:150] = 0.0
                data[0][:, 37:] = 0.0
            for isolated in (False, True):
                got = iset.patch_loglik(sset, boxes, data, isolated=isolated)
                for s in range(4):
                    want = 0.0
                    for b in range(2):
                        bx = boxes[b]
                        if bx[1] <= bx[0]:
                            continue
                        want += orc.patch_loglik(ob[b], H, W, typ[s], radec[s], shape[s], counts[s, b], bx,
                                                 data[b], 1 if isolated else 0)
                    np.testing.assert_allclose(got[s], want, rtol=RT_LL, err_msg="src %d boxes %s iso %s" % (s, boxes, isolated))
    finally:
        ctx.set_kernel("recurrence")
        ctx.set_tail_log("default")




@pytest.mark.parametrize("seed", _fuzz_seeds(10))
def test_fuzz_random_fields_vs_oracle(cel, ctx, orc, seed):
    """Seeded random small fields at the extremes the synthetic benchmark population never visits:
    frames of any size, sharp and broad PSFs, sky levels over six decades, galaxy scales from
    below a pixel to a third of the frame, degenerate axis ratios and profile mixes, counts from 1
    to 1e7, sources on and beyond the border.  Model image, log-likelihood, E-step reductions and
    conditional likelihoods on the sources' own boxes, all against the oracle."""
    from desi_mcmc_amd import synth
    rs = np.random.RandomState(1000 + seed)
    H, W = int(rs.randint(40, 260)), int(rs.randint(40, 300))
    B = int(rs.randint(1, 4))
    S = int(rs.randint(1, 50))
    bands = synth.make_bands(H, W, B)
    bands[:, 12:24] *= rs.choice([0.35, 1.0, 3.0])                  # PSF covariances
    bands[:, 0] *= 10.0 ** rs.uniform(-3, 3)                         # sky level
    bands[:, 36] = 0.0                                               # let the library derive the radius
    pix = np.column_stack([rs.uniform(-40, W + 40, S), rs.uniform(-40, H + 40, S)])
    typ = (rs.rand(S) < rs.rand()).astype(np.int32)
    radec = synth.pixel2equa(bands[0], pix)
    theta = np.where(rs.rand(S) < 0.2, rs.choice([0.0, 1.0], S), rs.rand(S))
    sigma = np.exp(rs.uniform(np.log(0.05), np.log(0.33 * min(H, W) * 0.396), S))   # arcsec; 0.396 arcsec per pixel
    shape = np.column_stack([theta, sigma, rs.uniform(0, 180, S), rs.uniform(0.03, 1.0, S)])
    counts = np.exp(rs.uniform(0.0, np.log(1e7), size=(S, B)))
    nelec = rs.poisson(np.clip(bands[:, 0], 1.0, 1e4)[:, None, None], size=(B, H, W)).astype(float)
    ctx.set_option(cel._lib.CEL_OPT_TILE_LAYOUT, int(__import__("os").environ.get("CEL_TEST_LAYOUT", "1")))     # render-tile layout under test
    try:
        iset = cel.ImageSet(ctx, bands, H, W, nelec=nelec)
    finally:
        ctx.set_option(cel._lib.CEL_OPT_TILE_LAYOUT, 1)
    sset = cel.SourceSet(ctx, S, B).set(typ, radec, counts, shape)
    # the arithmetic at the extremes, so at the STRICT threshold (the same fields at the shipping one:
    # test_fuzz_random_fields_vs_oracle_default_threshold, at what the rule guarantees)
    ctx.set_tail_log("strict")
    try:
        ll, llb = iset.render(sset, loglik=True)
        lam_got = iset.model_images()
        xt, ms, nz = iset.estep_stats(sset)
    finally:
        ctx.set_tail_log("default")
    ob = bands.copy()
    ob[:, 36] = [orc.checked_radius(ob[b], iset.band(b)[36]) for b in range(B)]
    o_lam, o_ll, o_st = orc.render_field(ob, H, W, typ, radec, counts, shape, nelec)
    np.testing.assert_allclose(lam_got, o_lam, rtol=RT_LAM_STRICT)
    # a band's log-likelihood is a sum of terms of both signs: the tolerance is relative to the sum of their magnitudes
    # (seed 2635 of a 3 000-seed run: a band whose terms cancel to 1e-3 of their size, 3e-11 of the result off)
    scale = (np.abs(nelec * np.log(o_lam)) + o_lam).sum(axis=(1, 2))
    assert np.all(np.abs(llb - o_ll) <= RT_LL * scale), (llb, o_ll, scale)
    assert iset.stats()["n_srcpix"] == o_st["n_srcpix"]
    oxt, oms, onz = orc.estep_stats(ob, H, W, typ, radec, counts, shape, nelec)
    np.testing.assert_allclose(xt, oxt, rtol=1e-10, atol=1e-9)
    np.testing.assert_allclose(ms, oms, rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(nz, onz, rtol=1e-11)
    # conditional likelihoods of each source on its own boxes, data = the observed pixels there
    boxes, status = iset.source_boxes(sset)
    for s in range(min(S, 6)):
        bx = boxes[:, s]
        data = [nelec[b, bx[b, 0]:bx[b, 1], bx[b, 2]:bx[b, 3]]
                if status[b, s] > 0 and bx[b, 1] > bx[b, 0] and bx[b, 3] > bx[b, 2] else None for b in range(B)]
        bxs = np.array([bx[b] if data[b] is not None else [0, 0, 0, 0] for b in range(B)])
        one = cel.SourceSet(ctx, 1, B).set(typ[s:s + 1], radec[s:s + 1], counts[s:s + 1], shape[s:s + 1])
        for isolated in (False, True):
            got = iset.patch_loglik(one, bxs, data, isolated=isolated)[0]
            want = sum(orc.patch_loglik(ob[b], H, W, typ[s], radec[s], shape[s], counts[s, b], bxs[b],
                                        np.ascontiguousarray(data[b]), 1 if isolated else 0)
                       for b in range(B) if data[b] is not None)
            # relative to the size of the terms, not of their sum (seed 5282 of a long run: terms of 1e2 that add up to 90
            # with 3.5e-8 between the two sums): the photon term's magnitudes and the mass from the oracle; on the observed
            # image an upper bound of |y log(lambda)| + lambda
            scale = 0.0
            for b in range(B):
                if data[b] is None:
                    continue
                if not isolated:
                    t = orc.patch_loglik_terms(ob[b], H, W, typ[s], radec[s], shape[s], counts[s, b], bxs[b], np.ascontiguousarray(data[b]))
                    scale += t[1] + t[2]
                else:
                    eps_b = ob[b, 0]
                    scale += (data[b].sum() * max(abs(np.log(eps_b)), abs(np.log(eps_b + counts[s, b]))) + eps_b * data[b].size + counts[s, b])
            assert abs(got - want) <= RT_LL * max(scale, abs(want)) + 1e-9, "seed %d src %d iso %s: %r %r (scale %g)" % (seed, s, isolated, got, want, scale)


@pytest.mark.parametrize("seed", _fuzz_seeds(10))
def test_fuzz_random_fields_vs_oracle_default_threshold(cel, ctx, orc, seed):
    """The same random extreme fields rendered at the library's DEFAULT drop threshold (T = 24).  What the rule guarantees:
    a skipped component adds less than eps * e^-T on its tile, so |d lambda| / lambda <= n * e^-T with n the components
    skipped on the pixel, at most 42 per source: asserted with n = 42 S (every component of every source); in these fields
    the error stays below 1e-9.  Log-likelihoods at 1e-9."""
    from desi_mcmc_amd import synth
    rs = np.random.RandomState(1000 + seed)
    H, W = int(rs.randint(40, 260)), int(rs.randint(40, 300))
    B = int(rs.randint(1, 4))
    S = int(rs.randint(1, 50))
    bands = synth.make_bands(H, W, B)
    bands[:, 12:24] *= rs.choice([0.35, 1.0, 3.0])
    bands[:, 0] *= 10.0 ** rs.uniform(-3, 3)
    bands[:, 36] = 0.0
    pix = np.column_stack([rs.uniform(-40, W + 40, S), rs.uniform(-40, H + 40, S)])
    typ = (rs.rand(S) < rs.rand()).astype(np.int32)
    radec = synth.pixel2equa(bands[0], pix)
    theta = np.where(rs.rand(S) < 0.2, rs.choice([0.0, 1.0], S), rs.rand(S))
    sigma = np.exp(rs.uniform(np.log(0.05), np.log(0.33 * min(H, W) * 0.396), S))
    shape = np.column_stack([theta, sigma, rs.uniform(0, 180, S), rs.uniform(0.03, 1.0, S)])
    counts = np.exp(rs.uniform(0.0, np.log(1e7), size=(S, B)))
    nelec = rs.poisson(np.clip(bands[:, 0], 1.0, 1e4)[:, None, None], size=(B, H, W)).astype(float)
    iset = cel.ImageSet(ctx, bands, H, W, nelec=nelec)
    sset = cel.SourceSet(ctx, S, B).set(typ, radec, counts, shape)
    assert ctx.get_option(cel._lib.CEL_OPT_TAIL_LOG) == cel._lib.TAIL_LOG_DEFAULT       # (what the suite runs at)
    ll, llb = iset.render(sset, loglik=True)
    lam = iset.model_images()
    ob = bands.copy()
    ob[:, 36] = [orc.checked_radius(ob[b], iset.band(b)[36]) for b in range(B)]
    o_lam, o_ll, _ = orc.render_field(ob, H, W, typ, radec, counts, shape, nelec)
    err = float(np.max(np.abs(lam / o_lam - 1.0)))
    assert err <= 42 * S * np.exp(-24.0) + 1e-12, (err, S)          # the rule's guarantee
    assert err < 1e-9, err                                           # what these fields show
    # (relative to the terms' magnitudes: 5 of 4 000 seeds hold a band whose terms cancel to a tenth of their size)
    scale = (np.abs(nelec * np.log(o_lam)) + o_lam).sum(axis=(1, 2))
    assert np.all(np.abs(llb - o_ll) <= 1e-9 * scale), (llb, o_ll, scale)


def test_integration_md_ctypes_binding(cel, orc):
    """The direct ctypes binding INTEGRATION.md shows (section 2 and the resident Gibbs calls),
    call for call, against the oracle: the document's signatures are the library's."""
    import ctypes as C
    from desi_mcmc_amd import _lib, synth
    L = C.CDLL(_lib.LIB_PATH)
    dp = C.POINTER(C.c_double)
    L.cel_last_error.restype = C.c_char_p

    def check(rc):
        if rc:
            raise (ValueError if rc == 1 else RuntimeError)(L.cel_last_error().decode())

    H, W, B, S = 96, 128, 2, 5
    bands = synth.make_bands(H, W, B)
    src = synth.make_sources(S, H, W, bands, frac_gal=0.6, seed=12)
    typ, radec, counts, shape = (np.ascontiguousarray(src[k]) for k in ("type", "radec", "counts", "shape"))
    nelec = np.ascontiguousarray(np.random.RandomState(2).poisson(300.0, size=(B, H, W)).astype(np.float64))
    ctx, img, srch = C.c_void_p(), C.c_void_p(), C.c_void_p()
    check(L.cel_ctx_create(0, None, C.byref(ctx)))
    check(L.cel_images_create(ctx, B, H, W, bands.ctypes.data_as(dp), C.byref(img)))
    check(L.cel_images_set_nelec(img, nelec.ctypes.data_as(C.c_void_p), 0))
    check(L.cel_sources_create(ctx, C.c_int64(S), B, C.byref(srch)))
    check(L.cel_sources_set(srch, C.c_int64(S), typ.ctypes.data_as(C.c_void_p), radec.ctypes.data_as(C.c_void_p),
                            counts.ctypes.data_as(C.c_void_p), shape.ctypes.data_as(C.c_void_p), 0))
    ll_band, ll = np.zeros(B), C.c_double()
    check(L.cel_render_field(img, srch, 1, ll_band.ctypes.data_as(dp), C.byref(ll)))
    lam = np.empty((B, H, W))
    check(L.cel_images_get_lambda(img, lam.ctypes.data_as(C.c_void_p), 0))
    ob = bands.copy()
    rec = np.zeros(37)
    for b in range(B):
        check(L.cel_images_get_band(img, b, rec.ctypes.data_as(dp)))
        ob[b, 36] = orc.checked_radius(ob[b], rec[36])
    o_lam, o_ll, _ = orc.render_field(ob, H, W, typ, radec, counts, shape, nelec)
    np.testing.assert_allclose(lam, o_lam, rtol=RT_LAM)
    np.testing.assert_allclose(ll_band, o_ll, rtol=RT_LL)
    assert abs(ll.value - o_ll.sum()) <= 1e-11 * abs(o_ll.sum())
    # the resident Gibbs calls
    noise = np.zeros(B)
    check(L.cel_photon_split(img, srch, C.c_uint64(7), None, None, 1, noise.ctypes.data_as(dp)))
    S_, total = C.c_int64(), C.c_int64()
    check(L.cel_samples_info(img, C.byref(S_), C.byref(total)))
    assert S_.value == S and total.value > 0
    sums = np.zeros((S, B))
    check(L.cel_samples_fetch(img, None, None, None, sums.ctypes.data_as(dp)))
    assert np.array_equal(sums.sum(axis=0) + noise, nelec.reshape(B, -1).sum(axis=1))
    P = 2 * S
    prop = C.c_void_p()
    check(L.cel_sources_create(ctx, C.c_int64(P), B, C.byref(prop)))
    rep = lambda a: np.ascontiguousarray(np.repeat(a, 2, axis=0))
    t2, r2, c2, s2 = rep(typ), rep(radec), rep(counts), rep(shape)
    check(L.cel_sources_set(prop, C.c_int64(P), t2.ctypes.data_as(C.c_void_p), r2.ctypes.data_as(C.c_void_p),
                            c2.ctypes.data_as(C.c_void_p), s2.ctypes.data_as(C.c_void_p), 0))
    owner = np.repeat(np.arange(S, dtype=np.int32), 2)
    llp = np.zeros(P)
    check(L.cel_patch_loglik_multi(img, prop, owner.ctypes.data_as(C.POINTER(C.c_int32)), C.c_int64(S),
                                   None, None, None, 1, 0, llp.ctypes.data_as(dp)))
    assert np.all(np.isfinite(llp)) and np.array_equal(llp[0::2], llp[1::2])      # identical proposals, identical scores
    # the device-resident per-source updates, as the document calls them
    new_radec, llh, stats = np.zeros((S, 2)), np.zeros(S), np.zeros(4, dtype=np.int64)
    check(L.cel_slice_locations(img, srch, None, C.c_double(1e-3), C.c_uint64(5), 4000, new_radec.ctypes.data_as(dp),
                                llh.ctypes.data_as(dp), stats.ctypes.data_as(C.POINTER(C.c_int64))))
    assert stats[0] >= 4 and stats[1] >= 4 * S and stats[2] > 0 and np.all(np.abs(new_radec - radec) < 1e-3)
    mass = np.zeros((S, B))
    check(L.cel_stamp_mass(img, srch, mass.ctypes.data_as(dp)))
    assert np.all(mass > 0.5) and np.all(mass < 1.01)
    rs = np.random.RandomState(4)
    numdir = 4
    dirs = rs.normal(size=(S, numdir, 4))
    dirs = np.ascontiguousarray(dirs / np.sqrt(np.sum(dirs ** 2, axis=2, keepdims=True)))
    th, llh2, stats2 = np.zeros((S, 4)), np.zeros(S), np.zeros(4, dtype=np.int64)
    check(L.cel_slice_sample(img, srch, 1, None, dirs.ctypes.data_as(dp), numdir, 1, 1000, C.c_double(1.0), C.c_double(180.0),
                             C.c_uint64(9), 20000, th.ctypes.data_as(dp), llh2.ctypes.data_as(dp),
                             stats2.ctypes.data_as(C.POINTER(C.c_int64))))
    gal = typ == 1
    assert np.array_equal(th[~gal], shape[~gal]) and np.all(np.any(th[gal] != shape[gal], axis=1))     # stars are left alone
    assert np.all((th[gal, 0] > 0) & (th[gal, 0] < 1) & (th[gal, 1] > 0) & (th[gal, 3] > 0) & (th[gal, 3] < 1)) and stats2[1] > 0
    for h in (prop, srch):
        check(L.cel_sources_destroy(h))
    check(L.cel_images_destroy(img))
    check(L.cel_ctx_destroy(ctx))


def test_multi_field_standin_dealt_to_one_rank(cel, ctx, orc):
    """BASELINE configs[3] stand-in at world size 1: K synthetic fields dealt by dist.field_shard, every
    field's log-likelihood against the oracle, and the job's sum (what bench.py --workload fields8_2048
    all-reduces) -- the world-2 deal is covered on CPU by tests/test_dist_gloo.py."""
    from desi_mcmc_amd import dist, synth
    K = 3
    mine = dist.field_shard(K, 1, 0)
    assert mine == [0, 1, 2]
    total = np.zeros(5)
    want = np.zeros(5)
    for k in mine:
        f = synth.SyntheticField(ctx, 150, 5, 200, 240, frac_gal=0.5, seed=42 + 1000 * k)
        _, llb = f.images.render(f.sources, loglik=True)
        _, o_ll, _ = orc.render_field(oracle_bands(f), f.H, f.W, f.src["type"], f.src["radec"], f.src["counts"],
                                      f.src["shape"], f.nelec)
        np.testing.assert_allclose(llb, o_ll, rtol=RT_LL)
        total += llb
        want += o_ll
    np.testing.assert_allclose(dist.allreduce_loglik(total), want, rtol=RT_LL)
    fields = [synth.SyntheticField(ctx, 150, 5, 200, 240, frac_gal=0.5, seed=42 + 1000 * k).src["radec"][0] for k in mine]
    assert len({tuple(r) for r in fields}) == K          # different fields, not one field K times


def test_tail_log_fast_preset_meets_the_1e6_bar(cel, orc, big_field):
    """CEL_OPT_TAIL_LOG = 20 (the documented fast preset, bench.py's second line): model pixels within
    1e-6 of the oracle on a mixed field, and within 1e-6 of the default threshold's pixels over the
    whole BASELINE-size field; log-likelihoods within 1e-8."""
    from desi_mcmc_amd import _lib, synth
    c2 = cel.Context(0)
    c2.set_tail_log("fast")
    assert c2.get_option(_lib.CEL_OPT_TAIL_LOG) == _lib.TAIL_LOG_FAST == 20.0
    f = synth.SyntheticField(c2, 400, 5, 300, 333, frac_gal=0.6, seed=77)
    ll, llb = f.images.render(f.sources, loglik=True)
    o_lam, o_ll, _ = orc.render_field(oracle_bands(f), f.H, f.W, f.src["type"], f.src["radec"], f.src["counts"],
                                      f.src["shape"], f.nelec)
    lam = f.images.model_images()
    np.testing.assert_allclose(lam, o_lam, rtol=1e-6)
    assert np.max(np.abs(lam / o_lam - 1.0)) > 1e-13          # the preset does skip more than the default
    np.testing.assert_allclose(llb, o_ll, rtol=1e-8)
    # the BASELINE-size field: fast preset against the default threshold
    bf = big_field
    bf.images.render(bf.sources, loglik=True)
    ll32, llb32 = bf.images.render(bf.sources, loglik=True)
    lam32 = bf.images.model_images()
    ctx0 = bf.images.ctx
    ctx0.set_tail_log("fast")
    try:
        ll20, llb20 = bf.images.render(bf.sources, loglik=True)
        lam20 = bf.images.model_images()
    finally:
        ctx0.set_tail_log("default")
    assert np.max(np.abs(lam20 / lam32 - 1.0)) < 1e-6
    np.testing.assert_allclose(llb20, llb32, rtol=1e-8)


def test_list_of_srcparams_end_to_end_after_single_source_moves(cel, orc):
    """celeste_likelihood[_multi_image](LIST of SrcParams, imgs) called again and again while single sources change -- the
    pattern of util/infer/mcmc_transitions.py:37-152 and celeste_mcmc.py:130: the device catalogue follows row by changed row
    (cel_sources_set_rows) and every value equals what a freshly built list gives, and the oracle."""
    from desi_mcmc_amd import celeste, synth
    ctx = cel.default_context(0)
    f = synth.SyntheticField(ctx, 300, 5, 256, 256, frac_gal=0.5, seed=31)
    imgs = synth.fits_images(f)
    fl5 = f.flux5()
    ps = [cel.SrcParams(u=f.src["radec"][s].copy(), a=int(f.src["type"][s]), fluxes=dict(zip(BANDS, fl5[s])),
                        theta=f.src["shape"][s, 0], sigma=f.src["shape"][s, 1], phi=f.src["shape"][s, 2], rho=f.src["shape"][s, 3])
          for s in range(f.S)]

    def fresh():
        return celeste.celeste_likelihood_multi_image(list(ps), imgs)       # a new list object: gathered and uploaded whole
    ll0 = celeste.celeste_likelihood_multi_image(ps, imgs)
    assert ll0 == fresh()
    ob = oracle_bands(f)
    rs = np.random.RandomState(2)
    for step in range(6):
        s = int(rs.randint(f.S))
        if step % 3 == 0:
            pos = ps[s].u
            pos[step % 2] += 2e-5
            ps[s].u = pos                                                   # in place, then assigned (mcmc_transitions.py:49-51)
        elif step % 3 == 1:
            ps[s].fluxes = {b: v * 1.5 for b, v in ps[s].fluxes.items()}
        else:
            ps[s].a = 1 - ps[s].a
            ps[s].theta, ps[s].sigma, ps[s].phi, ps[s].rho = 0.3, 1.1, 40.0, 0.7
        got = celeste.celeste_likelihood_multi_image(ps, imgs)
        assert got == fresh(), step
        assert got != ll0
        ll0 = got
        assert celeste.celeste_likelihood(ps, imgs[2]) == celeste.celeste_likelihood(list(ps), imgs[2])
    typ = np.array([p.a for p in ps], dtype=np.int32)
    radec = np.array([p.u for p in ps])
    counts = np.array([[p.fluxes[b] for b in BANDS] for p in ps]) / f.bands[:, 2][None, :] * f.bands[:, 1][None, :]
    shape = np.array([[p.theta, p.sigma, p.phi, p.rho] if p.a == 1 else [0, 0, 0, 0] for p in ps], dtype=float)
    _, o_ll, _ = orc.render_field(ob, f.H, f.W, typ, radec, counts, shape, f.nelec)
    np.testing.assert_allclose(ll0, o_ll.sum(), rtol=RT_LL)
    # the raw entry point: bad rows are refused
    sset = cel.SourceSet(ctx, 8, 5).set(typ[:8], radec[:8], counts[:8], shape[:8])
    with pytest.raises(ValueError):
        sset.set_rows([8], typ[:1], radec[:1], counts[:1], shape[:1])
    sset.set_rows([], typ[:0], radec[:0], counts[:0], shape[:0])


def test_image_set_cache_lru_budget_and_superset_reuse(cel, stamp_images):
    """the mirror's device image-set cache: least recently used sets are evicted -- and their device
    memory released at once -- beyond the byte / count budget; a caller touching SOME images of a
    resident set is handed that set, not a second copy; an evicted set fails loudly"""
    from desi_mcmc_amd import celeste
    rec, imgs = stamp_images
    for key in list(celeste._SETS):
        celeste._evict(key)
    full = celeste._image_set(tuple(imgs))
    assert len(celeste._SETS) == 1
    sub, pos = celeste._image_subset((imgs[3], imgs[1]))
    assert sub is full and pos == [3, 1] and len(celeste._SETS) == 1          # no second copy of the same pixels
    one = celeste._image_set((imgs[2],))
    assert one is not full and len(celeste._SETS) == 2
    # Gibbs changes an image's sky level: the next fetch pushes it to whichever set holds the image
    old = imgs[1].epsilon
    try:
        imgs[1].epsilon = old * 1.25
        again = celeste._image_set(tuple(imgs))
        assert again is full and full.eps[1] == old * 1.25 and full.band(1)[0] == old * 1.25
    finally:
        imgs[1].epsilon = old
        celeste._image_set(tuple(imgs))
    budget = celeste.CACHE_MAX_BYTES
    try:
        celeste.CACHE_MAX_BYTES = 24 * 51 * 51 * 5 + 1                          # room for the 5-band set only
        celeste._cache_trim()
        # sets somebody still holds (this test's `full` and `one`) are never closed under their holder
        assert len(celeste._SETS) == 2 and full.band(0)[0] == imgs[0].epsilon
        del sub, again, one
        celeste._cache_trim()
        assert list(celeste._SETS) == [tuple(id(i) for i in imgs)]               # the unreferenced one-band set went
        newest = celeste._image_set((imgs[0],))
        assert len(celeste._SETS) == 2                                           # `full` is held: over budget, but kept
        import weakref
        gone = weakref.ref(full)
        handle = full
        del full
        celeste._cache_trim(keep=(id(imgs[0]),))
        assert len(celeste._SETS) == 2                                           # `handle` still holds it
        del handle
        celeste._cache_trim(keep=(id(imgs[0]),))
        assert list(celeste._SETS) == [(id(imgs[0]),)] and gone() is None        # released at once, newest kept
        assert celeste._SETS[(id(imgs[0]),)][1] is newest
        closed = cel.ImageSet(cel.default_context(0), np.stack([imgs[0].band_record()]), 51, 51)
        closed.close()
        with pytest.raises(ValueError):
            closed.band(0)                                                       # closed: null handle, not a stale pointer
    finally:
        celeste.CACHE_MAX_BYTES = budget
    ll = celeste.celeste_likelihood_multi_image([], imgs)                        # re-created on demand
    np.testing.assert_allclose(ll, sum(np.sum(i.nelec * np.log(i.epsilon) - i.epsilon) for i in imgs), rtol=1e-13)


@pytest.mark.parametrize("seed", _fuzz_seeds(6))
def test_fuzz_star_fields_vs_oracle(cel, ctx, orc, seed):
    """random STAR-ONLY fields through the batched star pass: odd frame sizes, crowded tiles (more than
    one 64-star batch), stars on and off every edge, a PSF so sharp that the one-segment path must
    hand over to the general one, a caller-supplied huge bounding radius -- model pixels 1e-10,
    log-likelihoods 1e-11, and the photon split's strict-box totals through the same lists"""
    from desi_mcmc_amd import synth
    rs = np.random.RandomState(100 + seed)
    H, W = int(rs.randint(40, 300)), int(rs.randint(40, 300))
    S = int(rs.choice([1, 7, 150, 900]))
    f_bands = synth.make_bands(H, W, 5)
    if seed == 3:       # a very sharp first component: exponents beyond the one-segment bound
        f_bands[:, 12:16] = np.array([0.02, 0.0, 0.0, 0.025])[None, :]
    if seed == 4:       # a caller-imposed radius far beyond the PSF's own
        f_bands[:, 36] = 60.0
    pix = np.column_stack([rs.uniform(-30, W + 30, S), rs.uniform(-30, H + 30, S)])
    if seed == 5:       # everything on one tile: several batches of 64
        pix = np.column_stack([rs.uniform(10, 30, S), rs.uniform(5, 60, S)])
    typ = np.zeros(S, dtype=np.int32)
    counts = np.exp(rs.uniform(np.log(50.0), np.log(5e4), size=(S, 5)))
    radec = synth.pixel2equa(f_bands[0], pix)
    iset = cel.ImageSet(ctx, f_bands, H, W)
    ss = cel.SourceSet(ctx, S, 5).set(typ, radec, counts)
    iset.render(ss)
    ob = f_bands.copy()
    for b in range(5):
        ob[b, 36] = orc.checked_radius(ob[b], iset.band(b)[36])
    nelec = rs.poisson(iset.model_images()).astype(np.float64)
    iset.set_nelec(nelec)
    ll, llb = iset.render(ss, loglik=True)
    o_lam, o_ll, o_st = orc.render_field(ob, H, W, typ, radec, counts, np.zeros((S, 4)), nelec)
    np.testing.assert_allclose(iset.model_images(), o_lam, rtol=RT_LAM_STRICT)
    np.testing.assert_allclose(llb, o_ll, rtol=RT_LL)
    assert iset.stats()["n_srcpix"] == o_st["n_srcpix"]
    # the split conserves every photon on the same (stars-first) lists
    noise = iset.photon_split_resident(ss, seed=seed)
    np.testing.assert_array_equal(iset.sample_sums().sum(axis=0) + noise, nelec.reshape(5, -1).sum(axis=1))


@pytest.mark.parametrize("seed", range(6))
def test_star_tile_kernel_vs_oracle_and_general_kernel(cel, ctx, orc, seed):
    """k_render_stars (CEL_OPT_STAR_TILES; a catalogue without galaxies): the same random star fields as above --
    frames that are no multiple of a tile (edge tiles take the bounded epilogue), tiles with several 64-star
    batches, stars off every edge, a huge caller radius -- against the oracle (1e-10 / 1e-11) and against the
    general kernel (rounding); with and without stored model images; which kernel ran is read off the profile
    slots.  A sharp PSF or a huge radius (seeds 3, 4: no one-segment walk), one galaxy in the catalogue, or types that came from device memory keep the
    general kernel."""
    import ctypes as C
    from desi_mcmc_amd import synth
    L = cel._lib
    rs = np.random.RandomState(100 + seed)
    H, W = int(rs.randint(40, 300)), int(rs.randint(40, 300))
    S = int(rs.choice([1, 7, 150, 900]))
    f_bands = synth.make_bands(H, W, 5)
    if seed == 3:
        f_bands[:, 12:16] = np.array([0.02, 0.0, 0.0, 0.025])[None, :]
    if seed == 4:
        f_bands[:, 36] = 60.0
    pix = np.column_stack([rs.uniform(-30, W + 30, S), rs.uniform(-30, H + 30, S)])
    if seed == 5:
        pix = np.column_stack([rs.uniform(10, 30, S), rs.uniform(5, 60, S)])
    typ = np.zeros(S, dtype=np.int32)
    counts = np.exp(rs.uniform(np.log(50.0), np.log(5e4), size=(S, 5)))
    radec = synth.pixel2equa(f_bands[0], pix)
    iset = cel.ImageSet(ctx, f_bands, H, W)
    ss = cel.SourceSet(ctx, S, 5).set(typ, radec, counts)

    def run(mode, sources=ss, store=True):
        ctx.set_option(L.CEL_OPT_STAR_TILES, mode)
        ctx.profile(True)
        ll, llb = iset.render(sources, loglik=True, store=store)
        n_star, n_gen = ctx.profile_get("render_stars")[1] + ctx.profile_get("small_stars")[1], ctx.profile_get("render")[1]
        ctx.profile(False)
        return ll, llb, n_star, n_gen

    try:
        assert ctx.get_option(L.CEL_OPT_STAR_TILES) == 1.0
        with pytest.raises(ValueError):
            ctx.set_option(L.CEL_OPT_STAR_TILES, 4)
        ctx.set_option(L.CEL_OPT_STAR_TILES, 0)
        iset.render(ss)
        ob = f_bands.copy()
        for b in range(5):
            ob[b, 36] = orc.checked_radius(ob[b], iset.band(b)[36])
        nelec = rs.poisson(iset.model_images()).astype(np.float64)
        iset.set_nelec(nelec)
        ll0, llb0, ns0, ng0 = run(0)
        lam0 = iset.model_images()
        assert (ns0, ng0) == (0, 1)
        ll1, llb1, ns1, ng1 = run(3)                 # the tile-count rule alone: a frame of this size keeps the general kernel
        assert (ns1, ng1) == (0, 1) and ll1 == ll0
        # the default: at most 4096 stars on at most 2048 tiles take the ONE-launch path (k_small_stars: prep, binning, render
        # and the partials of the per-band sums in one kernel; counted in the star kernel's profile slot)
        iset.render(cel.SourceSet(ctx, 1, 5).set(typ[:1], radec[:1], counts[:1]))
        ll5, llb5, ns5, ng5 = run(1)
        lam5 = iset.model_images()
        # (1, 1): a part of a tile held more candidate stars than the kernel stages (900 stars on a small frame): the call was
        # rendered again on the general path, as is every later one on this image set
        fell_back = (ns5, ng5) == (1, 1)
        assert (ns5, ng5) == ((0, 1) if seed in (3, 4) else (1, 0)) or (fell_back and S >= 256)
        o_lam, o_ll, o_st = orc.render_field(ob, H, W, typ, radec, counts, np.zeros((S, 4)), nelec)
        np.testing.assert_allclose(lam5, o_lam, rtol=RT_LAM_STRICT)
        np.testing.assert_allclose(llb5, o_ll, rtol=RT_LL)
        np.testing.assert_allclose(lam5, lam0, rtol=1e-13)
        np.testing.assert_allclose(llb5, llb0, rtol=1e-13)
        st = iset.stats()                             # the kernel left k_prep's records behind: work counters, boxes, status
        assert st["n_srcpix"] == o_st["n_srcpix"] and st["n_gauss"] == o_st["n_gauss"]
        bx5, st5 = iset.source_boxes(ss)
        ctx.set_option(L.CEL_OPT_STAR_TILES, 0)
        ss0 = cel.SourceSet(ctx, S, 5).set(typ, radec, counts)       # (a set of its own: boxes are cached per set)
        iset.render(ss0)
        bx0, st0 = iset.source_boxes(ss0)
        assert np.array_equal(bx5, bx0) and np.array_equal(st5, st0)
        again = run(1)
        assert again[1].tolist() == llb5.tolist()                     # run to run: the same bits
        assert again[2:] == ((0, 1) if (fell_back or seed in (3, 4)) else (1, 0))
        ll6, llb6, ns6, _ = run(1, store=False)                       # log-likelihood only
        assert llb6.tolist() == llb5.tolist()
        ctx.set_option(L.CEL_OPT_STAR_TILES, 1)
        iset.render(cel.SourceSet(ctx, 1, 5).set(typ[:1], radec[:1], counts[:1]))
        iset.render(ss)                                               # model images only
        np.testing.assert_array_equal(iset.model_images(), lam5)
        ll2, llb2, ns2, ng2 = run(2)
        lam2 = iset.model_images()
        assert (ns2, ng2) == ((0, 1) if seed in (3, 4) else (1, 0))      # 3, 4: exponents beyond the one-segment bound
        np.testing.assert_allclose(lam2, o_lam, rtol=RT_LAM_STRICT)
        np.testing.assert_allclose(llb2, o_ll, rtol=RT_LL)
        np.testing.assert_allclose(lam2, lam0, rtol=1e-13)
        np.testing.assert_allclose(llb2, llb0, rtol=1e-13)
        # model images only (no log-likelihood asked for: the bounded epilogue without its loads)
        ctx.set_option(L.CEL_OPT_STAR_TILES, 2)
        iset.render(cel.SourceSet(ctx, 1, 5).set(typ[:1], radec[:1], counts[:1]))
        iset.render(ss)
        np.testing.assert_array_equal(iset.model_images(), lam2)
        # log-likelihood only: nothing is stored (the images keep what the last storing render left)
        iset.render(cel.SourceSet(ctx, 1, 5).set(typ[:1], radec[:1], counts[:1]))
        keep = iset.model_images()
        ll3, llb3, ns3, _ = run(2, store=False)
        assert ns3 == (0 if seed in (3, 4) else 1)
        np.testing.assert_array_equal(llb3, llb2)
        np.testing.assert_array_equal(iset.model_images(), keep)
        # a galaxy in the catalogue: the general kernel
        typ_g = typ.copy(); typ_g[0] = 1
        shape_g = np.zeros((S, 4)); shape_g[0] = [0.5, 2.0, 30.0, 0.7]
        sg = cel.SourceSet(ctx, S, 5).set(typ_g, radec, counts, shape_g)
        assert run(2, sources=sg)[2:] == (0, 1)
        # types uploaded from device memory: the composition is unknown, the general kernel
        hip = C.CDLL(L.LIB_PATH)                  # dlsym through the library finds the HIP runtime IT is linked to
        hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        hip.hipFree.argtypes = [C.c_void_p]
        host = [typ, np.ascontiguousarray(radec), np.ascontiguousarray(counts), np.zeros((S, 4))]
        dptr = []
        for h in host:
            d = C.c_void_p()
            assert hip.hipMalloc(C.byref(d), max(h.nbytes, 8)) == 0
            assert hip.hipMemcpy(d, h.ctypes.data, h.nbytes, 1) == 0
            dptr.append(d)
        sd = cel.SourceSet(ctx, S, 5)
        sd.set_device(S, *[d.value for d in dptr])
        sd.S = S
        ll4, llb4, ns4, ng4 = run(2, sources=sd)
        assert (ns4, ng4) == (0, 1)
        np.testing.assert_array_equal(llb4, llb0)
        for d in dptr:
            hip.hipFree(d)
    finally:
        ctx.set_option(L.CEL_OPT_STAR_TILES, 1)
        ctx.profile(False)


@pytest.mark.parametrize("S", [4096, 4097])
def test_binning_forms_at_the_small_catalogue_limit(cel, ctx, orc, S):
    """k_bin_direct (one wave per tile, per-tile list segments) takes catalogues of up to 4096 sources, the super-tile
    kernels everything larger: the same crowded mixed field at 4096 and at 4097 sources, both against the oracle -- model
    pixels, log-likelihoods, the number of source-pixels -- and the tile lists' total length by both forms."""
    from desi_mcmc_amd import synth
    f = synth.SyntheticField(ctx, S, 2, 200, 232, frac_gal=0.3, seed=11)
    ll, llb = f.images.render(f.sources, loglik=True)
    o_lam, o_ll, o_st = orc.render_field(oracle_bands(f), f.H, f.W, f.src["type"], f.src["radec"], f.src["counts"],
                                         f.src["shape"], f.nelec)
    np.testing.assert_allclose(f.images.model_images(), o_lam, rtol=RT_LAM)
    np.testing.assert_allclose(llb, o_ll, rtol=RT_LL)
    st = f.images.stats()
    assert st["n_srcpix"] == o_st["n_srcpix"]
    # the first 4096 sources through the other form give the same lists' length as a 4096-source catalogue
    if S == 4097:
        sub = cel.SourceSet(ctx, 4096, 2).set(f.src["type"][:4096], f.src["radec"][:4096], f.src["counts"][:4096], f.src["shape"][:4096])
        f.images.render(sub)
        n_direct = f.images.stats()["n_tile_entries"]
        f.images.render(f.sources)
        n_super = f.images.stats()["n_tile_entries"]
        assert n_super >= n_direct > 0


def test_bench_line_contract_on_the_small_star_workload():
    """bench.py end to end on BASELINE configs[1] (stars1k_512): one JSON line with the contract's keys,
    a roofline object measured live, the extra legs, and a bounded cpu_baseline"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "stars1k_512", "--steps", "10", "--warmup", "2",
                        "--cpu-sample", "50"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 10 and d["dtype"] == "f64" and d["vs_baseline"] is None
    assert d["config"]["workload"] == "stars1k_512" and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and rf["launches"] == 10
    np.testing.assert_allclose(rf["frac"], rf["achieved"] / rf["peak"], rtol=1e-12)
    np.testing.assert_allclose(rf["achieved"], rf["algorithmic_bytes_per_launch"] / (rf["kernel_ms"] * 1e-3) / 1e9, rtol=1e-9)
    assert 0 < rf["kernel_ms"] < d["ms_per_step"]
    assert d["value"] > 1e9 and d["work"]["n_gauss_evaluated_per_step"] > 0
    assert d["ms_per_step_with_source_upload"] > 0 and d["python_api_ms"] > 0 and d["python_api_loglik_rel_diff"] < 1e-12
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["one_thread"]["cores"] == 1
    assert "reference_semantics_fullframe" in cb


def test_rccl_collective_path_in_a_one_rank_group():
    """No multi-GPU box is available to these tests; this at least runs the RCCL code path itself --
    process-group init with backend nccl (= RCCL on ROCm), the B-double all-reduce on device tensors,
    the all-gather variant and the pipelined reducer bench.py uses -- in a one-rank group on the GPU."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = '''
import os, sys
sys.path.insert(0, %r)
os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29517",
                  HSA_ENABLE_IPC_MODE_LEGACY="0")
import numpy as np, torch, torch.distributed as td
from desi_mcmc_amd import dist
torch.cuda.set_device(0)
td.init_process_group(backend="nccl", rank=0, world_size=1)
# bench.py's first-run insurance over RCCL: the collective counts its ranks, every rank names its device
call = dist.roll_call(1, 0)
assert call["ranks_seen_by_collective"] == 1 and call["backend"] == "nccl" and len(call["device_uuid"]) == 1 and not call["device_uuid"][0].startswith("cpu"), call
try:
    dist.roll_call(2, 0)
    raise SystemExit("roll_call(2) passed in a one-rank group")
except RuntimeError as e:
    assert "reached 1 rank" in str(e)
print("roll call over rccl:", call)
x = np.array([1.5, -2.0, 3.25, 4.0, 1e9])
assert np.array_equal(dist.allreduce_loglik(x, device=0, force=True), x)
assert np.array_equal(dist.allreduce_loglik(x, device=0, deterministic=True, force=True), x)
red = dist.LoglikReducer(5, device=0, depth=2, force=True)
assert red.active and red.gpu
got = []
for k in range(4):
    red.submit(x * (k + 1))
    if len(red.pending) > 1:
        got.append(red.result())
got += red.drain()
assert len(got) == 4 and all(np.array_equal(g, x * (k + 1)) for k, g in enumerate(got))
# sums straight from the library's device memory (ImageSet.loglik_device_ptr): two fields' per-band log-likelihoods added on
# the device and all-reduced, pipelined one step deep -- nothing crosses PCIe before the collective; a small star field
# (sums formed on the host by the one-launch path) comes the same way
import desi_mcmc_amd as cel
from desi_mcmc_amd import synth
ctx = cel.default_context(0)
fa = synth.SyntheticField(ctx, 120, 5, 200, 240, frac_gal=0.5, seed=3)
fb = synth.SyntheticField(ctx, 90, 5, 160, 224, frac_gal=0.0, seed=4)
red = dist.LoglikReducer(5, device=0, depth=2, force=True)
want, got = [], []
for k in range(3):
    la = fa.images.render(fa.sources, loglik=True)[1]
    lb = fb.images.render(fb.sources, loglik=True)[1]
    want.append(la + lb)
    red.submit_device([fa.images, fb.images])
    if len(red.pending) > 1:
        got.append(red.result())
got += red.drain()
assert len(got) == 3 and all(np.array_equal(g, w) for g, w in zip(got, want)), (got, want)
assert ctx.profile_get("small_stars")[1] == 0
ctx.profile(True)
fb.images.render(fb.sources, loglik=True)
assert ctx.profile_get("small_stars")[1] == 1          # the star field did take the one-launch path
ctx.profile(False)
# the dealt chain's exchange on persistent pinned / device buffers
deal = dist.SourceDeal(7, 1, 0, device=0)
deal.world = 1
X = np.arange(21.0).reshape(7, 3)
g1 = deal._gather(X); g2 = deal._gather(X + 1)
assert np.array_equal(g1[0], X) and np.array_equal(g2[0], X + 1) and len(deal._bufs) == 1
td.barrier()
td.destroy_process_group()
print("rccl one-rank ok")
''' % root
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "rccl one-rank ok" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])


@pytest.mark.parametrize("seed", _fuzz_seeds(8))
def test_fuzz_patch_loglik_vs_oracle(cel, ctx, orc, seed):
    """random sources, random patch rectangles around and beside them, SPARSE photon patches (most pixels hold
    no photon, some rows and whole chunks none: the conditional form skips their logs and crops to the photon
    rectangle): both forms of cel_patch_loglik against the oracle"""
    from desi_mcmc_amd import synth
    rs = np.random.RandomState(500 + seed)
    H, W = 260, 330
    B = 2
    bands = synth.make_bands(H, W, B)
    P = 6
    pix = np.column_stack([rs.uniform(40, W - 40, P), rs.uniform(40, H - 40, P)])
    typ = (rs.rand(P) < 0.6).astype(np.int32)
    radec = synth.pixel2equa(bands[0], pix)
    shape = np.column_stack([rs.uniform(0.05, 0.95, P), rs.uniform(0.5, 4.0, P), rs.uniform(0, 180, P), rs.uniform(0.2, 1.0, P)])
    counts = 10.0 ** rs.uniform(2.5, 5.0, size=(P, B))
    nelec = rs.poisson(150.0, size=(B, H, W)).astype(float)
    iset = cel.ImageSet(ctx, bands, H, W, nelec=nelec)
    sset = cel.SourceSet(ctx, P, B).set(typ, radec, counts, shape)
    ob = bands.copy()
    ob[:, 36] = [orc.checked_radius(ob[b], iset.band(b)[36]) for b in range(B)]
    for case in range(3):
        # one rectangle per band, placed near a random source, of a random size (up to several chunks)
        boxes = np.zeros((B, 4), dtype=np.int64)
        data = []
        for b in range(B):
            c = pix[rs.randint(P)] + rs.uniform(-15, 15, 2)
            hh, ww = rs.randint(3, 110), rs.randint(3, 90)
            y0 = int(np.clip(c[1] - hh // 2, 0, H - hh)); x0 = int(np.clip(c[0] - ww // 2, 0, W - ww))
            boxes[b] = [y0, y0 + hh, x0, x0 + ww]
            z = rs.poisson(0.15 if case else 2.0, size=(hh, ww)).astype(float)
            if case == 2:                       # photons in a few rows only
                keep = np.zeros(hh, dtype=bool); keep[rs.randint(0, hh, size=max(1, hh // 8))] = True
                z[~keep] = 0.0
            data.append(z)
        for isolated in (False, True):
            got = iset.patch_loglik(sset, boxes, data, isolated=isolated)
            for s in range(P):
                want = sum(orc.patch_loglik(ob[b], H, W, typ[s], radec[s], shape[s], counts[s, b], boxes[b], data[b],
                                            1 if isolated else 0) for b in range(B))
                tol = RT_LL * abs(want)
                if not isolated:
                    # every source is scored on every rectangle: one 200 px from its photons has a unit stamp in the SUBNORMAL range
                    # there, where whether a pixel counts at all (m > 0) depends on the order of the arithmetic -- in the
                    # reference's own evaluators too.  The oracle prices that range (patch_loglik_terms' fourth number; seeds
                    # 101, 144, 904, 1285 of a 1 500-seed run: up to two photons' worth, 1 466 nats of -4.8e6); elsewhere the
                    # tolerance is relative to the photon term's magnitudes
                    t = np.array([orc.patch_loglik_terms(ob[b], H, W, typ[s], radec[s], shape[s], counts[s, b], boxes[b], data[b]) for b in range(B)])
                    tol = RT_LL * t[:, 1].sum() + 8 * np.finfo(float).eps * t[:, 2].sum() + t[:, 3].sum()
                assert abs(got[s] - want) <= tol, "seed %d case %d src %d iso %s: %r against %r (tolerance %g)" % (seed, case, s, isolated, got[s], want, tol)
