"""Pin the CPU oracle (oracle/celeste_oracle.c) to the reference.

Every expected value here comes from tests/golden/*.npz, i.e. from the reference's own code run
in the build container (tests/golden/make_golden.py).  CPU only.
Tolerances: the oracle restates the same fp64 arithmetic, so agreement is ~1e-13; the asserted
bound is 1e-10 relative (four orders inside the 1e-6 parity bar), boxes bit-exact.
"""
import numpy as np
import pytest

from conftest import load_golden, unpack_ragged
from oracle import oracle as orc

RTOL = 1e-10


@pytest.fixture(scope="module")
def bands():
    rec = load_golden("bands_253.npz")
    return rec, orc.pack_bands(rec)


def test_band_struct_layout():
    assert orc.lib().orc_band_doubles() == orc.BAND_DOUBLES


def test_profile_tables_match_reference():
    g = load_golden("radius.npz")
    ea, ev, da, dv = orc.profile_tables()
    for a, b in ((ea, g["exp_amp"]), (ev, g["exp_var"]), (da, g["dev_amp"]), (dv, g["dev_var"])):
        np.testing.assert_allclose(a, b, rtol=1e-15)


def test_wcs_roundtrip_and_cd(bands):
    rec, B = bands
    g = load_golden("wcs_points.npz")
    for bi in (0, 2):
        for p, e, pb, cd in zip(g["pix"], g["equa_b%d" % bi], g["pix_back_b%d" % bi], g["cd_b%d" % bi]):
            np.testing.assert_allclose(orc.pixel2equa(B[bi], p), e, rtol=1e-15)
            np.testing.assert_allclose(orc.equa2pixel(B[bi], e), pb, rtol=1e-12, atol=1e-10)
            np.testing.assert_allclose(orc.cd_at_pixel(B[bi], p[0], p[1]), cd, rtol=1e-9, atol=1e-18)


def test_cd_at_pixel_varies_across_wide_frame(bands):
    """Q9: cd_at_pixel is source-dependent (~4e-4 across 2048 px); the oracle must follow it."""
    rec, B = bands
    g = load_golden("wcs_points.npz")
    band = B[int(g["big_band"])].copy()
    band[24:26] = [2048 / 2.0, 64 / 2.0]  # rho = CRPIX - 1 of the virtual frame
    for p, e, cd in zip(g["big_pix"], g["big_equa"], g["big_cd"]):
        np.testing.assert_allclose(orc.pixel2equa(band, p), e, rtol=1e-15)
        np.testing.assert_allclose(orc.cd_at_pixel(band, p[0], p[1]), cd, rtol=1e-8, atol=1e-18)
    assert abs(g["big_cd"][0][0, 0] / g["big_cd"][1][0, 0] - 1) > 1e-6


def test_notebook_known_answer_pixel():
    """notebooks/RenderSources.ipynb cell 1: source 0 of cat-188.3444-63.4421 lands on (26.0470, 26.1040)."""
    g = load_golden("wcs_points.npz")
    band = np.zeros(orc.BAND_DOUBLES)
    band[24:26], band[26:28], band[32:36] = g["nb_rho"], g["nb_phi"], g["nb_ups_inv"].ravel()
    v = orc.equa2pixel(band, g["nb_u"])
    np.testing.assert_allclose(v, g["nb_pix"], rtol=1e-12)
    # the notebook output predates the `- 1` of fits_image.py:99 (1-based FITS pixel): it is
    # the current code's answer + 1 in both axes, to the 4 decimals the notebook printed
    np.testing.assert_allclose(v + 1.0, g["nb_pix_notebook"], atol=5e-5)


def test_reference_own_gmm_test_seed41():
    """CelestePy/test/test_gmm.py:63-105 -- K=42, 200x200 grid, seed 41."""
    g = load_golden("evaluator.npz")
    got = orc.gmm_like_2d(g["X"], g["ws"], g["means"], g["covs"])
    np.testing.assert_allclose(got, g["gmm_prob"], rtol=1e-9, atol=1e-300)
    ll = orc.mog_loglike(g["X"], g["means"], g["invcovs"], np.exp(g["logdets"]), g["ws"])
    np.testing.assert_allclose(ll, g["mog_loglike"], rtol=1e-10, atol=1e-10)
    # numpy restatement of mog.py agrees with the C one
    ll_np = orc.np_mog_loglike(g["X"][::37], g["means"], g["invcovs"], np.exp(g["logdets"]), g["ws"])
    np.testing.assert_allclose(ll_np, g["mog_loglike"][::37], rtol=1e-11, atol=1e-11)


def test_gmm_like_negative_weight_and_shapes():
    """Q8: the direct-sum evaluator accepts negative weights (mog_loglike would give NaN)."""
    x = np.array([[0.0, 0.0], [1.0, -1.0]])
    ws = np.array([1.2, -0.2])
    mus = np.zeros((2, 2))
    sigs = np.array([np.eye(2), 4 * np.eye(2)])
    got = orc.gmm_like_2d(x, ws, mus, sigs)
    exp = 1.2 * np.exp(-0.5 * (x ** 2).sum(1)) / (2 * np.pi) - 0.2 * np.exp(-0.125 * (x ** 2).sum(1)) / (8 * np.pi)
    np.testing.assert_allclose(got, exp, rtol=1e-14)


def test_bounding_radius(bands):
    rec, B = bands
    g = load_golden("radius.npz")
    for b in range(5):
        r = orc.bounding_radius(rec["weights"][b], rec["means"][b], rec["covars"][b], 0.001)
        np.testing.assert_allclose(r, g["psf_R_1e3"][b], rtol=1e-13)
        np.testing.assert_allclose(r, rec["R"][b], rtol=1e-13)
        r5 = orc.bounding_radius(rec["weights"][b], rec["means"][b], rec["covars"][b], 1e-5, center=(0.3, -0.2))
        np.testing.assert_allclose(r5, g["psf_R_1e5_c"][b], rtol=1e-13)
    # the survey's quoted r-band value
    np.testing.assert_allclose(rec["R"][2], 21.4426, atol=1e-4)


def test_star_stamps_incl_edges_and_q1(bands):
    rec, B = bands
    g = load_golden("star_stamps.npz")
    patches = unpack_ragged(g["flat"], g["offs"], g["shapes"])
    n_none = 0
    for i, (bi, u, box, none) in enumerate(zip(g["band"], g["u"], g["box"], g["is_none"])):
        patch, yl, xl = orc.source_patch(B[bi], 51, 51, 0, u)
        ok, v, obox = orc.star_box(B[bi], 51, 51, u)
        if none:
            n_none += 1
            assert not ok and patch is None
            continue
        assert ok
        assert (yl[0], yl[1], xl[0], xl[1]) == tuple(int(t) for t in box), (i, box)
        if patches[i].size == 0:
            assert patch is None
        else:
            np.testing.assert_allclose(patch, patches[i], rtol=RTOL, atol=1e-300)
    assert n_none >= 1  # the Q1 branch is exercised


def test_star_caller_limits_and_full_frame(bands):
    rec, B = bands
    g = load_golden("star_stamps.npz")
    v = orc.equa2pixel(B[2], g["lim_u"])
    box = np.array([g["lim_ylim"][0], g["lim_ylim"][1], g["lim_xlim"][0], g["lim_xlim"][1]], dtype=np.int32)
    patch = np.empty((box[1] - box[0], box[3] - box[2]))
    import ctypes as C
    dp = C.POINTER(C.c_double)
    band = np.ascontiguousarray(B[2])
    orc.lib().orc_star_patch(band.ctypes.data_as(dp), v.ctypes.data_as(dp),
                             box.ctypes.data_as(C.POINTER(C.c_int)), patch.ctypes.data_as(dp))
    np.testing.assert_allclose(patch, g["lim_patch"], rtol=RTOL)
    p, yl, xl = orc.source_patch(B[2], 51, 51, 0, g["lim_u"])
    full = np.zeros((51, 51))
    full[yl[0]:yl[1], xl[0]:xl[1]] = p
    np.testing.assert_allclose(full, g["full_image"], rtol=RTOL)


@pytest.mark.parametrize("tag", ["s", "b"])
def test_galaxy_tables_boxes_patches(bands, tag):
    rec, B = bands
    g = load_golden("galaxy_stamps.npz")
    H, W = (51, 51) if tag == "s" else (int(g["big_H"]), int(g["big_W"]))
    stride = int(g[tag + "_stride"])
    patches = unpack_ragged(g[tag + "_flat"], g[tag + "_offs"], g[tag + "_shapes"])
    for i in range(len(patches)):
        band = B[g[tag + "_band"][i]].copy()
        if tag == "b":
            band[24:26] = [W / 2.0, H / 2.0]
        th, u = g[tag + "_th"][i], g[tag + "_u"][i]
        pis, means, covs, pxy, tinv = orc.galaxy_table(band, th, u)
        np.testing.assert_allclose(pxy, g[tag + "_pix"][i], rtol=1e-12)
        np.testing.assert_allclose(tinv, g[tag + "_tinv"][i], rtol=1e-9)
        np.testing.assert_allclose(pis, g[tag + "_cw"][i], rtol=1e-13)
        np.testing.assert_allclose(means, g[tag + "_cm"][i], rtol=1e-13)
        np.testing.assert_allclose(covs, g[tag + "_cc"][i], rtol=1e-9)
        bound = orc.bounding_radius(pis, means, covs, 1e-5, center=pxy)
        np.testing.assert_allclose(bound, g[tag + "_bound"][i], rtol=1e-9)
        patch, yl, xl = orc.source_patch(band, H, W, 1, u, th)
        assert (yl[0], yl[1], xl[0], xl[1]) == tuple(int(t) for t in g[tag + "_box"][i])
        np.testing.assert_allclose(patch.sum(), g[tag + "_sum"][i], rtol=1e-10)
        np.testing.assert_allclose(patch[::stride, ::stride], patches[i], rtol=RTOL, atol=1e-300)


def test_mini_field_lambda_ll_and_patches():
    g = load_golden("mini_field.npz")
    B = orc.pack_bands(g)
    H, W = int(g["H"]), int(g["W"])
    counts = g["flux"] / g["calib"][None, :] * g["kappa"][None, :]  # celeste.py:80-81,94
    lam, ll, stats = orc.render_field(B, H, W, g["is_gal"], g["radec"], counts, g["shape"], g["nelec"])
    np.testing.assert_allclose(lam, g["lam"], rtol=RTOL)
    np.testing.assert_allclose(ll, g["ll_band"], rtol=1e-12)
    np.testing.assert_allclose(ll.sum(), g["ll"], rtol=1e-12)
    # per-source counts-scaled patches (gen_src_image_with_fluxes, celeste.py:84-96)
    patches = unpack_ragged(g["patch_flat"], g["patch_offs"], g["patch_shapes"])
    S = len(g["is_gal"])
    npix = 0
    for b in range(5):
        for s in range(S):
            i = b * S + s
            p, yl, xl = orc.source_patch(B[b], H, W, g["is_gal"][s], g["radec"][s], g["shape"][s])
            assert (yl[0], yl[1], xl[0], xl[1]) == tuple(g["patch_box"][i])
            np.testing.assert_allclose(p * counts[s, b], patches[i], rtol=RTOL, atol=1e-300)
            npix += p.size
    assert stats["n_srcpix"] == npix


def test_mini_field_stars_only_via_reference_gen_model_image():
    """The star-only golden came from the reference's own gen_model_image/celeste_likelihood."""
    g = load_golden("mini_field.npz")
    B = orc.pack_bands(g)
    H, W = int(g["H"]), int(g["W"])
    idx = g["star_idx"]
    counts = (g["flux"] / g["calib"][None, :] * g["kappa"][None, :])[idx]  # nmgy2counts, fits_image.py:183
    lam, ll, _ = orc.render_field(B, H, W, np.zeros(len(idx), np.int32), g["radec"][idx], counts,
                                  g["shape"][idx], g["nelec"])
    np.testing.assert_allclose(lam, g["star_lam"], rtol=RTOL)
    np.testing.assert_allclose(ll.sum(), g["star_ll"], rtol=1e-12)
    # the full-frame-per-source restatement of celeste.py:203-219 gives the same image
    for b in (0, 3):
        ff = orc.gen_model_image_fullframe(B[b], H, W, g["radec"][idx], counts[:, b])
        np.testing.assert_allclose(ff, g["star_lam"][b], rtol=RTOL)


def test_config1_real_stamps(bands):
    rec, B = bands
    g = load_golden("config1.npz")
    cat = g["cat"]
    counts = cat[:, 2:] * rec["kappa"][None, :]       # a is None: kappa * flux (celeste.py:52-55, Q2)
    S = cat.shape[0]
    lam, ll, _ = orc.render_field(B, 51, 51, np.zeros(S, np.int32), cat[:, :2], counts, np.zeros((S, 4)),
                                  rec["nelec"])
    np.testing.assert_allclose(lam, g["lam"], rtol=RTOL)
    np.testing.assert_allclose(ll, g["ll_band"], rtol=1e-12)
    np.testing.assert_allclose(ll.sum(), g["ll"], rtol=1e-12)
    # config 1 proper: one centred r-band star with catalogue flux
    c1 = g["one_flux"][2] / rec["calib"][2] * rec["kappa"][2]
    p, yl, xl = orc.source_patch(B[2], 51, 51, 0, g["one_u"])
    np.testing.assert_allclose(p * c1, g["one_patch"], rtol=RTOL)
    lam1, ll1, _ = orc.render_field(B[2:3], 51, 51, np.zeros(1, np.int32), g["one_u"][None, :],
                                    np.array([[c1]]), np.zeros((1, 4)), rec["nelec"][2:3])
    np.testing.assert_allclose(lam1[0], g["one_lam"], rtol=RTOL)
    np.testing.assert_allclose(ll1[0], g["one_ll"], rtol=1e-12)


def test_poisson_loglike_mask():
    rs = np.random.RandomState(0)
    m = rs.uniform(0.5, 5, 100)
    m[::9] = 0.0
    d = rs.poisson(3, 100).astype(float)
    mask = rs.rand(100) > 0.3
    good = (m > 0) & mask
    exp = np.sum(np.log(m[good]) * d[good]) - np.sum(m[good])   # sources.py:6-12
    np.testing.assert_allclose(orc.poisson_loglike(d, m, mask), exp, rtol=1e-13)


def test_source_conditional_loglik_golden():
    """Source.log_likelihood / log_likelihood_isolated (sources.py:134-237) run by the reference"""
    g = load_golden("source_ll.npz")
    B = orc.pack_bands(g)
    H, W = int(g["H"]), int(g["W"])
    used = g["bands_used"]
    for ci in range(int(g["ncases"])):
        kind = int(g["c%d_kind" % ci])
        zs = unpack_ragged(g["c%d_z" % ci], g["c%d_zoffs" % ci], g["c%d_zshapes" % ci])
        boxes = g["c%d_boxes" % ci]
        us, fl, sh = g["c%d_us" % ci], g["c%d_fl" % ci], g["c%d_sh" % ci]
        for mode, key in ((0, "ll0"), (1, "ll1")):
            exp = g["c%d_%s" % (ci, key)]
            for p in range(len(exp)):
                ll = 0.0
                for j, b in enumerate(used):
                    counts = fl[p, b] / g["calib"][b] * g["kappa"][b]      # flux_in_image, sources.py:120-129
                    data = zs[j] if mode == 0 else g["nelec"][b, boxes[j, 0]:boxes[j, 1], boxes[j, 2]:boxes[j, 3]]
                    ll += orc.patch_loglik(B[b], H, W, kind, us[p], sh[p], counts, boxes[j], data, mode)
                np.testing.assert_allclose(ll, exp[p], rtol=1e-12)
                if mode == 0:       # the same value from its terms taken apart (what the photon-list GPU test compares)
                    tt = np.zeros(4)
                    for j, b in enumerate(used):
                        tt += orc.patch_loglik_terms(B[b], H, W, kind, us[p], sh[p], fl[p, b] / g["calib"][b] * g["kappa"][b],
                                                     boxes[j], zs[j])
                    np.testing.assert_allclose(tt[0] - tt[2], exp[p], rtol=1e-12)
                    assert tt[1] >= abs(tt[0]) and tt[3] == 0.0
        if kind == 0:
            assert g["c%d_ll0" % ci][-1] < 0     # the overlap-miss proposal: -flux * sum(weights) only


def test_estep_statistics_golden():
    """celeste_em.py:38-91 reductions of the reference's gen_src_prob_layers (stars, 5 bands)"""
    g = load_golden("mini_field.npz")
    e = load_golden("estep.npz")
    B = orc.pack_bands(g)
    H, W = int(g["H"]), int(g["W"])
    idx = e["star_idx"]
    counts = (g["flux"] / g["calib"][None, :] * g["kappa"][None, :])[idx]
    xt, ms, nz = orc.estep_stats(B, H, W, np.zeros(len(idx), np.int32), g["radec"][idx], counts,
                                 g["shape"][idx], g["nelec"])
    np.testing.assert_allclose(xt, e["xtilde"], rtol=1e-11)
    np.testing.assert_allclose(ms, e["mass"], rtol=1e-11)
    np.testing.assert_allclose(nz, e["noise"], rtol=1e-12)
    # every observed photon is attributed to a source or to the sky
    np.testing.assert_allclose(xt.sum(axis=0) + nz, g["nelec"].sum(axis=(1, 2)), rtol=1e-12)


# ---- the older per-profile galaxy route ---------------------------------------
# The reference's celeste_fast.pyx does not build here, so these restatements are pinned through
# goldens the reference's OWN Python produced: the convolved tables of MixtureOfGaussians (A15:
# *_cw / *_cm / *_cc, galaxy-major) are the same table in another order -- the PSF-major A16
# output must be a permutation of them for the same W = Tinv Tinv^T -- and the per-profile image
# route (A18) must reproduce the reference's gen_galaxy_psf_image patches (A17) wherever the two
# routes' shape matrices coincide (a source at the WCS reference declination).
def _a16_perm():
    # galaxy-major index j*3 + k  ->  PSF-major index k*14 + j
    return np.array([(c % 14) * 3 + c // 14 for c in range(42)])


@pytest.mark.parametrize("tag", ["s", "b"])
def test_a16_mixture_params_are_a_permutation_of_reference_tables(bands, tag):
    rec, B = bands
    g = load_golden("galaxy_stamps.npz")
    ea, ev, da, dv = orc.profile_tables()
    perm = _a16_perm()
    for i in range(len(g[tag + "_th"])):
        bi = g[tag + "_band"][i]
        th = g[tag + "_th"][i]
        tinv = g[tag + "_tinv"][i]
        Wm = tinv @ tinv.T
        w, m, c = orc.galaxy_psf_mixture_params([th[0], 1. - th[0]], Wm, g[tag + "_pix"][i], rec["weights"][bi],
                                                rec["means"][bi], rec["covars"][bi], ea, ev, da, dv)
        np.testing.assert_allclose(w, g[tag + "_cw"][i][perm], rtol=1e-13)
        np.testing.assert_allclose(m, g[tag + "_cm"][i][perm], rtol=1e-13)
        np.testing.assert_allclose(c, g[tag + "_cc"][i][perm], rtol=1e-9, atol=1e-18)
        # the one-profile form with concatenated (theta-scaled) tables is the same table
        w2, m2, c2 = orc.galaxy_prof_psf_mixture_params(
            Wm, g[tag + "_pix"][i], rec["weights"][bi], rec["means"][bi], rec["covars"][bi],
            np.concatenate([th[0] * ea, (1. - th[0]) * da]), np.concatenate([ev, dv]))
        np.testing.assert_allclose(w2, w, rtol=1e-15)
        assert np.array_equal(m2, m) and np.array_equal(c2, c)


def test_a18_profile_images_reproduce_reference_patches_at_the_wcs_reference_dec(bands):
    """theta f_exp + (1-theta) f_dev through A16 + A5 + A7 with R from the CONSTANT Ups_n equals the
    reference's gen_galaxy_psf_image patch (golden, A17 route) when the source sits at the frame's
    reference declination, where cd_at_pixel == Ups_n; away from it the two differ at ~1e-4 (Q9)."""
    rec, B = bands
    g = load_golden("galaxy_stamps.npz")
    patches = unpack_ragged(g["s_flat"], g["s_offs"], g["s_shapes"])
    stride = int(g["s_stride"])
    n_same = 0
    for i in range(len(patches)):
        bi = g["s_band"][i]
        band = B[bi]
        th, u = g["s_th"][i], g["s_u"][i]
        box = [int(t) for t in g["s_box"][i]]
        cd = orc.cd_at_pixel(band, *g["s_pix"][i])
        R = orc.galaxy_tinv(th[1], th[3], th[2], rec["ups"][bi])
        fe, yl, xl = orc.galaxy_prof_psf_image(band, 51, 51, "exp", R, u, lims=box)
        fd, _, _ = orc.galaxy_prof_psf_image(band, 51, 51, "dev", R, u, lims=box)
        assert (yl, xl) == ((box[0], box[1]), (box[2], box[3]))
        f = th[0] * fe + (1. - th[0]) * fd
        rel = np.max(np.abs(cd - rec["ups"][bi]) / np.abs(rec["ups"][bi]).max())
        ref = patches[i]
        got = f[::stride, ::stride]
        if rel < 1e-9:
            n_same += 1
            np.testing.assert_allclose(got, ref, rtol=1e-7, atol=1e-300)
        else:
            # same mixture up to the ~1e-5..1e-3 change of the shape matrix
            big = ref > 1e-6 * ref.max()
            assert np.max(np.abs(got[big] / ref[big] - 1.0)) < 50 * rel + 1e-9
        # own box of the route: int() rule about v_s with the profile's 1e-5 bound
        pe, yle, xle = orc.galaxy_prof_psf_image(band, 51, 51, "exp", R, u)
        ea, ev, da, dv = orc.profile_tables()
        w, m, c = orc.galaxy_prof_psf_mixture_params(R @ R.T, g["s_pix"][i], rec["weights"][bi], rec["means"][bi],
                                                     rec["covars"][bi], ea, ev)
        bound = orc.bounding_radius(w, m, c, 1e-5, center=g["s_pix"][i])
        px, py = g["s_pix"][i]
        assert xle == (max(0, int(px - bound)), min(int(px + bound + 1), 51))
        assert yle == (max(0, int(py - bound)), min(int(py + bound + 1), 51))
        if pe is not None:
            np.testing.assert_allclose(pe, orc.gmm_like_2d(
                np.column_stack([a.ravel() for a in np.meshgrid(np.arange(xle[0], xle[1], dtype=float),
                                                                np.arange(yle[0], yle[1], dtype=float), indexing="xy")]),
                w, m, c).reshape(pe.shape), rtol=1e-13)
    # exactly on the reference declination cd_at_pixel == Ups_n: the two routes give the same patch
    for bi, th in ((2, g["s_th"][3]), (0, g["s_th"][30]), (4, g["s_th"][55])):
        band = B[bi]
        u = np.array([rec["phi"][bi][0] + 7e-4, rec["phi"][bi][1]])
        p17, yl, xl = orc.source_patch(band, 51, 51, 1, u, th)
        box = [yl[0], yl[1], xl[0], xl[1]]
        R = orc.galaxy_tinv(th[1], th[3], th[2], rec["ups"][bi])
        fe, _, _ = orc.galaxy_prof_psf_image(band, 51, 51, "exp", R, u, lims=box)
        fd, _, _ = orc.galaxy_prof_psf_image(band, 51, 51, "dev", R, u, lims=box)
        np.testing.assert_allclose(th[0] * fe + (1. - th[0]) * fd, p17, rtol=1e-9, atol=1e-300)   # cd_at_pixel is a finite difference: ~1e-12 off Ups_n


def test_galaxy_source_like_oracle_matches_a_numpy_statement(bands):
    rec, B = bands
    g = load_golden("galaxy_stamps.npz")
    rs = np.random.RandomState(5)
    for i in (0, 7, 20, 41):
        bi = g["s_band"][i]
        band = B[bi]
        th, u = g["s_th"][i], g["s_u"][i]
        box = [5, 47, 3, 50]
        Z = rs.poisson(3.0, size=(42, 47)).astype(float)
        flux = 1234.5
        R = orc.galaxy_tinv(th[1], th[3], th[2], rec["ups"][bi])
        fe, _, _ = orc.galaxy_prof_psf_image(band, 51, 51, "exp", R, u, lims=box)
        fd, _, _ = orc.galaxy_prof_psf_image(band, 51, 51, "dev", R, u, lims=box)
        lam = flux * (th[0] * fe + (1. - th[0]) * fd)
        ok = lam > 0
        want = np.sum(Z[ok] * np.log(lam[ok])) - np.sum(lam[ok])
        np.testing.assert_allclose(orc.galaxy_source_like(band, 51, 51, th, u, flux, box, Z), want, rtol=1e-12)


def test_patch_loglik_terms_of_a_per_profile_source():
    """type 2 of orc_patch_loglik_terms (shape = theta, W00, W01, W11) is galaxy_source_like's photon term
    (celeste_galaxy_conditionals.py:15-42: the value at Z minus the value at Z = 0) when W = R R^T of the same shape"""
    g = load_golden("mini_field.npz")
    B = orc.pack_bands(g)
    H, W = int(g["H"]), int(g["W"])
    rs = np.random.RandomState(7)
    th = np.array([0.35, 1.4, 50.0, 0.55])
    u = g["radec"][int(np.nonzero(g["is_gal"])[0][0])]
    for b in (0, 2, 4):
        R = orc.galaxy_tinv(th[1], th[3], th[2], B[b][28:32].reshape(2, 2))
        Wm = R @ R.T
        v = orc.equa2pixel(B[b], u)
        x0, y0 = int(v[0]) - 9, int(v[1]) - 7
        box = np.array([max(y0, 0), min(y0 + 17, H), max(x0, 0), min(x0 + 21, W)], dtype=np.int32)
        Z = rs.poisson(3.0, size=(box[1] - box[0], box[3] - box[2])).astype(np.float64)
        a = orc.galaxy_source_like(B[b], H, W, th, u, 900.0, box, Z)
        a0 = orc.galaxy_source_like(B[b], H, W, th, u, 900.0, box, np.zeros_like(Z))
        t = orc.patch_loglik_terms(B[b], H, W, 2, u, [th[0], Wm[0, 0], Wm[0, 1], Wm[1, 1]], 900.0, box, Z)
        np.testing.assert_allclose(t[0], a - a0, rtol=1e-11)
        np.testing.assert_allclose(t[2], 900.0 * B[b][3:6].sum(), rtol=1e-15)


def test_every_real_field_the_reference_ships():
    """tests/golden/real_fields.npz: the reference's own run on all 100 real fields in its tree (500 images, 221 catalogue
    sources as util/misc/init_utils.py:42-60 loads them) -- model images at 1e-10 (whole for data/stamps + data/real, every
    4th pixel of every 4th row elsewhere), per-image and multi-image log-likelihoods at 1e-12, every source's box exact,
    star stamps and gen_galaxy_psf_image patches (float limits, celeste_galaxy_conditionals.py:185-214)."""
    from conftest import real_fields
    g, fields = real_fields()
    assert len(fields) == 100 and sum(len(f["radec"]) for f in fields) == 221
    for f in fields:
        B = orc.pack_bands(f["rec"])
        S = len(f["radec"])
        counts = f["flux"] * f["rec"]["kappa"][None, :]          # a is None: kappa * flux (celeste.py:52-55, Q2)
        lam, ll, _ = orc.render_field(B, f["H"], f["W"], np.zeros(S, np.int32), f["radec"].reshape(S, 2), counts.reshape(S, 5),
                                      np.zeros((S, 4)), f["nelec"])
        if f["lam"] is not None:
            np.testing.assert_allclose(lam, f["lam"], rtol=RTOL, err_msg=f["name"])
        else:
            np.testing.assert_allclose(lam[:, ::4, ::4], f["lam_sub"], rtol=RTOL, err_msg=f["name"])
        np.testing.assert_allclose(ll, f["ll_band"], rtol=1e-12, err_msg=f["name"])
        np.testing.assert_allclose(ll.sum(), f["ll"], rtol=1e-12)
        for s in range(S):
            for b in range(5):
                ok, v, box = orc.star_box(B[b], f["H"], f["W"], f["radec"][s])
                assert (not ok) == bool(f["src_none"][s, b])
                if ok:
                    assert list(box) == list(f["src_box"][s, b]), (f["name"], s, b)
    by_index = {f["index"]: f for f in fields}
    stamps = unpack_ragged(g["st_flat"], g["st_offs"], g["st_shapes"])
    assert len(stamps) >= 40
    for k, want in enumerate(stamps):
        f = by_index[int(g["st_field"][k])]
        p, yl, xl = orc.source_patch(orc.pack_bands(f["rec"])[2], f["H"], f["W"], 0, f["radec"][int(g["st_src"][k])])
        assert [yl[0], yl[1], xl[0], xl[1]] == list(g["st_box"][k])
        np.testing.assert_allclose(p, want, rtol=RTOL)
    gal = unpack_ragged(g["g_flat"], g["g_offs"], g["g_shapes"])
    assert len(gal) == 36
    for k, want in enumerate(gal):
        f = by_index[int(g["g_field"][k])]
        p, yl, xl = orc.source_patch(orc.pack_bands(f["rec"])[int(g["g_band"][k])], f["H"], f["W"], 1, g["g_u"][k], g["g_th"][k])
        assert [float(yl[0]), float(yl[1]), float(xl[0]), float(xl[1])] == list(g["g_box"][k])
        np.testing.assert_allclose(p, want, rtol=RTOL)
