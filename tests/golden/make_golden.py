#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by RUNNING THE REFERENCE.

Run once, in the build container (where /root/reference is mounted):

    python tests/golden/make_golden.py

Every number written here is the output of the reference's own code
(CelestePy/{fits_image,celeste,celeste_galaxy_conditionals}.py,
util/dists/mog.py, util/like/gmm_like.py, util/bound/bounding_box.py) executed
through ``_refload`` (in-memory lib2to3, third-party stand-ins only; see its
docstring).  The fixtures are data: inputs + expected outputs.  No reference
source text is stored.

Fixtures
--------
bands_253.npz      A1  FitsImage fields for stamp-{ugriz}-253.1147-11.6072 (+ nelec, catalogue)
wcs_points.npz     A3/A4 equa2pixel / pixel2equa / cd_at_pixel samples (incl. the notebook's
                   known-answer pixel (26.0470, 26.1040), notebooks/RenderSources.ipynb)
evaluator.npz      A6/A7 the reference's own test (test/test_gmm.py:63-105, seed 41, K=42)
radius.npz         A5  calc_bounding_radius
star_stamps.npz    A8  gen_point_source_psf_image (patch + box, edge cases, Q1)
galaxy_stamps.npz  A14/A15/A17 gen_galaxy_transformation, component tables, gen_galaxy_psf_image
mini_field.npz     A9-A11 + Q3 extension: mixed star/galaxy field, 5 bands, 96x80
config1.npz        config 1: real stamps, catalogue sources, gen_model_image + celeste_likelihood
source_ll.npz      (f)1 Source.log_likelihood / log_likelihood_isolated (sources.py:134-237)
estep.npz          (f)4 gen_src_prob_layers reductions (celeste_em.py:38-91)
src_bound.npz      celeste.gen_psf_src_image_bound (celeste.py:193-199), the boxes of process_field.py:107-117,
                   FitsImage.make_pixel_grid / pixel_grid (fits_image.py:95,186-194)
slicesample.npz    config 5: slicesample (util/infer/slicesample.py:89-227) with every draw it made recorded
real_fields.npz    configs 1 / 4 on EVERY real field the reference ships (data/stamps 11, data/stamp_catalog 63,
                   data/galaxy_stamps 25, data/real 1 = 100 fields, 500 images, 221 catalogue sources): FitsImage
                   records + nelec, the catalogue as util/misc/init_utils.py:42-60 loads it, gen_model_image,
                   celeste_likelihood[_multi_image], every source's box, star stamps and gen_galaxy_psf_image patches
"""
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
warnings.simplefilter("ignore")

import _refload  # noqa: E402

_refload.install()

import CelestePy.fits_image as ref_fits  # noqa: E402
import CelestePy.celeste as ref_cel  # noqa: E402
import CelestePy.celeste_galaxy_conditionals as ref_gal  # noqa: E402
import CelestePy.util.dists.mog as ref_mog  # noqa: E402
import CelestePy.util.like as ref_like  # noqa: E402
import CelestePy.util.bound.bounding_box as ref_bb  # noqa: E402
from CelestePy.celeste_src import SrcParams  # noqa: E402

BANDS = ["u", "g", "r", "i", "z"]
STAMP_DIR = os.path.join(_refload.REF_ROOT, "data", "stamps")
FIELD = "253.1147-11.6072"


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrs)
    print("wrote %-20s %7.1f KB  %s" % (name, os.path.getsize(path) / 1024.0, sorted(arrs)))


def ref_images(field=FIELD):
    tmpl = os.path.join(STAMP_DIR, "stamp-%s-" + field + ".fits")
    return [ref_fits.FitsImage(b, fits_file_template=tmpl) for b in BANDS]


def band_record(img):
    """The A1 fields the hot path consumes, as a flat dict of arrays."""
    return dict(
        eps=img.epsilon, kappa=img.kappa, calib=img.calib,
        weights=img.weights, means=img.means, covars=img.covars,
        invcovars=img.invcovars, logdets=img.logdets,
        rho=img.rho_n, phi=img.phi_n, ups=img.Ups_n, ups_inv=img.Ups_n_inv, R=img.R)


def stack_bands(imgs):
    recs = [band_record(i) for i in imgs]
    return {k: np.array([r[k] for r in recs], dtype=np.float64) for k in recs[0]}


def virtual_images(name, H, W, nelec=None, field=FIELD):
    """Reference FitsImage objects for a synthetic H x W frame per band.

    Header = the real stamp's header with NAXIS/CRPIX moved to the frame centre
    (SURVEY 8d); pixel data chosen so that the reference's own
    ``round((img/CALIB + SKY) * GAIN)`` reproduces ``nelec`` exactly.
    """
    imgs = []
    for bi, b in enumerate(BANDS):
        hdr, _ = _refload.fits_image(os.path.join(STAMP_DIR, "stamp-%s-%s.fits" % (b, field)))
        hdr = dict(hdr)
        hdr["NAXIS1"], hdr["NAXIS2"] = W, H
        hdr["CRPIX1"], hdr["CRPIX2"] = W / 2.0 + 1.0, H / 2.0 + 1.0
        ne = np.zeros((H, W)) if nelec is None else nelec[bi]
        pix = (ne / hdr["GAIN"] - hdr["SKY"]) * hdr["CALIB"]
        _refload.VIRTUAL_FITS["virt:%s-%s" % (name, b)] = (hdr, pix)
    tmpl = "virt:" + name + "-%s"
    for b in BANDS:
        im = ref_fits.FitsImage(b, fits_file_template=tmpl)
        imgs.append(im)
    if nelec is not None:
        for bi, im in enumerate(imgs):
            assert np.array_equal(im.nelec, nelec[bi]), "nelec round trip failed"
    return imgs


def pack_ragged(patches):
    """list of 2-D arrays -> (flat values, offsets[n+1], shapes[n,2])"""
    offs = np.zeros(len(patches) + 1, dtype=np.int64)
    shapes = np.zeros((len(patches), 2), dtype=np.int64)
    for i, p in enumerate(patches):
        shapes[i] = p.shape
        offs[i + 1] = offs[i] + p.size
    flat = np.concatenate([np.asarray(p, dtype=np.float64).ravel() for p in patches]) if patches else np.zeros(0)
    return flat, offs, shapes


# ---------------------------------------------------------------------------
def gen_bands():
    imgs = ref_images()
    rec = stack_bands(imgs)
    rec["nelec"] = np.array([i.nelec for i in imgs])
    _, cat = _refload.fits_bintable(os.path.join(STAMP_DIR, "cat-%s.fits" % FIELD))
    rec["cat_radec"] = np.column_stack([cat["ra"], cat["dec"]]).astype(np.float64)
    rec["cat_flux"] = np.column_stack([cat["psfflux_" + b] for b in BANDS]).astype(np.float64)
    save("bands_253.npz", **rec)
    return imgs


def gen_wcs(imgs):
    rs = np.random.RandomState(7)
    pix = np.vstack([rs.uniform(-20, 70, size=(12, 2)), [[25.0, 25.0], [0.0, 0.0], [50.0, 50.0]]])
    out = dict(pix=pix)
    for bi in (0, 2):
        img = imgs[bi]
        equa = np.array([img.pixel2equa(p) for p in pix])
        out["equa_b%d" % bi] = equa
        out["pix_back_b%d" % bi] = np.array([img.equa2pixel(u) for u in equa])
        out["cd_b%d" % bi] = np.array([img.cd_at_pixel(p[0], p[1]) for p in pix])
    # a big virtual frame: cd_at_pixel varies measurably across 2048 px (Q9)
    big = virtual_images("wcsbig", 64, 2048)[2]
    bp = np.array([[0.0, 0.0], [2047.0, 63.0], [1024.0, 32.0], [100.5, 7.25]])
    out["big_pix"] = bp
    out["big_equa"] = np.array([big.pixel2equa(p) for p in bp])
    out["big_cd"] = np.array([big.cd_at_pixel(p[0], p[1]) for p in bp])
    out["big_band"] = np.array(2)
    # known-answer from notebooks/RenderSources.ipynb cell 1: source 0 of cat-188.3444-63.4421, r band
    f2 = "188.3444-63.4421"
    im2 = ref_fits.FitsImage("r", fits_file_template=os.path.join(STAMP_DIR, "stamp-%s-" + f2 + ".fits"))
    _, cat2 = _refload.fits_bintable(os.path.join(STAMP_DIR, "cat-%s.fits" % f2))
    u0 = np.array([cat2["ra"][0], cat2["dec"][0]], dtype=np.float64)
    out["nb_u"] = u0
    out["nb_rho"], out["nb_phi"], out["nb_ups_inv"] = im2.rho_n, im2.phi_n, im2.Ups_n_inv
    out["nb_pix"] = im2.equa2pixel(u0)
    out["nb_pix_notebook"] = np.array([26.0470, 26.1040])
    save("wcs_points.npz", **out)


def gen_evaluator():
    # inputs exactly as CelestePy/test/test_gmm.py:63-85
    np.random.seed(41)
    K = 42
    means = 2 * np.random.randn(K * 2).reshape(K, 2)
    covs = np.zeros((K, 2, 2))
    invcovs = np.zeros((K, 2, 2))
    logdets = np.zeros(K)
    for i in range(K):
        covs[i, :, :] = np.random.randn(2, 2)
        covs[i, :, :] = covs[i, :, :].dot(covs[i, :, :].T)
        invcovs[i, :, :] = np.linalg.inv(covs[i, :, :])
        sign, logdet = np.linalg.slogdet(covs[i, :, :])
        logdets[i] = logdet
    ws = np.random.rand(K)
    ws /= np.sum(ws)
    xgrid = np.linspace(-3, 3, 200)
    ygrid = np.linspace(-5, 5, 200)
    xx, yy = np.meshgrid(xgrid, ygrid)
    X = np.column_stack((xx.ravel(), yy.ravel()))
    # the reference's gmm_like_2d (its numpy fallback, util/like/__init__.py:14) and mog_loglike
    probs = ref_like.gmm_like_2d(X, ws, means, covs)
    assert ref_like.gmm_like_2d is ref_like.gmm_prob
    ll = ref_mog.mog_loglike(X, means, invcovs, np.exp(logdets), ws)
    # scipy cross-check, as the reference test does (test_gmm.py:95-101)
    from scipy.special import logsumexp
    from scipy.stats import multivariate_normal
    l2 = np.zeros((X.shape[0], K))
    for k in range(K):
        l2[:, k] = multivariate_normal(mean=means[k], cov=covs[k]).logpdf(X) + np.log(ws[k])
    slow = np.exp(logsumexp(l2, axis=1))
    assert np.allclose(probs, slow)
    save("evaluator.npz", X=X, ws=ws, means=means, covs=covs, invcovs=invcovs, logdets=logdets,
         gmm_prob=probs, mog_loglike=ll)


def galaxy_cmix(th, u, img):
    """The component table the reference builds inside gen_galaxy_psf_image (:196-203)."""
    px, py = img.equa2pixel(u)
    galmix = ref_mog.MixtureOfGaussians.convex_combine(ref_gal.galaxy_profs, [th[0], 1.0 - th[0]])
    Tinv = ref_gal.gen_galaxy_transformation(th[1], th[3], th[2], img.cd_at_pixel(px, py))
    amix = galmix.apply_affine(Tinv, np.array([px, py]))
    cmix = amix.convolve(img.psf)
    return px, py, Tinv, cmix


def gen_radius(imgs):
    out = {}
    out["psf_R_1e3"] = np.array([ref_bb.calc_bounding_radius(i.weights, i.means, i.covars, 0.001) for i in imgs])
    out["psf_R_1e5_c"] = np.array([ref_bb.calc_bounding_radius(i.weights, i.means, i.covars, 1e-5,
                                                                center=np.array([0.3, -0.2])) for i in imgs])
    # the exp/dev profile tables as the reference holds them (A13)
    out["exp_amp"], out["exp_var"] = ref_gal.galaxy_profs[0].pis, ref_gal.galaxy_profs[0].covs[:, 0, 0]
    out["dev_amp"], out["dev_var"] = ref_gal.galaxy_profs[1].pis, ref_gal.galaxy_profs[1].covs[:, 0, 0]
    save("radius.npz", **out)


def gen_star_stamps(imgs):
    rs = np.random.RandomState(11)
    pix = [[25.0, 25.0], [25.3, 24.8], [0.3, 50.2], [49.9, 0.1], [10.49999, 30.5], [-8.2, 12.0],
           [55.5, 57.25], [21.442606659470453, 29.557393340529547], [-30.0, 10.0], [30.0, 75.0]]
    pix += rs.uniform(0, 51, size=(6, 2)).tolist()
    pix = np.array(pix)
    bands, us, boxes, patches, is_none = [], [], [], [], []
    for bi in (0, 2, 4):
        img = imgs[bi]
        for p in pix:
            u = img.pixel2equa(p)
            patch, yl, xl = ref_cel.gen_point_source_psf_image(u, img)
            bands.append(bi)
            us.append(u)
            if patch is None:
                is_none.append(1)
                boxes.append([0, 0, 0, 0])
                patches.append(np.zeros((0, 0)))
            else:
                is_none.append(0)
                boxes.append([yl[0], yl[1], xl[0], xl[1]])
                patches.append(patch)
    # Q1: far off-frame sources for which the reference's overlap test returns None
    img = imgs[2]
    for p in ([-60.0, 10.0], [10.0, -60.0], [120.0, 10.0]):
        u = img.pixel2equa(np.array(p))
        patch, yl, xl = ref_cel.gen_point_source_psf_image(u, img)
        bands.append(2)
        us.append(u)
        is_none.append(int(patch is None))
        if patch is None:
            boxes.append([0, 0, 0, 0])
            patches.append(np.zeros((0, 0)))
        else:
            boxes.append([yl[0], yl[1], xl[0], xl[1]])
            patches.append(patch)
    flat, offs, shapes = pack_ragged(patches)
    # caller-supplied limits (celeste.py:145-152) + return_patch=False embedding (:169-176)
    u = imgs[2].pixel2equa(np.array([20.2, 30.7]))
    lim_patch, _, _ = ref_cel.gen_point_source_psf_image(u, imgs[2], xlim=(5, 40), ylim=(12, 51))
    full, yl, xl = ref_cel.gen_point_source_psf_image(u, imgs[2], return_patch=False)
    save("star_stamps.npz", band=np.array(bands), u=np.array(us), box=np.array(boxes, dtype=np.int64),
         is_none=np.array(is_none), flat=flat, offs=offs, shapes=shapes,
         lim_u=u, lim_xlim=np.array([5, 40]), lim_ylim=np.array([12, 51]), lim_patch=lim_patch,
         full_image=full)


GAL_SHAPES = np.array([
    # theta, sigma(arcsec), phi(deg), rho
    [0.40, 1.5, 30.0, 0.60],
    [0.05, 0.5, 0.0, 0.30],
    [0.95, 1.0, 90.0, 0.90],
    [0.50, 2.0, 135.0, 0.90],
    [0.70, 4.0, 45.0, 0.20],
    [0.30, 0.01, 10.0, 0.50],   # below the 1/30 arcsec floor (celeste_galaxy_conditionals.py:101)
    [0.60, 3.0, 170.0, 0.95],
])


def gen_galaxy_stamps(imgs):
    out = dict(shapes_in=GAL_SHAPES)
    # (a) on the real 51x51 stamps, (b) on a virtual 200(H) x 300(W) frame where big boxes fit
    big = virtual_images("galbig", 200, 300)
    frames = [("s", imgs, np.array([[25.3, 24.8], [3.5, 47.0], [40.0, 10.2]])),
              ("b", big, np.array([[150.2, 99.7], [20.5, 180.25], [290.0, 15.0]]))]
    for tag, frame, pixs in frames:
        # big-frame patches reach 200x300: keep every 2nd pixel (+ the full sum) to stay small
        stride = 1 if tag == "s" else 2
        sums = []
        bands, us, ths, boxes, patches, tinvs, pxs, bounds = [], [], [], [], [], [], [], []
        cw, cm, cc = [], [], []
        for bi in ((1, 2, 3) if tag == "s" else (2,)):
            img = frame[bi]
            for p in pixs:
                u = img.pixel2equa(p)
                for th in GAL_SHAPES:
                    patch, yl, xl = ref_gal.gen_galaxy_psf_image(th, u, img)
                    sums.append(patch.sum())
                    patch = patch[::stride, ::stride]
                    px, py, Tinv, cmix = galaxy_cmix(th, u, img)
                    bands.append(bi)
                    us.append(u)
                    ths.append(th)
                    boxes.append([yl[0], yl[1], xl[0], xl[1]])
                    patches.append(patch)
                    tinvs.append(Tinv)
                    pxs.append([px, py])
                    bounds.append(ref_bb.calc_bounding_radius(cmix.pis, cmix.means, cmix.covs,
                                                              error=1e-5, center=np.array([px, py])))
                    cw.append(cmix.pis)
                    cm.append(cmix.means)
                    cc.append(cmix.covs)
        flat, offs, shapes = pack_ragged(patches)
        out.update({tag + "_band": np.array(bands), tag + "_u": np.array(us), tag + "_th": np.array(ths),
                    tag + "_box": np.array(boxes, dtype=np.float64), tag + "_flat": flat, tag + "_offs": offs,
                    tag + "_shapes": shapes, tag + "_tinv": np.array(tinvs), tag + "_pix": np.array(pxs),
                    tag + "_bound": np.array(bounds), tag + "_cw": np.array(cw), tag + "_cm": np.array(cm),
                    tag + "_cc": np.array(cc), tag + "_sum": np.array(sums),
                    tag + "_stride": np.array(stride)})
    out["big_H"], out["big_W"] = np.array(200), np.array(300)
    save("galaxy_stamps.npz", **out)


def gen_mini_field():
    H, W = 80, 96
    blank = virtual_images("mini0", H, W)
    rs = np.random.RandomState(42)
    S = 12
    pix = np.column_stack([rs.uniform(-4, W + 4, S), rs.uniform(-4, H + 4, S)])
    is_gal = (np.arange(S) % 2 == 1).astype(np.int64)
    flux = np.exp(rs.uniform(np.log(1.0), np.log(100.0), size=(S, 5)))
    theta = rs.uniform(0.05, 0.95, S)
    sigma = np.exp(rs.uniform(np.log(0.5), np.log(4.0), S))
    rho = rs.uniform(0.2, 0.95, S)
    phi = rs.uniform(0.0, 180.0, S)
    radec = np.array([blank[2].pixel2equa(p) for p in pix])
    # sources, array-style fluxes: the convention of gen_src_image_with_fluxes (celeste.py:72-96)
    srcs = [SrcParams(u=radec[s], a=int(is_gal[s]), fluxes=flux[s], theta=theta[s], sigma=sigma[s],
                      phi=phi[s], rho=rho[s]) for s in range(S)]

    def render(imgs):
        lam, patches, boxes = [], [], []
        for img in imgs:
            f = np.zeros(img.nelec.shape)
            for src in srcs:
                p, yl, xl = ref_cel.gen_src_image_with_fluxes(src, img)
                y0, y1, x0, x1 = int(yl[0]), int(yl[1]), int(xl[0]), int(xl[1])
                f[y0:y1, x0:x1] += p          # Q3 extension: scatter-add each source's own patch
                patches.append(p)
                boxes.append([y0, y1, x0, x1])
            lam.append(img.epsilon + f)
        return np.array(lam), patches, boxes

    lam_true, _, _ = render(blank)
    nelec = np.random.RandomState(43).poisson(lam_true).astype(np.float64)
    imgs = virtual_images("mini1", H, W, nelec=nelec)
    lam, patches, boxes = render(imgs)
    assert np.array_equal(lam, lam_true)
    ll_band = np.array([np.sum(im.nelec * np.log(l) - l) for im, l in zip(imgs, lam)])
    flat, offs, shapes = pack_ragged(patches)
    out = stack_bands(imgs)
    out.update(H=np.array(H), W=np.array(W), nelec=nelec, radec=radec, is_gal=is_gal, flux=flux,
               shape=np.column_stack([theta, sigma, phi, rho]), lam=lam, ll_band=ll_band,
               ll=np.array(ll_band.sum()), patch_flat=flat, patch_offs=offs, patch_shapes=shapes,
               patch_box=np.array(boxes, dtype=np.int64))
    # stars only through the reference's own gen_model_image / celeste_likelihood (dict-style fluxes)
    stars = [SrcParams(u=radec[s], a=0, fluxes=dict(zip(BANDS, flux[s]))) for s in range(S)
             if not is_gal[s] and -4 < pix[s, 0] < W + 4]
    star_idx = np.array([s for s in range(S) if not is_gal[s]])
    inside = [s for s in star_idx]
    stars = [SrcParams(u=radec[s], a=0, fluxes=dict(zip(BANDS, flux[s]))) for s in inside]
    try:
        out["star_lam"] = np.array([ref_cel.gen_model_image(stars, im) for im in imgs])
        out["star_ll"] = np.array(ref_cel.celeste_likelihood_multi_image(stars, imgs))
        out["star_idx"] = np.array(inside)
    except Exception as e:  # Q1: an off-frame star makes the reference raise
        print("reference gen_model_image raised on the full star list (%s: %s); "
              "restricting to in-frame stars" % (type(e).__name__, e))
        inside = [s for s in star_idx if 0 <= pix[s, 0] < W and 0 <= pix[s, 1] < H]
        stars = [SrcParams(u=radec[s], a=0, fluxes=dict(zip(BANDS, flux[s]))) for s in inside]
        out["star_lam"] = np.array([ref_cel.gen_model_image(stars, im) for im in imgs])
        out["star_ll"] = np.array(ref_cel.celeste_likelihood_multi_image(stars, imgs))
        out["star_idx"] = np.array(inside)
    save("mini_field.npz", **out)


def gen_config1(imgs):
    _, cat = _refload.fits_bintable(os.path.join(STAMP_DIR, "cat-%s.fits" % FIELD))
    srcs = []
    for row in cat:
        fl = dict(zip(BANDS, [float(row["psfflux_" + b]) for b in BANDS]))
        if any(v < 0 for v in fl.values()):
            continue  # util/misc/init_utils.py:54-56
        srcs.append(SrcParams(u=np.array([row["ra"], row["dec"]], dtype=np.float64), fluxes=fl))
    keep = np.array([[s.u[0], s.u[1]] + [s.fluxes[b] for b in BANDS] for s in srcs])
    lam = np.array([ref_cel.gen_model_image(srcs, im) for im in imgs])
    ll_band = np.array([ref_cel.celeste_likelihood(srcs, im) for im in imgs])
    ll = ref_cel.celeste_likelihood_multi_image(srcs, imgs)
    # config 1 proper: ONE star at the stamp centre of the r band, catalogue flux of the brightest row
    r = imgs[2]
    b = int(np.argmax(keep[:, 4]))
    star = SrcParams(u=r.pixel2equa(np.array([25.0, 25.0])), a=0, fluxes=dict(zip(BANDS, keep[b, 2:])))
    one = ref_cel.gen_src_image(star, r)
    one_lam = ref_cel.gen_model_image([star], r)
    one_ll = ref_cel.celeste_likelihood([star], r)
    save("config1.npz", cat=keep, lam=lam, ll_band=ll_band, ll=np.array(ll),
         one_u=star.u, one_flux=keep[b, 2:], one_patch=one, one_lam=one_lam, one_ll=np.array(one_ll))


def gen_source_ll():
    """source_ll.npz: the reference's Source.log_likelihood / log_likelihood_isolated
    (CelestePy/sources.py:134-237) on fixed sample patches, for batches of proposals."""
    import types

    import CelestePy.sources as ref_src
    H, W = 80, 96
    rs = np.random.RandomState(5)
    nelec = rs.poisson(400.0, size=(5, H, W)).astype(np.float64)
    imgs = virtual_images("srcll", H, W, nelec=nelec)
    out = stack_bands(imgs)
    out.update(H=np.array(H), W=np.array(W), nelec=nelec, bands_used=np.array([1, 2, 3]))
    specs = [(0, [40.3, 33.8], [0.5, 1.0, 0.0, 0.5]),
             (1, [55.6, 41.2], [0.35, 1.2, 60.0, 0.55]),
             (0, [3.2, 70.5], [0.5, 1.0, 0.0, 0.5])]
    for ci, (kind, pix, shape) in enumerate(specs):
        u = imgs[2].pixel2equa(np.array(pix))
        flux = np.array([5.0, 12.0, 20.0, 26.0, 30.0])
        params = SrcParams(u=u, a=kind, fluxes=flux, theta=shape[0], sigma=shape[1], phi=shape[2], rho=shape[3])
        src = ref_src.Source(params, model=None)
        boxes, zs = [], []
        for bi in (1, 2, 3):
            xlim, ylim = ref_src.Source.get_bounding_box(params, imgs[bi])
            xlim, ylim = (int(xlim[0]), int(xlim[1])), (int(ylim[0]), int(ylim[1]))
            patch, _, _ = src.compute_model_patch(imgs[bi], xlim=xlim, ylim=ylim)
            z = rs.poisson(patch).astype(np.float64)
            samp = types.SimpleNamespace(data=z, x0=xlim[0], x1=xlim[1], y0=ylim[0], y1=ylim[1])
            src.sample_image_list.append((samp, imgs[bi], None))
            boxes.append([ylim[0], ylim[1], xlim[0], xlim[1]])
            zs.append(z)
        P = 7
        us = u[None, :] + rs.normal(0.0, 4e-5, size=(P, 2))
        us[0] = u
        fl = flux[None, :] * rs.uniform(0.5, 2.0, size=(P, 5))
        fl[0] = flux
        sh = np.array(shape)[None, :] * rs.uniform(0.8, 1.25, size=(P, 4))
        sh[:, 0] = np.clip(sh[:, 0], 0.02, 0.98)
        sh[:, 3] = np.clip(sh[:, 3], 0.1, 0.99)
        sh[0] = shape
        if kind == 0:
            us[P - 1] = imgs[2].pixel2equa(np.array([-70.0, 20.0]))      # fails the overlap test: psf_ns is None
        ll0 = np.array([src.log_likelihood(u=us[i], fluxes=fl[i], shape=sh[i]) for i in range(P)])
        n_iso = P - 1 if kind == 0 else P                                  # the isolated form asserts on a miss
        ll1 = np.array([src.log_likelihood_isolated(u=us[i], fluxes=fl[i], shape=sh[i]) for i in range(n_iso)])
        flat, offs, shapes = pack_ragged(zs)
        out.update({"c%d_kind" % ci: np.array(kind), "c%d_u" % ci: u, "c%d_flux" % ci: flux,
                    "c%d_shape" % ci: np.array(shape), "c%d_boxes" % ci: np.array(boxes, dtype=np.int64),
                    "c%d_z" % ci: flat, "c%d_zoffs" % ci: offs, "c%d_zshapes" % ci: shapes,
                    "c%d_us" % ci: us, "c%d_fl" % ci: fl, "c%d_sh" % ci: sh, "c%d_ll0" % ci: ll0,
                    "c%d_ll1" % ci: ll1})
    out["ncases"] = np.array(len(specs))
    save("source_ll.npz", **out)


def gen_estep():
    """estep.npz: the reductions celeste_em.py:38-91 takes of the reference's
    gen_src_prob_layers (celeste.py:222-234), stars of the mini field, all five bands."""
    g = dict(np.load(os.path.join(HERE, "mini_field.npz")))
    H, W = int(g["H"]), int(g["W"])
    imgs = virtual_images("estep", H, W, nelec=g["nelec"])
    idx = g["star_idx"]
    stars = [SrcParams(u=g["radec"][s], a=0, fluxes=dict(zip(BANDS, g["flux"][s]))) for s in idx]
    X = np.zeros((len(stars), 5))
    F = np.zeros((len(stars), 5))
    Z = np.zeros(5)
    for n, img in enumerate(imgs):
        probs = ref_cel.gen_src_prob_layers(stars, img)
        Z[n] = np.sum(img.nelec * probs[0])                                   # celeste_em.py:62 (x size)
        for s in range(len(stars)):
            X[s, n] = np.sum(probs[s + 1] * img.nelec)                        # celeste_em.py:85
            patch, _, _ = ref_cel.gen_point_source_psf_image(stars[s].u, img)
            F[s, n] = 0.0 if patch is None else np.sum(patch)                 # celeste_em.py:89
    save("estep.npz", star_idx=idx, xtilde=X, mass=F, noise=Z)


class _RecordingRandom(object):
    """Stands in for the `npr` of util/infer/slicesample.py:2: the same numpy generator calls, with
    every result noted in call order (kinds as oracle.slicesample_oracle.DRAW_*)."""

    def __init__(self, seed):
        self.rs = np.random.RandomState(seed)
        self.kinds, self.vals, self.perms = [], [], []

    def rand(self, *shape):
        assert not shape
        v = self.rs.rand()
        self.kinds.append(0)
        self.vals.append(v)
        return v

    def randn(self, n):
        v = self.rs.randn(n)
        self.kinds.extend([1] * n)
        self.vals.extend(v.tolist())
        return v

    def shuffle(self, ordering):
        from oracle.slicesample_oracle import shuffle_keys
        self.rs.shuffle(ordering)
        self.perms.append(list(ordering))
        self.kinds.extend([2] * len(ordering))
        self.vals.extend(shuffle_keys(ordering).tolist())


def gen_slicesample():
    """slicesample.npz: chains of the reference's slicesample (util/infer/slicesample.py:89-227) on
    closed-form targets, every option set tests/test_slicesample.py runs + the call
    Source.resample_location makes (sources.py:315-319) + the 4-D random-direction call of
    celeste_mcmc.py:229-239 + the bounded demo of slicesample.py:261-276.  Per call: the draws the
    reference made, in order, and what it returned."""
    import io
    import json
    from contextlib import redirect_stdout

    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle.slicesample_oracle import TARGETS
    import CelestePy.util.infer.slicesample as ref_ss
    cases = [
        ("gauss", dict(sigma=1.0, step_out=True, doubling_step=True)),
        ("bimodal", dict(sigma=1.0, step_out=True, doubling_step=True)),
        ("gauss", dict(sigma=0.4, step_out=True, doubling_step=False)),
        ("bimodal", dict(sigma=0.4, step_out=True, doubling_step=False)),
        ("gauss", dict(sigma=25.0, step_out=False)),
        ("bimodal", dict(sigma=25.0, step_out=False)),
        ("gauss", dict(sigma=0.7, step_out=True, doubling_step=True, compwise=False, numdir=3)),
        ("bimodal", dict(sigma=0.7, step_out=True, doubling_step=True, compwise=False, numdir=3)),
        ("gauss", dict(sigma=0.3, step_out=True, doubling_step=True, max_steps_out=3)),
        ("bimodal", dict(sigma=0.3, step_out=True, doubling_step=True, max_steps_out=3)),
        ("gauss", dict(step_out=False)),                                            # sources.py:315-319 (sigma stays 1.0)
        ("gauss4", dict(sigma=0.5, step_out=True, doubling_step=True, compwise=False, numdir=4)),   # celeste_mcmc.py:229-239
        ("gauss4", dict(sigma=0.5, step_out=True, doubling_step=True)),
        ("halfgauss", dict(sigma=0.1, step_out=True, doubling_step=True, compwise=False, numdir=4, lower_bound=0.0)),
        ("halfgauss", dict(sigma=0.6, step_out=True, doubling_step=False, lower_bound=0.0, upper_bound=50.0)),
    ]
    ncalls = 12
    out = dict(ncases=np.array(len(cases)), ncalls=np.array(ncalls))
    for ci, (tname, kw) in enumerate(cases):
        f = TARGETS[tname]
        D = 4 if tname in ("gauss4", "halfgauss") else 2
        x = np.abs(np.random.RandomState(100 + ci).randn(D)) + 0.05
        rec = _RecordingRandom(7000 + ci)
        ref_ss.npr = rec
        xs, lls, offs = [], [], [0]
        x0 = x.copy()
        for it in range(ncalls):
            with redirect_stdout(io.StringIO()):
                x, llh = ref_ss.slicesample(x.copy(), f, **kw)
            xs.append(np.array(x, dtype=np.float64))
            lls.append(float(llh))
            offs.append(len(rec.vals))
        out.update({"c%d_target" % ci: np.array(tname), "c%d_kw" % ci: np.array(json.dumps(kw)), "c%d_x0" % ci: x0,
                    "c%d_draw_kind" % ci: np.array(rec.kinds, dtype=np.int8), "c%d_draw_val" % ci: np.array(rec.vals),
                    "c%d_draw_off" % ci: np.array(offs, dtype=np.int64), "c%d_perms" % ci: np.array(rec.perms, dtype=np.int64),
                    "c%d_x" % ci: np.array(xs), "c%d_llh" % ci: np.array(lls)})
    ref_ss.npr = np.random
    # the log-prior slice_sample_skew adds to the likelihood (celeste_mcmc.py:209-214): inside and outside its support
    rs = np.random.RandomState(77)
    th = np.column_stack([rs.uniform(-0.2, 1.2, 200), rs.uniform(-0.5, 6.0, 200), rs.uniform(-0.5, 3.6, 200), rs.uniform(-0.2, 1.2, 200)])
    out["prior_th"] = th
    # galaxy_shape_prior_constrained (celeste_galaxy_conditionals.py:268-275) returns -inf outside its support and
    # otherwise calls fast_inv_gamma_lnpdf, a name its module never imports (NameError): the support test is
    # recorded from the function itself, the density from the function it means (util/like/like_list.py:18-27)
    import CelestePy.util.like.like_list as ref_ll
    lp = np.zeros(len(th))
    for i, t in enumerate(th):
        try:
            lp[i] = ref_gal.galaxy_shape_prior_constrained(*t)
            assert lp[i] == -np.inf
        except NameError:
            lp[i] = ref_ll.fast_inv_gamma_lnpdf(t[1] * t[1], a0=1., b0=1.)
    out["prior_lp"] = lp
    save("slicesample.npz", **out)


REAL_DIRS = ("stamps", "stamp_catalog", "galaxy_stamps", "real")


def real_field_list():
    """[(dir, tag, stamp template, catalogue file)] of every field with five band images and a catalogue, in sorted order
    (what load_imgs_and_catalog, util/misc/init_utils.py:9-40, is handed as a sorted glob of cat*.fits)"""
    import glob
    out = []
    for d in REAL_DIRS:
        D = os.path.join(_refload.REF_ROOT, "data", d)
        for c in sorted(glob.glob(os.path.join(D, "cat*.fits"))):
            base = os.path.basename(c)[:-5]
            tag = base.split("cat-")[1] if "cat-" in base else ""
            tmpl = os.path.join(D, "stamp-%s-" + tag + ".fits") if tag else os.path.join(D, "stamp-%s.fits")
            if all(os.path.exists(tmpl % b) for b in BANDS):
                out.append((d, tag, tmpl, c))
    return out


def catalogue_sources(cat_file):
    """get_sources_from_catalog (util/misc/init_utils.py:42-60): a row with any negative flux is skipped"""
    _, cat = _refload.fits_bintable(cat_file)
    srcs = []
    for row in cat:
        fl = dict(zip(BANDS, [float(row["psfflux_" + b]) for b in BANDS]))
        if any(v < 0 for v in fl.values()):
            continue
        srcs.append(SrcParams(u=np.array([row["ra"], row["dec"]], dtype=np.float64), fluxes=fl))
    return srcs


def gen_real_fields():
    """real_fields.npz: the reference's own FitsImage / gen_model_image / celeste_likelihood[_multi_image] /
    gen_point_source_psf_image / gen_galaxy_psf_image on every real field in its tree.  Model images are stored whole for
    data/stamps and data/real (12 fields), every 4th pixel of every 4th row for the others (the log-likelihoods pin the rest);
    star stamps for the r band of those 12 fields; one galaxy patch per field of data/stamps and data/galaxy_stamps."""
    fields = real_field_list()
    rs = np.random.RandomState(2024)
    recs, names, HW, nel, cat_off, cat_radec, cat_flux = [], [], [], [], [0], [], []
    lam_full, lam_sub, ll_band, ll = [], [], [], []
    src_box, src_none = [], []
    st_field, st_src, st_box, st_patch = [], [], [], []
    g_field, g_band, g_u, g_th, g_box, g_patch = [], [], [], [], [], []
    for fi, (d, tag, tmpl, catf) in enumerate(fields):
        imgs = [ref_fits.FitsImage(b, fits_file_template=tmpl) for b in BANDS]
        srcs = catalogue_sources(catf)
        H, W = imgs[0].nelec.shape
        assert all(im.nelec.shape == (H, W) for im in imgs)
        names.append("%s/%s" % (d, tag))
        HW.append([H, W])
        recs.append(stack_bands(imgs))
        ne = np.array([im.nelec for im in imgs])
        assert np.array_equal(ne, np.rint(ne)) and np.abs(ne).max() < 2 ** 31
        nel.append(ne.astype(np.int32).ravel())
        cat_off.append(cat_off[-1] + len(srcs))
        cat_radec += [s.u for s in srcs]
        cat_flux += [[s.fluxes[b] for b in BANDS] for s in srcs]
        lam = np.array([ref_cel.gen_model_image(srcs, im) for im in imgs])
        ll_band.append([ref_cel.celeste_likelihood(srcs, im) for im in imgs])
        ll.append(ref_cel.celeste_likelihood_multi_image(srcs, imgs))
        whole = d in ("stamps", "real")
        (lam_full if whole else lam_sub).append((lam if whole else lam[:, ::4, ::4]).ravel())
        for si, s in enumerate(srcs):
            for bi, im in enumerate(imgs):
                patch, yl, xl = ref_cel.gen_point_source_psf_image(s.u, im)
                src_none.append(int(patch is None))
                src_box.append([0, 0, 0, 0] if patch is None else [yl[0], yl[1], xl[0], xl[1]])
                if whole and bi == 2 and patch is not None:
                    st_field.append(fi); st_src.append(si); st_box.append([yl[0], yl[1], xl[0], xl[1]]); st_patch.append(patch)
        if d in ("stamps", "galaxy_stamps") and srcs:
            # a galaxy at the brightest catalogue source's place, on this field's PSF and WCS
            bi = fi % 5
            s = max(srcs, key=lambda q: q.fluxes["r"])
            th = np.array([rs.uniform(0.05, 0.95), np.exp(rs.uniform(np.log(0.3), np.log(3.0))), rs.uniform(0, 180), rs.uniform(0.2, 0.95)])
            patch, yl, xl = ref_gal.gen_galaxy_psf_image(th, s.u, imgs[bi])
            g_field.append(fi); g_band.append(bi); g_u.append(s.u); g_th.append(th)
            g_box.append([yl[0], yl[1], xl[0], xl[1]]); g_patch.append(patch)
    out = {k: np.array([r[k] for r in recs], dtype=np.float64) for k in recs[0]}        # (F, 5, ...)
    st_flat, st_offs, st_shapes = pack_ragged(st_patch)
    g_flat, g_offs, g_shapes = pack_ragged(g_patch)
    out.update(names=np.array(names), HW=np.array(HW, dtype=np.int64), nelec=np.concatenate(nel),
               cat_off=np.array(cat_off, dtype=np.int64), cat_radec=np.array(cat_radec, dtype=np.float64),
               cat_flux=np.array(cat_flux, dtype=np.float64), lam_full=np.concatenate(lam_full), lam_sub=np.concatenate(lam_sub),
               ll_band=np.array(ll_band, dtype=np.float64), ll=np.array(ll, dtype=np.float64),
               src_box=np.array(src_box, dtype=np.int64).reshape(-1, 5, 4), src_none=np.array(src_none, dtype=np.int8).reshape(-1, 5),
               st_field=np.array(st_field), st_src=np.array(st_src), st_box=np.array(st_box, dtype=np.int64),
               st_flat=st_flat, st_offs=st_offs, st_shapes=st_shapes,
               g_field=np.array(g_field), g_band=np.array(g_band), g_u=np.array(g_u), g_th=np.array(g_th),
               g_box=np.array(g_box, dtype=np.float64), g_flat=g_flat, g_offs=g_offs, g_shapes=g_shapes)
    save("real_fields.npz", **out)
    print("real fields: %d fields, %d sources, sum of ll = %.4f" % (len(fields), cat_off[-1], float(np.sum(ll))))


def gen_src_bound(imgs):
    """celeste.gen_psf_src_image_bound (celeste.py:193-199) for stars and galaxies on the real stamps' five images, the
    floor/ceil boxes its caller forms from it (experiments/fields/process_field.py:107-117), and FitsImage.make_pixel_grid /
    the pixel_grid attribute (fits_image.py:95, 186-194)."""
    rs = np.random.RandomState(23)
    S = 10
    H, W = imgs[0].nelec.shape
    pix = np.column_stack([rs.uniform(-5, W + 5, S), rs.uniform(-5, H + 5, S)])
    is_gal = (np.arange(S) % 2 == 1).astype(np.int64)
    flux = np.exp(rs.uniform(np.log(1.0), np.log(100.0), size=(S, 5)))
    shape = np.column_stack([rs.uniform(0.05, 0.95, S), np.exp(rs.uniform(np.log(0.3), np.log(5.0), S)),
                             rs.uniform(0.0, 180.0, S), rs.uniform(0.2, 0.95, S)])
    radec = np.array([imgs[2].pixel2equa(p) for p in pix])
    srcs = [SrcParams(u=radec[s], a=int(is_gal[s]), fluxes=flux[s], theta=shape[s, 0], sigma=shape[s, 1],
                      phi=shape[s, 2], rho=shape[s, 3]) for s in range(S)]
    bound = np.array([[ref_cel.gen_psf_src_image_bound(s, img) for s in srcs] for img in imgs])          # (5, S)
    boxes = []
    for b, img in enumerate(imgs):                       # process_field.py:107-117, line for line in meaning
        locs = np.vstack([img.equa2pixel(s.u) for s in srcs])
        boxes.append(np.column_stack([np.floor(locs[:, 0] - bound[b]), np.ceil(locs[:, 0] + bound[b]),
                                      np.floor(locs[:, 1] - bound[b]), np.ceil(locs[:, 1] + bound[b])]))
    grid = imgs[2].make_pixel_grid()
    assert np.array_equal(grid, imgs[2].pixel_grid)
    save("src_bound.npz", radec=radec, is_gal=is_gal, flux=flux, shape=shape, bound=bound, boxes=np.array(boxes),
         pixel_grid=grid, grid_HW=np.array([H, W]))


if __name__ == "__main__":
    if sys.argv[1:] == ["src_bound"]:
        gen_src_bound(ref_images())
        sys.exit(0)
    if sys.argv[1:] == ["real_fields"]:
        gen_real_fields()
        sys.exit(0)
    if sys.argv[1:] == ["slicesample"]:
        gen_slicesample()
        sys.exit(0)
    imgs = gen_bands()
    gen_wcs(imgs)
    gen_evaluator()
    gen_radius(imgs)
    gen_star_stamps(imgs)
    gen_galaxy_stamps(imgs)
    gen_mini_field()
    gen_config1(imgs)
    gen_source_ll()
    gen_estep()
    gen_slicesample()
    gen_src_bound(imgs)
    gen_real_fields()
