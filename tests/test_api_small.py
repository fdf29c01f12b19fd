"""The reference-named entry points no other test calls by name (found by listing the package's public names against tests/):
each against a sibling that IS pinned to the reference's goldens, or against the reference's own formula."""
import numpy as np
import pytest


def test_magnitudes_and_accessors():
    """celeste_src.py's unit helpers (CelestePy/util/data: 22.5 mag = 1 nanomaggy) and SrcParams.mags; FitsImage.contains keeps
    the reference's axis mix (fits_image.py:157-164: x against shape[0])"""
    import desi_mcmc_amd as cel
    from desi_mcmc_amd import celeste_src
    from desi_mcmc_amd.fits_image import FitsImage
    assert celeste_src.mags2nanomaggies(22.5) == 1.0 and abs(celeste_src.mags2nanomaggies(20.0) - 10.0) < 1e-12
    m = np.array([15.0, 18.3, 22.5, 25.0])
    np.testing.assert_allclose(celeste_src.nanomaggies2mags(celeste_src.mags2nanomaggies(m)), m, rtol=1e-14)
    p = cel.SrcParams(u=np.array([0.1, 0.2]), a=0, fluxes=np.array([1.0, 10.0, 100.0, 1.0, 1.0]))
    np.testing.assert_allclose(p.mags, [22.5, 20.0, 17.5, 22.5, 22.5], rtol=1e-14)
    img = FitsImage("r", np.zeros((40, 100)), epsilon=1., kappa=1., calib=1., weights=np.ones(3) / 3, means=np.zeros((3, 2)),
                    covars=np.tile(np.eye(2), (3, 1, 1)), rho_n=np.array([0.0, 0.0]), phi_n=np.array([0.0, 0.0]), Ups_n=np.eye(2))
    assert img.contains(np.array([30.0, 50.0]))                 # x = 30 < shape[0] = 40, y = 50 < shape[1] = 100
    assert not img.contains(np.array([95.0, 20.0]))             # x = 95 is inside the 100-column frame, but is tested against 40 rows + 50
    assert img.contains(np.array([89.0, 20.0])) and not img.contains(np.array([30.0, 151.0]))


@pytest.mark.gpu
def test_reference_named_stamp_and_photon_entry_points():
    import desi_mcmc_amd as cel
    from desi_mcmc_amd import celeste, celeste_mcmc, synth, models
    ctx = cel.default_context(0)
    f = synth.SyntheticField(ctx, 12, 5, 128, 160, frac_gal=0.5, seed=8)
    imgs = synth.fits_images(f)
    cat = cel.SrcCatalog((f.src["type"] == 1).astype(np.int64), f.src["radec"], f.flux5(), f.src["shape"])
    ps = [cel.SrcParams(u=p.u.copy(), a=p.a, fluxes=p.fluxes.copy(), theta=p.theta, sigma=p.sigma, phi=p.phi, rho=p.rho) for p in cat]
    star = [p for p in ps if p.a == 0][0]
    gal = [p for p in ps if p.a == 1][0]
    im = imgs[2]
    # gen_src_psf_image (celeste.py:64-70): the star or the galaxy stamp
    a, ya, xa = celeste.gen_src_psf_image(star, im)
    b, yb, xb = celeste.gen_point_source_psf_image(star.u, im)
    assert np.array_equal(a, b) and (ya, xa) == (yb, xb)
    a, ya, xa = celeste.gen_src_psf_image(gal, im)
    b, yb, xb = celeste.gen_galaxy_psf_image(gal, im)
    assert np.array_equal(a, b) and tuple(ya) == tuple(yb)
    # gen_point_source_psf_image_with_fluxes (celeste.py:72-82) = the unit stamp x flux / calib * kappa = gen_src_image_with_fluxes
    a, ya, xa = celeste.gen_point_source_psf_image_with_fluxes(star, im)
    unit, _, _ = celeste.gen_point_source_psf_image(star.u, im)
    np.testing.assert_allclose(a, unit * (star.flux_dict[im.band] / im.calib) * im.kappa, rtol=1e-15)
    b, yb, xb = celeste.gen_src_image_with_fluxes(star, im)
    np.testing.assert_allclose(a, b, rtol=1e-14)
    # sample_source_photons_single_image_cython (celeste_mcmc.py:98-150): one patch per source, every photon kept, the first row
    # and column of a patch empty (celeste_sample_sources.pyx:50-51)
    samp, noise = celeste_mcmc.sample_source_photons_single_image_cython(im, ps, seed=3)
    assert len(samp) == len(ps)
    tot = noise
    for sp in samp:
        if sp is None:
            continue
        assert sp.data.shape == (sp.y1 - sp.y0, sp.x1 - sp.x0)         # NativePatch's fields (celeste_sample_sources.pyx:31-42)
        assert not sp.data[0].any() and not sp.data[:, 0].any() and sp.data.min() >= 0
        tot += sp.data.sum()
    assert tot == im.nelec.sum()
    many, noises = celeste_mcmc.sample_source_photons_multi_image(imgs[:3], ps, seed=3)
    assert len(many) == 3 and len(noises) == 3 and all(len(row) == len(ps) for row in many)
    for n in range(3):
        assert sum(sp.data.sum() for sp in many[n] if sp is not None) + noises[n] == imgs[n].nelec.sum()
    # the model classes' small accessors
    m = models.Celeste()
    m.initialize_sources(init_src_params=ps)
    assert set(m.source_types) <= {"star", "galaxy"} and m.srcs[0].object_type in ("star", "galaxy")
    top, idx = m.get_brightest(object_type="star", num_srcs=2, band="r", return_idx=True)
    fl = np.array([s.params.flux_dict["r"] for s in m.srcs])
    stars = np.nonzero(m.source_types == "star")[0]
    assert list(idx) == list(stars[np.argsort(fl[stars])[::-1]][:2]) and top[0] is m.srcs[idx[0]]
    s0 = m.srcs[int(stars[0])]
    patch, yl, xl = s0.compute_scatter_on_pixels(im)
    ref, yr, xr = celeste.gen_point_source_psf_image(s0.params.u, im)
    assert np.array_equal(patch, ref)
    s0.store_sample()
    assert s0.flux_samples.shape == (1, 5) and s0.shape_samples.shape == (1, 4) and s0.location_samples.shape == (1, 2)
    s0.clear_sample_images()
    assert s0.sample_image_list == []


@pytest.mark.gpu
def test_caller_stream_and_device_resident_pixels():
    """cel_ctx_set_stream with a torch stream and cel_images_set_nelec(mem = CEL_DEVICE) with a torch tensor's address (the
    zero-copy interop INTEGRATION.md describes): the numbers of the library's own stream and a host upload.  In a process of
    its own that imports torch FIRST, as bench.py does: torch ships a HIP runtime, and whichever is loaded first serves both."""
    import os, subprocess, sys
    from conftest import ROOT
    code = '''
import sys
sys.path.insert(0, %r)
import numpy as np, torch
torch.cuda.init()
import desi_mcmc_amd as cel
from desi_mcmc_amd import synth
ctx = cel.Context(0)
f = synth.SyntheticField(ctx, 200, 3, 192, 224, frac_gal=0.5, seed=4)
ll0, llb0 = f.images.render(f.sources, loglik=True)
lam0 = f.images.model_images()
st = torch.cuda.Stream()
ctx.set_stream(st.cuda_stream)
t = torch.from_numpy(f.nelec).to("cuda")
torch.cuda.synchronize()
f.images.set_nelec_device(t.data_ptr())
ll1, llb1 = f.images.render(f.sources, loglik=True)
assert ll1 == ll0 and np.array_equal(llb1, llb0) and np.array_equal(f.images.model_images(), lam0)
t2 = t * 1.0
t2[0, 5, 7] += 3.0
torch.cuda.synchronize()
f.images.set_nelec_device(t2.data_ptr())
_, llb2 = f.images.render(f.sources, loglik=True)
assert llb2[0] != llb0[0] and np.array_equal(llb2[1:], llb0[1:])
np.testing.assert_allclose(llb2[0] - llb0[0], 3.0 * np.log(lam0[0, 5, 7]), rtol=1e-9)
ctx.set_stream(None)
f.images.set_nelec(f.nelec)
assert f.images.render(f.sources, loglik=True)[0] == ll0
print("interop ok")
''' % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "interop ok" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])


def _src_bound_scene():
    from conftest import load_golden
    import desi_mcmc_amd as cel
    g = load_golden("src_bound.npz")
    rec = load_golden("bands_253.npz")
    imgs = [cel.FitsImage.from_record("ugriz"[b], rec, b, rec["nelec"][b]) for b in range(5)]
    srcs = [cel.SrcParams(u=g["radec"][s], a=int(g["is_gal"][s]), fluxes=g["flux"][s], theta=g["shape"][s, 0],
                          sigma=g["shape"][s, 1], phi=g["shape"][s, 2], rho=g["shape"][s, 3]) for s in range(len(g["radec"]))]
    return g, imgs, srcs


def test_gen_psf_src_image_bound_and_pixel_grid_golden():
    """celeste.gen_psf_src_image_bound (celeste.py:193-199) and FitsImage.make_pixel_grid / pixel_grid (fits_image.py:95,
    186-194) against what the reference's own functions returned on the real stamps (golden src_bound.npz, made by
    tests/golden/make_golden.py); host arithmetic through cel_bounding_radius: runs without a GPU."""
    from desi_mcmc_amd import celeste
    g, imgs, srcs = _src_bound_scene()
    for b, img in enumerate(imgs):
        got = np.array([celeste.gen_psf_src_image_bound(s, img) for s in srcs])
        np.testing.assert_allclose(got, g["bound"][b], rtol=1e-12)
        assert all(got[s] == img.R for s in range(len(srcs)) if srcs[s].a == 0)
        # the boxes of experiments/fields/process_field.py:107-117 formed from it
        locs = np.vstack([img.equa2pixel(s.u) for s in srcs])
        boxes = np.column_stack([np.floor(locs[:, 0] - got), np.ceil(locs[:, 0] + got),
                                 np.floor(locs[:, 1] - got), np.ceil(locs[:, 1] + got)])
        assert np.array_equal(boxes, g["boxes"][b])
    grid = imgs[2].make_pixel_grid()
    assert grid.dtype == np.float64 and grid.flags["C_CONTIGUOUS"]
    assert np.array_equal(grid, g["pixel_grid"]) and np.array_equal(imgs[2].pixel_grid, g["pixel_grid"])
    H, W = (int(v) for v in g["grid_HW"])
    assert grid.shape == (H * W, 2) and tuple(grid[0]) == (1.0, 1.0) and tuple(grid[1]) == (2.0, 1.0) and tuple(grid[-1]) == (W, H)
    assert imgs[2].pixel_grid is imgs[2].pixel_grid                       # kept around, like the reference's attribute


@pytest.mark.gpu
def test_process_field_call_sequence_runs_against_the_mirror():
    """experiments/fields/process_field.py:84-117 -- the (f)3 caller pattern -- line for line against the mirror: main()'s
    accumulation of flux-scaled patches per band, then sample_source_photons_single_image's set-up (pixel locations,
    gen_psf_src_image_bound radii, floor/ceil boxes, gen_src_image_with_fluxes patches).  The accumulated image is
    gen_model_image minus the sky; every patch lies inside the box its caller draws photons from."""
    from desi_mcmc_amd import celeste
    from desi_mcmc_amd.celeste_galaxy_conditionals import gen_galaxy_psf_image
    from desi_mcmc_amd.util.bound.bounding_box import get_bounding_boxes_idx
    g, imgs, srcs = _src_bound_scene()
    BANDS = "ugriz"
    imgfits = {b: imgs[j] for j, b in enumerate(BANDS)}
    modelims = {b: np.zeros(imgfits[b].nelec.shape, dtype=float) for b in BANDS}
    for src_params in srcs:                                                # :86-103
        for j, band in enumerate(BANDS):
            if src_params.a == 0:
                f_s, ylim, xlim = celeste.gen_point_source_psf_image_with_fluxes(src_params, imgfits[band], return_patch=True)
                if f_s is None:
                    continue
                scale = 1.0                                                # the reference multiplies a flux-scaled star patch by the
            else:                                                          # flux again (:103); kept out of the comparison below
                f_s, ylim, xlim = gen_galaxy_psf_image(th=[src_params.theta, src_params.sigma, src_params.phi, src_params.rho],
                                                       u_s=src_params.u, img=imgfits[band])
                scale = imgfits[band].nmgy2counts(src_params.fluxes[j])
            modelims[band][int(ylim[0]):int(ylim[1]), int(xlim[0]):int(xlim[1])] += f_s * scale
    for j, band in enumerate(BANDS):
        lam = celeste.gen_model_image(srcs, imgfits[band])
        np.testing.assert_allclose(modelims[band] + imgfits[band].epsilon, lam, rtol=1e-9)
    img = imgfits["r"]                                                     # :107-120
    src_locs = np.vstack([img.equa2pixel(s.u) for s in srcs])
    imgR = np.array([celeste.gen_psf_src_image_bound(s, img) for s in srcs])
    np.testing.assert_allclose(imgR, g["bound"][2], rtol=1e-12)
    src_boxes = np.column_stack([np.floor(src_locs[:, 0] - imgR), np.ceil(src_locs[:, 0] + imgR),
                                 np.floor(src_locs[:, 1] - imgR), np.ceil(src_locs[:, 1] + imgR)])
    assert np.array_equal(src_boxes, g["boxes"][2])
    src_imgs = [celeste.gen_src_image_with_fluxes(s, img) for s in srcs]
    for s, (patch, ylim, xlim) in enumerate(src_imgs):
        if patch is None:
            continue
        assert patch.shape == (int(ylim[1] - ylim[0]), int(xlim[1] - xlim[0]))
        # a pixel of the patch is one the caller's box test hands to this source (:129-137)
        y, x = int(ylim[0]), int(xlim[0])
        assert s in get_bounding_boxes_idx(np.array([x, y]), src_boxes)
        y, x = int(ylim[1]) - 1, int(xlim[1]) - 1
        assert s in get_bounding_boxes_idx(np.array([x, y]), src_boxes)
