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
