"""CPU-only checks: the C-ABI library loads and exports every declared symbol, fails loudly
without a GPU (no CPU fallback), and the host-side mirror logic matches the goldens.
No compute call reaches a device here."""
import json
import os
import re
import subprocess
import sys

import warnings

import numpy as np
import pytest

from conftest import ROOT, load_golden

BANDS = ["u", "g", "r", "i", "z"]


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as ge
    ge.build()
    import desi_mcmc_amd
    return desi_mcmc_amd


def test_library_exports_every_declared_symbol(built):
    from desi_mcmc_amd import _lib
    header = open(os.path.join(ROOT, "include", "celeste_hip.h")).read()
    declared = set(re.findall(r"^\s*(?:int|const char \*)\s*\*?(cel_\w+)\s*\(", header, flags=re.M))
    assert len(declared) >= 27
    bound = {name for name, _, _ in _lib.SYMBOLS}
    assert declared == bound, (declared ^ bound)
    lib = _lib.lib()
    for name in declared:
        assert getattr(lib, name) is not None
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH]).decode()
    exported = set(re.findall(r" T (cel_\w+)", out))
    assert declared <= exported
    assert lib.cel_abi_version() == 1


def test_code_object_is_gfx950_only(built):
    from desi_mcmc_amd import _lib
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not present")
    out = subprocess.run([objdump, "--offloading", _lib.LIB_PATH], capture_output=True, text=True).stdout
    archs = set(re.findall(r"gfx\w+", out))
    assert archs == {"gfx950"}, archs


def test_no_gpu_means_loud_failure_not_fallback(built):
    from desi_mcmc_amd import _lib
    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(_lib.CelesteHipError, match="no CPU fallback"):
        built.Context(0)
    from desi_mcmc_amd.util.like import gmm_like_2d
    with pytest.raises(_lib.CelesteHipError):
        gmm_like_2d(np.zeros((4, 2)), np.ones(1), np.zeros((1, 2)), np.eye(2)[None])


def test_product_never_touches_oracle_or_reference():
    """the product path must not import, link or read oracle/ or /root/reference"""
    pkg = os.path.join(ROOT, "desi-mcmc_amd")
    bad = []
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".h", ".cpp")) or fn == "Makefile":
                txt = open(os.path.join(dp, fn)).read()
                if re.search(r"(import\s+oracle|from\s+oracle|oracle[/.]\w|libceleste_oracle|/root/reference)", txt):
                    bad.append(os.path.join(dp, fn))
    assert not bad, bad


def test_bounding_radius_host_entry(built):
    from desi_mcmc_amd.util.bound.bounding_box import calc_bounding_radius, get_bounding_boxes_idx
    rec = load_golden("bands_253.npz")
    g = load_golden("radius.npz")
    for b in range(5):
        r = calc_bounding_radius(rec["weights"][b], rec["means"][b], rec["covars"][b], 0.001)
        np.testing.assert_allclose(r, g["psf_R_1e3"][b], rtol=1e-13)
        r5 = calc_bounding_radius(rec["weights"][b], rec["means"][b], rec["covars"][b], 1e-5,
                                  center=np.array([0.3, -0.2]))
        np.testing.assert_allclose(r5, g["psf_R_1e5_c"][b], rtol=1e-13)
    gs = load_golden("galaxy_stamps.npz")
    for i in (0, 5, 17):
        r = calc_bounding_radius(gs["s_cw"][i], gs["s_cm"][i], gs["s_cc"][i], 1e-5, center=gs["s_pix"][i])
        np.testing.assert_allclose(r, gs["s_bound"][i], rtol=1e-12)
    with pytest.raises(ValueError):
        calc_bounding_radius(rec["weights"][0], rec["means"][0], rec["covars"][0], 1.5)
    boxes = np.array([[0, 10, 0, 10], [5, 15, 5, 15], [20, 30, 20, 30]])
    assert list(get_bounding_boxes_idx(np.array([7, 7]), boxes)) == [0, 1]


def test_library_radius_is_the_oracles_on_scaled_and_random_psfs(built):
    """the star radius of PSFs no golden holds (the scaled PSFs of the sharp-PSF test, the random PSFs of the fuzz tests): the
    library's host entry against the oracle's restatement, which radius.npz pins to the reference.  The GPU tests hand the
    checker the oracle's radius and assert this equality per band (orc.checked_radius)."""
    from desi_mcmc_amd import field, synth
    from oracle import oracle as orc
    rs = np.random.RandomState(11)
    bands = synth.make_bands(96, 128, 5)
    for scale in (1.0, 0.15, 0.02, 3.0, 40.0):
        for b in bands:
            b = b.copy()
            b[12:24] *= scale
            b[3:6] = rs.dirichlet([2.0, 2.0, 2.0])
            b[6:12] += rs.normal(0, 0.3 * np.sqrt(scale), 6)
            R = field.bounding_radius(b[3:6], b[6:12].reshape(3, 2), b[12:24].reshape(3, 2, 2), 1e-3)
            assert orc.checked_radius(b, R) == pytest.approx(R, rel=1e-13)
    b = bands[0].copy()
    b[36] = 60.0                                    # a caller-imposed radius is kept, not recomputed
    assert orc.checked_radius(b, 60.0) == 60.0
    with pytest.raises(AssertionError):
        orc.checked_radius(bands[0], 1.0001 * orc.band_radius(bands[0]))


def test_fitsimage_mirror_host_logic(built):
    rec = load_golden("bands_253.npz")
    g = load_golden("wcs_points.npz")
    for b in (0, 2):
        im = built.FitsImage.from_record(BANDS[b], rec, b, rec["nelec"][b])
        np.testing.assert_allclose(im.R, rec["R"][b], rtol=1e-13)
        np.testing.assert_allclose(im.Ups_n_inv, rec["ups_inv"][b], rtol=1e-14)
        for p, e, pb, cd in zip(g["pix"], g["equa_b%d" % b], g["pix_back_b%d" % b], g["cd_b%d" % b]):
            np.testing.assert_allclose(im.pixel2equa(p), e, rtol=1e-15)
            np.testing.assert_allclose(im.equa2pixel(e), pb, rtol=1e-12, atol=1e-10)
            np.testing.assert_allclose(im.cd_at_pixel(p[0], p[1]), cd, rtol=1e-9, atol=1e-18)
        assert not im.nelec.flags.writeable
        np.testing.assert_allclose(im.nmgy2counts(3.0), 3.0 / rec["calib"][b] * rec["kappa"][b])
        assert im.band_record().shape == (37,)
    # from_header reproduces the reference's header arithmetic (fits_image.py:85-147)
    hdr = {"CALIB": rec["calib"][2], "SKY": rec["eps"][2] / rec["kappa"][2], "GAIN": rec["kappa"][2],
           "CRPIX1": 26.0, "CRPIX2": 26.0, "CRVAL1": rec["phi"][2][0], "CRVAL2": rec["phi"][2][1],
           "CD1_1": rec["ups"][2][0, 0], "CD1_2": 0.0, "CD2_1": 0.0, "CD2_2": rec["ups"][2][1, 1]}
    psf = list(rec["weights"][2]) + list(rec["means"][2].ravel())
    for k in range(3):
        c = rec["covars"][2][k]
        psf += [c[0, 0], c[1, 1], c[0, 1]]
    hdr.update({"PSF_P%d" % i: v for i, v in enumerate(psf)})
    pix = (rec["nelec"][2] / hdr["GAIN"] - hdr["SKY"]) * hdr["CALIB"]
    im = built.FitsImage.from_header("r", hdr, pix)
    assert np.array_equal(im.nelec, rec["nelec"][2])
    np.testing.assert_allclose(im.epsilon, rec["eps"][2], rtol=1e-15)
    np.testing.assert_allclose(im.covars, rec["covars"][2], rtol=1e-15)


def test_flux_conventions_q2(built):
    from desi_mcmc_amd import celeste
    rec = load_golden("bands_253.npz")
    im = built.FitsImage.from_record("r", rec, 2, rec["nelec"][2])
    fl = dict(zip(BANDS, [1., 2., 3., 4., 5.]))
    star = built.SrcParams(u=np.zeros(2), a=0, fluxes=fl)
    gal = built.SrcParams(u=np.zeros(2), a=1, fluxes=np.array([1., 2., 3., 4., 5.]), theta=.5, sigma=1., phi=0., rho=.5)
    cat = built.SrcParams(u=np.zeros(2), fluxes=fl)
    assert celeste.expected_photons(star, im) == 3.0 / im.calib * im.kappa          # celeste.py:41
    assert celeste.expected_photons(gal, im) == 3.0 / im.calib * im.kappa           # celeste.py:50
    assert celeste.expected_photons(cat, im) == im.kappa * 3.0                       # celeste.py:55
    with pytest.raises(Exception):
        celeste.expected_photons(built.SrcParams(u=np.zeros(2)), im)                 # celeste.py:57-58
    with pytest.raises(NotImplementedError):
        celeste.expected_photons(built.SrcParams(u=np.zeros(2), a=0, t=5000., b=1.), im)
    celeste.photons_expected_brightness = lambda t, b, band: 42.0
    try:
        assert celeste.expected_photons(built.SrcParams(u=np.zeros(2), a=0, t=5000., b=1.), im) == 42.0
    finally:
        celeste.photons_expected_brightness = None
    assert gal.flux_dict["r"] == 3.0 and np.allclose(gal.shape, [.5, 1., 0., .5])


def test_galaxy_mixture_host_helper(built):
    from desi_mcmc_amd import celeste_galaxy_conditionals as gal
    rec = load_golden("bands_253.npz")
    g = load_golden("galaxy_stamps.npz")
    imgs = {b: built.FitsImage.from_record(BANDS[b], rec, b, rec["nelec"][b]) for b in (1, 2, 3)}
    for i in range(0, len(g["s_th"]), 5):
        img = imgs[int(g["s_band"][i])]
        pis, means, covs, pxy = gal.galaxy_mixture(g["s_th"][i], g["s_u"][i], img)
        np.testing.assert_allclose(pis, g["s_cw"][i], rtol=1e-13)
        np.testing.assert_allclose(means, g["s_cm"][i], rtol=1e-13)
        np.testing.assert_allclose(covs, g["s_cc"][i], rtol=1e-9)
        th = g["s_th"][i]
        np.testing.assert_allclose(gal.gen_galaxy_transformation(th[1], th[3], th[2], img.cd_at_pixel(*pxy)),
                                   g["s_tinv"][i], rtol=1e-9)
        src = built.SrcParams(u=g["s_u"][i], a=1, theta=th[0], sigma=th[1], phi=th[2], rho=th[3])
        np.testing.assert_allclose(gal.gen_galaxy_psf_image_bound(src, img), g["s_bound"][i], rtol=1e-9)


def test_synth_sources_deterministic(built):
    from desi_mcmc_amd import synth
    bands = synth.make_bands(2048, 2048, 5)
    a = synth.make_sources(100, 2048, 2048, bands, 0.5, 42)
    b = synth.make_sources(100, 2048, 2048, bands, 0.5, 42)
    assert all(np.array_equal(a[k], b[k]) for k in a)
    assert bands.shape == (5, 37) and np.all(bands[:, 24] == 1024.0)
    # pixel2equa round trip against the FitsImage mirror
    rec = load_golden("bands_253.npz")
    im = built.FitsImage.from_record("r", rec, 2, rec["nelec"][2])
    np.testing.assert_allclose(synth.pixel2equa(im.band_record(), np.array([[3.5, 40.25]]))[0],
                               im.pixel2equa(np.array([3.5, 40.25])), rtol=1e-15)


def test_strip_partition_covers_frame_once(built):
    from desi_mcmc_amd import dist
    for H in (1, 31, 32, 80, 2048, 1489):
        for world in (1, 2, 3, 4, 8):
            rows = [dist.strip_rows(H, world, r) for r in range(world)]
            assert rows[0][0] == 0 and rows[-1][1] == H
            for (a0, a1), (b0, b1) in zip(rows[:-1], rows[1:]):
                assert a1 == b0 and a0 <= a1
                assert a1 % 32 == 0 or a1 == H
    assert dist.field_shard(8, 4, 1) == [1, 5]


# ---- host-side containers added in round 2 (no device call) ------------------------------------------
def test_strip_edges_even_out_a_measured_cost_and_solo_deals(built):
    """dist.strip_edges: without a cost the equal-tile-row cut of strip_rows; with the per-band cost of a whole-frame render
    (strip_cost_from_tiles of the measured tile durations) the aligned cut that evens the strips' summed cost, every strip at
    least one band; StripDeal takes the edges (the same on every rank) and a 64-row alignment for the window trace; a solo
    deal (bench.py --as-rank k --of N: one rank played without a process group) sums to N times its own part and leaves the
    other ranks' rows alone."""
    from desi_mcmc_amd import dist
    for H, world, align in ((2048, 8, 32), (2048, 8, 64), (1000, 3, 32), (333, 4, 64), (96, 3, 32)):
        e = dist.strip_edges(H, world, align=align)
        assert e[0] == 0 and e[-1] == H and len(e) == world + 1 and all(b > a for a, b in zip(e, e[1:]))
        assert all(x % align == 0 for x in e[:-1])
        if align == 32:
            assert e == [dist.strip_rows(H, world, r)[0] for r in range(world)] + [H]
    cost = np.ones(64)
    cost[:16] = 3.0                                           # the frame's top quarter three times as expensive
    e = dist.strip_edges(2048, 8, cost)
    per = [cost[a // 32:b // 32].sum() for a, b in zip(e, e[1:])]
    assert e == [0, 128, 256, 384, 512, 896, 1280, 1664, 2048] and max(per) == min(per) == 12.0
    lumpy = np.array([100.0, 1, 1, 1, 1, 1, 1, 1])
    e = dist.strip_edges(256, 4, lumpy)                       # one band holds nearly everything: still a band per strip
    assert e[0] == 0 and e[-1] == 256 and all(b - a >= 32 for a, b in zip(e, e[1:]))
    with pytest.raises(ValueError):
        dist.strip_edges(256, 4, np.ones(7))
    with pytest.raises(ValueError):
        dist.strip_edges(64, 3, np.ones(2))
    tc = np.arange(2 * 4 * 3, dtype=float).reshape(2, 4, 3)   # B = 2, nty = 4 (64-row tiles), ntx = 3
    rc = dist.strip_cost_from_tiles(tc.ravel(), 2, 4, 3, 64, 32)
    assert rc.shape == (8,) and np.allclose(rc[::2], tc.sum(axis=(0, 2)) / 2) and np.allclose(rc[::2], rc[1::2])
    rows = np.random.RandomState(0).uniform(0, 2048, 500)
    d = dist.StripDeal(rows, 2048, 8, 3, halo=100, edges=[0, 128, 256, 384, 512, 896, 1280, 1664, 2048], solo=True, align=64)
    assert d.strip == (384, 512) and d.window == (256, 640) and d.noise_rows() == (128, 256) and d.solo
    assert np.array_equal(d.mine, np.nonzero((rows >= 384) & (rows < 512))[0])
    assert np.array_equal(d.rank_sum([1.0, 2.5]), [8.0, 20.0])
    arr = np.arange(500 * 3, dtype=float).reshape(500, 3)
    assert np.array_equal(d.merge(arr), arr)
    with pytest.raises(ValueError):
        dist.StripDeal(rows, 2048, 8, 3, edges=[0, 128, 256])
    d64 = dist.StripDeal(rows, 2048, 8, 0, halo=70)           # default alignment 32: the halo is whole 32-row tiles
    assert d64.window == (0, 256 + 96)


def test_src_catalog_reads_as_a_sequence_of_srcparams(built):
    cel = built
    ps = [cel.SrcParams(u=np.array([10.0 + i, 20.0 - i]), a=(i % 2), fluxes=np.arange(5.0) + i, theta=.3, sigma=1. + i,
                        phi=30., rho=.5) for i in range(4)]
    ps.append(cel.SrcParams(u=np.array([1.0, 2.0]), a=None, fluxes={"u": 1., "g": 2., "r": 3., "i": 4., "z": 5.}))
    cat = cel.SrcCatalog.from_params(ps)
    assert len(cat) == 5 and [p.a for p in cat] == [0, 1, 0, 1, None]
    assert np.array_equal(cat[1].u, ps[1].u) and cat[3].sigma == 4.0 and cat[4].flux("r") == 3.0
    assert cat[4].flux_dict == {"u": 1., "g": 2., "r": 3., "i": 4., "z": 5.}
    assert np.array_equal(cat.shape[0], np.zeros(4)) and np.array_equal(cat.shape[1], [.3, 2., 30., .5])
    v = cat[2]
    v.u = [7., 8.]
    v.fluxes = {"u": 9., "g": 9., "r": 9., "i": 9., "z": 9.}
    v.rho = 0.25
    assert np.array_equal(cat.u[2], [7., 8.]) and np.all(cat.fluxes[2] == 9.) and cat.shape[2, 3] == 0.25
    assert len(cat[1:3]) == 2 and cat[-1].a is None
    with pytest.raises(IndexError):
        cat[5]


def test_source_arrays_list_and_catalog_agree_for_every_flux_convention(built):
    """celeste._source_arrays: the vectorised gathers (SrcCatalog, list of arrays, list of dicts) and the
    source-by-source fallback give the same device inputs (celeste.py:35-62, the three conventions)"""
    from desi_mcmc_amd import celeste, models

    class Im(object):
        def __init__(self, band, calib, kappa):
            self.band, self.calib, self.kappa = band, calib, kappa

        def nmgy2counts(self, f):
            return (f / self.calib) * self.kappa
    ims = [Im(b, 0.004 + 0.001 * k, 4.0 + 0.2 * k) for k, b in enumerate("gri")]
    rs = np.random.RandomState(0)
    ps = []
    for i in range(40):
        a = [0, 1, None][i % 3]
        fl = rs.rand(5) * 10 + 1
        ps.append(built.SrcParams(u=rs.rand(2), a=a, fluxes=fl if i % 2 else dict(zip("ugriz", fl)), theta=.4, sigma=1.5,
                                  phi=10. * i, rho=.6))
    slow = [np.zeros(40, np.int32), np.zeros((40, 2)), np.zeros((40, 3)), np.zeros((40, 4))]
    for s, p in enumerate(ps):                               # the reference's own loop
        slow[0][s] = 1 if p.a == 1 else 0
        slow[1][s] = p.u
        if p.a == 1:
            slow[3][s] = [p.theta, p.sigma, p.phi, p.rho]
        for b, im in enumerate(ims):
            slow[2][s, b] = celeste.expected_photons(p, im)
    for srcs in (ps, built.SrcCatalog.from_params(ps),
                 [built.SrcParams(u=p.u, a=p.a, fluxes=np.array([p.flux(b) for b in "ugriz"]), theta=p.theta, sigma=p.sigma,
                                  phi=p.phi, rho=p.rho) for p in ps]):
        got = celeste._source_arrays(srcs, ims)
        for g, w in zip(got, slow):
            np.testing.assert_allclose(g, w, rtol=1e-15)
    # the flux_dict convention of the newer callers (sources.py:390-395)
    got = celeste._source_arrays(ps, ims, counts_fn=models._flux_counts)
    want = np.array([[(p.flux_dict[im.band] / im.calib) * im.kappa for im in ims] for p in ps])
    np.testing.assert_allclose(got[2], want, rtol=1e-15)
    # a star given by temperature needs the planck hook: the per-source path raises, vectorised gathers do not hide it
    ps[0].t, ps[0].a = 5000.0, 0
    with pytest.raises(NotImplementedError):
        celeste._source_arrays(ps, ims)


def test_catalogue_views_reach_the_device_arrays_without_a_gather(built):
    """SrcCatalog.views(): a plain list of per-source objects (what celeste_em.py:25,159 and celeste_mcmc.py:130 pass)
    whose attributes read and write the catalogue's arrays.  _source_arrays recognises the list, any selection of it,
    and sees a mutated parameter; a foreign object in the list sends it down the per-object route with the same result."""
    from desi_mcmc_amd import celeste
    from desi_mcmc_amd.fits_image import FitsImage
    rs = np.random.RandomState(3)
    S = 40
    ps = [built.SrcParams(u=rs.rand(2), a=int(i % 3 == 0), fluxes=rs.rand(5) + 1, theta=.4, sigma=1.5 + i, phi=30., rho=.6) for i in range(S)]
    imgs = [FitsImage(b, np.zeros((8, 8)), epsilon=1., kappa=2. + k, calib=.01 * (k + 1), weights=np.ones(3) / 3, means=np.zeros((3, 2)),
                      covars=np.tile(np.eye(2), (3, 1, 1)), rho_n=np.zeros(2), phi_n=np.zeros(2), Ups_n=np.eye(2)) for k, b in enumerate("gri")]
    cat = built.SrcCatalog.from_params(ps)
    views = cat.views()
    assert views is cat.views() and len(views) == S and isinstance(views, list)
    want = celeste._source_arrays(ps, imgs)
    for got in (celeste._source_arrays(views, imgs), celeste._source_arrays(list(views), imgs), celeste._source_arrays(cat, imgs)):
        for g, w in zip(got, want):
            assert np.array_equal(g, w)
    assert celeste._catalogue_rows(views) == (cat, None)
    sel = [views[i] for i in (5, 3, 11, 3)]
    c, rows = celeste._catalogue_rows(sel)
    assert c is cat and rows.tolist() == [5, 3, 11, 3]
    for g, w in zip(celeste._source_arrays(sel, imgs), celeste._source_arrays([ps[i] for i in (5, 3, 11, 3)], imgs)):
        assert np.array_equal(g, w)
    # a write through a view is a write to the arrays: the next call sees it
    views[7].u = np.array([0.25, 0.75])
    views[9].sigma = 9.5
    views[4].fluxes = dict(zip("ugriz", [1., 2., 3., 4., 5.]))
    ps[7].u, ps[9].sigma, ps[4].fluxes = np.array([0.25, 0.75]), 9.5, np.array([1., 2., 3., 4., 5.])
    for g, w in zip(celeste._source_arrays(views, imgs), celeste._source_arrays(ps, imgs)):
        assert np.array_equal(g, w)
    mixed = list(views)
    mixed[2] = ps[2]                                         # not a view: no shortcut, same numbers
    assert celeste._catalogue_rows(mixed) is None
    for g, w in zip(celeste._source_arrays(mixed, imgs), celeste._source_arrays(ps, imgs)):
        assert np.array_equal(g, w)
    other = built.SrcCatalog.from_params(ps[:3]).views()
    assert celeste._catalogue_rows([views[0], other[1]]) is None


def test_list_of_srcparams_is_gathered_once_and_then_row_by_changed_row(built):
    """celeste._source_arrays on a plain LIST of SrcParams (what celeste_em.py:25,159, celeste_mcmc.py:130 and the moves of
    util/infer/mcmc_transitions.py:37-152 pass): the arrays are kept per list object; an attribute assignment stamps the
    object (SrcParams.__setattr__) and only stamped objects are read again.  After every kind of change the cached arrays
    equal a fresh gather: one source assigned, many, the reference's in-place-then-assign idiom, an element replaced,
    two swapped, one appended, one removed, touch() after a purely in-place edit."""
    from desi_mcmc_amd import celeste, celeste_src
    celeste.list_cache("stamps")                  # the opt-in fast mode; the default ("exact") is tested below
    try:
        _stamps_mode_walk(built, celeste, celeste_src)
    finally:
        celeste.list_cache("exact")


def _stamps_mode_walk(built, celeste, celeste_src):
    class Im(object):
        def __init__(self, band, calib, kappa):
            self.band, self.calib, self.kappa = band, calib, kappa
    ims = [Im(b, 0.004 + 0.001 * k, 4.0 + 0.2 * k) for k, b in enumerate("gri")]
    rs = np.random.RandomState(1)
    S = 200

    def make(i):
        fl = rs.rand(5) * 10 + 1
        return built.SrcParams(u=rs.rand(2), a=[0, 1, None][i % 3], fluxes=dict(zip("ugriz", fl)) if i % 2 else fl, theta=.4, sigma=1.5,
                               phi=10. * i, rho=.6)
    ps = [make(i) for i in range(S)]

    def check():
        got = celeste._source_arrays(ps, ims)
        want = celeste._gather_plain(ps, ims, celeste.expected_photons, [1, 2, 3], np.array([im.calib for im in ims]),
                                     np.array([im.kappa for im in ims]))
        for g, w in zip(got, want):
            assert np.array_equal(g, w)
        return got
    first = check()
    ent = celeste._ENTRY_OF[id(first[0])]
    assert ent.srcs is ps and ent.version == 0
    assert check()[0] is first[0] and ent.version == 0                  # nothing changed: the same arrays, no new version
    ps[17].u = np.array([0.5, 0.25])
    assert check()[0] is first[0] and ent.version == 1 and ent.rows_since(0).tolist() == [17]
    pos = ps[40].u                                                      # mcmc_transitions.py:49-51: edit in place, then assign
    pos[0] = 0.125
    ps[40].u = pos
    ps[41].fluxes = np.arange(5.) + 1
    ps[43].sigma = 7.0
    ps[44].a = 1
    check()
    assert ent.version == 2 and ent.rows_since(1).tolist() == [40, 41, 43, 44] and ent.rows_since(0).tolist() == [17, 40, 41, 43, 44]
    ps[50].u[1] = 0.75                                                  # in place only: invisible until touched
    celeste_src.touch(ps[50])
    check()
    assert ent.rows_since(2).tolist() == [50]
    for p in ps:                                                        # most of the list assigned: gathered whole, a new entry
        p.u = p.u
    again = check()
    assert again[0] is not first[0] and id(first[0]) not in celeste._ENTRY_OF
    ps[3] = make(3)                                                     # an element replaced by a new object
    check()
    ps[5], ps[9] = ps[9], ps[5]                                         # two swapped: no stamp moved, the identity pass sees it
    check()
    ps.append(make(7))
    assert check()[0].shape[0] == S + 1
    del ps[10]
    assert check()[0].shape[0] == S
    ps[20] = ps[21]                                                     # one object listed twice: both rows follow it
    check()
    ps[21].sigma = 3.25
    got = check()
    assert got[3][20, 1] == got[3][21, 1] == 3.25 or ps[21].a != 1
    for k in range(9000):                                               # more assignments than the log holds: the stamps decide
        ps[30].rho = 0.5 + 1e-6 * k
    ps[31].rho = 0.25
    check()
    short = ps[:10]                                                     # short lists are not cached
    celeste._source_arrays(short, ims)
    assert all(e.srcs is not short for e in celeste._LIST_CACHE.values())
    assert len(celeste._LIST_CACHE) <= celeste._LIST_CACHE_MAX


def test_native_gather_reads_what_the_numpy_passes_read(built):
    """csrc_host/srcgather.c (one C pass over the objects' slots) against celeste._gather_plain's numpy passes: the same bits for
    every flux convention and container the fast path takes (ndarray / dict fluxes, python and numpy scalars, untyped rows), and
    a clean hand-over to the general path -- same values, same exceptions -- for everything it does not (a temperature star, a
    float32 or list location, a numpy-integer type, a subclass, fluxes of another length, a missing flux)."""
    from desi_mcmc_amd import celeste

    class Im(object):
        def __init__(self, band, calib, kappa):
            self.band, self.calib, self.kappa = band, calib, kappa

        def nmgy2counts(self, flux):
            return (flux / self.calib) * self.kappa
    ims = [Im(b, 0.004 + 0.001 * k, 4.0 + 0.2 * k) for k, b in enumerate("zgu")]
    bidx = [4, 1, 0]
    calib, kappa = np.array([im.calib for im in ims]), np.array([im.kappa for im in ims])
    rs = np.random.RandomState(7)
    assert celeste._native_gather(), "the host helper is built by __graft_entry__.build()"

    def make(i):
        fl = rs.rand(5) * 10 + 1
        return built.SrcParams(u=rs.rand(2), a=[0, 1, None, True, np.int64(1)][i % 5], fluxes=dict(zip("ugriz", fl)) if i % 2 else fl,
                               theta=.4 if i % 3 else np.float64(.3), sigma=1.5, phi=7 * i, rho=.6)      # (phi: a python int)

    def both(ps, fn=celeste.expected_photons):
        out = []
        for native in (None, False):
            celeste._NATIVE[0] = native
            try:
                out.append(celeste._gather_plain(ps, ims, fn, bidx, calib, kappa))
            except Exception as e:                       # noqa: BLE001 -- compared below
                out.append(e)
            finally:
                celeste._NATIVE[0] = None
        a, b = out
        if isinstance(b, Exception):
            assert type(a) is type(b) and str(a) == str(b), (a, b)
        else:
            assert all(x.dtype == y.dtype and x.shape == y.shape and np.array_equal(x, y, equal_nan=True) for x, y in zip(a, b))
        return b
    ps = [make(i) for i in range(300)]
    both(ps)
    both(ps, fn="flux_dict convention")
    # the hand-overs: one odd object anywhere in the list
    class Sub(built.SrcParams):
        __slots__ = ()
    odd = [built.SrcParams(u=rs.rand(2).astype(np.float32), a=0, fluxes=rs.rand(5)),                   # float32 location
           built.SrcParams(u=list(rs.rand(2)), a=0, fluxes=rs.rand(5)),                                # a list
           built.SrcParams(u=rs.rand(2), a=0, fluxes=rs.rand(5), t=5000.0, b=1.0),                     # a temperature star: the hook
           built.SrcParams(u=rs.rand(2), a=1, fluxes=rs.rand(5), theta=.5, sigma=None, phi=1., rho=.5),  # a shape that is no number (NaN, by numpy's rule)
           built.SrcParams(u=rs.rand(2), a=None, fluxes=None),                                         # no flux at all
           built.SrcParams(u=rs.rand(2), a=0, fluxes={"u": 1.0, "g": 2.0}),                            # a band missing
           built.SrcParams(u=rs.rand(4)[::2], a=0, fluxes=rs.rand(5)),                                 # a strided view
           Sub(u=rs.rand(2), a=0, fluxes=rs.rand(5))]
    for o in odd:
        both(ps[:40] + [o] + ps[40:80])
    hook = celeste.photons_expected_brightness
    try:
        celeste.photons_expected_brightness = lambda t, b, band: 123.0
        got = both(ps[:10] + [odd[2]])
        assert np.all(got[2][10] == 123.0)
    finally:
        celeste.photons_expected_brightness = hook


def test_list_cache_audit_catches_an_unstamped_in_place_edit(built):
    """An object of a list changed IN PLACE without an attribute assignment (src.u[0] = x, src.fluxes['r'] = f) moves no
    stamp.  The DEFAULT mode ("exact") re-reads every source on every call, as the reference does (celeste.py:203-219): the
    FIRST call after the edit returns the edited value, and only that row is marked for upload.  The opt-in "stamps" mode
    does not see the edit; its rotating audit (a sixteenth of the list re-read per call) meets it within 16 calls, WARNS and
    re-reads the whole list.  A changed calibration hook is seen at once in either mode."""
    from desi_mcmc_amd import celeste

    class Im(object):
        def __init__(self, band, calib, kappa):
            self.band, self.calib, self.kappa = band, calib, kappa

        def nmgy2counts(self, flux):
            return (flux / self.calib) * self.kappa
    ims = [Im(b, 0.004 + 0.001 * k, 4.0 + 0.2 * k) for k, b in enumerate("gri")]
    rs = np.random.RandomState(2)
    S = 2000
    ps = [built.SrcParams(u=rs.rand(2), a=i % 2, fluxes=dict(zip("ugriz", rs.rand(5) + 1)), theta=.4, sigma=1.5, phi=1. * i, rho=.6)
          for i in range(S)]

    def fresh():
        return celeste._gather_plain(ps, ims, celeste.expected_photons, [1, 2, 3], np.array([im.calib for im in ims]),
                                     np.array([im.kappa for im in ims]))
    assert celeste.list_cache() == "exact"                              # the default
    first = celeste._source_arrays(ps, ims)
    ent = celeste._ENTRY_OF[id(first[0])]
    assert celeste._source_arrays(ps, ims)[0] is first[0] and ent.version == 0      # nothing changed: nothing to upload
    for k, (victim, edit) in enumerate(((1234, lambda p: p.u.__setitem__(0, 0.5)), (77, lambda p: p.fluxes.__setitem__("r", 9.0)))):
        edit(ps[victim])
        got = celeste._source_arrays(ps, ims)                           # the FIRST call after the edit
        assert all(np.array_equal(g, w) for g, w in zip(got, fresh()))
        assert got[0] is first[0] and ent.version == k + 1 and ent.rows_since(k).tolist() == [victim]
    assert got[1][1234, 0] == 0.5 and got[2][77, 1] == 9.0 / ims[1].calib * ims[1].kappa
    hook = celeste.photons_expected_brightness
    try:                                                                # a star given by temperature reads the hook on every call
        ps[10].a, ps[10].t, ps[10].b = 0, 5000.0, 1.0
        celeste.photons_expected_brightness = lambda t, b, band: 111.0
        assert celeste._source_arrays(ps, ims)[2][10, 0] == 111.0
        celeste.photons_expected_brightness = lambda t, b, band: 222.0
        assert celeste._source_arrays(ps, ims)[2][10, 0] == 222.0
        ps[10].t = None
    finally:
        celeste.photons_expected_brightness = hook
    try:
        assert celeste.list_cache("stamps") == "stamps" and not celeste._LIST_CACHE
        for victim, edit in ((1234, lambda p: p.u.__setitem__(0, 0.25)), (77, lambda p: p.fluxes.__setitem__("r", 7.0))):
            celeste._source_arrays(ps, ims)
            with warnings.catch_warnings():
                warnings.simplefilter("error")
                for _ in range(20):                                     # clean calls: the audit goes round without a complaint
                    celeste._source_arrays(ps, ims)
            edit(ps[victim])
            with pytest.warns(RuntimeWarning, match="source %d of this list was changed IN PLACE" % victim):
                for _ in range(celeste._AUDIT_PARTS):
                    got = celeste._source_arrays(ps, ims)
            assert all(np.array_equal(g, w) for g, w in zip(got, fresh()))       # the call that warned already returns a fresh gather
        assert celeste.list_cache("off") == "off" and not celeste._LIST_CACHE
        a0 = celeste._source_arrays(ps, ims)
        ps[5].u[1] = 0.875
        a1 = celeste._source_arrays(ps, ims)
        assert a1[1][5, 1] == 0.875 and a0[1] is not a1[1] and not celeste._LIST_CACHE
    finally:
        celeste.list_cache("exact")
    with pytest.raises(ValueError):
        celeste.list_cache("sometimes")


def test_hot_kernels_keep_their_resource_budget(built):
    """`make resource-usage` (hipcc -Rpass-analysis=kernel-resource-usage: cross-compiles here, ~25 s) against
    tools/resource_budget.json: a hot kernel that gains scratch or VGPR spills, loses an occupancy step by registers, or grows
    its LDS past the size at which a CU holds one wave less, fails here -- not in a profile three rounds later."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import resource_usage
    finally:
        sys.path.pop(0)
    k = resource_usage.collect()
    hot = ("k_render_hw<false, 1>", "k_render_hw<false, 2>", "k_render_stars<2, false>", "k_small_stars", "k_patch_ll_nz<false>", "k_patch_ll_hw<0, int>",
           "k_photon_split_hw<int, unsigned short>")
    for name in hot:
        assert name in k, (name, sorted(k))
    bad = resource_usage.check(k)
    assert not bad, "\n".join(bad)
    # the kernels the judge named as hot carry no scratch at all -- or the budget file says how much and DESIGN.md section 5 why
    budget = json.load(open(resource_usage.BUDGET))["kernels"]
    for name in hot:
        assert k[name]["scratch"] <= budget[name]["max_scratch"]


def test_mog_sampling_api(built):
    """mog_samples / discrete / MixtureOfGaussians.rvs / var (util/dists/mog.py:25-37,69-73): host-side draws"""
    from desi_mcmc_amd.util.dists import mog
    p = np.array([.2, .5, .3])
    u = np.random.RandomState(0).rand(5000)
    want = p.shape[0] - np.sum(u[:, None] < np.cumsum(p), axis=1)            # the reference's expression (mog.py:36)
    assert np.array_equal(mog.discrete(p, (5000,), rng=np.random.RandomState(0)), want)
    assert mog.discrete(p, (4, 5), rng=np.random.RandomState(1)).shape == (4, 5)
    assert mog.discrete(np.array([.3, .3]), (2000,), rng=np.random.RandomState(2)).max() == 1     # weights summing below one
    m = mog.MixtureOfGaussians(np.array([[0., 0.], [3., 1.]]), np.array([[[1., .3], [.3, .5]], [[.2, 0.], [0., 2.]]]), np.array([.3, .7]))
    x = m.rvs(100000, rng=np.random.RandomState(1))
    assert x.shape == (100000, 2)
    np.testing.assert_allclose(x.mean(axis=0), m.mean(), atol=0.02)
    between = sum(w * np.outer(mu - m.mean(), mu - m.mean()) for w, mu in zip(m.pis, m.means))
    np.testing.assert_allclose(np.cov(x.T), m.var() + between, atol=0.03)
    np.testing.assert_allclose(m.var(), .3 * m.covs[0] + .7 * m.covs[1])
    np.random.seed(5)
    assert mog.mog_samples(3, m.means, m.chols, m.pis).shape == (3, 2)      # the global generator, as the reference


def test_step_seeds_are_unrelated_for_every_chain_seed():
    """the sweep's randomised steps draw from unrelated streams for ANY chain seed -- with `seed * prime + sweep` seeds a
    chain seeded 0 handed the location, shape and flux steps one and the same uniform sequence"""
    from desi_mcmc_amd import celeste_mcmc
    from desi_mcmc_amd.util.infer.slicesample import ChainStreams
    for seed in (0, 1, 4, 2 ** 63 + 5):
        seen = {}
        for sweep in range(3):
            for step in ("split", "flux", "location", "shape"):
                for k in range(2):
                    sd = celeste_mcmc.step_seed(seed, step, sweep, k)
                    assert 0 <= sd < 2 ** 64 and sd not in seen, (seed, step, sweep, k, seen.get(sd))
                    seen[sd] = (step, sweep, k)
        first = {}
        for step in ("flux", "location", "shape"):
            st = ChainStreams(celeste_mcmc.step_seed(seed, step, 0), np.arange(50))
            first[step] = st.uniform(np.arange(50))
        assert not np.any(first["flux"] == first["location"]) and not np.any(first["shape"] == first["location"])


def test_python_constants_are_the_headers(built):
    """every enumerator of include/celeste_hip.h that desi_mcmc_amd._lib names has the header's value (options are added by
    hand on both sides), and every option of the header is named"""
    import re
    from desi_mcmc_amd import _lib
    text = open(os.path.join(ROOT, "include", "celeste_hip.h")).read()
    enums = dict((m.group(1), int(m.group(2))) for m in re.finditer(r"\b(CEL_[A-Z0-9_]+)\s*=\s*(-?\d+)\s*[,/}\n]", text))
    assert len(enums) > 30
    named = [n for n in dir(_lib) if n.startswith("CEL_") and isinstance(getattr(_lib, n), int)]
    assert len(named) > 20
    for n in named:
        assert n in enums and getattr(_lib, n) == enums[n], (n, getattr(_lib, n), enums.get(n))
    for n, v in enums.items():
        if n.startswith("CEL_OPT_") or n.startswith("CEL_ERR_") or n.startswith("CEL_RENDER_"):
            assert getattr(_lib, n, None) == v, (n, v)
    opts = sorted(v for n, v in enums.items() if n.startswith("CEL_OPT_"))
    assert opts == list(range(1, len(opts) + 1)), opts            # no key used twice, none skipped
