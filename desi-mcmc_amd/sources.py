"""Per-source conditional likelihoods: mirror of the hot methods of CelestePy/sources.py.

`Source` keeps the reference's method names and argument meaning for the calls that sit on the
render path -- compute_scatter_on_pixels / compute_model_patch (sources.py:351-395),
flux_in_image (:120-129), get_bounding_box (:83-96), log_likelihood / log_likelihood_isolated /
location_likelihood (:134-237) -- and adds log_likelihood_batch, which scores many proposals in
one launch (the reference's slice sampler and HMC call log_likelihood 10-50 times per source per
sweep, sources.py:308-319).  Sampling itself (resample_*, Gamma draws, slice sampling) is host
control flow outside this path and is not reproduced here.
"""
import numpy as np

from . import celeste as _celeste
from . import celeste_galaxy_conditionals as gal_funs

BANDS = ['u', 'g', 'r', 'i', 'z']


def poisson_loglike(data, model_img, mask):
    """sum log(m) d - sum m over m > 0 & mask  -- sources.py:6-12"""
    assert model_img.shape == mask.shape
    assert data.shape == model_img.shape
    good_pix = (model_img > 0.) & (mask != 0)
    return np.sum(np.log(model_img[good_pix]) * data[good_pix]) - np.sum(model_img[good_pix])


class SamplePatch(object):
    """What the reference's NativePatch carries (celeste_sample_sources.pyx:31-42): patch data and
    its place in the field, x0/x1/y0/y1."""

    def __init__(self, data, ylim, xlim):
        self.data = np.ascontiguousarray(data, dtype=np.float64)
        self.y0, self.y1 = int(ylim[0]), int(ylim[1])
        self.x0, self.x1 = int(xlim[0]), int(xlim[1])


def log_likelihood_sweep(srcs, us, fluxes=None, shapes=None, isolated=False):
    """Source.log_likelihood for proposals of MANY sources in one device launch -- what a whole
    sweep of per-source location / flux / shape updates needs (sources.py:242-349).
        srcs   list of Source, all with sample images on the same image objects
        us     (S, P, 2) proposed locations; fluxes (S, P, 5) / shapes (S, P, 4) or None (current)
    -> ll (S, P)"""
    S = len(srcs)
    us = np.asarray(us, dtype=np.float64)
    P = us.shape[1]
    imgs = None
    for s in srcs:
        cur = tuple(id(fi) for (_, fi, _) in s.sample_image_list)
        if s.sample_image_list:
            if imgs is None:
                imgs = tuple(fi for (_, fi, _) in s.sample_image_list)
            elif not set(cur) <= set(id(i) for i in imgs):
                raise ValueError("log_likelihood_sweep: sources sampled on different image sets")
    if imgs is None:
        return np.zeros((S, P))
    if len(imgs) > 16 or any(im.nelec.shape != imgs[0].nelec.shape for im in imgs):
        raise ValueError("log_likelihood_sweep needs <= 16 same-shape images; use Source.log_likelihood_batch")
    iset = _celeste._image_set(imgs)
    B = len(imgs)
    pos = {id(im): b for b, im in enumerate(imgs)}
    typ = np.repeat(np.array([1 if s.is_galaxy() else 0 for s in srcs], dtype=np.int32), P)
    fl = np.empty((S, P, 5))
    sh = np.zeros((S, P, 4))
    for i, s in enumerate(srcs):
        fl[i] = np.array([s.params.flux_dict[b] for b in BANDS]) if fluxes is None else fluxes[i]
        if s.is_galaxy():
            sh[i] = np.asarray(s.params.shape, dtype=np.float64) if shapes is None else shapes[i]
    counts = np.stack([(fl[..., BANDS.index(im.band)] / im.calib) * im.kappa for im in imgs], axis=-1).reshape(S * P, B)
    boxes = np.zeros((S, B, 4), dtype=np.int32)
    patches = [[None] * B for _ in range(S)]
    for i, s in enumerate(srcs):
        for (samp, im, _) in s.sample_image_list:
            b = pos[id(im)]
            boxes[i, b] = [samp.y0, samp.y1, samp.x0, samp.x1]
            patches[i][b] = im.nelec[samp.y0:samp.y1, samp.x0:samp.x1] if isolated else np.array(samp.data)
    sset = iset._sources(typ, us.reshape(S * P, 2), counts, sh.reshape(S * P, 4))
    owner = np.repeat(np.arange(S, dtype=np.int32), P)
    return iset.patch_loglik_multi(sset, owner, boxes, patches, isolated=isolated).reshape(S, P)


class Source(object):
    """Holds one source's parameters and its sample images; scores parameter proposals."""

    def __init__(self, params, model=None):
        self.params = params
        self.model = model
        self.sample_image_list = []      # (samp_img, fits_img, pixel_grid) like the reference

    def clear_sample_images(self):
        self.sample_image_list = []

    def is_star(self):
        return self.params.a == 0

    def is_galaxy(self):
        return self.params.a == 1

    @staticmethod
    def get_bounding_box(params, img):
        """(xlim, ylim), float limits  -- sources.py:83-96"""
        if params.is_star():
            bound = img.R
        elif params.is_galaxy():
            bound = gal_funs.gen_galaxy_psf_image_bound(params, img)
        else:
            raise ValueError("source type unknown")
        px, py = img.equa2pixel(params.u)
        xlim = (np.max([0, np.floor(px - bound)]), np.min([img.nelec.shape[1], np.ceil(px + bound)]))
        ylim = (np.max([0, np.floor(py - bound)]), np.min([img.nelec.shape[0], np.ceil(py + bound)]))
        return xlim, ylim

    def flux_in_image(self, fits_image, fluxes=None):
        """nanomaggies -> photon counts in this image  -- sources.py:120-129"""
        if fluxes is not None:
            f = fluxes[BANDS.index(fits_image.band)]
        else:
            f = self.params.flux_dict[fits_image.band]
        return (f / fits_image.calib) * fits_image.kappa

    def compute_scatter_on_pixels(self, fits_image, u=None, shape=None, xlim=None, ylim=None,
                                  pixel_grid=None, force_type=None):
        """unit-flux photon scatter image of this source  -- sources.py:351-388"""
        u = self.params.u if u is None else u
        render_star = self.is_star() if force_type is None else (force_type == 'star')
        render_gal = self.is_galaxy() if force_type is None else (force_type == 'galaxy')
        if render_star:
            return _celeste.gen_point_source_psf_image(u, fits_image, xlim=xlim, ylim=ylim, pixel_grid=pixel_grid)
        elif render_gal:
            if shape is None:
                shape = self.params.shape
            return gal_funs.gen_galaxy_psf_image(shape, u, fits_image, xlim=xlim, ylim=ylim,
                                                 check_overlap=True, unconstrained=False, return_patch=True)
        raise NotImplementedError("only stars and galaxies have photon scattering images")

    def compute_model_patch(self, fits_image, u=None, xlim=None, ylim=None):
        """counts-scaled patch  -- sources.py:390-395"""
        patch, ylim, xlim = self.compute_scatter_on_pixels(fits_image, u=u, xlim=xlim, ylim=ylim)
        band_flux = (self.params.flux_dict[fits_image.band] / fits_image.calib) * fits_image.kappa
        return band_flux * patch, ylim, xlim

    # ---- likelihoods ---------------------------------------------------------------------------
    def log_likelihood_batch(self, us=None, fluxes=None, shapes=None, isolated=False):
        """ll of P proposals at once.  Each of us (P,2), fluxes (P,5), shapes (P,4) may be None
        (= the source's current value for every proposal).  -> ndarray (P,)"""
        P = max([len(v) for v in (us, fluxes, shapes) if v is not None] + [1])
        cur_flux = np.array([self.params.flux_dict[b] for b in BANDS], dtype=np.float64)
        us = np.tile(np.asarray(self.params.u, dtype=np.float64), (P, 1)) if us is None else np.asarray(us, float)
        fluxes = np.tile(cur_flux, (P, 1)) if fluxes is None else np.asarray(fluxes, dtype=np.float64)
        if self.is_galaxy():
            shapes = np.tile(np.asarray(self.params.shape, dtype=np.float64), (P, 1)) if shapes is None \
                else np.asarray(shapes, dtype=np.float64)
        else:
            shapes = np.zeros((P, 4))
        assert np.all(~np.isnan(fluxes)), 'passing in NAN fluxes.'
        if not self.sample_image_list:
            return np.zeros(P)
        imgs = tuple(fi for (_, fi, _) in self.sample_image_list)
        ll = np.zeros(P)
        typ = np.full(P, 1 if self.is_galaxy() else 0, dtype=np.int32)
        i = 0
        while i < len(imgs):       # consecutive same-shape images share a device image set
            j = i + 1
            while j < len(imgs) and j - i < 16 and imgs[j].nelec.shape == imgs[i].nelec.shape:
                j += 1
            group = imgs[i:j]
            # a resident set that already holds these images (the field's) is reused; its other
            # bands get an empty box = "no sample image in that band"
            iset, pos = _celeste._image_subset(group)
            counts = np.zeros((P, iset.B))
            boxes = np.zeros((iset.B, 4), dtype=np.int32)
            patches = [None] * iset.B
            for k, (samp, im, _) in zip(pos, self.sample_image_list[i:j]):
                counts[:, k] = (fluxes[:, BANDS.index(im.band)] / im.calib) * im.kappa
                boxes[k] = [samp.y0, samp.y1, samp.x0, samp.x1]
                if isolated:
                    patches[k] = im.nelec[samp.y0:samp.y1, samp.x0:samp.x1]        # sources.py:204
                else:
                    patches[k] = np.array(samp.data)
            sset = iset._sources(typ, us, counts, shapes)
            ll += iset.patch_loglik(sset, boxes, patches, isolated=isolated)
            i = j
        return ll

    def log_likelihood(self, u=None, fluxes=None, shape=None):
        """conditional likelihood given the photon-sampled images  -- sources.py:134-183"""
        return float(self.log_likelihood_batch(None if u is None else [u], None if fluxes is None else [fluxes],
                                               None if shape is None else [shape])[0])

    def location_likelihood(self, u):
        return self.log_likelihood(u=u)

    def log_likelihood_isolated(self, u=None, fluxes=None, shape=None):
        """likelihood if this were the only source on its patch  -- sources.py:188-237"""
        return float(self.log_likelihood_batch(None if u is None else [u], None if fluxes is None else [fluxes],
                                               None if shape is None else [shape], isolated=True)[0])
