"""Per-source conditional likelihoods: mirror of the hot methods of CelestePy/sources.py.

`Source` keeps the reference's method names and argument meaning for the calls that sit on the
render path -- compute_scatter_on_pixels / compute_model_patch (sources.py:351-395),
flux_in_image (:120-129), get_bounding_box (:83-96), log_likelihood / log_likelihood_isolated /
location_likelihood (:134-237) -- and for the per-source Gibbs updates that drive it: resample /
resample_fluxes / resample_location (:242-349), the star <-> galaxy move's image_like
(:275-306), make_bbox_dict / get_active_sources / generate_background_patch (:434-483).  It adds
log_likelihood_batch, which scores many proposals in one launch (the reference's slice sampler
calls log_likelihood 10-50 times per source per sweep, sources.py:308-319).

These per-object methods are the small-catalogue API (one device launch per likelihood call, sample
patches on the host, as the reference holds them).  A whole catalogue is updated by
celeste_mcmc.ModelGibbs -- the same steps for every source at once, patches resident on the device
-- which CelesteBase.resample_model uses.

Reference slips on this stretch, handled as follows:
  * resample_location calls slicesample without importing it (NameError) and passes `step=`, which
    slicesample does not read: sigma stays 1.0 (degrees) and the bounds are never applied.  The call
    is reproduced as written (`sigma=` may be given to override);
  * resample_fluxes passes the cached pixel grid as `u` (sources.py:338); the intended call -- the
    unit stamp at the current location on the source's own box -- is what is summed here.
"""
import numpy as np

from . import celeste as _celeste
from . import celeste_galaxy_conditionals as gal_funs
from .util.infer.slicesample import slicesample

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
        # a ~20 x 20 pixel box about the first guess (sources.py:23-27)
        self.u_lower = np.asarray(self.params.u, dtype=np.float64) - .0025
        self.u_upper = np.asarray(self.params.u, dtype=np.float64) + .0025
        self.du = self.u_upper - self.u_lower
        self.loc_samps, self.flux_samps, self.shape_samps, self.ll_samps = [], [], [], []

    def clear_sample_images(self):
        self.sample_image_list = []

    @property
    def object_type(self):
        return "star" if self.is_star() else ("galaxy" if self.is_galaxy() else "none")

    # ---- kept samples (sources.py:98-118) --------------------------------------------------------
    @property
    def location_samples(self):
        return np.array(self.loc_samps)

    @property
    def flux_samples(self):
        return np.array(self.flux_samps)

    @property
    def shape_samples(self):
        return np.array(self.shape_samps)

    @property
    def loglike_samples(self):
        return np.array(self.ll_samps)

    def store_sample(self):
        self.loc_samps.append(np.array(self.params.u, copy=True))
        self.flux_samps.append(np.array([self.params.flux_dict[b] for b in BANDS]))
        self.shape_samps.append(np.array(self.params.shape, copy=True))

    def store_loglike(self):
        self.ll_samps.append(self.log_likelihood())

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

    # ---- resampling (sources.py:242-349) ---------------------------------------------------------
    def resample(self, rng=None):
        assert len(self.sample_image_list) != 0, "resample source needs sampled images"
        self.resample_fluxes(rng=rng)
        self.resample_location(rng=rng)

    def resample_fluxes(self, rng=None):
        """fluxes u,g,r,i,z given everything else: Gamma(a_0 + photons, 1 / (b_0 + sum(unit stamp) *
        kappa / calib)) per band  -- sources.py:321-349"""
        rng = np.random if rng is None else rng
        a_0, b_0 = 5., .005
        band_counts = {b: 0 for b in BANDS}
        psf_sums = {b: 0 for b in BANDS}
        for src_img, fits_img, pixel_grid in self.sample_image_list:
            band_counts[fits_img.band] += np.sum(np.array(src_img.data))
            psf_ns, ylim, xlim = self.compute_scatter_on_pixels(fits_img)     # see the module docstring
            if psf_ns is not None:
                psf_sums[fits_img.band] += np.sum(psf_ns) * fits_img.kappa / fits_img.calib
        a_n = a_0 + np.array([band_counts[b] for b in BANDS])
        b_n = b_0 + np.array([psf_sums[b] for b in BANDS])
        self.params.fluxes = rng.gamma(a_n, 1. / b_n)

    def resample_location(self, u=None, rng=None, **slice_args):
        """conditionally resample the location by slice sampling  -- sources.py:308-319"""
        if u is None:
            u = np.array(self.params.u, dtype=np.float64, copy=True)
        kw = dict(step_out=False, upper_bound=self.u_upper, lower_bound=self.u_lower)
        kw.update(slice_args)                 # `step=self.du/5` of the reference is not a slicesample argument
        if "seed" not in kw:
            kw["seed"] = int((np.random if rng is None else rng).randint(0, 2 ** 31 - 1))
        u, ll = slicesample(u, lambda uu: self.location_likelihood(uu), **kw)
        self.params.u = u
        return u

    def resample_shape(self):
        """shape/extent of a galaxy: not implemented in the reference either (sources.py:321-325)"""
        return

    # ---- star <-> galaxy move (sources.py:247-306) ------------------------------------------------
    def image_like(self, src, img):
        """Poisson log-likelihood of `img` on this source's bounding box with `src` rendered on the
        stored background  -- the closure of calculate_acceptance_logprob, sources.py:277-291"""
        xlim, ylim = self.bounding_boxes[img]
        background_img = self.background_image_dict[img]
        data_img = img.nelec[ylim[0]:ylim[1], xlim[0]:xlim[1]]
        invvar = getattr(img, "invvar", None)
        mask_img = np.ones(data_img.shape) if invvar is None else invvar[ylim[0]:ylim[1], xlim[0]:xlim[1]]
        model_img, _, _ = src.compute_model_patch(img, xlim=xlim, ylim=ylim)
        return poisson_loglike(data=data_img, model_img=background_img + model_img, mask=mask_img)

    def calculate_acceptance_logprob(self, proposal, logprob_proposal, logprob_reverse, logdet, images):
        """sources.py:275-306; the priors (model.logprior) are the caller's"""
        curr_like = np.sum([self.image_like(self, img) for img in images])
        curr_logprior = self.model.logprior(self.params)
        proposal_source = self.model._source_type(proposal, self.model)
        prop_like = np.sum([self.image_like(proposal_source, img) for img in images])
        prop_logprior = self.model.logprior(proposal_source.params)
        return (prop_like + prop_logprior) - (curr_like + curr_logprior) + \
               (logprob_reverse - logprob_proposal) + logdet

    def resample_type(self, proposal_fun=None, rng=None):
        """star vs galaxy by a reversible jump  -- sources.py:247-261.  Needs model.prior_sample /
        model.logprior (priors are outside this path) and bounding_boxes / background_image_dict."""
        rng = np.random if rng is None else rng
        proposal_fun = self.propose_other_type_prior if proposal_fun is None else proposal_fun
        proposal, logpdf, logreverse, logdet = proposal_fun()
        fimgs = [self.model.field_list[0].img_dict[b] for b in self.model.bands]
        accept_logprob = self.calculate_acceptance_logprob(proposal, logpdf, logreverse, logdet, fimgs)
        if np.log(rng.rand()) < accept_logprob:
            self.params = proposal

    def propose_other_type_prior(self):
        """prior-based proposal  -- sources.py:263-273"""
        if self.is_star():
            params, logprob = self.model.prior_sample('galaxy', u=self.params.u)
        elif self.is_galaxy():
            params, logprob = self.model.prior_sample('star', u=self.params.u)
        logreverse = self.model.logprior(self.params)
        return params, logprob, logreverse, 0.


# ---- source utility functions (sources.py:430-483) --------------------------------------------------
def make_bbox_dict(params, images, pixel_radius=None):
    """{img: (xlim, ylim)}: the area a source's model affects  -- sources.py:434-456"""
    if pixel_radius is None:
        raise NotImplementedError

    def image_bbox(params, img):
        img_ymax, img_xmax = img.nelec.shape
        px, py = img.equa2pixel(params.u)
        xlim = (np.max([0, int(np.floor(px - pixel_radius))]), np.min([img_xmax, int(np.ceil(px + pixel_radius))]))
        ylim = (np.max([0, int(np.floor(py - pixel_radius))]), np.min([img_ymax, int(np.ceil(py + pixel_radius))]))
        return xlim, ylim
    return {img: image_bbox(params, img) for img in images}


def get_active_sources(source, source_list, image):
    """sources whose bounding box intersects `source`'s in `image`  -- sources.py:458-474"""
    def intersect(sa, sb, image):
        xlima, ylima = sa.bounding_boxes[image]
        xlimb, ylimb = sb.bounding_boxes[image]
        widtha, heighta = xlima[1] - xlima[0], ylima[1] - ylima[0]
        widthb, heightb = xlimb[1] - xlimb[0], ylimb[1] - ylimb[0]
        return (np.abs(xlima[0] - xlimb[0]) * 2 < (widtha + widthb)) and \
               (np.abs(ylima[0] - ylimb[0]) * 2 < (heighta + heightb))
    return [s for s in source_list if intersect(s, source, image) and s is not source]


def generate_background_patch(source, source_list, image):
    """epsilon + every active source's model patch on `source`'s bounding box  -- sources.py:476-483.
    All active sources are rendered on the box in ONE device call."""
    active_sources = get_active_sources(source, source_list, image)
    xlim, ylim = source.bounding_boxes[image]
    y0, y1, x0, x1 = int(ylim[0]), int(ylim[1]), int(xlim[0]), int(xlim[1])
    background = np.zeros((y1 - y0, x1 - x0)) + image.epsilon
    if active_sources:
        iset = _celeste._image_set((image,))
        counts_fn = lambda p, im: (p.flux_dict[im.band] / im.calib) * im.kappa      # noqa: E731  (sources.py:393-394)
        typ, radec, counts, shape = _celeste._source_arrays([s.params for s in active_sources], (image,), counts_fn=counts_fn)
        sset = iset._sources(typ, radec, counts, shape)
        boxes = np.tile(np.array([[y0, y1, x0, x1]], dtype=np.int32), (len(active_sources), 1))
        patches, _ = iset.stamps(sset, 0, scaled=True, boxes_in=boxes)
        for p in patches:
            if p is not None:
                background += p
    return background
