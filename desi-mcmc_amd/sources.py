"""Per-source conditional likelihoods: mirror of the hot methods of CelestePy/sources.py.

`Source` keeps the reference's method names and argument meaning for the calls that sit on the
render path -- compute_scatter_on_pixels / compute_model_patch (sources.py:351-395),
flux_in_image (:120-129), get_bounding_box (:83-96), log_likelihood / log_likelihood_isolated /
location_likelihood (:134-237) -- and for the per-source Gibbs updates that drive it: resample /
resample_fluxes / resample_location (:242-349), the star <-> galaxy move (:247-306: resample_type,
calculate_acceptance_logprob, image_like -- current and proposed source scored in one device call),
make_bbox_dict / get_active_sources / generate_background_patch (:434-483, array forms).  It adds
log_likelihood_batch, which scores many proposals in one launch (the reference's slice sampler
calls log_likelihood 10-50 times per source per sweep, sources.py:308-319).

These per-object methods are the small-catalogue API (one device launch per likelihood call, sample
patches on the host, as the reference holds them).  A whole catalogue is updated by
celeste_mcmc.ModelGibbs -- the same steps for every source at once, patches resident on the device
-- which CelesteBase.resample_model uses.

Reference slips on this stretch, handled as follows:
  * resample_location calls slicesample without importing it (NameError) and passes `step=`, which
    slicesample does not read: sigma stays 1.0 (degrees) and the bounds are never applied.  The default
    here is the interval the call intends (du / 5 = 0.001 deg); `sigma=1.0` runs it as executed;
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
    assert data.shape == model_img.shape == mask.shape
    keep = np.logical_and(model_img > 0., mask != 0)
    m = model_img[keep]
    return np.sum(np.log(m) * data[keep]) - np.sum(m)


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
        if not (params.is_star() or params.is_galaxy()):
            raise ValueError("source type unknown")
        reach = img.R if params.is_star() else gal_funs.gen_galaxy_psf_image_bound(params, img)
        centre = np.asarray(img.equa2pixel(params.u), dtype=np.float64)          # x, y
        lo = np.maximum(np.floor(centre - reach), 0.)
        hi = np.minimum(np.ceil(centre + reach), np.array(img.nelec.shape[::-1], dtype=np.float64))
        return (lo[0], hi[0]), (lo[1], hi[1])

    def flux_in_image(self, fits_image, fluxes=None):
        """nanomaggies -> photon counts in this image  -- sources.py:120-129"""
        band = fits_image.band
        nmgy = self.params.flux_dict[band] if fluxes is None else fluxes[BANDS.index(band)]
        return (nmgy / fits_image.calib) * fits_image.kappa

    def compute_scatter_on_pixels(self, fits_image, u=None, shape=None, xlim=None, ylim=None,
                                  pixel_grid=None, force_type=None):
        """unit-flux photon scatter image of this source  -- sources.py:351-388"""
        kind = force_type or self.object_type
        where = self.params.u if u is None else u
        if kind == 'star':
            return _celeste.gen_point_source_psf_image(where, fits_image, xlim=xlim, ylim=ylim, pixel_grid=pixel_grid)
        if kind == 'galaxy':
            return gal_funs.gen_galaxy_psf_image(self.params.shape if shape is None else shape, where, fits_image,
                                                 xlim=xlim, ylim=ylim, check_overlap=True, unconstrained=False,
                                                 return_patch=True)
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
        kappa / calib)) per band  -- sources.py:321-349 (a_0 = 5, b_0 = .005)"""
        rng = np.random if rng is None else rng
        photons, rate = np.full(5, 5.), np.full(5, .005)
        for samp, im, _ in self.sample_image_list:
            k = BANDS.index(im.band)
            photons[k] += np.sum(samp.data)
            stamp = self.compute_scatter_on_pixels(im)[0]                     # see the module docstring (Q14)
            if stamp is not None:
                rate[k] += np.sum(stamp) * im.kappa / im.calib
        self.params.fluxes = rng.gamma(photons, 1. / rate)

    def resample_location(self, u=None, rng=None, **slice_args):
        """conditionally resample the location by slice sampling  -- sources.py:308-319"""
        if u is None:
            u = np.array(self.params.u, dtype=np.float64, copy=True)
        # `step=self.du/5` (0.001 deg) of the reference's call is not an argument its slicesample reads, so what
        # the reference runs is sigma = 1.0 deg (Q13).  As in ModelGibbs the default here is the call's intent;
        # pass sigma=1.0 for the literal behaviour.
        kw = dict(step_out=False, sigma=float(np.mean(self.du)) / 5, upper_bound=self.u_upper, lower_bound=self.u_lower)
        kw.update(slice_args)
        if "seed" not in kw:
            kw["seed"] = int((np.random if rng is None else rng).randint(0, 2 ** 31 - 1))
        u, ll = slicesample(u, lambda uu: self.location_likelihood(uu), **kw)
        self.params.u = u
        return u

    def resample_shape(self, rng=None, logprior=None, phi_period=180., **slice_args):
        """shape / extent of a galaxy.  A TODO in the reference's Source (sources.py:321-325); the step run here is
        its older sampler's slice_sample_skew (celeste_mcmc.py:209-243): one slicesample update of (theta, sigma,
        phi, rho) along random directions with stepping out by doubling, on log-prior + conditional likelihood,
        then phi wrapped into [0, phi_period).  Stars return at once, as in the reference."""
        if self.is_star():
            return
        if logprior is None:
            logprior = lambda th: gal_funs.galaxy_shape_prior_constrained(th[0], th[1], th[2], th[3], phi_period)   # noqa: E731

        def skew_likelihood(th):
            lp = logprior(th)
            return lp + self.log_likelihood(shape=th) if np.isfinite(lp) else -np.inf
        kw = dict(step_out=True, doubling_step=True, compwise=False, numdir=4)
        kw.update(slice_args)
        if "seed" not in kw:
            kw["seed"] = int((np.random if rng is None else rng).randint(0, 2 ** 31 - 1))
        th, _ = slicesample(np.array(self.params.shape, dtype=np.float64), skew_likelihood, **kw)
        th[2] = (th[2] + phi_period) % phi_period
        self.params.shape = th
        return th

    # ---- star <-> galaxy move (sources.py:247-306) ------------------------------------------------
    # Re-designed for the device: the reference renders, per image, the current and the proposed source on
    # the bounding box and sums a masked Poisson term in numpy (the closure of :277-291, twice per image).
    # Here every candidate (current, proposed, or any number of proposals of either type) is scored on
    # every image of a same-shape group by ONE call of cel_patch_loglik in mode 4: the observed box and
    # the stored background cross the ABI as two planes per band, masked pixels marked by a NaN count.
    def image_like_batch(self, candidates, images):
        """sum over `images` of poisson_loglike(observed box, background + candidate's model patch, invvar mask)
        for each SrcParams of `candidates` (stars and galaxies may be mixed)  -> (P,)"""
        images = list(images)
        P = len(candidates)
        total = np.zeros(P)
        by_shape = {}
        for im in images:
            by_shape.setdefault(im.nelec.shape, []).append(im)
        for group in by_shape.values():
            for lo in range(0, len(group), 16):
                part = tuple(group[lo:lo + 16])
                iset, pos = _celeste._image_subset(part)
                boxes = np.zeros((iset.B, 4), dtype=np.int32)
                planes = [None] * iset.B
                for k, im in zip(pos, part):
                    (x0, x1), (y0, y1) = self.bounding_boxes[im]
                    y0, y1, x0, x1 = int(y0), int(y1), int(x0), int(x1)
                    boxes[k] = (y0, y1, x0, x1)
                    obs = np.array(im.nelec[y0:y1, x0:x1], dtype=np.float64)
                    invvar = getattr(im, "invvar", None)
                    if invvar is not None:
                        obs[invvar[y0:y1, x0:x1] == 0] = np.nan        # the mask is its own signal: a negative count is data (sources.py:9)
                    planes[k] = np.stack([obs, np.asarray(self.background_image_dict[im], dtype=np.float64)])
                flux_counts = lambda q, im: (q.flux_dict[im.band] / im.calib) * im.kappa      # noqa: E731  (sources.py:393-394)
                typ, radec, counts_part, shape = _celeste._source_arrays(list(candidates), part, counts_fn=flux_counts)
                counts = np.zeros((P, iset.B))
                counts[:, pos] = counts_part
                total += iset.patch_loglik_planes(iset._sources(typ, radec, counts, shape), boxes, planes)
        return total

    def image_like(self, src, img):
        """the closure of calculate_acceptance_logprob (sources.py:277-291) for one source on one image"""
        return float(self.image_like_batch([src.params], [img])[0])

    def calculate_acceptance_logprob(self, proposal, logprob_proposal, logprob_reverse, logdet, images):
        """log acceptance ratio of the type move (sources.py:275-306): current and proposed source scored in
        one device call; the prior terms are the model's (priors are outside this path)"""
        like_cur, like_prop = self.image_like_batch([self.params, proposal], images)
        prior_cur, prior_prop = self.model.logprior(self.params), self.model.logprior(proposal)
        return (like_prop + prior_prop) - (like_cur + prior_cur) + (logprob_reverse - logprob_proposal) + logdet

    def propose_other_type_prior(self):
        """a draw from the prior of the OTHER type at the current location (sources.py:263-273)
        -> (params, log q(proposal), log q(reverse), log |det|)"""
        other = 'galaxy' if self.is_star() else 'star'
        drawn, logq = self.model.prior_sample(other, u=self.params.u)
        return drawn, logq, self.model.logprior(self.params), 0.

    def resample_type(self, proposal_fun=None, rng=None):
        """Metropolis-Hastings flip star <-> galaxy (sources.py:247-261).  Needs model.prior_sample /
        model.logprior (or a proposal_fun), self.bounding_boxes and self.background_image_dict.
        -> True when the proposal was accepted"""
        draw = (proposal_fun or self.propose_other_type_prior)()
        field_imgs = [self.model.field_list[0].img_dict[b] for b in self.model.bands]
        log_alpha = self.calculate_acceptance_logprob(*draw, images=field_imgs)
        accepted = bool(np.log((np.random if rng is None else rng).rand()) < log_alpha)
        if accepted:
            self.params = draw[0]
        return accepted


# ---- source utility functions (sources.py:430-483) --------------------------------------------------
def make_bbox_dict(params, images, pixel_radius=None):
    """{img: (xlim, ylim)}: the square of half-width pixel_radius about the source, cut to each image
    (sources.py:434-456; like the reference, a radius must be given)"""
    if pixel_radius is None:
        raise NotImplementedError
    images = list(images)
    centre = np.array([im.equa2pixel(params.u) for im in images], dtype=np.float64).reshape(len(images), 2)   # x, y
    size = np.array([im.nelec.shape[::-1] for im in images], dtype=np.int64).reshape(len(images), 2)         # W, H
    lo = np.maximum(np.floor(centre - pixel_radius).astype(np.int64), 0)
    hi = np.minimum(np.ceil(centre + pixel_radius).astype(np.int64), size)
    return {im: ((int(lo[i, 0]), int(hi[i, 0])), (int(lo[i, 1]), int(hi[i, 1]))) for i, im in enumerate(images)}


def get_active_sources(source, source_list, image):
    """the other sources whose bounding box meets `source`'s in `image` under the reference's test
    (sources.py:458-474: twice the distance of the boxes' lower corners against the summed extents, per
    axis) -- one array comparison over the whole list"""
    others = [s for s in source_list if s is not source]
    if not others:
        return []
    lim = np.array([s.bounding_boxes[image] for s in [source] + others], dtype=np.float64)      # (n, axis, lo/hi)
    corner, extent = lim[:, :, 0], lim[:, :, 1] - lim[:, :, 0]
    meets = np.all(2.0 * np.abs(corner[1:] - corner[0]) < extent[1:] + extent[0], axis=1)
    return [s for s, m in zip(others, meets) if m]


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
