"""Catalogue-level model classes: mirror of the render / likelihood surface of CelestePy/models.py.

`CelesteBase` keeps the reference's field list, source list, the two methods that sit on the
render path -- render_model_image and img_log_likelihood (models.py:88-108) -- on top of the
device-resident image sets, Field.resample_photons (models.py:123-160) on top of the device
photon split, and the Gibbs sweep resample_model / resample_sources (models.py:75-83).

resample_model runs the sweep for the whole catalogue on the device (celeste_mcmc.ModelGibbs: the
photon split of every field, then every source's flux and location update in lock-step) and
writes the new parameters back into the Source objects.  `resample_sources` alone, after
Field.resample_photons has filled the sources' sample_image_list with host-side patches, is the
reference's per-object loop (one Source.resample per source).

Quirk Q4: the reference's render_model_image re-uses its `xlim` / `ylim` loop variables,
so every source after the first is rendered onto the FIRST source's box and the result is
cropped to it (models.py:95-100).  The intended semantics -- every source on its own box -- are
implemented; with one source, or with caller-imposed limits, both agree.
"""
import numpy as np

from . import celeste as _celeste
from .sources import Source

BANDS = ['u', 'g', 'r', 'i', 'z']


def _flux_counts(src_params, image):
    """flux_dict convention of the newer callers: (flux / calib) * kappa  (sources.py:390-395)"""
    return (src_params.flux_dict[image.band] / image.calib) * image.kappa


class Field(object):
    """image data of a single field, keyed by band  -- models.py:110-121"""

    def __init__(self, img_dict):
        self.img_dict = img_dict
        for k, img in self.img_dict.items():
            img.epsilon = np.median(img.nelec)      # models.py:115-117: noise level := median
        self.a_0 = 5
        self.b_0 = .005

    def resample_photons(self, srcs, verbose=False, seed=None, rng=None):
        """resample photons, store source-specific sample images, resample each image's noise level
        -- models.py:123-160.  One device pass splits the photons of all the field's bands."""
        from . import celeste_mcmc as cel_mcmc
        rng = np.random if rng is None else rng
        for src in srcs:
            src.clear_sample_images()
        bands = list(self.img_dict.keys())
        imgs = [self.img_dict[b] for b in bands]
        same = all(im.nelec.shape == imgs[0].nelec.shape for im in imgs)
        groups = [imgs] if same and len(imgs) <= 16 else [[im] for im in imgs]
        noise_sums = {}
        k = 0
        for group in groups:
            samp, noise = cel_mcmc.sample_source_photons_multi_image(
                group, [s.params for s in srcs], seed=None if seed is None else seed + k)
            for n, img in enumerate(group):
                for src, samp_img in zip(srcs, samp[n]):
                    if samp_img is not None:
                        # the reference caches a pixel grid per sample image (models.py:146-150);
                        # the device evaluators need none
                        src.sample_image_list.append((samp_img, img, None))
                noise_sums[bands[k]] = noise[n]
                k += 1
        for band, img in self.img_dict.items():
            a_n = self.a_0 + noise_sums[band]
            b_n = self.b_0 + img.nelec.size
            img.epsilon = rng.gamma(a_n, 1. / b_n)           # models.py:156-160
        return noise_sums


class CelesteBase(object):
    """Main model class: a list of Source objects and a list of fields  -- models.py:15-108"""
    _source_type = Source

    def __init__(self, gal_flux_prior_distn=None, star_flux_prior_distn=None):
        self.field_list = []
        self.bands = list(BANDS)
        self.srcs = []
        self.star_flux_prior_distn = star_flux_prior_distn
        self.gal_flux_prior_distn = gal_flux_prior_distn

    def initialize_sources(self, init_srcs=None, init_src_params=None, photoobj_df=None, catalogue=False):
        """models.py:62-73.  catalogue=True (an addition): the parameters are packed into ONE SrcCatalog and every
        Source gets a view of its row (same attribute names, reads and writes go to the arrays), so that
        `[s.params for s in model.srcs]` reaches the device without a per-object gather -- for sources given by
        fluxes (a view has no black-body t / b)."""
        if init_srcs is not None:
            self.srcs = init_srcs
        elif init_src_params is not None:
            if catalogue:
                from .celeste_src import SrcCatalog
                init_src_params = SrcCatalog.from_params(init_src_params).views()
            self.srcs = [self._source_type(s, self) for s in init_src_params]
        else:
            raise NotImplementedError("photoObj tables need the reference's data-acquisition layer")

    def add_field(self, img_dict):
        for k in img_dict.keys():
            assert k in self.bands, "Celeste model doesn't support band %s" % k
        self.field_list.append(Field(img_dict))

    @property
    def source_types(self):
        return np.array([{0: "star", 1: "galaxy"}.get(s.params.a, "none") for s in self.srcs])

    def get_brightest(self, object_type='star', num_srcs=1, band='r', return_idx=False):
        fluxes = np.array([s.params.flux_dict[band] for s in self.srcs])
        type_idx = np.where(self.source_types == object_type)[0]
        type_idx = type_idx[np.argsort(fluxes[type_idx])[::-1]][:num_srcs]
        blist = [self.srcs[i] for i in type_idx]
        return (blist, type_idx) if return_idx else blist

    # ---- resample methods: models.py:75-83 ---------------------------------------------------------
    def gibbs(self, seed=None, slice_args=None, rebuild=False):
        """the catalogue-wide device sampler over this model's fields and sources (built once; the
        Source objects are re-read when `rebuild` is set or their number changed)"""
        from . import celeste_mcmc as cel_mcmc
        g = getattr(self, "_gibbs", None)
        if g is None or rebuild or g.S != len(self.srcs):
            if seed is None:
                seed = int(np.random.randint(0, 2 ** 31 - 1))
            g = cel_mcmc.ModelGibbs.from_images([f.img_dict for f in self.field_list], [s.params for s in self.srcs],
                                                seed=seed, slice_args=slice_args)
            for gf, field in zip(g.fields, self.field_list):
                gf.a_0, gf.b_0 = field.a_0, field.b_0
            self._gibbs = g
        return g

    def sync_sources(self):
        """write the sampler's state (locations, fluxes) back into the Source objects"""
        g = self._gibbs
        for i, s in enumerate(self.srcs):
            s.params.u = g.u[i].copy()
            s.params.fluxes = g.fluxes[i].copy()

    def resample_model(self, n_sweeps=1, seed=None, slice_args=None, sync=True):
        """resample each field's photons, then every source  -- models.py:75-79.  n_sweeps > 1 runs
        that many sweeps back to back on the device before the Source objects are updated."""
        if not self.srcs or not self.field_list:
            return
        g = self.gibbs(seed=seed, slice_args=slice_args)
        if getattr(self, "_gibbs_dirty", True):
            # the Source objects may have been edited since the last sweep
            fresh = type(g).from_images([f.img_dict for f in self.field_list], [s.params for s in self.srcs], seed=g.seed)
            g.typ, g.u, g.fluxes, g.shape = fresh.typ, fresh.u, fresh.fluxes, fresh.shape
        for _ in range(int(n_sweeps)):
            g.sweep()
        self._gibbs_dirty = sync
        if sync:
            self.sync_sources()

    def resample_sources(self, rng=None):
        """one Source.resample per source, on the sample images Field.resample_photons stored
        -- models.py:81-83"""
        for src in self.srcs:
            src.resample(rng=rng)
        self._gibbs_dirty = True

    # ---- the render path -----------------------------------------------------------------------
    def render_model_image(self, fimg, xlim=None, ylim=None, exclude=None):
        """epsilon + every source's model patch  -- models.py:88-102 (intended semantics, Q4)"""
        source_list = [s for s in self.srcs if s is not exclude]
        params = [s.params for s in source_list]
        if xlim is None and ylim is None:
            iset = _celeste._image_set((fimg,))
            typ, radec, counts, shape = _celeste._source_arrays(params, (fimg,), counts_fn=_flux_counts)
            iset.render(iset._sources(typ, radec, counts, shape), loglik=False)
            return iset.model_images()[0]
        # caller-imposed limits: every source is evaluated on that box (models.py:96 passes the
        # limits to compute_model_patch) and the image is cropped to it (:99-100)
        y0, y1, x0, x1 = int(ylim[0]), int(ylim[1]), int(xlim[0]), int(xlim[1])
        mod = np.ones((y1 - y0, x1 - x0)) * fimg.epsilon
        if params:
            iset = _celeste._image_set((fimg,))
            typ, radec, counts, shape = _celeste._source_arrays(params, (fimg,), counts_fn=_flux_counts)
            sset = iset._sources(typ, radec, counts, shape)
            boxes = np.tile(np.array([[y0, y1, x0, x1]], dtype=np.int32), (len(params), 1))
            patches, _ = iset.stamps(sset, 0, scaled=True, boxes_in=boxes)
            for p in patches:
                if p is not None:
                    mod += p
        return mod

    def img_log_likelihood(self, fimg, mod_img=None):
        """sum log(m) nelec - sum m  -- models.py:104-108; fused on the device when mod_img is None"""
        if mod_img is None:
            iset = _celeste._image_set((fimg,))
            typ, radec, counts, shape = _celeste._source_arrays([s.params for s in self.srcs], (fimg,),
                                                                counts_fn=_flux_counts)
            total, _ = iset.render(iset._sources(typ, radec, counts, shape), loglik=True)
            return total
        return np.sum(np.log(mod_img) * fimg.nelec) - np.sum(mod_img)

    def log_likelihood(self):
        """sum of img_log_likelihood over every image of every field (one device pass per field)"""
        ll = 0.0
        params = [s.params for s in self.srcs]
        for field in self.field_list:
            imgs = [field.img_dict[b] for b in self.bands if b in field.img_dict]
            ll += _celeste.celeste_likelihood_multi_image(_FluxDictView.wrap(params), imgs)
        return ll


class _FluxDictView(object):
    """Presents SrcParams with array-style fluxes to celeste.expected_photons, which indexes
    `fluxes[band]` like the reference's gen_src_image (celeste.py:41,50)."""
    __slots__ = ("_p", "fluxes")

    def __init__(self, p):
        self._p = p
        self.fluxes = p.flux_dict

    def __getattr__(self, name):
        return getattr(self._p, name)

    @staticmethod
    def wrap(params):
        return [_FluxDictView(p) for p in params]


class Celeste(CelesteBase):
    _source_type = Source
