"""Mirror of the runnable Gibbs step of CelestePy/celeste_mcmc.py: the photon split.

sample_source_photons_single_image_cython (celeste_mcmc.py:98-150) renders every source's
counts-scaled patch (gen_src_image_with_fluxes) and calls the Cython multinomial split
(celeste_sample_sources.pyx:61-156).  Here both happen in one device pass (cel_photon_split);
the (samp_imgs, noise_sum) return shape is kept.  The rest of celeste_mcmc.py is not runnable in
the reference as written and is not mirrored.

Random numbers: the reference draws from randomkit's MT19937 through numpy's RandomState; the
device uses a counter-based Philox generator.  Results match in distribution, not draw by draw.
"""
import numpy as np

from . import celeste as _celeste
from .sources import SamplePatch


_M64 = (1 << 64) - 1
# one tag per randomised step of a sweep: no two steps share a stream key for any seed (with seed = 0 the plain
# `seed * prime + sweep` forms all collapsed to `sweep`, and a source's location, shape and flux draws were one sequence)
STEP_TAGS = dict(split=0x53504C4954000001, flux=0x464C555800000002, location=0x4C4F434154000003, shape=0x5348415045000004)


def _mix64(z):
    """SplitMix64's finaliser on a Python int"""
    z = (z + 0x9E3779B97F4A7C15) & _M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


def step_seed(seed, step, sweep, k=0):
    """the stream seed of one step (`split`, `flux`, `location`, `shape`) of sweep number `sweep` (of field k): a hash of
    all of them, so that the Gibbs blocks draw from unrelated streams whatever the chain's seed is"""
    return _mix64((_mix64(_mix64(int(seed) & _M64) ^ STEP_TAGS[step]) + int(sweep)) & _M64 ^ (int(k) * 0xD6E8FEB86659FD93 & _M64))


def _flux_counts(src, image):
    return (src.flux_dict[image.band] / image.calib) * image.kappa      # celeste.py:80-81,94


def sample_source_photons_multi_image(imgs, srcs, seed=None):
    """The split for several same-shape images in one pass.
    -> (samp_imgs[n][s] SamplePatch or None, noise_sums[n])"""
    if seed is None:
        seed = np.random.randint(0, 2 ** 31 - 1)
    imgs = tuple(imgs)
    iset = _celeste._image_set(imgs)
    typ, radec, counts, shape = _celeste._source_arrays(srcs, imgs, counts_fn=_flux_counts)
    sset = iset._sources(typ, radec, counts, shape)
    patches, boxes, noise = iset.photon_split(sset, seed)
    out = []
    for n in range(len(imgs)):
        row = []
        for s in range(len(srcs)):
            p = patches[n][s]
            if p is None:
                row.append(None)
            else:
                y0, y1, x0, x1 = boxes[n, s]
                row.append(SamplePatch(p, (y0, y1), (x0, x1)))
        out.append(row)
    return out, noise


def sample_source_photons_single_image_cython(img, srcs, seed=None):
    """Given a single photon-count image and a list of sources, sample source-specific images
    using the Poisson/multinomial representation  -- celeste_mcmc.py:98-150.
    returns (samp_imgs: list of SamplePatch (x0,x1,y0,y1,data) or None, noise_sum)"""
    samp, noise = sample_source_photons_multi_image((img,), srcs, seed)
    return samp[0], noise[0]


def gamma_by_stream(a, seed, ids):
    """Standard Gamma(a) variates, one per element of `a`, each from ITS OWN counter-based stream (seed, ids[i]) -- so that a
    source's flux draw does not depend on which other sources are drawn beside it, by which rank, or in what order (numpy's
    generator hands a sequential stream to a rejection sampler: the numbers an element gets depend on every element before
    it).  Marsaglia & Tsang (2000): x ~ N(0,1), v = (1 + c x)^3, accept when log u < x^2/2 + d - d v + d log v with
    d = a - 1/3, c = 1 / sqrt(9 d); a < 1 through Gamma(a + 1) U^(1/a).  Host-side, vectorised: a round over the elements
    still waiting for an acceptance (~5 % per round)."""
    from .util.infer.slicesample import ChainStreams
    a = np.asarray(a, dtype=np.float64)
    st = ChainStreams(seed, np.asarray(ids))
    n = a.shape[0]
    boost = a < 1.0
    aa = np.where(boost, a + 1.0, a)
    d = aa - 1.0 / 3.0
    c = 1.0 / np.sqrt(9.0 * d)
    out = np.empty(n)
    todo = slice(None)                   # the first round takes every element: views, not gathers
    left = n
    while left:
        x, u = st.normal(todo), st.uniform(todo)
        dt = d[todo]
        t = 1.0 + c[todo] * x
        v = t * t * t
        x2 = x * x
        # the squeeze u < 1 - 0.0331 x^4 implies the full test (Marsaglia & Tsang, step 3): ~92 % accept without a logarithm
        ok = (v > 0.0) & (u < 1.0 - 0.0331 * x2 * x2)
        rest = np.nonzero((v > 0.0) & ~ok)[0]
        if rest.size:
            vr = v[rest]
            ok[rest] = np.log(u[rest]) < 0.5 * x2[rest] + dt[rest] * (1.0 - vr + np.log(vr))
        if isinstance(todo, slice):
            out[ok] = dt[ok] * v[ok]
            todo = np.nonzero(~ok)[0]
        else:
            out[todo[ok]] = dt[ok] * v[ok]
            todo = todo[~ok]
        left = todo.size
    if boost.any():
        idx = np.nonzero(boost)[0]
        out[idx] *= st.uniform(idx) ** (1.0 / a[idx])
    return out


# ---- the whole Gibbs sweep, catalogue-wide and device-resident -------------------------------------
class GibbsField(object):
    """One field's images on the device plus what the sweep needs to know about them."""

    def __init__(self, iset, band_index, calib, kappa, npix, a_0=5, b_0=.005, trace_iset=None):
        self.iset = iset
        # a strip-partitioned chain (dist.StripDeal): `iset` holds this rank's window (strip + halo), `trace_iset` its strip
        # alone -- what this rank adds to the field's log-likelihood; npix stays the whole frame's
        self.trace_iset = trace_iset
        self.band_index = np.asarray(band_index, dtype=np.int64)      # which of u,g,r,i,z each image is
        self.calib = np.asarray(calib, dtype=np.float64)
        self.kappa = np.asarray(kappa, dtype=np.float64)
        self.npix = float(npix)
        self.a_0, self.b_0 = a_0, b_0                                  # Gamma prior of the sky level (models.py:119-121)
        self.epsilon = np.array(iset.eps, copy=True)
        self.sset = None
        self.prop = None
        self.sub = None             # this rank's sources of a dealt chain (resample_fluxes)
        self.has_patch = None
        self._sums = None           # photons per (source, image) of the current split: fetched when somebody reads `sums`

    @property
    def sums(self):
        """photons per (source, image) of the resident split, read back on first use (the device flux step never reads them:
        400 KB per sweep that stay where they are)"""
        if self._sums is None:
            self._sums = self.iset.sample_sums()
        return self._sums

    @sums.setter
    def sums(self, value):
        self._sums = value


def strip_gibbs_field(ctx, bands, nelec, rows, boxes, status, world, rank, band_index=None, slack=48, device=None, edges=None,
                      solo=False, window_trace=True):
    """This rank's part of ONE chain partitioned by row strips (dist.StripDeal).
        bands (B, 37) cel_band records, nelec (B, H, W) the whole frame's pixels (every rank can read them: only the window is
        uploaded), rows (S,) the sources' pixel rows, boxes (B, S, 4) / status (B, S) their boxes on the whole frame
        (ImageSet.source_boxes): the halo is as tall as this rank's sources' boxes reach beyond its strip, plus `slack` rows
        for the few pixels they move per sweep.
        window_trace (round 5): the chain's log-likelihood trace is rendered on the WINDOW image set, whose log-likelihood adds
        the strip's tile rows only (cel_images_set_noise_rows: the rows this rank owns) -- so the model image the next photon
        split needs is already on the device (its totals come from k_strict_totals instead of a render, the flux step's stamp
        masses from the split's own sums): what the single-rank chain does.  Needs strip edges and halo on 64-row tiles (the
        default layout's); otherwise, and with window_trace=False, a second image set of the strip alone renders the trace.
    -> (StripDeal, GibbsField over the window [with the strip as its trace image set])"""
    from . import dist as _dist
    from . import field as _field
    B, H, W = nelec.shape
    align = 64 if window_trace else _dist.TILE_ROWS
    if edges is not None and any(int(e) % 64 for e in list(edges)[:-1]):
        align, window_trace = _dist.TILE_ROWS, False
    if edges is None and int(world) > -(-H // 64):
        align, window_trace = _dist.TILE_ROWS, False
    probe = _dist.StripDeal(rows, H, world, rank, edges=edges, solo=solo, align=align)       # (edges: dist.strip_edges, the same on every rank)
    mine = probe.mine
    has = status[:, mine] > 0
    reach = max(int(np.max(np.where(has, probe.strip[0] - boxes[:, mine, 0], 0), initial=0)),
                int(np.max(np.where(has, boxes[:, mine, 1] - probe.strip[1], 0), initial=0)), 0)
    deal = _dist.StripDeal(rows, H, world, rank, halo=reach + slack, device=device, edges=probe.edges, solo=solo, align=align)
    w0, w1 = deal.window
    win = _field.ImageSet(ctx, bands, w1 - w0, W, nelec=np.ascontiguousarray(nelec[:, w0:w1]))
    win.set_window(w0, H)
    win.set_noise_rows(*deal.noise_rows())
    y0, y1 = deal.strip
    strip = None
    if not window_trace:
        strip = _field.ImageSet(ctx, bands, max(y1 - y0, 1), W, nelec=np.ascontiguousarray(nelec[:, y0:max(y1, y0 + 1)]))
        strip.set_window(y0, H)
    bands = np.asarray(bands)
    gf = GibbsField(win, list(range(B)) if band_index is None else band_index, bands[:, 2], bands[:, 1], H * W, trace_iset=strip)
    return deal, gf


class ModelGibbs(object):
    """CelesteBase.resample_model (CelestePy/models.py:75-83) for a whole catalogue at once:

        for every field:  Field.resample_photons (models.py:123-160) -- the photon split of all its
                          images in one device pass, sample patches kept on the device, and the
                          sky level of every image redrawn from its Gamma conditional;
        for every source: Source.resample (sources.py:242-245) = resample_fluxes (:321-349), then
                          resample_location (:308-319) by slice sampling -- all sources in
                          lock-step: round k scores the k-th slice evaluation of every unfinished
                          source in ONE launch of cel_patch_loglik_multi against the resident patches.

    State lives in arrays (type[S], u[S,2], fluxes[S,5] in nanomaggies, shape[S,4]); nothing per
    source runs in Python.  A source that has no sample patch in any image is left alone (the
    reference asserts there, sources.py:243).

    Random numbers: Philox on the device for the split (keyed by pixel and source), one SplitMix64
    stream per source for the slice sampler and per (source, band) for the flux Gamma draws, numpy's
    generator for the sky levels' -- all derived from `seed`; a chain is reproducible and does not
    depend on batching, nor on how it is dealt over ranks."""

    BANDS = ['u', 'g', 'r', 'i', 'z']

    def __init__(self, fields, typ, u, fluxes, shape, seed=0, flux_a_0=5., flux_b_0=.005, slice_args=None, engine="auto",
                 deal=None, shape_args=None, shape_logprior=None, phi_period=180., shape_mass="reference", conditional="reference"):
        self.fields = list(fields)
        self.typ = np.ascontiguousarray(typ, dtype=np.int32)
        self.S = self.typ.shape[0]
        self.u = np.array(u, dtype=np.float64).reshape(self.S, 2)
        self.fluxes = np.array(fluxes, dtype=np.float64).reshape(self.S, 5)
        self.shape = np.array(shape, dtype=np.float64).reshape(self.S, 4)
        self.seed = int(seed)
        self.rng = np.random.RandomState(self.seed & 0x7FFFFFFF)
        self.flux_a_0, self.flux_b_0 = flux_a_0, flux_b_0
        # Source.resample_location's call (sources.py:312-317): no stepping out, step = du / 5 = 0.001
        # degrees.  The reference's slicesample does not read `step=`, so what it effectively runs is an
        # interval of sigma = 1.0 degree, with which a faint source can leave the frame for good (DESIGN Q13,
        # Q15).  The default here is the call's INTENT (sigma = 0.001 deg); slice_args="literal" asks for the
        # call as the reference executes it.
        self.slice_args = self.slice_preset(slice_args)
        # the galaxies' shape step (celeste_mcmc.py:224-243, slice_sample_skew): slicesample over (theta, sigma, phi,
        # rho) along random directions with stepping out by doubling; sigma stays slicesample's default 1.0 there.
        # The log-prior added to the conditional likelihood is the caller's (priors are outside this path); the
        # default is the reference's galaxy_shape_prior_constrained with phi in the renderer's unit (degrees, Q7),
        # and phi is wrapped into [0, phi_period) after the move as :241 wraps it into [0, pi).
        self.shape_args = dict(step_out=True, doubling_step=True, compwise=False, numdir=4)
        self.shape_args.update(shape_args or {})
        self.phi_period = float(phi_period)
        # The conditional likelihood the shape step scores (Source.log_likelihood(shape=), sources.py:134-183) charges a source
        # its FULL expected photons, band_flux * sum(psf weights), whatever the shape ("model_outside ... should be small",
        # :166-170) -- but the photons it is given were split on the source's box, and how much of a proposal's stamp lies on
        # its box depends on the shape: for an extended de Vaucouleurs-dominated galaxy at high signal to noise the constant
        # term leaves sigma 10-20 % low (tests/test_calibration.py: the calibration test that found it; DESIGN Q20).
        # shape_mass="reference" keeps the reference's term; "exact" charges counts * (the proposal's unit stamp summed over its
        # own box, cel_stamp_mass) in the shape step (host engine).  That is one of three corrections: conditional="exact"
        # below makes all of them.
        if shape_mass not in ("reference", "exact"):
            raise ValueError("shape_mass must be 'reference' or 'exact'")
        self.shape_mass = shape_mass
        # conditional="exact": every block of the sweep samples the Gibbs conditional of the model the RENDERER draws from -- a
        # source's photons lie on its own box and nowhere else (celeste.py:217-219).  Three departures from the reference:
        # the photons are split on whole boxes (CEL_OPT_SPLIT_FULL_BOX; celeste_sample_sources.pyx:50-51 leaves a box's first
        # row and column out); the location AND the shape step charge counts * (the proposal's stamp mass on ITS box); and a
        # proposal whose box does not cover every photon of the source has probability zero (the reference scores the fixed
        # data patch whatever the proposal's box, sources.py:134-183).  Calibrated at 192 replicates and pi P = pi on a sigma
        # grid (tests/test_calibration.py, DESIGN Q20).  Host engine; a change of the integer box needs the ring between the two
        # boxes free of the source's photons, so big galaxies mix more slowly across box sizes than under the reference's rules.
        if conditional not in ("reference", "exact"):
            raise ValueError("conditional must be 'reference' or 'exact'")
        self.conditional = conditional
        if conditional == "exact":
            self.shape_mass = "exact"
        self._default_shape_prior = shape_logprior is None
        if shape_logprior is None:
            from .celeste_galaxy_conditionals import galaxy_shape_prior_constrained
            shape_logprior = lambda TH: galaxy_shape_prior_constrained(TH[:, 0], TH[:, 1], TH[:, 2], TH[:, 3], self.phi_period)   # noqa: E731
        self.shape_logprior = shape_logprior
        # where the slice sampler's state machine runs: "device" (cel_slice_locations: nothing but a
        # counter crosses PCIe per round; one field, the reference call's options), "host" (the numpy
        # engine of util/infer/slicesample.py: every option, any number of fields), "auto" = device
        # when it applies.  Both draw the same per-chain streams: a chain takes the same trajectory.
        if engine not in ("auto", "device", "host"):
            raise ValueError("engine must be auto, device or host")
        self.engine = engine
        self.sweeps = 0
        self.timing = dict(split=0.0, flux=0.0, location=0.0, rounds=0, evals=0, shape=0.0, shape_rounds=0, shape_evals=0)
        # ONE chain over several GPUs: `deal` (dist.SourceDeal) names the sources this rank
        # updates.  Every rank runs the same photon split (counter-based draws: the replicas are bitwise equal) and
        # the same host draws (same seed), updates the fluxes and locations of ITS sources only, and the ranks
        # exchange the new rows with one all-gather per sweep.  The chain is the single-rank chain, bit for bit.
        self.deal = deal
        if deal is not None and deal.S != self.S:
            raise ValueError("the deal is over %d sources, the catalogue has %d" % (deal.S, self.S))
        if deal is not None and getattr(deal, "kind", "") == "strips" and self.conditional == "exact":
            raise ValueError("conditional='exact' runs on whole frames (a replicated deal or one rank): its box tests are not window-relative")
        self.noise_sums = None
        self.active = np.ones(self.S, dtype=bool)
        # the flux conditionals' Gamma variates: on the device (cel_gamma_streams) or, host_gamma=True, by the numpy form of
        # the same sampler (gamma_by_stream: the same streams and decisions, values equal to rounding)
        import os
        self.host_gamma = os.environ.get("CEL_HOST_GAMMA") == "1"
        # the flux step as ONE device call (cel_flux_conditionals) where it applies; CEL_HOST_FLUX=1 / device_flux = False: the
        # host form (sums and masses read back, variates drawn on the device, fluxes formed in numpy): the same bits
        self.device_flux = os.environ.get("CEL_HOST_FLUX") != "1"

    # -- helpers ---------------------------------------------------------------------------------
    SLICE_INTENDED = dict(step_out=False, sigma=1e-3)
    SLICE_LITERAL = dict(step_out=False)                       # sigma stays slicesample's default, 1.0 degree

    @classmethod
    def slice_preset(cls, slice_args):
        """None -> the intended call; "literal" -> the reference's effective call; a dict is laid over the
        intended call's step_out=False (its sigma is then the caller's, or slicesample's default 1.0)"""
        if slice_args is None or slice_args == "intended":
            return dict(cls.SLICE_INTENDED)
        if isinstance(slice_args, str):
            if slice_args != "literal":
                raise ValueError("slice_args preset must be 'intended' or 'literal'")
            return dict(cls.SLICE_LITERAL)
        out = dict(step_out=False)
        out.update(slice_args)
        return out

    @classmethod
    def from_images(cls, img_dicts, params, **kw):
        """fields given as the reference holds them: a list of {band letter: FitsImage} dicts
        (CelesteBase.add_field) and a list of SrcParams."""
        fields = []
        for img_dict in img_dicts:
            bands = [b for b in cls.BANDS if b in img_dict]
            imgs = [img_dict[b] for b in bands]
            iset = _celeste._image_set(tuple(imgs))
            f = GibbsField(iset, [cls.BANDS.index(b) for b in bands], [im.calib for im in imgs],
                           [im.kappa for im in imgs], imgs[0].nelec.size)
            f.images = imgs
            fields.append(f)
        S = len(params)
        typ = np.array([1 if p.a == 1 else 0 for p in params], dtype=np.int32)
        u = np.array([p.u for p in params], dtype=np.float64).reshape(S, 2)
        fl = np.array([[p.flux_dict[b] for b in cls.BANDS] for p in params], dtype=np.float64).reshape(S, 5)
        sh = np.array([[p.theta, p.sigma, p.phi, p.rho] if p.a == 1 else [0., 0., 0., 0.] for p in params],
                      dtype=np.float64).reshape(S, 4)
        return cls(fields, typ, u, fl, sh, **kw)

    def step_seed(self, step, k=0):
        return step_seed(self.seed, step, self.sweeps, k)

    def counts(self, f, fluxes=None, idx=None):
        """flux in nanomaggies -> expected photons in every image of field f  (sources.py:120-129)"""
        fl = self.fluxes if fluxes is None else fluxes
        if idx is not None:
            fl = fl[idx]
        return fl[:, f.band_index] / f.calib[None, :] * f.kappa[None, :]

    def _sources(self, f):
        """the catalogue at the chain's current state on the device.  It is uploaded only when it differs from what the
        device holds (three calls per sweep ask for it, one has anything new): besides the copies saved, an unchanged
        SourceSet keeps its identity, which is what lets the next photon split take its totals image from the trace
        render's model image instead of rendering it again (CEL_OPT_SPLIT_REUSE)."""
        from . import field as _field
        if f.sset is None or f.sset.capacity < self.S:
            f.sset = _field.SourceSet(f.iset.ctx, max(self.S, 1), f.iset.B)
            f._uploaded = None
        cur = (self.typ, self.u, self.counts(f), self.shape)
        last = getattr(f, "_uploaded", None)
        if last is not None and f.sset.S == self.S and all(a.shape == b.shape and np.array_equal(a, b) for a, b in zip(cur, last)):
            return f.sset
        f.sset.set(*cur)
        f._uploaded = tuple(np.array(a, copy=True) for a in cur)
        return f.sset

    # -- Field.resample_photons: models.py:123-160 ---------------------------------------------------
    def resample_photons(self):
        import time
        t0 = time.perf_counter()
        self._split_photons()
        self._resample_sky()
        self.timing["split"] += time.perf_counter() - t0
        return self.noise_sums

    def _split_photons(self):
        """the photon split of every field at the chain's current state (celeste_sample_sources.pyx:61-156): sample patches on the
        device, photons per (source, image), the sky photons per image"""
        self.noise_sums = []
        any_patch = np.zeros(self.S, dtype=bool)
        from . import _lib
        for k, f in enumerate(self.fields):
            seed = self.step_seed("split", k)
            if self.conditional == "exact":
                # the split of the model the renderer draws from: a source takes part on its whole box (the reference's rule
                # leaves every box's first row and column out, celeste_sample_sources.pyx:50-51)
                ctx = f.iset.ctx
                was = ctx.get_option(_lib.CEL_OPT_SPLIT_FULL_BOX)
                ctx.set_option(_lib.CEL_OPT_SPLIT_FULL_BOX, 1)
                try:
                    noise = f.iset.photon_split_resident(self._sources(f), seed)
                finally:
                    ctx.set_option(_lib.CEL_OPT_SPLIT_FULL_BOX, was)
            else:
                noise = f.iset.photon_split_resident(self._sources(f), seed)
            if self.deal is not None and self.deal.kind == "strips":
                # this rank split its window and counted its strip's sky photons: the frame's sum over the ranks; and its own
                # sources' boxes have to lie inside the window (their patches must be complete)
                # (one collective for both; every rank raises if any rank's window cuts a box)
                bx, stt = f.iset.source_boxes(f.sset)
                noise = self.deal.check_boxes(bx, stt, extra=noise)
            f.sums = None                                              # photons per (source, image): read back on first use
            f.has_patch = f.iset.sample_box_areas() > 0
            any_patch |= f.has_patch.any(axis=1)
            if self.conditional == "exact":
                f.photon_rects = f.iset.photon_rects()                 # where each source's photons lie: what its box must cover
            self.noise_sums.append(noise)
        self.active = any_patch

    def _resample_sky(self):
        """the noise parameter of every image from its Gamma conditional given the sky photons of the current split
        (models.py:155-160)"""
        for f, noise in zip(self.fields, self.noise_sums):
            a_n = f.a_0 + noise
            b_n = f.b_0 + f.npix
            f.epsilon = self.rng.gamma(a_n, 1. / b_n)
            for b in range(f.iset.B):
                f.iset.set_epsilon(b, f.epsilon[b])
                if f.trace_iset is not None:
                    f.trace_iset.set_epsilon(b, f.epsilon[b])
                if getattr(f, "images", None) is not None:
                    f.images[b].epsilon = float(f.epsilon[b])

    # -- Source.resample_fluxes: sources.py:321-349 --------------------------------------------------
    def _device_flux_applies(self):
        """the whole flux step on the device (cel_flux_conditionals): one field whose resident split belongs to the catalogue on
        the device, every source this process's own, the default conditional and the device's Gamma streams, scalar priors"""
        f = self.fields[0]
        return (self.device_flux and len(self.fields) == 1 and (self.deal is None or self.deal.world == 1) and not self.host_gamma and
                self.conditional != "exact" and np.isscalar(self.flux_a_0) and np.isscalar(self.flux_b_0) and
                getattr(f, "sset", None) is not None and getattr(f, "_uploaded", None) is not None and f.sset.S == self.S and
                hasattr(f.iset, "flux_conditionals"))

    def resample_fluxes(self):
        import time
        t0 = time.perf_counter()
        if self._device_flux_applies():
            # Round 6: sums, masses, Gamma variates and the new counts never leave the device (the host form below read the
            # photon sums and the masses back, sent the Gamma shapes up, read the variates back and uploaded the whole catalogue
            # again before the location step: 0.45 ms of copies per sweep).  The values are the host form's bit for bit
            # (tests/test_gibbs.py::test_device_flux_step_is_the_host_flux_step).
            f = self.fields[0]
            new, act = f.iset.flux_conditionals(f.sset, self.step_seed("flux"), self.flux_a_0, self.flux_b_0, f.band_index, f.calib, f.kappa)
            if not np.array_equal(act, self.active):
                raise RuntimeError("cel_flux_conditionals: the device's patches are not those of the split this chain made")
            self.fluxes = np.where(self.active[:, None], new, self.fluxes)
            # the device's catalogue holds the new counts already: what _sources compares with is brought up to date, not uploaded
            f._uploaded = (f._uploaded[0], f._uploaded[1], self.counts(f), f._uploaded[3])
            self.timing["flux"] += time.perf_counter() - t0
            return self.fluxes
        band_counts = np.zeros((self.S, 5))
        for f in self.fields:
            for b in range(f.iset.B):
                band_counts[:, f.band_index[b]] += f.sums[:, b]
        a_n = self.flux_a_0 + band_counts

        mine = None if self.deal is None or self.deal.world == 1 else self.deal.mine

        # The device's part -- the sum of every source's unit stamp on its own box -- runs beside the host's draws: the last
        # field's kernel is queued (stamp_mass_begin), the Gamma variates are drawn, then its values are collected.  (Fields
        # before the last are summed synchronously: two pending calls may not share a context.  Round 2 used a worker thread
        # for the overlap: its hand-over through the interpreter lock made the step vary between 2 and 4 ms.)
        whole = {}                                  # fields whose masses come for the whole catalogue although this rank owns a part

        def field_sources(f):
            if mine is None:
                return f.sset
            if f.iset.stamp_mass_ready(f.sset):     # the replicated split has summed every source's stamps already: free, and
                whole[id(f)] = True                 # every rank holds the numbers the single-rank chain holds
                return f.sset
            from . import field as _field          # this rank's sources only; a (source, band) value does not depend on the batch
            if getattr(f, "sub", None) is None or f.sub.capacity < mine.size:
                f.sub = _field.SourceSet(f.iset.ctx, max(mine.size, 1), f.iset.B)
            f.sub.set(self.typ[mine], self.u[mine], self.counts(f, idx=mine), self.shape[mine])
            return f.sub

        def add_mass(psf_sums, f, m):
            if mine is None:
                mass = m * f.has_patch
            elif whole.get(id(f)):
                mass = np.zeros((self.S, f.iset.B))
                mass[mine] = m[mine] * f.has_patch[mine]
            else:
                mass = np.zeros((self.S, f.iset.B))
                mass[mine] = m * f.has_patch[mine]
            for b in range(f.iset.B):
                psf_sums[:, f.band_index[b]] += mass[:, b] * (f.kappa[b] / f.calib[b])

        psf_sums = np.zeros((self.S, 5))
        for f in self.fields[:-1]:
            add_mass(psf_sums, f, f.iset.stamp_mass(field_sources(f)))
        last = self.fields[-1]
        last.iset.stamp_mass_begin(field_sources(last))
        try:
            # Gamma(a_n, 1 / b_n) = standard Gamma(a_n) * (1 / b_n); every (source, band) draws from its own stream -- on the
            # device, queued behind the mass kernel (cel_gamma_streams: gamma_by_stream's sampler, streams and decisions, ~10 us;
            # the numpy draws beside the kernel took 1.7 ms on an idle host and 3.5 ms on a busy one: the step's time moved
            # with the host)
            if self.host_gamma:
                g = gamma_by_stream(a_n.ravel(), self.step_seed("flux"), np.arange(self.S * 5)).reshape(self.S, 5)
            else:
                g = last.iset.ctx.gamma_streams(a_n.ravel(), self.step_seed("flux")).reshape(self.S, 5)
        finally:
            m_last = last.iset.stamp_mass_end()     # whatever the draw does, the pending call is collected
        add_mass(psf_sums, last, m_last)
        new = g * (1. / (self.flux_b_0 + psf_sums))
        self.fluxes = np.where(self.active[:, None], new, self.fluxes)       # rows of other ranks' sources: merged at the sweep's end
        self.timing["flux"] += time.perf_counter() - t0
        return self.fluxes

    # -- Source.resample_location: sources.py:308-319, all sources in lock-step ------------------------
    def location_loglik(self, idx, U):
        """Source.location_likelihood (sources.py:185-186) of chain idx[i] at U[i], summed over fields"""
        from . import field as _field
        idx = np.asarray(idx, dtype=np.int64)
        P = idx.shape[0]
        ll = np.zeros(P)
        typ, shape = self.typ[idx], self.shape[idx]
        owner = idx.astype(np.int32)
        for f in self.fields:
            if f.prop is None or f.prop.capacity < P:
                f.prop = _field.SourceSet(f.iset.ctx, max(2 * self.S, P, 16), f.iset.B)
            cts = getattr(f, "_counts", None)            # (S, B), fixed while the locations are sampled
            pc = self.counts(f, idx=idx) if cts is None else cts[idx]
            f.prop.set(typ, U, pc, shape)
            ll += f.iset.patch_loglik_resident(f.prop, owner)
            if self.conditional == "exact":
                ll += self._exact_terms(f, idx, pc)
        return ll

    def _exact_terms(self, f, sel, pc):
        """what turns Source.log_likelihood's value for the proposals in f.prop (chains `sel`, expected photons `pc`) into the
        exact conditional: -counts * (the proposal's stamp mass on its own box) in place of -counts * sum(psf weights) where
        the source has a patch, and -inf where a photon of the source lies outside the proposal's box"""
        wsum = getattr(f, "_wsum", None)
        if wsum is None:                                               # (the PSF weights of an image set do not change)
            wsum = f._wsum = np.array([f.iset.band(b)[3:6].sum() for b in range(f.iset.B)])
        mass = f.iset.stamp_mass(f.prop)
        out = -(pc * (mass - wsum[None, :]) * f.has_patch[sel]).sum(axis=1)
        rects = getattr(f, "photon_rects", None)
        if rects is not None:
            bx, st = f.iset.source_boxes(f.prop)                       # (B, P, 4) = y0, y1, x0, x1
            bx, st, r = bx.transpose(1, 0, 2), st.T, rects[sel]
            held = r[..., 1] > r[..., 0]                               # the (source, band) pairs that hold a photon at all
            inside = (st > 0) & (bx[..., 0] <= r[..., 0]) & (bx[..., 1] >= r[..., 1]) & (bx[..., 2] <= r[..., 2]) & (bx[..., 3] >= r[..., 3])
            out = np.where((held & ~inside).any(axis=1), -np.inf, out)
        return out

    # -- the galaxies' shapes: celeste_mcmc.py:209-243 (skew_likelihood, slice_sample_skew) ---------------------------
    def shape_logprob(self, idx, TH):
        """skew_likelihood (celeste_mcmc.py:209-222) of chain idx[i] at shape TH[i] = (theta, sigma, phi, rho): the
        log-prior, and where that is finite the source's conditional likelihood given its photons
        (Source.log_likelihood(shape=...), sources.py:134-183) -- one launch for all of them"""
        from . import field as _field
        idx = np.asarray(idx, dtype=np.int64)
        TH = np.asarray(TH, dtype=np.float64).reshape(idx.shape[0], 4)
        out = np.asarray(self.shape_logprior(TH), dtype=np.float64).copy()
        ok = np.isfinite(out)                       # outside the prior's support nothing is rendered (:213-214)
        if not ok.any():
            return out
        sel = idx[ok]
        ll = np.zeros(sel.shape[0])
        owner = sel.astype(np.int32)
        for f in self.fields:
            if f.prop is None or f.prop.capacity < sel.shape[0]:
                f.prop = _field.SourceSet(f.iset.ctx, max(2 * self.S, sel.shape[0], 16), f.iset.B)
            cts = getattr(f, "_counts", None)
            pc = self.counts(f, idx=sel) if cts is None else cts[sel]
            f.prop.set(self.typ[sel], self.u[sel], pc, TH[ok])
            ll += f.iset.patch_loglik_resident(f.prop, owner)
            if self.shape_mass == "exact":
                ll += self._exact_terms(f, sel, pc)
        out[ok] += ll
        return out

    def resample_shapes(self):
        """slice_sample_skew (celeste_mcmc.py:224-243) for every galaxy at once: one slicesample update of
        (theta, sigma, phi, rho) per galaxy -- random directions, stepping out by doubling -- all galaxies in
        lock-step against the resident photon patches; phi is wrapped afterwards (:241).  Stars are skipped
        (Source.resample_shape, sources.py:321-325)."""
        import time
        from .util.infer.slicesample import ChainStreams, slicesample_lockstep
        t0 = time.perf_counter()
        mine = self.active if self.deal is None else (self.active & self.deal.mask)
        gal = np.nonzero(mine & (self.typ == 1))[0]
        seed = self.step_seed("shape")
        if self._shape_engine_on_device():
            # the state machine on the device (cel_slice_sample): the directions are drawn here, from each chain's
            # normal stream, exactly as the host engine draws them; nothing but counters crosses PCIe per round
            f = self.fields[0]
            sset = self._sources(f)
            a = self.shape_args
            dirs = None if a.get("compwise", True) else ChainStreams(seed, np.arange(self.S)).directions(int(a.get("numdir", 2)), 4)
            ids = np.where(mine & (self.typ == 1), np.arange(self.S), -1).astype(np.int32)
            new, _, st = f.iset.slice_sample(sset, 1, a.get("sigma", 1.0), seed, dirs=dirs, step_out=a.get("step_out", True),
                                             max_steps_out=a.get("max_steps_out", 1000), phi_max=self.phi_period, chain_ids=ids)
            new[:, 2] = np.where(ids >= 0, (new[:, 2] + self.phi_period) % self.phi_period, new[:, 2])
            self.shape = new
            f._uploaded = None                                 # (the device holds the unwrapped angles)
            self.timing["shape_rounds"] += st["rounds"]
            self.timing["shape_evals"] += st["evals"]
            self.timing["shape"] += time.perf_counter() - t0
            return self.shape
        for f in self.fields:
            f._counts = self.counts(f)
        if gal.size:
            st = {}
            new, _ = slicesample_lockstep(self.shape[gal], lambda i, TH: self.shape_logprob(gal[i], TH),
                                          seed=seed, chain_ids=gal, stats=st, **self.shape_args)
            new[:, 2] = (new[:, 2] + self.phi_period) % self.phi_period
            self.shape[gal] = new
            self.timing["shape_rounds"] += st["rounds"]
            self.timing["shape_evals"] += st["evals"]
        for f in self.fields:
            f._counts = None
        self.timing["shape"] += time.perf_counter() - t0
        return self.shape

    def _shape_engine_on_device(self):
        """the device runs the shape step when there is one field, the log-prior is the built-in one and the options are
        those of slice_sample_skew's family: no stepping out or doubling, component-wise or random directions"""
        a = self.shape_args
        ok = (len(self.fields) == 1 and self._default_shape_prior and (not a.get("step_out", True) or a.get("doubling_step", True))
              and set(a) <= {"step_out", "doubling_step", "compwise", "numdir", "sigma", "max_steps_out"}
              and a.get("accept", "reference") == "reference" and self.shape_mass == "reference" and self.conditional == "reference")
        if self.engine == "device" and not ok:
            raise ValueError("the device shape sampler runs one field, the built-in log-prior, stepping out by doubling or none")
        return ok and self.engine != "host"

    def _device_engine_applies(self):
        a = self.slice_args
        return (len(self.fields) == 1 and not a.get("step_out", True) and a.get("compwise", True)
                and set(a) <= {"step_out", "compwise", "sigma"} and self.conditional == "reference")

    def resample_locations(self):
        import time
        from .util.infer.slicesample import slicesample_lockstep
        t0 = time.perf_counter()
        use_device = self.engine == "device" or (self.engine == "auto" and self._device_engine_applies())
        if use_device:
            if not self._device_engine_applies():
                raise ValueError("the device slice sampler runs one field with step_out=False, compwise=True")
            f = self.fields[0]
            sset = self._sources(f)                        # the catalogue with the fluxes just drawn (other ranks' rows are
            # stale until the merge: a chain reads only its own source's counts)
            ids = None if self.deal is None or self.deal.world == 1 else self.deal.chain_ids()
            new_u, _, st = f.iset.slice_locations(sset, self.slice_args.get("sigma", 1.0), self.step_seed("location"),
                                                  chain_ids=ids)
            self.u = new_u
            if getattr(f, "_uploaded", None) is not None:      # the device's catalogue moved with the chains: it IS the new state
                f._uploaded = (f._uploaded[0], new_u.copy(), f._uploaded[2], f._uploaded[3])
            self.timing["rounds"] += st["rounds"]
            self.timing["evals"] += st["evals"]
            self.timing["loc_bytes"] = self.timing.get("loc_bytes", 0) + st["algorithmic_bytes"]
            self.timing["loc_launches"] = self.timing.get("loc_launches", 0) + st["launches"]
            self.timing["location"] += time.perf_counter() - t0
            return self.u
        act = np.nonzero(self.active if self.deal is None else (self.active & self.deal.mask))[0]
        for f in self.fields:
            f._counts = self.counts(f)
        if act.size:
            st = {}
            new_u, _ = slicesample_lockstep(self.u[act], lambda i, U: self.location_loglik(act[i], U),
                                            seed=self.step_seed("location"), chain_ids=act, stats=st,
                                            **self.slice_args)
            self.u[act] = new_u
            self.timing["rounds"] += st["rounds"]
            self.timing["evals"] += st["evals"]
        for f in self.fields:
            f._counts = None
        self.timing["location"] += time.perf_counter() - t0
        return self.u

    def sweep(self, shapes=False):
        """CelesteBase.resample_model: every field's photons, then every source (fluxes, location) -- and, with
        shapes=True, the galaxies' shapes as sample_galaxy_params does (celeste_mcmc.py:166-243;
        Source.resample_shape is a stub in the reference, sources.py:321-325, so it is off by default)"""
        solo = self.deal is not None and getattr(self.deal, "solo", False) and self.deal.world > 1
        if solo:        # one rank of an N-rank chain played alone (bench.py --as-rank): the other ranks' rows keep their values
            before = (self.u.copy(), self.fluxes.copy(), self.shape.copy())
        self.resample_photons()
        self.resample_fluxes()
        self.resample_locations()
        if shapes:
            self.resample_shapes()
        self.merge_ranks()
        if solo:
            other = ~self.deal.mask
            self.u[other], self.fluxes[other], self.shape[other] = before[0][other], before[1][other], before[2][other]
        self.sweeps += 1

    def sweep_reversed(self, shapes=False):
        """The sweep's blocks in REVERSE order -- (shapes,) locations, fluxes, sky levels, photons -- on the augmented state (sources, sky
        levels, the current photon split): the time reversal of sweep().  Every block is reversible with respect to its
        conditional (Gibbs draws; slice updates whose axis order is shuffled per call), so a chain run backwards from a state
        is a chain run with this.  It needs a photon split of the current state to start from (_split_photons()).  Used by the
        calibration test (tests/test_calibration.py) to put the true parameters at a random position of a stationary chain;
        a single rank."""
        if self.deal is not None and self.deal.world > 1:
            raise ValueError("sweep_reversed runs a single-rank chain")
        if not self.noise_sums:
            raise ValueError("sweep_reversed needs a photon split of the current state: call _split_photons() first")
        if shapes:
            self.resample_shapes()
        self.resample_locations()
        self.resample_fluxes()
        self._resample_sky()
        self._split_photons()
        self.sweeps += 1

    def merge_ranks(self):
        """one chain over several GPUs: every rank's new fluxes and locations to every rank (one all-gather)"""
        if self.deal is None or self.deal.world == 1:
            return
        import time
        t0 = time.perf_counter()
        both = self.deal.merge(np.concatenate([self.u, self.fluxes, self.shape], axis=1))
        self.u, self.fluxes, self.shape = (np.ascontiguousarray(both[:, :2]), np.ascontiguousarray(both[:, 2:7]),
                                           np.ascontiguousarray(both[:, 7:]))
        self.timing["merge"] = self.timing.get("merge", 0.0) + time.perf_counter() - t0

    def log_likelihood(self):
        """sum of img_log_likelihood over every image of every field at the current state (models.py:104-108)"""
        tot = 0.0
        strips = self.deal is not None and self.deal.kind == "strips"
        for f in self.fields:
            if strips and f.trace_iset is None:
                # the window image set adds its own rows' terms only (strip_gibbs_field, window_trace): one render gives this
                # rank's share of the trace AND the model image the next split and flux step start from
                ll, _ = f.iset.render(self._sources(f), loglik=True)
            elif strips:        # this rank's strip of every image; the strips' sums added over the ranks
                from . import field as _field
                if getattr(f, "trace_sset", None) is None or f.trace_sset.capacity < self.S:
                    f.trace_sset = _field.SourceSet(f.trace_iset.ctx, max(self.S, 1), f.trace_iset.B)
                ll, _ = f.trace_iset.render(f.trace_sset.set(self.typ, self.u, self.counts(f), self.shape), loglik=True)
            else:
                ll, _ = f.iset.render(self._sources(f), loglik=True)
            tot += ll
        if strips:
            tot = float(self.deal.rank_sum([tot])[0])
        return tot
