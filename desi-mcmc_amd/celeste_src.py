"""SrcParams: the per-source record the hot path consumes (input layout only).

Field names follow CelestePy/celeste_src.py:57-94 so that catalogues built for the reference
can be handed over unchanged:
    a      0 star, 1 galaxy, None = untyped catalogue row
    u      (ra, dec) in degrees
    fluxes per-band flux in nanomaggies: dict keyed by band letter OR length-5 array (ugriz)
    t, b   black-body temperature / brightness (needs a planck hook; see celeste.py)
    theta, sigma, phi, rho   galaxy shape: exp fraction, r_e [arcsec], angle [DEGREES, as the
           code of celeste_galaxy_conditionals.py:97 uses it], axis ratio
"""
import numpy as np

BANDS = ["u", "g", "r", "i", "z"]


def mags2nanomaggies(mags):
    return np.power(10., (np.asarray(mags) - 22.5) / -2.5)


def nanomaggies2mags(nanos):
    return (-2.5) * np.log10(nanos) + 22.5


# A process-wide modification clock: every attribute assignment on a SrcParams ticks it and stamps the object.  The render
# / likelihood functions keep the arrays they gathered from a LIST of SrcParams and, called again with the same list,
# re-read only the objects whose stamp moved (celeste._source_arrays): the reference's callers evaluate the likelihood
# of a list after changing ONE source (util/infer/mcmc_transitions.py:37-152, celeste_mcmc.py:130).
_CLOCK = [0]
_set = object.__setattr__
# ... and a bounded log of (stamp, id(object)) of the latest assignments: a cached list of 10 000 sources finds the one object
# that was assigned to since its last look by reading the log's tail instead of every object's stamp (celeste._cached_list_arrays;
# when the log no longer reaches back to that look, it reads the stamps)
import collections as _collections
_LOG = _collections.deque(maxlen=8192)


def clock():
    return _CLOCK[0]


def touch(src):
    """mark `src` as modified.  Needed only after changing one of its containers IN PLACE (src.u[0] = x,
    src.fluxes['r'] = f) without assigning the attribute afterwards; `src.u = u` -- what the reference's own moves do
    after their in-place edits (mcmc_transitions.py:49-51) -- stamps by itself."""
    _CLOCK[0] += 1
    _set(src, "_stamp", _CLOCK[0])
    _LOG.append((_CLOCK[0], id(src)))


def stamped_since(clock0, limit=64):
    """ids of the objects assigned to after the clock read `clock0` (at most limit + 1 of them: the caller then reads the
    stamps instead), or None when the log does not reach back that far"""
    if _CLOCK[0] - clock0 > limit:                 # (every assignment ticks the clock once)
        return None
    if not _LOG or _LOG[0][0] > clock0 + 1:
        return None if _CLOCK[0] != clock0 else []
    out = []
    for stamp, oid in reversed(_LOG):
        if stamp <= clock0:
            break
        out.append(oid)
    return out


class SrcParams(object):
    __slots__ = ("a", "u", "b", "t", "v", "theta", "phi", "sigma", "rho", "fluxes", "ell", "d", "header", "_stamp")

    def __init__(self, u, a=None, b=None, t=None, v=None, theta=None, phi=None, sigma=None, rho=None,
                 fluxes=None, ell=None, d=None, header=None):
        _set(self, "u", u); _set(self, "a", a); _set(self, "b", b); _set(self, "t", t); _set(self, "v", v)
        _set(self, "theta", theta); _set(self, "phi", phi); _set(self, "sigma", sigma); _set(self, "rho", rho)
        _set(self, "fluxes", fluxes); _set(self, "ell", ell); _set(self, "d", d); _set(self, "header", header)
        _CLOCK[0] += 1
        _set(self, "_stamp", _CLOCK[0])

    def __setattr__(self, name, value):
        _set(self, name, value)
        _CLOCK[0] += 1
        _set(self, "_stamp", _CLOCK[0])
        _LOG.append((_CLOCK[0], id(self)))

    def __eq__(self, other):
        return isinstance(other, SrcParams) and np.array_equal(self.u, other.u) and self.b == other.b

    def __hash__(self):
        return id(self)

    def flux(self, band):
        """Flux in `band` for either flux layout (dict by letter, or ugriz array)."""
        if isinstance(self.fluxes, dict):
            return self.fluxes[band]
        return self.fluxes[BANDS.index(band)]

    @property
    def flux_dict(self):
        if isinstance(self.fluxes, dict):
            return dict(self.fluxes)
        return dict(zip(BANDS, self.fluxes))

    @property
    def mags(self):
        return nanomaggies2mags(np.array([self.flux(b) for b in BANDS]))

    @property
    def shape(self):
        return np.array([self.theta, self.sigma, self.phi, self.rho])

    @shape.setter
    def shape(self, shape):
        self.theta, self.sigma, self.phi, self.rho = shape

    def is_star(self):
        return self.a == 0

    def is_galaxy(self):
        return self.a == 1

    def __str__(self):
        kind = {0: "StrSrc", 1: "GalSrc"}.get(self.a, "NoType")
        return "%s: u=(%2.2f, %2.2f)" % (kind, self.u[0], self.u[1])


class SrcCatalog(object):
    """A catalogue of sources held as arrays (structure of arrays) that still reads as a sequence
    of SrcParams: `len(cat)`, `cat[i]`, iteration -- each element is a view whose attributes read
    and write the arrays.  The render / likelihood functions (celeste.gen_model_image,
    celeste_likelihood[_multi_image], ...) take it wherever the reference takes a list of
    SrcParams and hand its arrays to the device without touching the sources one by one: at
    10 000 sources the per-object gather of a plain list costs more than the render itself.

        a        (S,) int: 0 star, 1 galaxy, -1 = None (untyped catalogue row, celeste.py:52-62)
        u        (S, 2) ra, dec in degrees
        fluxes   (S, 5) nanomaggies in u, g, r, i, z
        shape    (S, 4) theta, sigma, phi (degrees), rho
    """

    def __init__(self, a, u, fluxes, shape=None):
        self.a = np.array([-1 if v is None else int(v) for v in a] if not isinstance(a, np.ndarray) else a, dtype=np.int64)
        S = self.a.shape[0]
        self.u = np.array(u, dtype=np.float64).reshape(S, 2)
        self.fluxes = np.array(fluxes, dtype=np.float64).reshape(S, 5)
        self.shape = np.zeros((S, 4)) if shape is None else np.array(shape, dtype=np.float64).reshape(S, 4)

    @classmethod
    def from_params(cls, params):
        """pack a list of SrcParams (fluxes as a dict or a ugriz array)"""
        params = list(params)
        a = [p.a for p in params]
        u = [p.u for p in params]
        fl = [[p.flux(b) for b in BANDS] for p in params]
        sh = [[p.theta, p.sigma, p.phi, p.rho] if p.a == 1 else [0., 0., 0., 0.] for p in params]
        return cls(a, u, fl, sh)

    def __len__(self):
        return self.a.shape[0]

    def views(self):
        """The catalogue as a plain LIST of per-source objects with the attribute names of SrcParams, each a
        window on this catalogue's arrays: what a caller that wants `list[SrcParams]` semantics (celeste_em.py:25,159,
        celeste_mcmc.py:130 pass such lists) should hold.  Writing `srcs[i].u = ...` writes the arrays, and the
        render / likelihood functions recognise the list (or any sub-list of it) and take the arrays without
        touching the objects one by one.  The same objects are returned on every call."""
        if getattr(self, "_views", None) is None or len(self._views) != len(self):
            self._views = [_SrcView(self, i) for i in range(len(self))]
        return self._views

    def __getitem__(self, i):
        if isinstance(i, slice):
            return SrcCatalog(self.a[i], self.u[i], self.fluxes[i], self.shape[i])
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError(i)
        return _SrcView(self, i)

    def __iter__(self):
        for i in range(len(self)):
            yield _SrcView(self, i)


class _SrcView(object):
    """one row of a SrcCatalog with the attribute names of SrcParams"""
    __slots__ = ("_c", "_i")
    t = b = v = ell = d = header = None

    def __init__(self, cat, i):
        self._c, self._i = cat, i

    a = property(lambda s: None if s._c.a[s._i] < 0 else int(s._c.a[s._i]),
                 lambda s, v: s._c.a.__setitem__(s._i, -1 if v is None else int(v)))
    u = property(lambda s: s._c.u[s._i], lambda s, v: s._c.u.__setitem__(s._i, v))
    fluxes = property(lambda s: s._c.fluxes[s._i], lambda s, v: s._c.fluxes.__setitem__(
        s._i, [v[b] for b in BANDS] if isinstance(v, dict) else v))
    shape = property(lambda s: s._c.shape[s._i], lambda s, v: s._c.shape.__setitem__(s._i, v))
    theta = property(lambda s: s._c.shape[s._i, 0], lambda s, v: s._c.shape.__setitem__((s._i, 0), v))
    sigma = property(lambda s: s._c.shape[s._i, 1], lambda s, v: s._c.shape.__setitem__((s._i, 1), v))
    phi = property(lambda s: s._c.shape[s._i, 2], lambda s, v: s._c.shape.__setitem__((s._i, 2), v))
    rho = property(lambda s: s._c.shape[s._i, 3], lambda s, v: s._c.shape.__setitem__((s._i, 3), v))

    def flux(self, band):
        return self._c.fluxes[self._i, BANDS.index(band)]

    @property
    def flux_dict(self):
        return dict(zip(BANDS, self._c.fluxes[self._i]))

    def is_star(self):
        return self.a == 0

    def is_galaxy(self):
        return self.a == 1
