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


class SrcParams(object):
    __slots__ = ("a", "u", "b", "t", "v", "theta", "phi", "sigma", "rho", "fluxes", "ell", "d", "header")

    def __init__(self, u, a=None, b=None, t=None, v=None, theta=None, phi=None, sigma=None, rho=None,
                 fluxes=None, ell=None, d=None, header=None):
        self.u, self.a, self.b, self.t, self.v = u, a, b, t, v
        self.theta, self.phi, self.sigma, self.rho = theta, phi, sigma, rho
        self.fluxes, self.ell, self.d, self.header = fluxes, ell, d, header

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
