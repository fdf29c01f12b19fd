"""Deterministic synthetic fields for the parity tests at scale and for bench.py.

Images: per-band PSF / CALIB / GAIN / SKY of the SDSS stamp field 253.1147-11.6072
(sdss_bands.py), WCS CD = diag(-1.1e-4, +1.1e-4) deg/px, CRVAL = (253.11475, 11.60716),
CRPIX at the frame centre.  Sources from RandomState(seed): uniform pixel positions, fluxes
log-uniform in [1, 100] nmgy per band, galaxies with theta ~ U(.05,.95), r_e ~ logU(0.5", 4"),
rho ~ U(.2,.95), phi ~ U(0,180) deg.  nelec = RandomState(seed+1).poisson(lambda_true) with
lambda_true rendered by the HIP path itself.
"""
import numpy as np

from . import field as _field
from . import sdss_bands as sb

CONFIGS = {
    # name: (S, B, H, W, galaxy fraction)  -- BASELINE.json configs[0..2]
    "stamp51": (1, 1, 51, 51, 0.0),
    "stars1k_512": (1000, 5, 512, 512, 0.0),
    "mixed10k_2048": (10000, 5, 2048, 2048, 0.5),
    # not a BASELINE config: the star-only regime of the same frame, where the HBM roof binds
    # (3-component stamps: ~9 fp64 ops per source-pixel against 16 B per image pixel)
    "stars10k_2048": (10000, 5, 2048, 2048, 0.0),
    "stars2k_4096": (2000, 5, 4096, 4096, 0.0),
}


def make_bands(H, W, nbands=5):
    """(B, 37) cel_band records for an H x W frame (R <= 0: the library computes it)."""
    recs = []
    ups = sb.CD
    ups_inv = np.linalg.inv(ups)
    for b in range(nbands):
        k = b % 5
        recs.append(_field.pack_band(sb.SKY[k] * sb.GAIN[k], sb.GAIN[k], sb.CALIB[k], sb.PSF_WEIGHTS[k],
                                     sb.PSF_MEANS[k], sb.PSF_COVARS[k], [W / 2.0, H / 2.0], sb.CRVAL, ups,
                                     ups_inv, 0.0))
    return np.stack(recs)


def pixel2equa(band, pix):
    """vectorised fits_image.py:176-181 for a (37,) band record and (S,2) pixels"""
    rho, phi, ups = band[24:26], band[26:28], band[28:32].reshape(2, 2)
    iwc = (pix - rho[None, :]) @ ups.T
    return np.column_stack([iwc[:, 0] / np.cos(phi[1] / 180. * np.pi) + phi[0], iwc[:, 1] + phi[1]])


def make_sources(S, H, W, bands, frac_gal=0.5, seed=42):
    """-> dict(type[S] i32, radec[S,2], counts[S,B], shape[S,4], flux[S,B], pix[S,2])"""
    rs = np.random.RandomState(seed)
    B = bands.shape[0]
    pix = np.column_stack([rs.uniform(0, W, S), rs.uniform(0, H, S)])
    typ = (rs.rand(S) < frac_gal).astype(np.int32)
    flux = np.exp(rs.uniform(np.log(1.0), np.log(100.0), size=(S, 5)))[:, [b % 5 for b in range(B)]]
    theta = rs.uniform(0.05, 0.95, S)
    sigma = np.exp(rs.uniform(np.log(0.5), np.log(4.0), S))
    rho = rs.uniform(0.2, 0.95, S)
    phi = rs.uniform(0.0, 180.0, S)
    shape = np.column_stack([theta, sigma, phi, rho])
    counts = flux / bands[None, :, 2] * bands[None, :, 1]       # nmgy2counts, fits_image.py:183-184
    return dict(type=typ, radec=pixel2equa(bands[0], pix), counts=counts, shape=shape, flux=flux, pix=pix)


class SyntheticField(object):
    """A device-resident synthetic field: .images (ImageSet), .sources (SourceSet), .src (host dict)."""

    def __init__(self, ctx, S, B, H, W, frac_gal=0.5, seed=42, with_nelec=True):
        self.S, self.B, self.H, self.W = S, B, H, W
        self.bands = make_bands(H, W, B)
        self.src = make_sources(S, H, W, self.bands, frac_gal, seed)
        self.images = _field.ImageSet(ctx, self.bands, H, W)
        self.sources = _field.SourceSet(ctx, max(S, 1), B).set(self.src["type"], self.src["radec"],
                                                               self.src["counts"], self.src["shape"])
        self.nelec = None
        if with_nelec:
            self.images.render(self.sources, loglik=False)
            lam = self.images.model_images()
            self.nelec = np.random.RandomState(seed + 1).poisson(lam).astype(np.float64)
            self.images.set_nelec(self.nelec)

    def flux5(self):
        """(S, 5) fluxes in nanomaggies by band letter u,g,r,i,z (band b of this field is letter b % 5)"""
        out = np.zeros((self.S, 5))
        for b in range(self.B):
            out[:, b % 5] = self.src["flux"][:, b]
        return out

    @classmethod
    def from_config(cls, ctx, name, seed=42, with_nelec=True):
        S, B, H, W, fg = CONFIGS[name]
        return cls(ctx, S, B, H, W, fg, seed, with_nelec)


def fits_images(field):
    """The field's band images as FitsImage objects (what the reference-API functions take)."""
    from .fits_image import FitsImage
    out = []
    for b in range(field.B):
        r = field.bands[b]
        out.append(FitsImage("ugriz"[b % 5], field.nelec[b] if field.nelec is not None else np.zeros((field.H, field.W)),
                             epsilon=r[0], kappa=r[1], calib=r[2], weights=r[3:6], means=r[6:12].reshape(3, 2),
                             covars=r[12:24].reshape(3, 2, 2), rho_n=r[24:26], phi_n=r[26:28], Ups_n=r[28:32].reshape(2, 2)))
    return out
