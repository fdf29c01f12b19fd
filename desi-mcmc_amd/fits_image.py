"""FitsImage: the per-band image record the hot path consumes.

Mirror of CelestePy/fits_image.py:15-223 as an *input layout*: same attribute names
(nelec, epsilon, kappa, calib, weights, means, covars, invcovars, logdets, rho_n, phi_n,
Ups_n, Ups_n_inv, R, band, shape) and the same scalar WCS helpers, so that code written
against the reference's FitsImage reads the same here.  FITS I/O through fitsio / astropy /
tractor is out of scope; a plain-FITS primary-HDU reader covers the reference's stamp files.
"""
import numpy as np

from . import field as _field

BANDS = ["u", "g", "r", "i", "z"]


def _read_primary_hdu(path):
    """Minimal FITS reader: header cards + BITPIX -64/-32 image of the primary HDU."""
    with open(path, "rb") as f:
        raw = f.read()
    hdr, pos, done = {}, 0, False
    while not done:
        block = raw[pos:pos + 2880]
        pos += 2880
        for i in range(36):
            card = block[i * 80:(i + 1) * 80].decode("ascii")
            key = card[:8].strip()
            if key == "END":
                done = True
                break
            if card[8:10] != "= ":
                continue
            body = card[10:]
            if body.lstrip().startswith("'"):
                s = body.lstrip()
                hdr[key] = s[1:s.find("'", 1)].strip()
            else:
                tok = body.split("/")[0].strip()
                if tok in ("T", "F"):
                    hdr[key] = tok == "T"
                else:
                    try:
                        hdr[key] = int(tok)
                    except ValueError:
                        hdr[key] = float(tok.replace("D", "E"))
    n1, n2 = int(hdr["NAXIS1"]), int(hdr["NAXIS2"])
    dt = {-64: ">f8", -32: ">f4"}[int(hdr["BITPIX"])]
    img = np.frombuffer(raw[pos:pos + n1 * n2 * int(dt[2])], dtype=dt).reshape(n2, n1).astype(np.float64)
    return hdr, img


class FitsImage(object):
    """One band image.  Build with FitsImage.from_header(...) / from_file(...) / from_record(...)."""

    def __init__(self, band, nelec, epsilon, kappa, calib, weights, means, covars, rho_n, phi_n, Ups_n,
                 darkvar=None, header=None):
        self.band = band
        self.header = header
        self.nelec = np.array(nelec, dtype=np.float64)
        self.nelec.flags.writeable = False               # fits_image.py:96
        self.shape = self.nelec.shape
        self.rho_n = np.asarray(rho_n, dtype=np.float64)  # CRPIX - 1        fits_image.py:99
        self.phi_n = np.asarray(phi_n, dtype=np.float64)  # CRVAL            fits_image.py:100
        self.Ups_n = np.asarray(Ups_n, dtype=np.float64).reshape(2, 2)
        self.Ups_n_inv = np.linalg.inv(self.Ups_n)
        self.use_wcs = False
        self.kappa = float(kappa)
        self.epsilon = float(epsilon)
        self.epsilon0 = self.epsilon
        self.darkvar = darkvar
        self.calib = float(calib)
        self.weights = np.asarray(weights, dtype=np.float64).reshape(3)
        self.means = np.asarray(means, dtype=np.float64).reshape(3, 2)
        self.covars = np.asarray(covars, dtype=np.float64).reshape(3, 2, 2)
        self.invcovars = np.array([np.linalg.inv(c) for c in self.covars])
        self.logdets = np.array([np.linalg.slogdet(c)[1] for c in self.covars])
        # star bounding radius, 1 - 1e-3 of the PSF mass (fits_image.py:151-155)
        self.R = _field.bounding_radius(self.weights, self.means, self.covars, 0.001)

    @property
    def psf_mog(self):
        """the PSF as a MixtureOfGaussians (fits_image.py:147); `psf` is the name
        gen_galaxy_psf_image reads (celeste_galaxy_conditionals.py:200)"""
        if getattr(self, "_psf_mog", None) is None:
            from .util.dists.mog import MixtureOfGaussians
            self._psf_mog = MixtureOfGaussians(means=self.means, covs=self.covars, pis=self.weights)
        return self._psf_mog

    psf = psf_mog

    # ---- constructors -------------------------------------------------------------------
    @classmethod
    def from_header(cls, band, header, img):
        """header: dict with CALIB, SKY, GAIN, CRPIX1/2, CRVAL1/2, CD*, PSF_P0..17 (fits_image.py:85-147)."""
        dn = np.asarray(img, dtype=np.float64) / header["CALIB"] + header["SKY"]
        nelec = np.round(dn * header["GAIN"])
        psf = [header["PSF_P%d" % i] for i in range(18)]
        cv = np.array(psf[9:]).reshape(3, 3)               # [var_x, var_y, cov_xy] per row
        covars = np.array([[[c[0], c[2]], [c[2], c[1]]] for c in cv])
        return cls(band, nelec, epsilon=header["SKY"] * header["GAIN"], kappa=header["GAIN"],
                   calib=header["CALIB"], weights=psf[0:3], means=np.array(psf[3:9]).reshape(3, 2),
                   covars=covars, rho_n=np.array([header["CRPIX1"], header["CRPIX2"]]) - 1,
                   phi_n=[header["CRVAL1"], header["CRVAL2"]],
                   Ups_n=[[header["CD1_1"], header["CD1_2"]], [header["CD2_1"], header["CD2_2"]]],
                   darkvar=header.get("DARKVAR"), header=header)

    @classmethod
    def from_file(cls, band, filename=None, fits_file_template=None):
        path = fits_file_template % band if fits_file_template else filename
        hdr, img = _read_primary_hdu(path)
        return cls.from_header(band, hdr, img)

    @classmethod
    def from_record(cls, band, rec, b, nelec):
        """rec: dict of per-band stacked arrays with the keys of field.BAND_KEYS."""
        return cls(band, nelec, rec["eps"][b], rec["kappa"][b], rec["calib"][b], rec["weights"][b],
                   rec["means"][b], rec["covars"][b], rec["rho"][b], rec["phi"][b], rec["ups"][b])

    # ---- the C-ABI record ------------------------------------------------------------------
    def band_record(self):
        return _field.pack_band(self.epsilon, self.kappa, self.calib, self.weights, self.means, self.covars,
                                self.rho_n, self.phi_n, self.Ups_n, self.Ups_n_inv, self.R)

    # ---- scalar WCS helpers (host side; the device repeats them per source in k_prep) -------
    def contains(self, s_equa, pad=50):
        v_s = self.equa2pixel(s_equa)                      # fits_image.py:157-164 (axis mix kept)
        return (v_s[0] > -pad) and (v_s[0] < self.nelec.shape[0] + pad) and \
               (v_s[1] > -pad) and (v_s[1] < self.nelec.shape[1] + pad)

    def equa2pixel(self, s_equa):
        phi1rad = self.phi_n[1] / 180. * np.pi             # fits_image.py:166-174
        s_iwc = np.array([(s_equa[0] - self.phi_n[0]) * np.cos(phi1rad), (s_equa[1] - self.phi_n[1])])
        return np.dot(self.Ups_n_inv, s_iwc) + self.rho_n

    def pixel2equa(self, s_pixel):
        phi1rad = self.phi_n[1] / 180. * np.pi             # fits_image.py:176-181
        s_iwc = np.dot(self.Ups_n, np.asarray(s_pixel) - self.rho_n)
        return np.array([s_iwc[0] / np.cos(phi1rad) + self.phi_n[0], s_iwc[1] + self.phi_n[1]])

    def nmgy2counts(self, flux):
        return (flux / self.calib) * self.kappa            # fits_image.py:183-184

    def make_pixel_grid(self):
        """(H*W, 2) stack of the 1-BASED (x, y) pixel coordinates, x fastest (C order of the H x W frame): fits_image.py:186-194"""
        y_grid = np.arange(self.nelec.shape[0], dtype=np.float64) + 1
        x_grid = np.arange(self.nelec.shape[1], dtype=np.float64) + 1
        xx, yy = np.meshgrid(x_grid, y_grid, indexing='xy')
        return np.column_stack((xx.ravel(order='C'), yy.ravel(order='C')))

    @property
    def pixel_grid(self):
        """the grid the reference builds in its constructor and keeps (fits_image.py:95); here on first read -- 67 MB for a
        2048^2 frame that nothing on the render path looks at"""
        if getattr(self, "_pixel_grid", None) is None:
            self._pixel_grid = self.make_pixel_grid()
        return self._pixel_grid

    @pixel_grid.setter
    def pixel_grid(self, value):
        self._pixel_grid = value

    def cd_at_pixel(self, x, y):
        ra0, dec0 = self.pixel2equa(np.array([x, y]))      # fits_image.py:196-216
        step = 10.
        rax, decx = self.pixel2equa(np.array([x + step, y]))
        ray, decy = self.pixel2equa(np.array([x, y + step]))
        cosd = np.cos(dec0 * (np.pi / 180.))
        return np.array([[(rax - ra0) / step * cosd, (ray - ra0) / step * cosd],
                         [(decx - dec0) / step, (decy - dec0) / step]])
