"""Mirror of the galaxy-stamp API of CelestePy/celeste_galaxy_conditionals.py:90-256.

gen_galaxy_psf_image keeps the reference's signature and (patch, ylim, xlim) return; the
stamp itself is evaluated by the HIP kernels.  The small 2x2 / 42-component host helpers are
kept because callers use them directly (bounds, transformations); they are not the hot loop.
"""
import numpy as np

from . import celeste as _celeste
from . import field as _field
from . import mixture_profiles as mp

BANDS = ['u', 'g', 'r', 'i', 'z']


def gen_galaxy_ra_dec_basis(sig_s, rho_s, phi_s):
    """r_e unit vectors -> (dRA, dDec) degrees  -- celeste_galaxy_conditionals.py:90-107.
    phi_s is in DEGREES (the code's convention, :97)."""
    phi = (90. - phi_s) * np.pi / 180.
    re_deg = max(1. / 30, sig_s) / 3600.
    cp, sp = np.cos(phi), np.sin(phi)
    return re_deg * np.array([[cp, sp * rho_s], [-sp, cp * rho_s]])


def gen_galaxy_transformation(sig_s, rho_s, phi_s, Ups_n):
    """Tinv: effective radii -> pixels  -- celeste_galaxy_conditionals.py:109-125"""
    G = gen_galaxy_ra_dec_basis(sig_s, rho_s, phi_s)
    T = np.dot(np.linalg.inv(G), Ups_n)
    return np.linalg.inv(T)


def galaxy_mixture(th, u_s, img):
    """(pis[42], means[42,2], covs[42,2,2], (px,py)) of profile (x) PSF, galaxy-major order
    (celeste_galaxy_conditionals.py:193-203 with util/dists/mog.py:75-100)."""
    theta_s, sig_s, phi_s, rho_s = th[0:4]
    px, py = img.equa2pixel(u_s)
    Tinv = gen_galaxy_transformation(sig_s, rho_s, phi_s, img.cd_at_pixel(px, py))
    W = np.dot(Tinv, Tinv.T)
    amp = np.concatenate([theta_s * mp.exp_amp, (1. - theta_s) * mp.dev_amp])
    var = np.concatenate([mp.exp_var, mp.dev_var])
    pis = (amp[:, None] * img.weights[None, :]).reshape(-1)
    means = np.reshape(np.array([px, py])[None, None, :] + img.means[None, :, :] + np.zeros((14, 1, 1)), (-1, 2))
    covs = np.reshape(var[:, None, None, None] * W[None, None] + img.covars[None], (-1, 2, 2))
    return pis, means, covs, (px, py)


def gen_galaxy_psf_image_bound(src, img):
    """bounding radius of a galaxy's stamp (error 1e-5)  -- celeste_galaxy_conditionals.py:217-232"""
    pis, means, covs, (px, py) = galaxy_mixture(src.shape, src.u, img)
    return _field.bounding_radius(pis, means, covs, 1e-5, center=(px, py))


def gen_galaxy_psf_image(th, u_s, img, xlim=None, ylim=None, check_overlap=True, unconstrained=True,
                         return_patch=True):
    """unit-flux exp+dev galaxy stamp convolved with the image PSF
    -- celeste_galaxy_conditionals.py:185-214.  Returns (patch, ylim, xlim).

    return_patch=False embeds the patch in a zero frame (the reference accepts the flag and
    ignores it)."""
    th = np.asarray(th, dtype=np.float64)
    patch, (y0, y1), (x0, x1) = _celeste._one_stamp(img, 1, u_s, th[0:4], xlim, ylim)
    if xlim is None and ylim is None:
        # the reference's limits are floats (np.floor / np.ceil, :208-211)
        xlim, ylim = (float(x0), float(x1)), (float(y0), float(y1))
    if patch is None:
        assert (ylim[1] > ylim[0]) and (xlim[1] > xlim[0]), "bad limits."   # util/dists/mog.py:103
    if return_patch:
        return patch, ylim, xlim
    full = np.zeros(img.nelec.shape)
    full[int(ylim[0]):int(ylim[1]), int(xlim[0]):int(xlim[1])] = patch
    return full, (0, full.shape[0]), (0, full.shape[1])
