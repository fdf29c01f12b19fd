"""Mirror of the galaxy API of CelestePy/celeste_galaxy_conditionals.py:15-256.

gen_galaxy_psf_image (the current renderer, :185-214) and the older per-profile route --
gen_galaxy_prof_psf_image (:134-182) with its callers galaxy_source_like / galaxy_source_like_grad
(:15-88) -- keep the reference's signatures and return tuples; stamps and likelihoods are
evaluated by the HIP kernels.  The small 2x2 / 42-component host helpers are kept because callers
use them directly (bounds, transformations); they are not the hot loop.

The older route differs from the current one: its shape matrix R comes from the
CONSTANT img.Ups_n (not cd_at_pixel), it convolves ONE profile with the PSF in PSF-major order
(celeste_fast.pyx:100-140), and its box is the int() box of a star with a 1e-5 bound.  In the
reference it cannot run as written: celeste_fast does not build against numpy >= 1.20,
`galaxy_prof_dict[prof].amp / .var` do not exist on MixtureOfGaussians (:155-156; the profile's
normalised amplitudes and variances are meant), and galaxy_source_like multiplies the
(patch, ylim, xlim) tuple by a float (:34-36).  The intended semantics are implemented: both
profiles on the SAME limits (those of the photon patch), lam = image_flux * (theta f_exp +
(1 - theta) f_dev), ll = sum Z log(lam) - sum lam over the pixels the model reaches.
"""
import numpy as np

from . import celeste as _celeste
from . import field as _field
from . import mixture_profiles as mp

BANDS = ['u', 'g', 'r', 'i', 'z']


# galaxy profile objects, each a mixture of gaussians  -- celeste_galaxy_conditionals.py:129-131
def _profile_mog(amp, var):
    from .util.dists.mog import MixtureOfGaussians
    return MixtureOfGaussians(means=np.zeros((len(amp), 2)), covs=var[:, None, None] * np.eye(2)[None], pis=amp)


class _LazyProfiles(dict):
    """galaxy_prof_dict: built on first use (MixtureOfGaussians needs numpy only, but importing this
    module must not cost anything on the hot path)"""

    def __missing__(self, key):
        if key == 'exp':
            self[key] = _profile_mog(mp.exp_amp, mp.exp_var)
        elif key == 'dev':
            self[key] = _profile_mog(mp.dev_amp, mp.dev_var)
        else:
            raise KeyError(key)
        return self[key]

    def has_key(self, key):
        return key in ('exp', 'dev')


galaxy_prof_dict = _LazyProfiles()
_PROF = {'exp': (mp.exp_amp, mp.exp_var, 1.0), 'dev': (mp.dev_amp, mp.dev_var, 0.0)}


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


# ---- the older per-profile route: celeste_galaxy_conditionals.py:134-182, 235-256, 15-88 ------------
def _prof_shape(theta, R):
    R = np.asarray(R, dtype=np.float64).reshape(2, 2)
    W = np.dot(R, R.T)                                     # :151
    return [theta, W[0, 0], W[0, 1], W[1, 1]]


def gen_galaxy_prof_psf_image(prof_type, R, u, img, return_patch=True, xlim=None, ylim=None):
    """unit-flux stamp of ONE profile ('exp' or 'dev') with shape matrix R, convolved with the image
    PSF  -- celeste_galaxy_conditionals.py:134-182.  Returns (patch, (y0, y1), (x0, x1)), or the
    patch embedded in a zero frame with the frame's limits when return_patch is False (:178-182)."""
    assert galaxy_prof_dict.has_key(prof_type), "unknown galaxy profile type"
    patch, (y0, y1), (x0, x1) = _celeste._one_stamp(img, 2, u, _prof_shape(_PROF[prof_type][2], R), xlim, ylim)
    if xlim is not None and ylim is not None:
        (y0, y1), (x0, x1) = ylim, xlim
    if patch is None:                                      # empty box: the reference reshapes 0 pixels
        patch = np.zeros((max(int(y1) - int(y0), 0), max(int(x1) - int(x0), 0)))
    if return_patch:
        return patch, (y0, y1), (x0, x1)
    psf_grid = np.zeros(img.nelec.shape)
    psf_grid[int(y0):int(y1), int(x0):int(x1)] = patch
    return psf_grid, (0, psf_grid.shape[0]), (0, psf_grid.shape[1])


def gen_galaxy_prof_psf_image_bound(prof_type, R, u, img, ERROR=.01):
    """radius holding 1 - ERROR of the profile (x) PSF mass  -- celeste_galaxy_conditionals.py:235-256"""
    from . import celeste_fast
    assert galaxy_prof_dict.has_key(prof_type), "unknown galaxy profile type"
    v_s = img.equa2pixel(u)
    R = np.asarray(R, dtype=np.float64).reshape(2, 2)
    amp, var, _ = _PROF[prof_type]
    weights, means, covars = celeste_fast.gen_galaxy_prof_psf_mixture_params(
        W=np.dot(R, R.T), v_s=v_s, image_ws=img.weights, image_means=img.means, image_covars=img.covars,
        gal_prof_amp=amp, gal_prof_sigs=var)
    return _field.bounding_radius(weights, means, covars, ERROR, center=v_s)


def _patch_limits(Z, img, lims):
    """limits (y0, y1, x0, x1) of one photon array: a SamplePatch-like object carries its own; a
    full-frame array is the frame; anything else needs explicit limits"""
    if hasattr(Z, "y0"):
        return (int(Z.y0), int(Z.y1), int(Z.x0), int(Z.x1)), np.asarray(Z.data, dtype=np.float64)
    Z = np.asarray(Z, dtype=np.float64)
    if lims is not None:
        (y0, y1), (x0, x1) = lims
        box = (int(y0), int(y1), int(x0), int(x1))
    elif Z.shape == img.nelec.shape:
        box = (0, Z.shape[0], 0, Z.shape[1])
    else:
        raise ValueError("galaxy_source_like: a photon patch smaller than the image needs its limits "
                         "(pass limits=[(ylim, xlim), ...] or objects with y0, y1, x0, x1)")
    if Z.shape != (box[1] - box[0], box[3] - box[2]):
        raise ValueError("galaxy_source_like: photon patch shape %s does not match its limits" % (Z.shape,))
    return box, Z


def _source_like_batch(ths, Z_s, images, limits=None):
    """galaxy_source_like for P parameter vectors at once (one device launch per image) -> ll (P,)"""
    ths = np.atleast_2d(np.asarray(ths, dtype=np.float64))
    P = ths.shape[0]
    ll = np.zeros(P)
    typ = np.full(P, 2, dtype=np.int32)
    for n, img in enumerate(images):
        box, Z = _patch_limits(Z_s[n], img, None if limits is None else limits[n])
        iset, pos = _celeste._image_subset((img,))
        k = pos[0]
        shapes = np.zeros((P, 4))
        for p in range(P):
            theta_s, sig_s, phi_s, rho_s = ths[p, 0:4]
            R_s = gen_galaxy_transformation(sig_s, rho_s, phi_s, img.Ups_n)       # :33 (constant CD)
            shapes[p] = _prof_shape(theta_s, R_s)
        counts = np.zeros((P, iset.B))
        counts[:, k] = (ths[:, 6 + BANDS.index(img.band)] / img.calib) * img.kappa   # :39
        boxes = np.zeros((iset.B, 4), dtype=np.int32)
        boxes[k] = box
        patches = [None] * iset.B
        patches[k] = Z
        sset = iset._sources(typ, ths[:, 4:6], counts, shapes)
        ll += iset.patch_loglik(sset, boxes, patches, mode=2)                      # :40-41
    return ll


def galaxy_source_like(th, Z_s, images, check_overlap=True, unconstrained=True, limits=None):
    """log probability of galaxy-specific photons Z_s given th = [theta, sigma, phi, rho, ra, dec,
    b_u, b_g, b_r, b_i, b_z]  -- celeste_galaxy_conditionals.py:15-42 (intended semantics, see the
    module docstring).  Z_s[n]: a full-frame array, a patch with `limits[n] = (ylim, xlim)`, or an
    object with data / y0 / y1 / x0 / x1 (a sample patch)."""
    return float(_source_like_batch(np.asarray(th, dtype=np.float64)[None, :], Z_s, images, limits)[0])


def galaxy_source_like_grad(th, Z_s, images, check_overlap=True, unconstrained=False, limits=None):
    """gradient of galaxy_source_like in th  -- celeste_galaxy_conditionals.py:44-88: analytic in theta
    and the fluxes, central differences (step 1e-5) in sigma, phi, rho, ra, dec.  The analytic terms
    are the reference's as written (:63-67: they use the flux in nanomaggies where the image flux in
    counts belongs; documented quirk).  The ten finite-difference likelihoods of an
    image are ONE device launch."""
    th = np.asarray(th, dtype=np.float64)
    theta_s, sig_s, phi_s, rho_s = th[0:4]
    u_s = th[4:6]
    bs = dict(zip(BANDS, th[-5:]))
    grad_theta_s = 0.
    grad_bs = dict(zip(BANDS, np.zeros(len(BANDS))))
    for n, img in enumerate(images):
        box, Z = _patch_limits(Z_s[n], img, None if limits is None else limits[n])
        ylim, xlim = (box[0], box[1]), (box[2], box[3])
        R_s = gen_galaxy_transformation(sig_s, rho_s, phi_s, img.Ups_n)
        f_nms_exp, _, _ = gen_galaxy_prof_psf_image('exp', R_s, u_s, img, xlim=xlim, ylim=ylim)
        f_nms_dev, _, _ = gen_galaxy_prof_psf_image('dev', R_s, u_s, img, xlim=xlim, ylim=ylim)
        f_nms = theta_s * f_nms_exp + (1. - theta_s) * f_nms_dev
        f_nms_diff = f_nms_exp - f_nms_dev
        ok = f_nms > 0.
        grad_theta_s += np.sum((Z[ok] / f_nms[ok] - bs[img.band]) * f_nms_diff[ok])      # :63
        grad_bs[img.band] += 1. / bs[img.band] * np.sum(Z) - np.sum(f_nms)              # :66
    numerical_inds = [1, 2, 3, 4, 5]
    ths = np.tile(th, (10, 1))
    for i, th_i in enumerate(numerical_inds):
        ths[2 * i, th_i] += 1e-5
        ths[2 * i + 1, th_i] -= 1e-5
    lls = _source_like_batch(ths, Z_s, images, limits)
    grad_RU = (lls[0::2] - lls[1::2]) / (2. * 1e-5)                                      # :72-80
    return np.concatenate([[grad_theta_s], grad_RU, np.array([grad_bs[b] for b in BANDS])])


def galaxy_shape_prior_constrained(theta, sig, phi, rho, phi_max=180.):
    """log prior of a galaxy's shape, -inf outside theta in (0,1), sig > 0, phi in (0, phi_max), rho in (0,1);
    inside, the unnormalised inverse-gamma(1, 1) log-density of sig^2 -- celeste_galaxy_conditionals.py:268-275
    with util/like/like_list.py:18-27.  Arrays broadcast.  The reference bounds phi by pi (its samplers think in
    radians) while its renderer reads phi in DEGREES (:97, Q7); the default here is the renderer's unit, 180;
    phi_max=np.pi is the literal bound."""
    theta, sig, phi, rho = np.broadcast_arrays(*[np.asarray(v, dtype=np.float64) for v in (theta, sig, phi, rho)])
    inside = (theta > 0.) & (theta < 1.) & (sig > 0.) & (phi > 0.) & (phi < phi_max) & (rho > 0.) & (rho < 1.)
    out = np.full(theta.shape, -np.inf)
    s2 = np.where(inside, sig * sig, 1.0)
    out[inside] = (-2. * np.log(s2) - 1. / s2)[inside]
    return out if out.ndim else float(out)
