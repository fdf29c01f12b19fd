"""The convolved-mixture table builders of CelestePy/celeste_fast.pyx (names, argument order and
PSF-major output layout kept), built on the device behind cel_galaxy_mixture_params.

    gen_galaxy_prof_psf_mixture_params   celeste_fast.pyx:100-140
    gen_galaxy_psf_mixture_params        celeste_fast.pyx:29-94
Both return (weights (K,), means (K, 2), covars (K, 2, 2)) with K = K_psf * J and component
index k * J + j (PSF component k outer, profile component j inner; for the two-profile form the
exp components come before the dev components inside every k).
"""
import numpy as np

from . import _lib as L
from . import field as _field


def _params(W, v_s, image_ws, image_means, image_covars, amp, sigs, device=0):
    W, v_s = L.f64(W).reshape(-1, 4), L.f64(v_s).reshape(-1, 2)
    N = W.shape[0]
    if v_s.shape[0] != N:
        raise ValueError("W and v_s must describe the same number of sources")
    image_ws, image_means, image_covars = L.f64(image_ws), L.f64(image_means), L.f64(image_covars)
    Kp = image_ws.shape[0]
    if image_means.shape != (Kp, 2) or image_covars.shape != (Kp, 2, 2):
        raise ValueError("PSF arrays must be (K,), (K, 2), (K, 2, 2)")
    amp, sigs = L.f64(amp), L.f64(sigs)
    J = amp.shape[0]
    if sigs.shape != (J,):
        raise ValueError("profile amplitudes and variances must have the same length")
    K = Kp * J
    weights, means, covars = np.zeros((N, K)), np.zeros((N, K, 2)), np.zeros((N, K, 2, 2))
    ctx = _field.default_context(device)
    L.check(L.lib().cel_galaxy_mixture_params(ctx._h, N, L.dptr(W), L.dptr(v_s), L.dptr(image_ws), L.dptr(image_means),
                                              L.dptr(image_covars), Kp, L.dptr(amp), L.dptr(sigs), J, L.dptr(weights),
                                              L.dptr(means), L.dptr(covars)))
    return weights, means, covars


def gen_galaxy_prof_psf_mixture_params(W, v_s, image_ws, image_means, image_covars, gal_prof_amp, gal_prof_sigs):
    """one profile (x) PSF  -- celeste_fast.pyx:100-140.  W = R R^T (2, 2), v_s = pixel position (2,)"""
    w, m, c = _params(np.asarray(W).reshape(1, 4), np.asarray(v_s).reshape(1, 2), image_ws, image_means, image_covars,
                      gal_prof_amp, gal_prof_sigs)
    return w[0], m[0], c[0]


def gen_galaxy_psf_mixture_params(thetas, W, v_s, image_ws, image_means, image_covars, gal_exp_amp, gal_exp_sigs,
                                  gal_dev_amp, gal_dev_sigs):
    """thetas[0] * exp + thetas[1] * dev, each (x) PSF  -- celeste_fast.pyx:29-94"""
    thetas = np.asarray(thetas, dtype=np.float64)
    amp = np.concatenate([thetas[0] * np.asarray(gal_exp_amp, dtype=np.float64),
                          thetas[1] * np.asarray(gal_dev_amp, dtype=np.float64)])
    sigs = np.concatenate([np.asarray(gal_exp_sigs, dtype=np.float64), np.asarray(gal_dev_sigs, dtype=np.float64)])
    w, m, c = _params(np.asarray(W).reshape(1, 4), np.asarray(v_s).reshape(1, 2), image_ws, image_means, image_covars,
                      amp, sigs)
    return w[0], m[0], c[0]


def gen_galaxy_prof_psf_mixture_params_batch(W, v_s, image_ws, image_means, image_covars, gal_prof_amp, gal_prof_sigs):
    """The same table for N sources in one device call: W (N, 2, 2), v_s (N, 2) -> (N, K), (N, K, 2), (N, K, 2, 2)"""
    return _params(W, v_s, image_ws, image_means, image_covars, gal_prof_amp, gal_prof_sigs)
