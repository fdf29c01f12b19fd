"""celeste_em: the EM driver of CelestePy/celeste_em.py:17-180 on top of the device reductions.

The reference's E-step materialises gen_src_prob_layers -- an (S+1, H, W) tensor per image -- only
to reduce it to three sums per (source, image); here the sums come straight from the device
(celeste.estep_statistics -> cel_estep_stats), so the loop runs at 10 000 sources x 2048^2 where the
layers would need 335 GB per band.  The M-step is the reference's: the sky level's closed form
(:60-63), a one-dimensional profile likelihood in the temperature maximised with scipy's fmin
(:111-141), the closed-form brightness (:143-151).

Black-body photometry (CelestePy/planck.py) is outside this path: pass a `planck` object with
    photons_per_joule(t, band), lens_area, exposure_duration, sun_wattage, m_per_ly
(the reference's module has exactly these) -- the same object also supplies the render's expected
photons for stars given by temperature (celeste.photons_expected_brightness hook).

Reference slip kept out: the sky update reads `src_probs` -- the LAST image's layers, a leaked loop
variable -- for every image (celeste_em.py:62); each image's own sky responsibility is used here.
"""
import numpy as np

from . import celeste as _celeste


def _expected_brightness(planck):
    def f(t, b, band):             # planck.py:155-158
        lens_watts = planck.lens_area * b * planck.sun_wattage / (planck.m_per_ly ** 2)
        return planck.photons_per_joule(t, band) * lens_watts * planck.exposure_duration
    return f


def celeste_em(srcs, imgs, maxiter=20, debug=False, verbose=True, planck=None):
    """maximizes the log likelihood over the (temperature, brightness) of a fixed set of point sources
    -- celeste_em.py:17-180.  srcs: list of SrcParams with .t / .b; imgs: list of FitsImage.
    Returns (ll_trace, converged)."""
    from scipy.optimize import fmin
    if planck is None:
        raise NotImplementedError("celeste_em needs a planck object (photons_per_joule, lens_area, exposure_duration, "
                                  "sun_wattage, m_per_ly): black-body photometry is outside this path")
    hook = getattr(planck, "photons_expected_brightness", None) or _expected_brightness(planck)
    old_hook = _celeste.photons_expected_brightness
    _celeste.photons_expected_brightness = hook
    say = (lambda msg, level=1: print(msg)) if verbose else (lambda msg, level=1: None)
    detail = verbose > 1
    try:
        prev_ll = _celeste.celeste_likelihood_multi_image(srcs, imgs)
        ll_trace = [prev_ll]
        imgbands = np.array([img.band for img in imgs])
        uniquebands = np.unique(imgbands)
        # lens_watts per unit brightness: what turns photons per joule into expected photons (planck.py:155-158)
        fac = 1. / (planck.lens_area * planck.exposure_duration * planck.sun_wattage / (planck.m_per_ly ** 2))

        def band_efficiency(t):
            """photons per joule of a black body at temperature t in every image's band (celeste_em.py:96-106)"""
            I_ts = np.zeros(len(imgbands))
            for b in uniquebands:
                I_ts[imgbands == b] = planck.photons_per_joule(t, b)
            return I_ts

        say("EM start: log-likelihood %.2f over %d images, %d sources" % (prev_ll, len(imgs), len(srcs)))
        em_iter = -1
        for em_iter in range(maxiter):
            # E-step: the three reductions of the responsibility layers (celeste_em.py:38-58, 85, 89), on the device
            X_all, F_all, Z = _celeste.estep_statistics(srcs, imgs)
            # M-step, sky levels (:60-63): the photons the E-step left to the sky, per pixel
            for i, img in enumerate(imgs):
                before = img.epsilon
                img.epsilon = Z[i] / img.nelec.size
                if detail:
                    say("[%d] image %d sky %.3f -> %.3f" % (em_iter, i, before, img.epsilon))
            # M-step, sources (:113-141): temperature by a 1-D search on the profiled objective, brightness in closed form
            for s in range(len(srcs)):
                X_tildes, sum_fs = X_all[s], F_all[s]

                def temperature_objective(temp):
                    I_ts = band_efficiency(temp)
                    return X_tildes.dot(np.log(I_ts)) - np.log(I_ts.dot(sum_fs)) * X_tildes.sum()

                t_hat = fmin(lambda t: -temperature_objective(np.atleast_1d(t)[0]), srcs[s].t, disp=False)[0]
                b_hat = fac * (1. / band_efficiency(t_hat).dot(sum_fs)) * X_tildes.sum()
                if detail:
                    say("[%d] source %d: T %.1f -> %.1f, brightness %.3g -> %.3g" % (em_iter, s, srcs[s].t, t_hat, srcs[s].b, b_hat))
                srcs[s].t = t_hat
                srcs[s].b = b_hat
            ll = _celeste.celeste_likelihood_multi_image(srcs, imgs)
            ll_trace.append(ll)
            say("[%d] log-likelihood %.2f (%+.3f)" % (em_iter, ll, ll - prev_ll))
            if prev_ll > ll:
                say("[%d] warning: the log-likelihood went DOWN, %.4f -> %.4f" % (em_iter, prev_ll, ll))
            if ll - prev_ll < 1:                                        # the reference's stopping rule (:174-177)
                say("converged after %d of at most %d iterations" % (em_iter + 1, maxiter))
                break
            prev_ll = ll
        return ll_trace, em_iter < maxiter
    finally:
        _celeste.photons_expected_brightness = old_hook
