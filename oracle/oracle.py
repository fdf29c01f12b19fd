"""ctypes front-end of the CPU parity oracle (oracle/celeste_oracle.c) + a numpy restatement
of the mixture evaluator.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package never imports this module.  Parity status: pinned
against tests/golden/*.npz (outputs of the reference run in the build container).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libceleste_oracle.so")
BAND_DOUBLES = 37  # 3+3+6+12+2+2+4+4+1, the orc_band layout
K_GAL = 42

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def build(force=False):
    """Compile the C oracle with gcc (a few hundred ms)."""
    src = os.path.join(_HERE, "celeste_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.orc_band_doubles.restype = C.c_int
        assert L.orc_band_doubles() == BAND_DOUBLES
        L.orc_max_threads.restype = C.c_int
        L.orc_bounding_radius.restype = C.c_double
        L.orc_bounding_radius_rsq.restype = C.c_double
        L.orc_galaxy_box.restype = C.c_double
        L.orc_source_patch.restype = C.c_int64
        L.orc_poisson_loglike.restype = C.c_double
        # never more threads than this process may run on (the GPU boxes report every core of the host
        # but grant a share of 16): an oversubscribed OpenMP team turns every barrier into a scheduler wait
        try:
            avail = len(os.sched_getaffinity(0))
        except AttributeError:
            avail = os.cpu_count() or 1
        L.orc_set_threads(C.c_int(max(1, min(L.orc_max_threads(), avail, int(os.environ.get("ORC_MAX_THREADS", "16"))))))
        _lib = L
    return _lib


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


def pack_bands(rec, idx=None):
    """dict of per-band arrays (tests/golden band records) -> (B, 41) float64 in orc_band order."""
    keys = ("eps", "kappa", "calib")
    B = len(np.atleast_1d(rec["eps"]))
    out = np.zeros((B, BAND_DOUBLES))
    for b in range(B):
        out[b, 0:3] = [np.atleast_1d(rec[k])[b] for k in keys]
        out[b, 3:6] = rec["weights"][b]
        out[b, 6:12] = np.asarray(rec["means"][b]).ravel()
        out[b, 12:24] = np.asarray(rec["covars"][b]).ravel()
        out[b, 24:26] = rec["rho"][b]
        out[b, 26:28] = rec["phi"][b]
        out[b, 28:32] = np.asarray(rec["ups"][b]).ravel()
        out[b, 32:36] = np.asarray(rec["ups_inv"][b]).ravel()
        out[b, 36] = np.atleast_1d(rec["R"])[b]
    return out if idx is None else out[idx]


def max_threads():
    return lib().orc_max_threads()


def set_threads(n):
    lib().orc_set_threads(C.c_int(int(n)))


def equa2pixel(band, u):
    b, bp = _d(band)
    u, up = _d(u)
    v = np.zeros(2)
    lib().orc_equa2pixel(bp, up, v.ctypes.data_as(_dp))
    return v


def pixel2equa(band, p):
    b, bp = _d(band)
    p, pp = _d(p)
    u = np.zeros(2)
    lib().orc_pixel2equa(bp, pp, u.ctypes.data_as(_dp))
    return u


def cd_at_pixel(band, x, y):
    b, bp = _d(band)
    cd = np.zeros(4)
    lib().orc_cd_at_pixel(bp, C.c_double(x), C.c_double(y), cd.ctypes.data_as(_dp))
    return cd.reshape(2, 2)


def bounding_radius(w, mu, cov, error, center=(0.0, 0.0), rsq=None):
    w, wp = _d(w)
    mu, mp = _d(mu)
    cov, cp = _d(cov)
    c, ccp = _d(center)
    if rsq is None:
        return lib().orc_bounding_radius(wp, mp, cp, C.c_int(len(w)), C.c_double(error), ccp)
    return lib().orc_bounding_radius_rsq(wp, mp, cp, C.c_int(len(w)), C.c_double(rsq), ccp)


def band_radius(band):
    """fits_image.py:151-155: FitsImage.R = calc_bounding_radius(weights, means, covars, error=1e-3, center=(0, 0)) of a packed
    band record (orc_band layout)"""
    band = np.asarray(band, dtype=np.float64)
    return bounding_radius(band[3:6], band[6:12].reshape(3, 2), band[12:24].reshape(3, 2, 2), 1e-3, center=(0.0, 0.0))


def checked_radius(band, lib_R, rtol=1e-13):
    """The star radius for a band record handed to the oracle: the ORACLE's own (orc_bounding_radius, pinned to the
    reference's calc_bounding_radius by golden radius.npz), after asserting that the library under test derived the same
    number from the same PSF -- so a wrong R in the library cannot move the boxes of library and checker alike."""
    band = np.asarray(band, dtype=np.float64)
    if band[36] != 0.0:      # a radius that came with the record (a caller's own, or the reference's from a golden): the library
        if not abs(float(lib_R) - band[36]) <= rtol * abs(band[36]):     # must hold that number
            raise AssertionError("star radius: the record says %.17g, library holds %.17g" % (band[36], float(lib_R)))
        return float(band[36])
    R = band_radius(band)
    if not abs(float(lib_R) - R) <= rtol * abs(R):
        raise AssertionError("star radius: library %.17g, oracle %.17g" % (float(lib_R), R))
    return R


def gmm_like_2d(x, ws, mus, sigs):
    x, xp = _d(x)
    ws, wp = _d(ws)
    mus, mp = _d(mus)
    sigs, sp = _d(sigs)
    out = np.empty(x.shape[0])
    lib().orc_gmm_like_2d(out.ctypes.data_as(_dp), xp, C.c_int64(x.shape[0]), wp, mp, sp, C.c_int(len(ws)))
    return out


def mog_loglike(x, means, icovs, dets, pis):
    x, xp = _d(x)
    means, mp = _d(means)
    icovs, ip = _d(icovs)
    dets, dp = _d(dets)
    pis, pp = _d(pis)
    out = np.empty(x.shape[0])
    lib().orc_mog_loglike(out.ctypes.data_as(_dp), xp, C.c_int64(x.shape[0]), mp, ip, dp, pp, C.c_int(len(pis)))
    return out


def profile_tables():
    ea, ev, da, dv = np.zeros(6), np.zeros(6), np.zeros(8), np.zeros(8)
    lib().orc_profile_tables(*[a.ctypes.data_as(_dp) for a in (ea, ev, da, dv)])
    return ea, ev, da, dv


def galaxy_tinv(sig, rho, phi, cd):
    cd, cp = _d(cd)
    T = np.zeros(4)
    lib().orc_galaxy_tinv(C.c_double(sig), C.c_double(rho), C.c_double(phi), cp, T.ctypes.data_as(_dp))
    return T.reshape(2, 2)


def galaxy_table(band, th, u):
    b, bp = _d(band)
    th, tp = _d(th)
    u, up = _d(u)
    pis, means, covs = np.zeros(K_GAL), np.zeros((K_GAL, 2)), np.zeros((K_GAL, 2, 2))
    pxy, tinv = np.zeros(2), np.zeros(4)
    lib().orc_galaxy_table(bp, tp, up, *[a.ctypes.data_as(_dp) for a in (pis, means, covs, pxy, tinv)])
    return pis, means, covs, pxy, tinv.reshape(2, 2)


def source_patch(band, H, W, typ, u, shape=(0.5, 1.0, 0.0, 0.5)):
    """Unit-flux patch of one source in one band -> (patch or None, (y0,y1), (x0,x1))."""
    b, bp = _d(band)
    u, up = _d(u)
    sh, sp = _d(shape)
    box = np.zeros(4, dtype=np.int32)
    n = lib().orc_source_patch(bp, C.c_int(H), C.c_int(W), C.c_int(int(typ)), up, sp,
                               box.ctypes.data_as(_ip), None)
    if n <= 0:
        return None, (int(box[0]), int(box[1])), (int(box[2]), int(box[3]))
    patch = np.empty((box[1] - box[0], box[3] - box[2]))
    lib().orc_source_patch(bp, C.c_int(H), C.c_int(W), C.c_int(int(typ)), up, sp,
                           box.ctypes.data_as(_ip), patch.ctypes.data_as(_dp))
    return patch, (int(box[0]), int(box[1])), (int(box[2]), int(box[3]))


def star_box(band, H, W, u):
    b, bp = _d(band)
    u, up = _d(u)
    v = np.zeros(2)
    box = np.zeros(4, dtype=np.int32)
    ok = lib().orc_star_box(bp, C.c_int(H), C.c_int(W), up, v.ctypes.data_as(_dp), box.ctypes.data_as(_ip))
    return bool(ok), v, box


def render_field(bands, H, W, typ, radec, counts, shape, nelec=None):
    """-> (lambda (B,H,W), ll_band (B) or None, stats dict)"""
    bands, bp = _d(bands)
    B = bands.shape[0]
    typ = np.ascontiguousarray(typ, dtype=np.int32)
    radec, rp = _d(radec)
    counts, cp = _d(counts)
    shape, sp = _d(shape)
    S = typ.shape[0]
    assert radec.shape == (S, 2) and counts.shape == (S, B) and shape.shape == (S, 4)
    lam = np.empty((B, H, W))
    ll = np.zeros(B)
    stats = np.zeros(2)
    if nelec is not None:
        nelec, np_ = _d(nelec)
        assert nelec.shape == (B, H, W)
    else:
        np_ = None
    lib().orc_render_field(bp, C.c_int(B), C.c_int(H), C.c_int(W), C.c_int64(S),
                           typ.ctypes.data_as(C.POINTER(C.c_int32)), rp, cp, sp, np_,
                           lam.ctypes.data_as(_dp), ll.ctypes.data_as(_dp), stats.ctypes.data_as(_dp))
    return lam, (ll if nelec is not None else None), dict(n_srcpix=stats[0], n_gauss=stats[1])


def gen_model_image_fullframe(band, H, W, radec, counts_b):
    b, bp = _d(band)
    radec, rp = _d(radec)
    counts_b, cp = _d(counts_b)
    lam = np.empty((H, W))
    lib().orc_gen_model_image_fullframe(bp, C.c_int(H), C.c_int(W), C.c_int64(radec.shape[0]), rp, cp,
                                        lam.ctypes.data_as(_dp))
    return lam


def patch_loglik(band, H, W, typ, u, shape, counts, box, data, mode=0):
    """One image's term of Source.log_likelihood (mode 0) / log_likelihood_isolated (mode 1)."""
    b, bp = _d(band)
    u, up = _d(u)
    sh, sp = _d(shape)
    box = np.ascontiguousarray(box, dtype=np.int32)
    data, dp = _d(data)
    lib().orc_patch_loglik.restype = C.c_double
    return lib().orc_patch_loglik(bp, C.c_int(H), C.c_int(W), C.c_int(int(typ)), up, sp, C.c_double(float(counts)),
                                  box.ctypes.data_as(_ip), dp, C.c_int(int(mode)))


def patch_loglik_terms(band, H, W, typ, u, shape, counts, box, data):
    """mode 0's terms apart -> (sum_{m>0} log(m) z, sum_{m>0} |log(m) z|, counts * sum(psf weights), the worth of one
    subnormal quantum in the photon term where the unit stamp is a subnormal number); type 2 takes
    shape = (theta, W00, W01, W11), the per-profile route's source as the product's ABI holds it"""
    b, bp = _d(band)
    u, up = _d(u)
    sh, sp = _d(shape)
    box = np.ascontiguousarray(box, dtype=np.int32)
    data, dp = _d(data)
    out = np.zeros(4)
    lib().orc_patch_loglik_terms(bp, C.c_int(H), C.c_int(W), C.c_int(int(typ)), up, sp, C.c_double(float(counts)),
                                 box.ctypes.data_as(_ip), dp, out.ctypes.data_as(_dp))
    return out


def galaxy_prof_psf_mixture_params(W, v_s, image_ws, image_means, image_covars, amp, sigs):
    """celeste_fast.pyx:100-140 -> (weights, means, covars), PSF-major"""
    W, Wp = _d(W)
    v_s, vp = _d(v_s)
    iw, iwp = _d(image_ws)
    im, imp = _d(image_means)
    ic, icp = _d(image_covars)
    amp, ap = _d(amp)
    sigs, sp = _d(sigs)
    K = len(iw) * len(amp)
    w, m, c = np.zeros(K), np.zeros((K, 2)), np.zeros((K, 2, 2))
    lib().orc_galaxy_prof_psf_mixture_params(Wp, vp, iwp, imp, icp, C.c_int(len(iw)), ap, sp, C.c_int(len(amp)),
                                             w.ctypes.data_as(_dp), m.ctypes.data_as(_dp), c.ctypes.data_as(_dp))
    return w, m, c


def galaxy_psf_mixture_params(thetas, W, v_s, image_ws, image_means, image_covars, exp_amp, exp_sigs, dev_amp, dev_sigs):
    """celeste_fast.pyx:29-94 -> (weights, means, covars)"""
    th, tp = _d(thetas)
    W, Wp = _d(W)
    v_s, vp = _d(v_s)
    iw, iwp = _d(image_ws)
    im, imp = _d(image_means)
    ic, icp = _d(image_covars)
    ea, eap = _d(exp_amp)
    es, esp = _d(exp_sigs)
    da, dap = _d(dev_amp)
    ds, dsp = _d(dev_sigs)
    K = len(iw) * (len(ea) + len(da))
    w, m, c = np.zeros(K), np.zeros((K, 2)), np.zeros((K, 2, 2))
    lib().orc_galaxy_psf_mixture_params(tp, Wp, vp, iwp, imp, icp, C.c_int(len(iw)), eap, esp, C.c_int(len(ea)),
                                        dap, dsp, C.c_int(len(da)), w.ctypes.data_as(_dp), m.ctypes.data_as(_dp),
                                        c.ctypes.data_as(_dp))
    return w, m, c


def galaxy_prof_psf_image(band, H, W, prof, R, u, lims=None):
    """celeste_galaxy_conditionals.py:134-182 -> (patch or None, (y0,y1), (x0,x1)); prof 'exp' | 'dev';
    lims = (y0, y1, x0, x1) or None (own int() box)"""
    b, bp = _d(band)
    R, Rp = _d(R)
    u, up = _d(u)
    pi = {"exp": 0, "dev": 1}[prof]
    box = np.zeros(4, dtype=np.int32)
    lp = None
    if lims is not None:
        lims = np.ascontiguousarray(lims, dtype=np.int32)
        lp = lims.ctypes.data_as(_ip)
    lib().orc_galaxy_prof_psf_image.restype = C.c_int64
    n = lib().orc_galaxy_prof_psf_image(bp, C.c_int(H), C.c_int(W), C.c_int(pi), Rp, up, lp, box.ctypes.data_as(_ip), None)
    if n <= 0:
        return None, (int(box[0]), int(box[1])), (int(box[2]), int(box[3]))
    patch = np.empty((box[1] - box[0], box[3] - box[2]))
    lib().orc_galaxy_prof_psf_image(bp, C.c_int(H), C.c_int(W), C.c_int(pi), Rp, up, lp, box.ctypes.data_as(_ip),
                                    patch.ctypes.data_as(_dp))
    return patch, (int(box[0]), int(box[1])), (int(box[2]), int(box[3]))


def galaxy_source_like(band, H, W, th, u, image_flux, box, Z):
    """one image's term of celeste_galaxy_conditionals.py:15-42 on the limits box = (y0, y1, x0, x1)"""
    b, bp = _d(band)
    th, tp = _d(th)
    u, up = _d(u)
    box = np.ascontiguousarray(box, dtype=np.int32)
    Z, zp = _d(Z)
    lib().orc_galaxy_source_like.restype = C.c_double
    return lib().orc_galaxy_source_like(bp, C.c_int(H), C.c_int(W), tp, up, C.c_double(float(image_flux)),
                                        box.ctypes.data_as(_ip), zp)


def estep_stats(bands, H, W, typ, radec, counts, shape, nelec):
    """-> (xtilde[S,B], mass[S,B], noise[B]) : celeste_em.py:38-91 reductions"""
    lam, _, _ = render_field(bands, H, W, typ, radec, counts, shape, nelec)
    bands, bp = _d(bands)
    B = bands.shape[0]
    typ = np.ascontiguousarray(typ, dtype=np.int32)
    S = typ.shape[0]
    radec, rp = _d(radec)
    counts, cp = _d(counts)
    shape, sp = _d(shape)
    nelec, np_ = _d(nelec)
    lam, lp = _d(lam)
    xt, ms, nz = np.zeros((S, B)), np.zeros((S, B)), np.zeros(B)
    lib().orc_estep_stats(bp, C.c_int(B), C.c_int(H), C.c_int(W), C.c_int64(S), typ.ctypes.data_as(C.POINTER(C.c_int32)),
                          rp, cp, sp, np_, lp, xt.ctypes.data_as(_dp), ms.ctypes.data_as(_dp), nz.ctypes.data_as(_dp))
    return xt, ms, nz


def poisson_loglike(data, model, mask=None):
    data, dp = _d(data)
    model, mp = _d(model)
    if mask is not None:
        mask = np.ascontiguousarray(mask, dtype=np.uint8)
        mkp = mask.ctypes.data_as(C.POINTER(C.c_uint8))
    else:
        mkp = None
    return lib().orc_poisson_loglike(dp, mp, mkp, C.c_int64(data.size))


# ---------------------------------------------------------------------------
# numpy restatement of the evaluator (util/dists/mog.py:5-21), used to cross-check the C one
# ---------------------------------------------------------------------------
def np_mog_loglike(x, means, icovs, dets, pis):
    from scipy.special import logsumexp
    xx = np.atleast_2d(x)
    centered = xx[:, :, None] - means.T[None, :, :]
    solved = np.einsum("ijk,lji->lki", icovs, centered)
    logprobs = -0.5 * np.sum(solved * centered, axis=1) - np.log(2 * np.pi) - 0.5 * np.log(dets) + np.log(pis)
    return logsumexp(logprobs, axis=1)
