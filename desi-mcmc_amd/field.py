"""Device-resident handles over the C ABI: a context, a set of band images, a set of sources.

This is the host-side "fast lane" the reference-API mirror (celeste.py) is built on, and what
bulk callers (MCMC sweeps, bench.py) use directly so that nothing is re-uploaded per call.
"""
import ctypes as C
import weakref

import numpy as np

from . import _lib as L

BAND_KEYS = ("eps", "kappa", "calib", "weights", "means", "covars", "rho", "phi", "ups", "ups_inv", "R")


def pack_band(eps, kappa, calib, weights, means, covars, rho, phi, ups, ups_inv, R=0.0):
    """-> 37 doubles in cel_band order (include/celeste_hip.h)."""
    out = np.zeros(L.BAND_DOUBLES)
    out[0:3] = [eps, kappa, calib]
    out[3:6] = np.asarray(weights, dtype=np.float64).ravel()
    out[6:12] = np.asarray(means, dtype=np.float64).ravel()
    out[12:24] = np.asarray(covars, dtype=np.float64).ravel()
    out[24:26] = np.asarray(rho, dtype=np.float64).ravel()
    out[26:28] = np.asarray(phi, dtype=np.float64).ravel()
    out[28:32] = np.asarray(ups, dtype=np.float64).ravel()
    out[32:36] = np.asarray(ups_inv, dtype=np.float64).ravel()
    out[36] = R
    return out


def pack_bands(rec):
    """dict of per-band stacked arrays (keys BAND_KEYS) -> (B, 37)."""
    B = len(np.atleast_1d(rec["eps"]))
    return np.stack([pack_band(*[np.asarray(rec[k])[b] for k in BAND_KEYS]) for b in range(B)])


def _destroy_child(destroy, handle, ctx):
    """finalizer of an object that lives on a Context: `ctx` rides along so that the Context cannot be collected -- and
    cel_ctx_destroy run -- before this has (objects in one reference cycle are finalized in no particular order, and a child's
    destructor synchronises its context's stream)"""
    destroy(handle)


class Context(object):
    """One HIP device + stream.  `stream` is a raw hipStream_t (int), e.g.
    torch.cuda.current_stream().cuda_stream; None lets the library create its own."""

    def __init__(self, device=0, stream=None):
        self._h = C.c_void_p()
        L.check(L.lib().cel_ctx_create(int(device), C.c_void_p(stream) if stream else None, C.byref(self._h)))
        self.device = int(device)
        self._finalizer = weakref.finalize(self, L.lib().cel_ctx_destroy, self._h)

    def set_stream(self, stream):
        L.check(L.lib().cel_ctx_set_stream(self._h, C.c_void_p(stream) if stream else None))

    def synchronize(self):
        L.check(L.lib().cel_ctx_synchronize(self._h))

    def set_option(self, key, value):
        L.check(L.lib().cel_ctx_set_option(self._h, int(key), float(value)))

    def get_option(self, key):
        v = C.c_double(0.0)
        L.check(L.lib().cel_ctx_get_option(self._h, int(key), C.byref(v)))
        return v.value

    # convenience
    def set_kernel(self, name):
        self.set_option(L.CEL_OPT_KERNEL, {"direct": 0, "recurrence": 1}[name])

    def set_tail_log(self, T):
        """drop threshold T of CEL_OPT_TAIL_LOG (a number sets the field render's AND the per-source kernels'); presets:
        "default" = the library's defaults (24 for the field render, 32 for the per-source kernels), "strict" = 32 for both
        (what the 1e-10 parity tests run at), "fast" = 20 (1e-6 parity only)"""
        T = {"default": float("nan"), "strict": L.TAIL_LOG_STRICT, "fast": L.TAIL_LOG_FAST}.get(T, T)
        self.set_option(L.CEL_OPT_TAIL_LOG, T)

    def profile(self, on=True):
        """HIP-event timing of the kernels: True / 1 = every kernel, 2 = the evaluating kernels only (the small launches
        around a render go unbracketed: an event pair costs the host ~10 us per launch), False / 0 = off.  Resets the sums."""
        self.set_option(L.CEL_OPT_PROFILE, float(int(on)))
        L.check(L.lib().cel_profile_reset(self._h))

    def profile_get(self, kernel):
        ms, n = C.c_double(0.0), C.c_int64(0)
        L.check(L.lib().cel_profile_get(self._h, L.KERNELS[kernel], C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def profile_render(self):
        """(mean ms, launches, kernel name) of the field render since the last reset: k_render_stars when the star-tile
        kernel took the launches (CEL_OPT_STAR_TILES; a catalogue without galaxies), k_small_stars when the one-launch path
        of a small star field did (that kernel is the whole step: prep, binning, render, partial sums), the general kernel otherwise"""
        ms_s, n_s = self.profile_get("render_stars")
        ms_g, n_g = self.profile_get("render")
        ms_f, n_f = self.profile_get("small_stars")
        if n_f > max(n_s, n_g):
            return ms_f, n_f, "k_small_stars"
        if n_s > n_g:
            return ms_s, n_s, "k_render_stars"
        return ms_g, n_g, "k_render_hw"

    def gamma_streams(self, a, seed):
        """standard Gamma(a[i]) variates, element i from its own streams keyed by (seed, i), drawn on the device
        (cel_gamma_streams; celeste_mcmc.gamma_by_stream(a, seed, arange(n)) is the same sampler on the host)"""
        a = L.f64(a).ravel()
        out = np.empty_like(a)
        L.check(L.lib().cel_gamma_streams(self._h, a.shape[0], L.dptr(a), C.c_uint64(int(seed) & (2 ** 64 - 1)), L.dptr(out)))
        return out

    def gmm_like_2d(self, x, ws, mus, sigs, probs=None):
        """probs[n] = sum_k ws[k] N(x[n]; mus[k], sigs[k])  (gmm_like_fast.pyx:130-176)."""
        x, ws, mus, sigs = L.f64(x), L.f64(ws), L.f64(mus), L.f64(sigs)
        if x.ndim != 2 or x.shape[1] != 2:
            raise ValueError("x must be N x 2")
        if mus.shape[0] != sigs.shape[0] or mus.shape[0] != ws.shape[0]:
            raise ValueError("Means, covariances and weights must have same first dimension!")
        if mus.ndim != 2 or sigs.ndim != 3 or mus.shape[1] != sigs.shape[1] or mus.shape[1] != sigs.shape[2] \
                or mus.shape[1] != 2:
            raise ValueError("Means and inverse covariance shapes don't jive!")
        if probs is None:
            probs = np.zeros(x.shape[0], dtype=np.float64)
        if probs.dtype != np.float64 or not probs.flags.c_contiguous or probs.shape != (x.shape[0],):
            raise ValueError("probs must be a C-contiguous float64 buffer of length N")
        L.check(L.lib().cel_gmm_like_2d(self._h, x.ctypes.data, x.shape[0], L.dptr(ws), L.dptr(mus), L.dptr(sigs),
                                        int(ws.shape[0]), probs.ctypes.data, L.CEL_HOST))
        return probs


_default_ctx = {}


def default_context(device=0):
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
    return _default_ctx[device]


class SourceSet(object):
    """S sources on the device: type, (ra,dec), counts[S,B], shape[S,4]."""

    def __init__(self, ctx, capacity, B):
        self.ctx, self.B, self.capacity = ctx, int(B), int(capacity)
        self._h = C.c_void_p()
        L.check(L.lib().cel_sources_create(ctx._h, int(capacity), int(B), C.byref(self._h)))
        self._finalizer = weakref.finalize(self, _destroy_child, L.lib().cel_sources_destroy, self._h, ctx)
        self.S = 0

    def set(self, typ, radec, counts, shape=None):
        typ = np.ascontiguousarray(typ, dtype=np.int32)
        S = typ.shape[0]
        radec, counts = L.f64(radec).reshape(S, 2), L.f64(counts).reshape(S, self.B)
        shape = np.zeros((S, 4)) if shape is None else L.f64(shape).reshape(S, 4)
        L.check(L.lib().cel_sources_set(self._h, S, typ.ctypes.data, radec.ctypes.data, counts.ctypes.data,
                                        shape.ctypes.data, L.CEL_HOST))
        self.S = S
        return self

    def set_rows(self, rows, typ, radec, counts, shape):
        """replace the rows `rows` of the catalogue on the device (cel_sources_set_rows): what changed since the last upload"""
        rows = np.ascontiguousarray(rows, dtype=np.int32)
        n = rows.shape[0]
        typ = np.ascontiguousarray(typ, dtype=np.int32).reshape(n)
        radec, counts, shape = L.f64(radec).reshape(n, 2), L.f64(counts).reshape(n, self.B), L.f64(shape).reshape(n, 4)
        L.check(L.lib().cel_sources_set_rows(self._h, n, rows.ctypes.data_as(L.c_int32_p), typ.ctypes.data_as(L.c_int32_p),
                                             L.dptr(radec), L.dptr(counts), L.dptr(shape)))
        return self

    def set_device(self, S, typ_ptr, radec_ptr, counts_ptr, shape_ptr):
        """Same, from raw device pointers (e.g. torch tensors' data_ptr())."""
        L.check(L.lib().cel_sources_set(self._h, int(S), C.c_void_p(typ_ptr), C.c_void_p(radec_ptr),
                                        C.c_void_p(counts_ptr), C.c_void_p(shape_ptr), L.CEL_DEVICE))
        self.S = int(S)
        return self


class ImageSet(object):
    """B band images of one H x W field, resident on the device."""

    def __init__(self, ctx, bands, H, W, nelec=None):
        bands = L.f64(bands).reshape(-1, L.BAND_DOUBLES)
        self.ctx, self.B, self.H, self.W = ctx, bands.shape[0], int(H), int(W)
        self._h = C.c_void_p()
        L.check(L.lib().cel_images_create(ctx._h, self.B, self.H, self.W, L.dptr(bands), C.byref(self._h)))
        self._finalizer = weakref.finalize(self, _destroy_child, L.lib().cel_images_destroy, self._h, ctx)
        self._srcs = None
        self.eps = bands[:, 0].copy()          # host mirror of the sky levels on the device
        if nelec is not None:
            self.set_nelec(nelec)

    def close(self):
        """Release the device memory now (otherwise: when the object is collected).  Any later call
        on this object fails with ValueError (null handle)."""
        self._finalizer()
        self._h = C.c_void_p(None)

    def set_nelec(self, nelec):
        nelec = L.f64(nelec)
        if nelec.size != self.B * self.H * self.W:
            raise ValueError("nelec must have B*H*W = %d elements" % (self.B * self.H * self.W))
        L.check(L.lib().cel_images_set_nelec(self._h, nelec.ctypes.data, L.CEL_HOST))

    def set_nelec_device(self, ptr):
        L.check(L.lib().cel_images_set_nelec(self._h, C.c_void_p(ptr), L.CEL_DEVICE))

    def set_window(self, y0, full_H):
        """This set holds rows [y0, y0+H) of a full_H-row frame (row-strip partition, dist.py)."""
        L.check(L.lib().cel_images_set_window(self._h, int(y0), int(full_H)))

    def set_noise_rows(self, y0, y1):
        """rows [y0, y1) of this set (window-relative) that the set OWNS (a strip inside its halo): photon_split's noise sums count
        them, and render(loglik=True) adds the Poisson terms of their tiles only.  For a log-likelihood the rows must begin and
        end on RENDER-tile rows -- 64 in the default 32x64 layout, so dist.StripDeal(align=64) / strip_edges(align=64), not the
        32-row TILE_ROWS the split alone needs; cel_render_field refuses other rows (CEL_ERR_INVALID) rather than add a partial
        tile.  A set with owned rows never takes the one-launch small-star path."""
        L.check(L.lib().cel_images_set_noise_rows(self._h, int(y0), int(y1)))
        self._noise_rows = (int(y0), int(y1))

    def set_epsilon(self, band, eps):
        L.check(L.lib().cel_images_set_epsilon(self._h, int(band), float(eps)))
        self.eps[int(band)] = float(eps)

    def band(self, b):
        out = np.zeros(L.BAND_DOUBLES)
        L.check(L.lib().cel_images_get_band(self._h, int(b), L.dptr(out)))
        return out

    def device_ptrs(self, nelec=True):
        """(device address of the observed pixels, of the model images).  Asking for the observed pixels' address tells the
        library that the caller may write them at any time: it stops assuming their range and keeps no Poisson partials
        between renders (the dirty-tile render of a log-likelihood is off for this set); nelec=False asks for the model
        images' address alone -> (None, address)"""
        a, b = C.c_void_p(), C.c_void_p()
        L.check(L.lib().cel_images_device_ptrs(self._h, C.byref(a) if nelec else None, C.byref(b)))
        return a.value, b.value

    def loglik_device_ptr(self):
        """device address of the B per-band log-likelihoods of the last render(loglik=True): what dist.LoglikReducer
        all-reduces without a trip through host memory"""
        p = C.c_void_p()
        L.check(L.lib().cel_images_loglik_device(self._h, C.byref(p)))
        return p.value

    # ---- the hot path ----
    def _sources(self, typ, radec, counts, shape):
        S = len(typ)
        self._list_state = None             # (celeste._device_sources: which cached list the device copy mirrors)
        if self._srcs is None or self._srcs.capacity < S:
            self._srcs = SourceSet(self.ctx, max(S, 16), self.B)
        return self._srcs.set(typ, radec, counts, shape)

    def render(self, sources, loglik=False, store=True):
        """gen_model_image for all bands (+ fused celeste_likelihood).
        -> (ll_total, ll_band[B]) when loglik else None.  Model images stay on the device;
        fetch with .model_images().  On a set with owned rows (set_noise_rows) the log-likelihood is that of the owned rows'
        tiles, and those rows must lie on render-tile rows (see set_noise_rows)."""
        flags = (L.CEL_RENDER_LOGLIK if loglik else 0) | (0 if store else L.CEL_RENDER_NO_STORE)
        if loglik:
            llb = np.zeros(self.B)
            tot = C.c_double(0.0)
            L.check(L.lib().cel_render_field(self._h, sources._h, flags, L.dptr(llb), C.byref(tot)))
            return tot.value, llb
        L.check(L.lib().cel_render_field(self._h, sources._h, flags, None, None))
        return None

    def model_images(self):
        out = np.empty((self.B, self.H, self.W))
        L.check(L.lib().cel_images_get_lambda(self._h, out.ctypes.data, L.CEL_HOST))
        return out

    def split_rates(self):
        """the totals image the last photon split drew from (B, H, W): diagnostic (cel_debug_split_rates)"""
        out = np.empty((self.B, self.H, self.W))
        L.check(L.lib().cel_debug_split_rates(self._h, L.dptr(out)))
        return out

    def stats(self):
        a, b, c = C.c_double(), C.c_double(), C.c_double()
        L.check(L.lib().cel_field_stats(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return dict(n_srcpix=a.value, n_gauss=b.value, n_tile_entries=c.value)

    def tile_timing(self):
        """diagnostic stamps of the last render under CEL_OPT_TILE_TIMING -> (T, 3) uint64: start, end
        (100 MHz ticks), packed work counters (include/celeste_hip.h: cel_debug_tile_timing)"""
        n = C.c_int64(0)
        L.check(L.lib().cel_debug_tile_timing(self._h, None, C.byref(n)))
        buf = np.zeros((n.value, 3), dtype=np.uint64)
        L.check(L.lib().cel_debug_tile_timing(self._h, buf.ctypes.data, C.byref(n)))
        return buf

    def last_render_dirty_tiles(self):
        """-1 when the last render of this set rendered every tile; otherwise the number of tiles its incremental render
        (CEL_OPT_INCREMENTAL: only the tiles the changed rows' boxes touch) rendered -- diagnostic (cel_debug_last_render)"""
        n = C.c_int64(0)
        L.check(L.lib().cel_debug_last_render(self._h, C.byref(n)))
        return n.value

    def source_boxes(self, sources):
        """(boxes[B,S,4] = y0,y1,x0,x1, status[B,S]) for every band."""
        S = sources.S
        boxes = np.zeros((self.B, S, 4), dtype=np.int32)
        status = np.zeros((self.B, S), dtype=np.int32)
        L.check(L.lib().cel_source_boxes(self._h, sources._h, boxes.ctypes.data_as(L.c_int32_p),
                                         status.ctypes.data_as(L.c_int32_p)))
        return boxes, status

    def photon_split(self, sources, seed):
        """Gibbs photon split (celeste_sample_sources.pyx:61-156) for every band.
        -> (patches[b][s] 2-D arrays or None, boxes[B,S,4], noise_sum[B])"""
        S = sources.S
        boxes, status = self.source_boxes(sources)
        area = np.where(status > 0, (boxes[..., 1] - boxes[..., 0]).astype(np.int64) * (boxes[..., 3] - boxes[..., 2]), 0)
        offs = np.zeros(self.B * S + 1, dtype=np.int64)
        np.cumsum(area.T.ravel(), out=offs[1:])              # source-major: index s*B + b
        flat = np.zeros(max(int(offs[-1]), 1))
        noise = np.zeros(self.B)
        L.check(L.lib().cel_photon_split(self._h, sources._h, C.c_uint64(int(seed) & (2 ** 64 - 1)),
                                         offs.ctypes.data_as(L.c_int64_p), flat.ctypes.data, L.CEL_HOST, L.dptr(noise)))
        out = []
        for b in range(self.B):
            row = []
            for s in range(S):
                i = s * self.B + b
                if status[b, s] > 0:
                    row.append(flat[offs[i]:offs[i + 1]].reshape(boxes[b, s, 1] - boxes[b, s, 0],
                                                                 boxes[b, s, 3] - boxes[b, s, 2]))
                else:
                    row.append(None)
            out.append(row)
        return out, boxes, noise

    # ---- device-resident Gibbs pieces: nothing but proposals and scalars crosses PCIe -------------
    def photon_split_resident(self, sources, seed):
        """The photon split with the sample patches kept in device memory as int32 (1.6 GB at 10 000
        sources x 5 bands x 2048^2 never leave the GPU).  -> noise_sum[B]"""
        noise = np.zeros(self.B)
        L.check(L.lib().cel_photon_split(self._h, sources._h, C.c_uint64(int(seed) & (2 ** 64 - 1)), None, None,
                                         L.CEL_DEVICE, L.dptr(noise)))
        return noise

    def patch_loglik_resident(self, proposals, owner, isolated=False):
        """Conditional log-likelihoods of proposals against the resident sample patches
        (isolated=True: against the observed image on the same boxes).  owner[p] = index of the
        source (of the split) that proposal p belongs to.  -> ll[P]"""
        owner = np.ascontiguousarray(owner, dtype=np.int32)
        if owner.shape != (proposals.S,):
            raise ValueError("owner must have one entry per proposal")
        S, tot = C.c_int64(0), C.c_int64(0)
        L.check(L.lib().cel_samples_info(self._h, C.byref(S), C.byref(tot)))
        out = np.zeros(proposals.S)
        L.check(L.lib().cel_patch_loglik_multi(self._h, proposals._h, owner.ctypes.data_as(L.c_int32_p), S.value,
                                               None, None, None, L.CEL_DEVICE, 1 if isolated else 0, L.dptr(out)))
        return out

    def slice_locations(self, sources, sigma, seed, chain_ids=None, max_rounds=4000):
        """Source.resample_location for every source at once, on the device (cel_slice_locations).
        The sources' locations are updated in place on the device.  -> (radec[S,2], llh[S], dict(rounds, evals))"""
        S = sources.S
        radec, llh = np.zeros((S, 2)), np.zeros(S)
        stats = np.zeros(4, dtype=np.int64)
        ids = None
        if chain_ids is not None:
            ids = np.ascontiguousarray(chain_ids, dtype=np.int32)
            if ids.shape != (S,):
                raise ValueError("chain_ids must have one entry per source")
        try:
            L.check(L.lib().cel_slice_locations(self._h, sources._h, None if ids is None else ids.ctypes.data_as(L.c_int32_p),
                                                float(sigma), C.c_uint64(int(seed) & (2 ** 64 - 1)), int(max_rounds),
                                                L.dptr(radec), L.dptr(llh), stats.ctypes.data_as(L.c_int64_p)))
        except ValueError as e:
            if "Slice sampler" in str(e):
                raise Exception(str(e))          # the sampler's own failures are plain Exceptions in the reference
            raise
        return radec, llh, dict(rounds=int(stats[0]), evals=int(stats[1]), algorithmic_bytes=int(stats[2]), launches=int(stats[3]))

    def slice_sample(self, sources, param, sigma, seed, dirs=None, step_out=True, max_steps_out=1000, phi_max=180.,
                     chain_ids=None, max_rounds=20000):
        """slicesample with random directions / stepping out by doubling for every source's location (param 0) or every
        galaxy's shape (param 1) on the device (cel_slice_sample).  dirs (S, numdir, D) unit directions or None =
        component-wise.  The sampled parameter is updated in place on the device.
        -> (x[S,D], llh[S], dict(rounds, evals))"""
        S = sources.S
        D = 4 if param else 2
        x, llh = np.zeros((S, D)), np.zeros(S)
        stats = np.zeros(4, dtype=np.int64)
        ids = None
        if chain_ids is not None:
            ids = np.ascontiguousarray(chain_ids, dtype=np.int32)
            if ids.shape != (S,):
                raise ValueError("chain_ids must have one entry per source")
        numdir = 0
        if dirs is not None:
            dirs = L.f64(dirs)
            if dirs.ndim != 3 or dirs.shape[0] != S or dirs.shape[2] != D:
                raise ValueError("dirs must be (S, numdir, %d)" % D)
            numdir = dirs.shape[1]
        try:
            L.check(L.lib().cel_slice_sample(self._h, sources._h, int(param), None if ids is None else ids.ctypes.data_as(L.c_int32_p),
                                             None if dirs is None else L.dptr(dirs), int(numdir), 1 if step_out else 0,
                                             int(max_steps_out), float(sigma), float(phi_max), C.c_uint64(int(seed) & (2 ** 64 - 1)),
                                             int(max_rounds), L.dptr(x), L.dptr(llh), stats.ctypes.data_as(L.c_int64_p)))
        except ValueError as e:
            if "Slice sampler" in str(e):
                raise Exception(str(e))
            raise
        return x, llh, dict(rounds=int(stats[0]), evals=int(stats[1]), launches=int(stats[3]))

    def sample_sums(self):
        """photons attributed to every (source, band) by the resident split -> (S, B)"""
        S, tot = C.c_int64(0), C.c_int64(0)
        L.check(L.lib().cel_samples_info(self._h, C.byref(S), C.byref(tot)))
        sums = np.zeros((S.value, self.B))
        L.check(L.lib().cel_samples_fetch(self._h, None, None, None, L.dptr(sums)))
        return sums

    def sample_box_areas(self):
        """pixels in the resident split's patch of every (source, band) -> (S, B) int64 (0: no patch)"""
        S, tot = C.c_int64(0), C.c_int64(0)
        L.check(L.lib().cel_samples_info(self._h, C.byref(S), C.byref(tot)))
        offs = np.zeros(S.value * self.B + 1, dtype=np.int64)
        L.check(L.lib().cel_samples_fetch(self._h, None, offs.ctypes.data_as(L.c_int64_p), None, None))
        return np.diff(offs).reshape(S.value, self.B)

    def fetch_samples(self):
        """host copies of the resident split: (boxes[S,B,4] = y0,y1,x0,x1, offsets[S*B+1], data)"""
        S, tot = C.c_int64(0), C.c_int64(0)
        L.check(L.lib().cel_samples_info(self._h, C.byref(S), C.byref(tot)))
        boxes = np.zeros((S.value, self.B, 4), dtype=np.int32)
        offs = np.zeros(S.value * self.B + 1, dtype=np.int64)
        data = np.zeros(max(tot.value, 1))
        L.check(L.lib().cel_samples_fetch(self._h, boxes.ctypes.data_as(L.c_int32_p), offs.ctypes.data_as(L.c_int64_p),
                                          data.ctypes.data, None))
        return boxes, offs, data[:tot.value]

    def photon_rects(self):
        """(S, B, 4) = y0, y1, x0, x1: the smallest rectangle of each (source, band) patch of the resident split that holds
        all of its photons; zeros for a patch without a photon"""
        S, tot = C.c_int64(0), C.c_int64(0)
        L.check(L.lib().cel_samples_info(self._h, C.byref(S), C.byref(tot)))
        rects = np.zeros((S.value, self.B, 4), dtype=np.int32)
        L.check(L.lib().cel_samples_photon_rects(self._h, rects.ctypes.data_as(L.c_int32_p)))
        return rects

    def stamp_mass(self, sources):
        """sum of every source's unit stamp over its own box -> (S, B)  (sources.py:336-339)"""
        out = np.zeros((sources.S, self.B))
        self._mass_pending = False              # (a call queued earlier and never collected is dropped)
        L.check(L.lib().cel_stamp_mass(self._h, sources._h, L.dptr(out)))
        return out

    def stamp_mass_ready(self, sources):
        """True when stamp_mass(sources) would read the masses off the sums of the photon split that has just run on this
        catalogue (CEL_OPT_SPLIT_REUSE = 2) instead of evaluating the stamps"""
        r = C.c_int(0)
        L.check(L.lib().cel_stamp_mass_ready(self._h, sources._h, C.byref(r)))
        return bool(r.value)

    def stamp_mass_begin(self, sources):
        """queue stamp_mass and return: the device sums the stamps while the host does something else (no other call on
        this context before stamp_mass_end)"""
        L.check(L.lib().cel_stamp_mass_begin(self._h, sources._h))
        self._mass_shape = (sources.S, self.B)
        self._mass_pending = True

    def stamp_mass_end(self):
        out = np.zeros(self._mass_shape)
        self._mass_pending = False
        L.check(L.lib().cel_stamp_mass_end(self._h, L.dptr(out)))
        return out

    def flux_conditionals(self, sources, seed, a0, b0, band_letter, calib, kappa):
        """Source.resample_fluxes for the whole catalogue on the device (cel_flux_conditionals): needs the resident split of
        `sources`; the catalogue's expected counts are rewritten on the device for the sources that have a patch.
        -> (flux_new[S,5], active[S] bool)"""
        S = sources.S
        letter = np.ascontiguousarray(band_letter, dtype=np.int32)
        cal, kap = L.f64(calib), L.f64(kappa)
        if letter.shape != (self.B,) or cal.shape != (self.B,) or kap.shape != (self.B,):
            raise ValueError("band_letter, calib and kappa must have one entry per image")
        new = np.zeros((S, 5))
        act = np.zeros(S, dtype=np.int32)
        L.check(L.lib().cel_flux_conditionals(self._h, sources._h, C.c_uint64(int(seed) & (2 ** 64 - 1)), float(a0), float(b0),
                                              letter.ctypes.data_as(L.c_int32_p), L.dptr(cal), L.dptr(kap), L.dptr(new),
                                              act.ctypes.data_as(L.c_int32_p)))
        return new, act.astype(bool)

    def estep_stats(self, sources):
        """E-step reductions (celeste_em.py:38-91) -> (xtilde[S,B], mass[S,B], noise[B])."""
        S = sources.S
        xt, ms, nz = np.zeros((S, self.B)), np.zeros((S, self.B)), np.zeros(self.B)
        L.check(L.lib().cel_estep_stats(self._h, sources._h, L.dptr(xt), L.dptr(ms), L.dptr(nz)))
        return xt, ms, nz

    def patch_loglik(self, sources, boxes, patches, isolated=False, mode=None):
        """Conditional log-likelihood of each of the P proposals in `sources` on fixed patches.
        boxes: (B,4) int y0,y1,x0,x1 (empty box = band without a sample image);
        patches: list of B arrays (or None) of the box shapes.  -> ll[P]
        (Source.log_likelihood / log_likelihood_isolated, sources.py:134-237)"""
        boxes = np.ascontiguousarray(boxes, dtype=np.int32).reshape(self.B, 4)
        offs = np.zeros(self.B + 1, dtype=np.int64)
        flat = []
        for b in range(self.B):
            y0, y1, x0, x1 = boxes[b]
            n = int(y1 - y0) * int(x1 - x0) if (y1 > y0 and x1 > x0) else 0
            if n:
                p = L.f64(patches[b])
                if p.shape != (y1 - y0, x1 - x0):
                    raise ValueError("band %d: patch shape %s does not match its box" % (b, p.shape))
                flat.append(p.ravel())
            offs[b + 1] = offs[b] + n
        data = np.concatenate(flat) if flat else np.zeros(1)
        out = np.zeros(sources.S)
        L.check(L.lib().cel_patch_loglik(self._h, sources._h, boxes.ctypes.data_as(L.c_int32_p),
                                         offs.ctypes.data_as(L.c_int64_p), data.ctypes.data, L.CEL_HOST,
                                         (1 if isolated else 0) if mode is None else int(mode), L.dptr(out)))
        return out

    def patch_loglik_planes(self, sources, boxes, planes):
        """mode 4 of cel_patch_loglik: sum over bands of sum_{z unmasked, m+bg>0} log(m + bg) z - (m + bg) for each of the
        P proposals in `sources`.  boxes (B,4) y0,y1,x0,x1 (empty: band not scored); planes[b] = (2, ny, nx): the
        observed counts (NaN = masked pixel; negative counts are data) and the background everything else contributes.  -> ll[P]
        (poisson_loglike of sources.py:6-12 as the star <-> galaxy move uses it, :277-291)"""
        boxes = np.ascontiguousarray(boxes, dtype=np.int32).reshape(self.B, 4)
        offs = np.zeros(self.B + 1, dtype=np.int64)
        flat = []
        for b in range(self.B):
            y0, y1, x0, x1 = boxes[b]
            n = int(y1 - y0) * int(x1 - x0) if (y1 > y0 and x1 > x0) else 0
            if n:
                p = L.f64(planes[b])
                if p.shape != (2, y1 - y0, x1 - x0):
                    raise ValueError("band %d: planes of shape %s do not match the box" % (b, p.shape))
                flat.append(p.ravel())
            offs[b + 1] = offs[b] + 2 * n
        data = np.concatenate(flat) if flat else np.zeros(1)
        out = np.zeros(sources.S)
        L.check(L.lib().cel_patch_loglik(self._h, sources._h, boxes.ctypes.data_as(L.c_int32_p),
                                         offs.ctypes.data_as(L.c_int64_p), data.ctypes.data, L.CEL_HOST, 4, L.dptr(out)))
        return out

    def patch_loglik_multi(self, sources, owner, boxes, patches, isolated=False):
        """patch_loglik for proposals of many sources at once.  owner[p]: which patch set proposal p
        is scored on; boxes (NB, B, 4); patches[set][band] arrays or None.  -> ll[P]"""
        boxes = np.ascontiguousarray(boxes, dtype=np.int32).reshape(-1, self.B, 4)
        NB = boxes.shape[0]
        owner = np.ascontiguousarray(owner, dtype=np.int32)
        if owner.shape != (sources.S,):
            raise ValueError("owner must have one entry per proposal")
        offs = np.zeros(NB * self.B + 1, dtype=np.int64)
        flat = []
        for o in range(NB):
            for b in range(self.B):
                y0, y1, x0, x1 = boxes[o, b]
                n = int(y1 - y0) * int(x1 - x0) if (y1 > y0 and x1 > x0) else 0
                if n:
                    p = L.f64(patches[o][b])
                    if p.shape != (y1 - y0, x1 - x0):
                        raise ValueError("set %d band %d: patch shape %s does not match its box" % (o, b, p.shape))
                    flat.append(p.ravel())
                offs[o * self.B + b + 1] = offs[o * self.B + b] + n
        data = np.concatenate(flat) if flat else np.zeros(1)
        out = np.zeros(sources.S)
        L.check(L.lib().cel_patch_loglik_multi(self._h, sources._h, owner.ctypes.data_as(L.c_int32_p), NB,
                                               boxes.ctypes.data_as(L.c_int32_p), offs.ctypes.data_as(L.c_int64_p),
                                               data.ctypes.data, L.CEL_HOST, 1 if isolated else 0, L.dptr(out)))
        return out

    def stamp_boxes(self, sources, band):
        S = sources.S
        boxes = np.zeros((S, 4), dtype=np.int32)
        status = np.zeros(S, dtype=np.int32)
        L.check(L.lib().cel_stamp_boxes(self._h, sources._h, int(band),
                                        boxes.ctypes.data_as(L.c_int32_p), status.ctypes.data_as(L.c_int32_p)))
        return boxes, status

    def stamps(self, sources, band, scaled=False, boxes_in=None):
        """Per-source stamps in one band -> (list of 2-D arrays or None, boxes[S,4] = y0,y1,x0,x1)."""
        S = sources.S
        # the sources' own boxes + status (one k_prep + a 20-byte-per-source copy, shared with the
        # render call below: the library keeps them until the sources change)
        boxes, status = self.stamp_boxes(sources, band)
        if boxes_in is not None:
            boxes = np.ascontiguousarray(boxes_in, dtype=np.int32).reshape(S, 4)
            # a star that fails the reference's overlap test is (None, None, None) whatever limits
            # the caller imposes (celeste.py:130-135): status -1 stays a miss
            status = (((boxes[:, 1] > boxes[:, 0]) & (boxes[:, 3] > boxes[:, 2])) & (status != -1)).astype(np.int32)
        area = np.where(status > 0, (boxes[:, 1] - boxes[:, 0]).astype(np.int64) * (boxes[:, 3] - boxes[:, 2]), 0)
        offs = np.zeros(S + 1, dtype=np.int64)
        np.cumsum(area, out=offs[1:])
        flat = np.zeros(max(int(offs[-1]), 1))
        L.check(L.lib().cel_render_stamps(
            self._h, sources._h, int(band), 1 if scaled else 0,
            boxes.ctypes.data_as(L.c_int32_p) if boxes_in is not None else None,
            offs.ctypes.data_as(L.c_int64_p), flat.ctypes.data, L.CEL_HOST))
        out = []
        for s in range(S):
            if status[s] > 0:
                out.append(flat[offs[s]:offs[s + 1]].reshape(boxes[s, 1] - boxes[s, 0], boxes[s, 3] - boxes[s, 2]))
            else:
                out.append(None)
        return out, boxes


def bounding_radius(weights, means, covars, error, center=(0.0, 0.0)):
    """calc_bounding_radius (util/bound/bounding_box.py:9-31) through the C ABI (host arithmetic)."""
    w, mu, cov, c = L.f64(weights), L.f64(means), L.f64(covars), L.f64(center)
    out = C.c_double(0.0)
    L.check(L.lib().cel_bounding_radius(L.dptr(w), L.dptr(mu), L.dptr(cov), int(w.shape[0]), float(error),
                                        L.dptr(c), C.byref(out)))
    return out.value
