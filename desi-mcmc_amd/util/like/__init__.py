"""gmm_like_2d with the wrapper signature of CelestePy/util/like/__init__.py:5-14.

The reference tries its Cython kernel and silently falls back to numpy; here the HIP kernel is
the only implementation and a missing library is an error, not a fallback.
"""
from ... import field as _field


def gmm_like_2d(x, ws, mus, sigs, probs=None, device=0):
    """probs[n] = sum_k ws[k] N(x[n]; mus[k], sigs[k]) -- util/like/gmm_like_fast.pyx:130-176.
    Raises ValueError on shape mismatch like the Cython kernel (:146-149)."""
    return _field.default_context(device).gmm_like_2d(x, ws, mus, sigs, probs)
