"""In-memory loader for the read-only reference tree (golden generation ONLY).

This module exists solely so that ``make_golden.py`` can execute the reference's
own Python (CelestePy) in *this* container and record its outputs as fixtures.
It is never imported by the product, the tests proper, ``bench.py`` or
``smoke()``: ``/root/reference`` does not exist on the GPU box.

What it does
------------
* A meta-path finder maps ``CelestePy.<m>``, bare ``<m>`` (Python-2 implicit
  relative imports such as ``import celeste_galaxy_conditionals``) and
  ``util.<m>`` to the files under ``/root/reference/CelestePy``.
* Each file's text is read as-is, tabs expanded, run through the stdlib
  ``lib2to3`` fixers **in memory** (the reference is Python 2), compiled and
  executed.  Nothing is written to disk and no reference text enters the repo.
* Third-party packages that are absent from this image are replaced by
  harness-side stand-ins that carry *no* arithmetic of the hot path:
    autograd.numpy -> numpy, autograd.scipy.misc.logsumexp -> scipy.special,
    fitsio -> a 40-line FITS primary-HDU reader, astropy.wcs / tractor.sdss /
    pyprind -> inert stubs, CelestePy.celeste_fast / gmm_like_fast (Cython,
    does not build against numpy 2.x) -> absent, so the reference's own
    ``util/like/__init__.py`` falls back to its numpy ``gmm_prob`` exactly as
    it does for any user without the compiled extension.
"""
import importlib.abc
import importlib.machinery
import importlib.util
import os
import sys
import types
import warnings

import numpy as np

REF_ROOT = "/root/reference"
PKG_ROOT = os.path.join(REF_ROOT, "CelestePy")


# --------------------------------------------------------------------------
# numpy aliases removed in numpy >= 1.24 that the reference still spells
# --------------------------------------------------------------------------
def _patch_numpy():
    for name, val in (("float", float), ("int", int), ("bool", bool), ("object", object)):
        if not hasattr(np, name):
            setattr(np, name, val)
    if not hasattr(np, "row_stack"):
        np.row_stack = np.vstack
    if not hasattr(np, "Inf"):                     # util/infer/slicesample.py:4 (removed in numpy 2.0)
        np.Inf = np.inf


# --------------------------------------------------------------------------
# FITS reader (primary + image extensions, BITPIX -64/-32/16/32, bintables)
# --------------------------------------------------------------------------
def _parse_value(tok):
    tok = tok.strip()
    if tok.startswith("'"):
        return tok.strip("'").strip()
    if tok in ("T", "F"):
        return tok == "T"
    try:
        return int(tok)
    except ValueError:
        pass
    try:
        return float(tok.replace("D", "E"))
    except ValueError:
        return tok


def read_fits_hdus(path):
    """Return a list of (header dict, raw data bytes) for each HDU."""
    with open(path, "rb") as f:
        raw = f.read()
    pos, hdus = 0, []
    while pos < len(raw):
        hdr, done = {}, False
        while not done:
            block = raw[pos:pos + 2880]
            pos += 2880
            for i in range(36):
                card = block[i * 80:(i + 1) * 80].decode("ascii")
                key = card[:8].strip()
                if key == "END":
                    done = True
                    break
                if card[8:10] == "= ":
                    body = card[10:]
                    if body.lstrip().startswith("'"):
                        s = body.lstrip()
                        end = s.find("'", 1)
                        val = s[1:end].strip()
                    else:
                        val = _parse_value(body.split("/")[0])
                    hdr[key] = val
        nax = int(hdr.get("NAXIS", 0))
        n = 1 if nax else 0
        for i in range(nax):
            n *= int(hdr["NAXIS%d" % (i + 1)])
        nbytes = abs(int(hdr["BITPIX"])) // 8 * n + int(hdr.get("PCOUNT", 0))
        hdus.append((hdr, raw[pos:pos + nbytes]))
        pos += (nbytes + 2879) // 2880 * 2880
    return hdus


def fits_image(path, ext=0):
    hdr, data = read_fits_hdus(path)[ext]
    dt = {-64: ">f8", -32: ">f4", 16: ">i2", 32: ">i4"}[int(hdr["BITPIX"])]
    arr = np.frombuffer(data, dtype=dt).reshape(int(hdr["NAXIS2"]), int(hdr["NAXIS1"]))
    return hdr, arr.astype(arr.dtype.newbyteorder("="))


def fits_bintable(path, ext=1):
    """Tiny BINTABLE reader for the stamp catalogues (D and E columns only)."""
    hdr, data = read_fits_hdus(path)[ext]
    nf = int(hdr["TFIELDS"]) if "TFIELDS" in hdr else sum(1 for k in hdr if k.startswith("TTYPE"))
    fields = []
    for i in range(1, nf + 1):
        form = hdr["TFORM%d" % i].strip()
        fields.append((hdr["TTYPE%d" % i].strip(), {"D": ">f8", "E": ">f4", "J": ">i4", "K": ">i8"}[form[-1]]))
    rec = np.frombuffer(data[: int(hdr["NAXIS1"]) * int(hdr["NAXIS2"])], dtype=np.dtype(fields))
    return hdr, rec


# A registry of virtual FITS files so that the reference's FitsImage can be
# pointed at synthetic frames (bigger than the 51x51 stamps) without touching
# the filesystem: name -> (header dict, image array)
VIRTUAL_FITS = {}


class _FitsioHDU:
    def __init__(self, arr):
        self._arr = arr

    def read(self):
        return np.array(self._arr, dtype=np.float64)


class _FitsioFITS:
    def __init__(self, path, *a, **k):
        self._path = path

    def __getitem__(self, ext):
        if self._path in VIRTUAL_FITS:
            return _FitsioHDU(VIRTUAL_FITS[self._path][1])
        return _FitsioHDU(fits_image(self._path, ext)[1])


def _fitsio_read_header(path, ext=0):
    if path in VIRTUAL_FITS:
        return dict(VIRTUAL_FITS[path][0])
    return fits_image(path, ext)[0]


def _fitsio_read(path, ext=1):
    return fits_bintable(path, ext)[1]


# --------------------------------------------------------------------------
# stand-ins for third-party packages missing from the image
# --------------------------------------------------------------------------
def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _install_stubs():
    import scipy.special

    _patch_numpy()
    # autograd -> numpy
    ag = _module("autograd", grad=lambda f, *a, **k: (lambda *x, **y: (_ for _ in ()).throw(
        NotImplementedError("autograd.grad is not available in the golden harness"))))
    ag.numpy = np
    sys.modules["autograd.numpy"] = np
    sys.modules["autograd.numpy.linalg"] = np.linalg
    sys.modules["autograd.numpy.random"] = np.random
    agsp = _module("autograd.scipy")
    agmisc = _module("autograd.scipy.misc", logsumexp=scipy.special.logsumexp)
    agsp.misc = agmisc
    ag.scipy = agsp
    # fitsio -> minimal reader
    _module("fitsio", FITS=_FitsioFITS, read_header=_fitsio_read_header, read=_fitsio_read)

    # astropy.wcs -> inert (FitsImage.use_wcs is never set by the reference)
    class _WCS:
        def __init__(self, *a, **k):
            pass

    ap = _module("astropy")
    ap.wcs = _module("astropy.wcs", WCS=_WCS)
    # tractor.sdss, pyprind -> inert
    tr = _module("tractor")
    tr.sdss = _module("tractor.sdss")

    class _Bar:
        def __init__(self, *a, **k):
            pass

        def update(self, *a, **k):
            pass

    _module("pyprind", ProgBar=_Bar)

    # Cython extension that does not build against numpy 2.x: importable but inert
    def _absent(*a, **k):
        raise NotImplementedError("CelestePy.celeste_fast is a Cython extension that does not "
                                  "build in this image; the golden harness never calls it")

    cf = _module("CelestePy.celeste_fast", gen_galaxy_prof_psf_mixture_params=_absent,
                 gen_galaxy_psf_mixture_params=_absent)
    sys.modules["celeste_fast"] = cf


# --------------------------------------------------------------------------
# the finder/loader
# --------------------------------------------------------------------------
_ABSENT = {"celeste_fast", "gmm_like_fast", "celeste_sample_sources"}  # Cython: do not build here
_refactor_tool = None


def _to_py3(src, filename):
    global _refactor_tool
    if _refactor_tool is None:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            from lib2to3 import refactor
        # fix_import would rewrite py2 implicit-relative imports into `from . import a.b`
        # (invalid); the finders below resolve those names instead.
        fixers = [f for f in refactor.get_fixers_from_package("lib2to3.fixes")
                  if not f.endswith(".fix_import")]
        _refactor_tool = refactor.RefactoringTool(fixers)
    return str(_refactor_tool.refactor_string(src.expandtabs(8) + "\n", filename))


def _resolve(fullname):
    """Map a module name to (canonical name, path, is_package) or None."""
    parts = fullname.split(".")
    if parts[0] == "CelestePy":
        parts = parts[1:]
    if not parts:
        return ("CelestePy", os.path.join(PKG_ROOT, "__init__.py"), True)
    if parts[-1] in _ABSENT:
        return None
    base = os.path.join(PKG_ROOT, *parts)
    if os.path.isdir(base) and os.path.exists(os.path.join(base, "__init__.py")):
        return ("CelestePy." + ".".join(parts), os.path.join(base, "__init__.py"), True)
    if os.path.exists(base + ".py"):
        return ("CelestePy." + ".".join(parts), base + ".py", False)
    return None


class _RefLoader(importlib.abc.Loader):
    def __init__(self, canonical, path, is_pkg):
        self.canonical, self.path, self.is_pkg = canonical, path, is_pkg

    def create_module(self, spec):
        # alias: `celeste`, `util.like`, `CelestePy.util.like` are ONE module
        if self.canonical in sys.modules and self.canonical != spec.name:
            return sys.modules[self.canonical]
        return None

    def exec_module(self, module):
        if getattr(module, "_ref_loaded", False):
            return
        module._ref_loaded = True
        sys.modules.setdefault(self.canonical, module)
        with open(self.path) as f:
            src = f.read()
        if self.canonical == "CelestePy":
            src = ""  # the package __init__ star-imports celeste_em (needs planck data); skip it
        code = compile(_to_py3(src, self.path), self.path, "exec")
        module.__file__ = self.path
        exec(code, module.__dict__)


class _RefFinder(importlib.abc.MetaPathFinder):
    TOP = None

    def find_spec(self, fullname, path=None, target=None):
        if self.TOP is None:
            type(self).TOP = {os.path.splitext(n)[0] for n in os.listdir(PKG_ROOT)} | {"CelestePy"}
        head = fullname.split(".")[0]
        if head not in self.TOP:
            return None
        # relative-looking names inside util subpackages (e.g. `like_list`, `gmm_like`)
        r = _resolve(fullname)
        if r is None and path:
            for p in path:
                cand = os.path.join(p, fullname.split(".")[-1] + ".py")
                if p.startswith(PKG_ROOT) and os.path.exists(cand):
                    rel = os.path.relpath(cand[:-3], PKG_ROOT).replace(os.sep, ".")
                    r = ("CelestePy." + rel, cand, False)
        if r is None:
            return None
        canonical, fpath, is_pkg = r
        spec = importlib.machinery.ModuleSpec(
            fullname, _RefLoader(canonical, fpath, is_pkg), origin=fpath, is_package=is_pkg)
        if is_pkg:
            spec.submodule_search_locations = [os.path.dirname(fpath)]
        return spec


class _SubdirFinder(importlib.abc.MetaPathFinder):
    """Python-2 implicit relative imports inside util/* (`from like_list import *`)."""

    def find_spec(self, fullname, path=None, target=None):
        if "." in fullname:
            return None
        for sub in ("util/like", "util/dists", "util/bound", "util/infer", "util/misc", "util/data"):
            cand = os.path.join(PKG_ROOT, sub, fullname + ".py")
            if os.path.exists(cand) and fullname not in _ABSENT:
                canonical = "CelestePy." + sub.replace("/", ".") + "." + fullname
                return importlib.machinery.ModuleSpec(
                    fullname, _RefLoader(canonical, cand, False), origin=cand)
        return None


_installed = False


def install():
    """Make `import CelestePy...` / `import celeste` resolve to the reference."""
    global _installed
    if _installed:
        return
    if not os.path.isdir(PKG_ROOT):
        raise RuntimeError("reference tree not present: goldens can only be regenerated "
                           "in the build container")
    _install_stubs()
    sys.meta_path.insert(0, _RefFinder())
    sys.meta_path.append(_SubdirFinder())
    _installed = True
