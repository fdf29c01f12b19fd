import os
import sys

import numpy as np
import pytest

# The suite runs at the library's SHIPPING defaults (CEL_OPT_TAIL_LOG: 24 for the field render, 32 for the per-source
# kernels): no environment override.  A test that wants the strict threshold asks for it (ctx.set_tail_log("strict"), the
# `strict_ctx` fixture of the GPU test modules) and says why.
os.environ.pop("CEL_TAIL_LOG", None)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


import contextlib


@contextlib.contextmanager
def tail_log(ctx, preset):
    """run a block at another drop threshold ("strict" = 32 for every kernel: the 1e-10 tolerances; "fast"; a number) and
    put the context back on the library's shipping defaults afterwards"""
    ctx.set_tail_log(preset)
    try:
        yield ctx
    finally:
        ctx.set_tail_log("default")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


@pytest.fixture(scope="session")
def golden():
    return load_golden


def band_slice(rec, b):
    """Per-band record dict -> the same dict restricted to band b (kept 1-long)."""
    keys = ("eps", "kappa", "calib", "weights", "means", "covars", "rho", "phi", "ups", "ups_inv", "R")
    return {k: np.asarray(rec[k])[b:b + 1] for k in keys}


def unpack_ragged(flat, offs, shapes):
    return [flat[offs[i]:offs[i + 1]].reshape(shapes[i]) for i in range(len(shapes))]


BAND_KEYS = ("eps", "kappa", "calib", "weights", "means", "covars", "invcovars", "logdets", "rho", "phi", "ups", "ups_inv", "R")


def real_fields(dirs=None):
    """tests/golden/real_fields.npz as a list of per-field dicts: every real field the reference ships (data/stamps,
    data/stamp_catalog, data/galaxy_stamps, data/real), its FitsImage records, pixels and catalogue, and what the
    reference's own gen_model_image / celeste_likelihood[_multi_image] / gen_point_source_psf_image returned for it."""
    g = load_golden("real_fields.npz")
    out = []
    pix = full = sub = 0
    for fi, name in enumerate(g["names"]):
        H, W = (int(v) for v in g["HW"][fi])
        c0, c1 = int(g["cat_off"][fi]), int(g["cat_off"][fi + 1])
        whole = str(name).split("/")[0] in ("stamps", "real")
        hs, ws = len(range(0, H, 4)), len(range(0, W, 4))
        f = dict(index=fi, name=str(name), dir=str(name).split("/")[0], H=H, W=W,
                 rec={k: g[k][fi] for k in BAND_KEYS},
                 nelec=g["nelec"][pix:pix + 5 * H * W].reshape(5, H, W).astype(np.float64),
                 radec=g["cat_radec"][c0:c1], flux=g["cat_flux"][c0:c1], src_box=g["src_box"][c0:c1], src_none=g["src_none"][c0:c1],
                 ll_band=g["ll_band"][fi], ll=float(g["ll"][fi]), cat0=c0,
                 lam=g["lam_full"][full:full + 5 * H * W].reshape(5, H, W) if whole else None,
                 lam_sub=None if whole else g["lam_sub"][sub:sub + 5 * hs * ws].reshape(5, hs, ws))
        pix += 5 * H * W
        if whole:
            full += 5 * H * W
        else:
            sub += 5 * hs * ws
        if dirs is None or f["dir"] in dirs:
            out.append(f)
    return g, out


def fuzz_seeds(default):
    """the seeds of a fuzz test: `default` of them in the suite; CEL_FUZZ_SEEDS / CEL_FUZZ_FIRST ask for a longer run
    (profiles/r05_fuzz_run.txt)"""
    import os
    first = int(os.environ.get("CEL_FUZZ_FIRST", "0"))
    return range(first, first + int(os.environ.get("CEL_FUZZ_SEEDS", str(default))))
