"""Mirror of the runnable Gibbs step of CelestePy/celeste_mcmc.py: the photon split.

sample_source_photons_single_image_cython (celeste_mcmc.py:98-150) renders every source's
counts-scaled patch (gen_src_image_with_fluxes) and calls the Cython multinomial split
(celeste_sample_sources.pyx:61-156).  Here both happen in one device pass (cel_photon_split);
the (samp_imgs, noise_sum) return shape is kept.  The rest of celeste_mcmc.py is not runnable in
the reference as written and is not mirrored.

Random numbers: the reference draws from randomkit's MT19937 through numpy's RandomState; the
device uses a counter-based Philox generator.  Results match in distribution, not draw by draw.
"""
import numpy as np

from . import celeste as _celeste
from .sources import SamplePatch


def _flux_counts(src, image):
    return (src.flux_dict[image.band] / image.calib) * image.kappa      # celeste.py:80-81,94


def sample_source_photons_multi_image(imgs, srcs, seed=None):
    """The split for several same-shape images in one pass.
    -> (samp_imgs[n][s] SamplePatch or None, noise_sums[n])"""
    if seed is None:
        seed = np.random.randint(0, 2 ** 31 - 1)
    imgs = tuple(imgs)
    iset = _celeste._image_set(imgs)
    typ, radec, counts, shape = _celeste._source_arrays(srcs, imgs, counts_fn=_flux_counts)
    sset = iset._sources(typ, radec, counts, shape)
    patches, boxes, noise = iset.photon_split(sset, seed)
    out = []
    for n in range(len(imgs)):
        row = []
        for s in range(len(srcs)):
            p = patches[n][s]
            if p is None:
                row.append(None)
            else:
                y0, y1, x0, x1 = boxes[n, s]
                row.append(SamplePatch(p, (y0, y1), (x0, x1)))
        out.append(row)
    return out, noise


def sample_source_photons_single_image_cython(img, srcs, seed=None):
    """Given a single photon-count image and a list of sources, sample source-specific images
    using the Poisson/multinomial representation  -- celeste_mcmc.py:98-150.
    returns (samp_imgs: list of SamplePatch (x0,x1,y0,y1,data) or None, noise_sum)"""
    samp, noise = sample_source_photons_multi_image((img,), srcs, seed)
    return samp[0], noise[0]
