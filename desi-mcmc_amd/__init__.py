"""MI355X-native render + Poisson log-likelihood path of CelestePy (HIPS/DESI-MCMC).

The package directory is `desi-mcmc_amd/`; import it as `desi_mcmc_amd` (alias module at the
repo root).  Public surface = the reference's module-function API:

    desi_mcmc_amd.celeste                       gen_point_source_psf_image, gen_src_image,
                                                gen_model_image, celeste_likelihood[_multi_image], ...
    desi_mcmc_amd.celeste_galaxy_conditionals   gen_galaxy_psf_image, gen_galaxy_transformation
    desi_mcmc_amd.util.like                     gmm_like_2d
    desi_mcmc_amd.util.bound.bounding_box       calc_bounding_radius
    desi_mcmc_amd.fits_image / celeste_src      FitsImage / SrcParams input records
    desi_mcmc_amd.field                         device-resident Context / ImageSet / SourceSet

All arithmetic runs in hand-written HIP kernels (csrc/celeste_hip.hip) behind the C ABI in
include/celeste_hip.h.  There is no CPU fallback.
"""
from . import _lib  # noqa: F401
from .celeste_src import SrcCatalog, SrcParams  # noqa: F401
from .field import Context, ImageSet, SourceSet, default_context  # noqa: F401
from .fits_image import FitsImage  # noqa: F401

__all__ = ["SrcParams", "SrcCatalog", "FitsImage", "Context", "ImageSet", "SourceSet", "default_context"]
