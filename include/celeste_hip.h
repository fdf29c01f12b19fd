/*
 * celeste_hip.h -- C ABI of the MI355X (gfx950) render + Poisson log-likelihood path.
 *
 * The reference (HIPS/DESI-MCMC, CelestePy) has no FFI layer for this path: the boundary
 * is the module-function API of CelestePy/celeste.py and celeste_galaxy_conditionals.py,
 * plus one native seam (CelestePy/util/like/__init__.py:5-14, which swaps the Cython
 * gmm_like_2d in for the numpy gmm_prob).  Each entry point below names the reference
 * interface it replaces (file:line relative to the reference root).  The Python mirror in
 * desi-mcmc_amd/ binds exactly these symbols with ctypes; INTEGRATION.md shows the stub a
 * CelestePy maintainer would add.
 *
 * Conventions
 *   - plain C: pointers and sizes only; no C++/torch/numpy types cross this boundary;
 *   - every function returns a cel_status; nothing throws across the ABI;
 *     cel_last_error() gives the message of the calling thread's last failure;
 *   - all arithmetic and all image / parameter buffers are IEEE fp64, images row-major [y][x];
 *   - `mem` says where a caller buffer lives: CEL_HOST (pageable or pinned host memory) or
 *     CEL_DEVICE (a HIP device pointer on the context's device, e.g. torch.Tensor.data_ptr());
 *   - device memory for images, sources, bins and partial sums is owned by the library;
 *   - work is enqueued on the context's HIP stream; calls that return values to the host
 *     synchronise that stream, the *_async forms do not;
 *   - threads: a context and the objects created on it belong to one host thread at a time; different contexts may be
 *     driven from different threads at once (tools/dbg/two_threads.py: results equal to the threads run alone);
 *   - there is NO CPU fallback: without a HIP device cel_ctx_create fails with
 *     CEL_ERR_NO_DEVICE.
 */
#ifndef CELESTE_HIP_H
#define CELESTE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CEL_ABI_VERSION 1

typedef enum {
    CEL_OK = 0,
    CEL_ERR_INVALID = 1,   /* bad argument / shape mismatch: the Python mirror raises ValueError
                              (gmm_like_fast.pyx:146-149) */
    CEL_ERR_HIP = 2,       /* a HIP runtime call failed */
    CEL_ERR_NOMEM = 3,
    CEL_ERR_NO_DEVICE = 4  /* no usable gfx950 device */
} cel_status;

enum { CEL_HOST = 0, CEL_DEVICE = 1 };

/* cel_render_field flags */
enum {
    CEL_RENDER_LOGLIK = 1,   /* also reduce sum(nelec*log(lambda) - lambda) per band */
    CEL_RENDER_NO_STORE = 2  /* do not write the model images (log-lik only)        */
};

/* cel_ctx_set_option keys */
enum {
    CEL_OPT_KERNEL = 1,    /* 0 = direct exp per Gaussian-pixel, 1 = row-recurrence (default).
                              Selects the form of every evaluating kernel: field render, stamps,
                              conditional log-likelihoods, photon split, E-step reductions        */
    CEL_OPT_TAIL_LOG = 2,  /* T >= 0: a mixture component is skipped on an image tile when its
                              contribution stays below eps * e^-T everywhere on the part of the
                              tile its source covers (eps = the band's sky level, so the bound is
                              relative to lambda >= eps).  0 = never skip.
                              |d lambda| / lambda <= n_skipped * e^-T: the field render's default is
                              T = 24 (n * 3.8e-11: three orders inside the 1e-6 parity bar for a
                              thousand skips on one pixel; measured on the benchmark field: every
                              pixel within 1e-10 of the oracle, the log-likelihood unchanged in all 16
                              digits, the kernel 13 % shorter than at 32).
                              The per-source kernels whose output has no sky in it (stamps, the
                              conditional log-likelihood, the E-step sums, the photon split's stamps)
                              use T against the source's own smallest value on the tile instead of
                              eps -- the relative error of every pixel stays below n_components * e^-T
                              -- and default to T = 32.  Setting the option sets BOTH thresholds; NaN
                              restores the two defaults; cel_ctx_get_option returns the render's
                              (the per-source kernels': CEL_OPT_TAIL_LOG_SOURCE).  CEL_TAIL_LOG in the
                              environment, when inside [0, 300], is the initial value of both       */
    CEL_OPT_TAIL_LOG_SOURCE = 12, /* the per-source kernels' threshold alone (set / get; NaN: its default, 32) */
    CEL_OPT_INCREMENTAL = 14, /* 1 (default): a cel_render_field whose image set still holds the model image, tile lists and Poisson
                               partials of an earlier state of the SAME catalogue, from which at most 64 rows were changed by
                               cel_sources_set_rows since, renders only the tiles those rows' old and new boxes touch -- each
                               from its complete source list, so every pixel and the log-likelihood are the full render's bit
                               for bit (source preparation and binning still run in full).  The reference's single-source moves
                               (util/infer/mcmc_transitions.py:37-152) evaluate the whole likelihood after changing one source.
                               A render after cel_sources_set, or with nothing changed, renders every tile.  0 = always every tile */
    CEL_OPT_SPLIT_FULL_BOX = 15, /* 0 (default): the photon split gives a source photons only STRICTLY inside its box on the low side, as
                               the reference does (celeste_sample_sources.pyx:50-51: the first row and column of every sample patch
                               stay 0) -- although the renderer adds the source on its whole box (celeste.py:217-219).  1 = on the
                               whole box: the split of the model the renderer draws from, which the exact conditionals of
                               ModelGibbs(conditional="exact") need (DESIGN Q20: with the reference's rule a big galaxy's first box
                               row and column -- up to 10^-4 of its photons -- are modelled but never attributed).  The stamp-mass
                               short cut of CEL_OPT_SPLIT_REUSE = 2 is not taken then */
    CEL_OPT_SLICE_FUSE = 16, /* 0 (default): a round of cel_slice_locations is three launches (likelihoods, chain step).  N > 1: a round
                                of at most N likelihood blocks is ONE launch when every patch of the call is scored at its photons --
                                the block that finishes a chain's last job of the round consumes the chain's log-likelihoods, names
                                its next point and writes that point's records; 1: every round.  The chains are the same bit for
                                bit either way.  Measured in round 6: no gain at any N, hence off */
    CEL_OPT_TILE_PARTS = 13, /* how many one-wave blocks share a render tile of the general 32 x 64 kernel.  0 (default) = by
                               the frame's size: 4 for at most 512 tiles, 2 for at most 3 072, else 1 -- a frame of few tiles
                               (one rank's row strip of a field cut 8 ways, a 51 x 51 real field) finishes when its heaviest
                               tile does; with PARTS blocks per tile each takes every PARTS-th source of the tile's list and the
                               last to finish adds their accumulators in part order.  1 / 2 / 4 = always that many.  Values
                               agree to rounding between settings and are reproducible bit for bit under each */
    CEL_OPT_PROFILE = 3,   /* 1 = bracket every kernel launch with HIP events; 2 = the evaluating kernels only (render,
                              conditional likelihoods, split, mass, E-step: not the prep / binning / reduction launches
                              around a render -- an event pair costs the host ~10 us per launch); 3 = as 2 on a SAMPLE of
                              the launches, every fourth of a kernel (for steps of a few tens of microseconds)     */
    CEL_OPT_TILE_ORDER = 4,/* launch order of the render tiles; never changes results.  0 = index order,
                              1 (default) = heaviest first by the durations the tiles had in the previous
                              render of the same number of sources (the binning pass's estimate when there
                              is none), 2 = heaviest first by the estimate only                         */
    CEL_OPT_TILE_ROWS = 5, /* rows per render tile, 32 (default) or 64; read by cel_images_create  */
    CEL_OPT_TILE_TIMING = 6,/* diagnostic: 1 = k_render stamps every tile's start/end wall clock   */
    CEL_OPT_TILE_LAYOUT = 7,/* render tile geometry, read by cel_images_create:
                               0 = 64 columns x TILE_ROWS rows, one lane per column;
                               1 (default) = 32 columns x 64 rows, two component groups per column;
                               2 = 16 columns x 128 rows, four component groups per column (fewer
                               recurrence seeds, more per-tile set-up: measured 4 % slower than 1
                               on the benchmark field, kept for fields of tall narrow boxes)      */
    CEL_OPT_PHOTON_LISTS = 9, /* how the conditional likelihoods read a device-resident photon split: 0 (default) = per
                               patch, at the pixels that hold a photon or densely, whichever the layout pass estimates
                               cheaper; 1 = every patch at its photons; 2 = never (no lists are built).  The values
                               agree to rounding; set it before cel_photon_split */
    CEL_OPT_STAR_TILES = 10,/* which kernel renders the field of a catalogue WITHOUT galaxies (known when the types
                               came from host memory): 0 = the general one (k_render_hw); 1 (default) = k_render_stars
                               -- the same 32 x 64 tiles taken in two column halves, 8 KB of accumulator instead of 16,
                               three waves per SIMD instead of two -- when the frame has more than 2048 tiles (the
                               general kernel's wave slots; fewer do not fill the extra waves), and for a catalogue of at
                               most 4096 stars on at most 2048 tiles (BASELINE configs[1]) ONE launch that prepares,
                               bins, renders and reduces (k_small_stars: every quarter of a tile's columns a wave of
                               its own, no tile lists); 2 = k_render_stars at any size; 3 = rule 1 without the
                               one-launch path.  Tile layout 1 and the row-recurrence only.  The two kernels add a pixel's stars in different orders: values
                               agree to rounding (1e-15), and which one runs depends on the call's inputs only */
    CEL_OPT_SPLIT_REUSE = 11,/* what a resident cel_photon_split takes from work already done.  1 = its totals image (every
                               pixel's rate under the split's strict boxes): when the model image of exactly these sources and
                               sky levels is on the device -- the chain's trace render came last -- it is formed from that image
                               by subtracting the sources' first box row and column (k_strict_totals: 0.35 ms instead of a
                               1.3 ms render at configs[4]).  The two agree to ~1e-10 of a pixel's rate on those pixels
                               (exactly elsewhere): equal in distribution, not photon for photon.  2 (default) = that, and the
                               two kernels also add up every unit stamp they evaluate, so that a cel_stamp_mass[_begin] of the
                               same catalogue right after the split reads the masses off those sums instead of evaluating
                               every stamp again (1.7 ms at configs[4]); galaxies fainter than a sixteenth of a sky pixel (counts <
                               eps / 16; stars: eps / 1024) still go through the mass kernel.  The sums carry the split's drop rule:
                               masses agree with the mass kernel's to 1e-11 eps / counts at most for a galaxy, 1.5e-13 eps / counts for
                               a star (measured; 1e-14 for a bright source): 1.6e-10 at the thresholds.  0 = totals always rendered from scratch, masses
                               always by the mass kernel */
    CEL_OPT_DEBUG = 8       /* diagnostics.  The shipped library accepts two result-preserving bits: 64 = the E-step
                               takes its per-source form, 128 = CEL_OPT_TILE_TIMING's third word carries the
                               row-waste counters of tools/row_waste.py.  The timing-only ABLATION bits (render:
                               1 = skip the star walk, 2 = the star seeds + walk, 4 = the epilogue's log, 8 =
                               everything after the tile header, 16 = the epilogue's global loads / stores, 32 =
                               every source; photon split: 1 = no draws, 2 = every draw 1, 4 = no stamp walk) exist
                               only in a -DCEL_ABLATE build (`make -C desi-mcmc_amd/csrc ablate`, loaded by
                               tools/ablate_render.py); the shipped library refuses them: CEL_ERR_INVALID */
};

/* kernels reported by cel_profile_get */
enum {
    CEL_K_PREP = 0, CEL_K_BIN = 1, CEL_K_RENDER = 2, CEL_K_REDUCE = 3, CEL_K_STAMPS = 4,
    CEL_K_GMM = 5,
    CEL_K_PATCH_LL = 6,     /* k_patch_ll[_hw]: cel_patch_loglik[_multi] and every round of cel_slice_locations */
    CEL_K_SPLIT = 7,        /* k_photon_split[_hw] */
    CEL_K_MASS = 8,         /* cel_stamp_mass */
    CEL_K_ESTEP = 9,        /* cel_estep_stats */
    CEL_K_RENDER_STARS = 10,/* k_render_stars: the field render of a catalogue without galaxies (CEL_OPT_STAR_TILES) */
    CEL_K_SMALL_STARS = 11,/* k_small_stars: a small star field's whole step in one launch (CEL_OPT_STAR_TILES = 1) */
    CEL_K_TOTALS = 12,     /* k_strict_totals: the photon split's totals image from the model image on the device (CEL_OPT_SPLIT_REUSE) */
    CEL_K_COUNT = 13
};

typedef struct cel_ctx cel_ctx;
typedef struct cel_images cel_images;
typedef struct cel_sources cel_sources;

/* One band image's parameters = the FitsImage fields the path reads
 * (CelestePy/fits_image.py:85-155).  37 doubles, no padding. */
typedef struct {
    double eps;        /* epsilon = SKY*GAIN                     fits_image.py:113 */
    double kappa;      /* GAIN                                   fits_image.py:112 */
    double calib;      /* CALIB, nmgy per count                  fits_image.py:116 */
    double w[3];       /* PSF mixture weights                    fits_image.py:129 */
    double mu[6];      /* PSF means, (x,y) per component         fits_image.py:130 */
    double cov[12];    /* PSF covariances, 2x2 per component     fits_image.py:135-137 */
    double rho[2];     /* CRPIX - 1                              fits_image.py:99  */
    double phi[2];     /* CRVAL                                  fits_image.py:100 */
    double ups[4];     /* CD matrix                              fits_image.py:101 */
    double ups_inv[4]; /* inv(CD)                                fits_image.py:103 */
    double R;          /* star bounding radius (error 1e-3)      fits_image.py:151-155;
                          <= 0: computed by the library with cel_bounding_radius */
} cel_band;

/* ---- library / context ------------------------------------------------------------- */
int cel_abi_version(void);
const char *cel_last_error(void);
int cel_device_count(int *n);

/* device: HIP ordinal.  stream: a hipStream_t to enqueue on (NULL = the library creates one). */
int cel_ctx_create(int device, void *stream, cel_ctx **out);
/* fails with CEL_ERR_INVALID (and destroys nothing) while image sets or source sets created on the context are alive: their own
 * destructors use the context's stream */
int cel_ctx_destroy(cel_ctx *ctx);
int cel_ctx_set_stream(cel_ctx *ctx, void *stream);
int cel_ctx_synchronize(cel_ctx *ctx);
int cel_ctx_set_option(cel_ctx *ctx, int key, double value);
int cel_ctx_get_option(cel_ctx *ctx, int key, double *value);

/* ---- images: B bands of one H x W field -------------------------------------------- */
/* Replaces constructing B FitsImage objects (fits_image.py:48-155) as far as the path reads them. */
int cel_images_create(cel_ctx *ctx, int B, int H, int W, const cel_band *bands, cel_images **out);
int cel_images_destroy(cel_images *img);
/* nelec: B*H*W observed electron counts, FitsImage.nelec (fits_image.py:86-93); stays on device.  The call also reduces the
 * image's range on the device (one pass, synchronous): images within 0 ... 65 535 let cel_photon_split keep its photons-left
 * plane in 16 bits (a seventh wave per CU); any other image takes the 32-bit instantiation, with the same draws */
int cel_images_set_nelec(cel_images *img, const double *nelec, int mem);
/* Gibbs resamples the sky level (models.py:156-160) */
int cel_images_set_epsilon(cel_images *img, int band, double eps);
/* Declare that this image set holds rows [y0, y0 + H) of a full_H-row frame (row-strip partition
 * of one field across GPUs).  Source boxes and the overlap test are formed against
 * the full frame (celeste.py:130-140) and then cut to the window, so the strips tile the frame's
 * model image exactly.  WCS (rho) stays that of the full frame. */
int cel_images_set_window(cel_images *img, int y0, int full_H);
/* The rows [y0, y1) of THIS image set (window-relative) that it OWNS; default: every row.  cel_photon_split's noise sums count
 * the left-over photons of these rows only, and cel_render_field's log-likelihood adds the Poisson terms of these rows only
 * (they must then begin and end on render-tile rows, 64 in the default layout; the model image is rendered on every row).  One Gibbs chain partitioned over GPUs by row strips ("the photon split ... uses the same spatial
 * partition"): a rank's image set holds its strip plus a halo as tall as its own sources' boxes reach, so that their sample
 * patches are complete; the sky photons of the halo rows belong to the neighbours' sums (Field.resample_photons,
 * CelestePy/models.py:155-160, needs the frame's total: the ranks all-reduce their strips' sums). */
int cel_images_set_noise_rows(cel_images *img, int y0, int y1);
int cel_images_get_band(cel_images *img, int band, cel_band *out);
/* copy the last rendered model images (B*H*W) out */
int cel_images_get_lambda(cel_images *img, double *out, int mem);
/* raw device pointers of the library-owned B*H*W buffers (for zero-copy consumers).  Asking for `nelec` tells the library that
 * the caller may write the observed image in place: it stops assuming the range cel_images_set_nelec found (the photon split's
 * 16-bit photons-left plane) until the next cel_images_set_nelec, and from then on keeps no per-tile Poisson sums between
 * renders (CEL_OPT_INCREMENTAL renders every tile when a log-likelihood is asked for).  nelec = NULL asks for neither */
int cel_images_device_ptrs(cel_images *img, void **nelec, void **lambda);
/* device pointer of the B per-band log-likelihoods of the LAST render with CEL_RENDER_LOGLIK (doubles, valid until the next
 * render, photon split or E-step call on this image set -- they reduce into the same buffer; the stream has been synchronised
 * when the render returned): what a multi-GPU caller hands to
 * its all-reduce without a trip through host memory (the one collective of the path). */
int cel_images_loglik_device(cel_images *img, void **ll_band);

/* ---- sources ------------------------------------------------------------------------ */
/* Replaces a python list of SrcParams (celeste_src.py:57-94) as far as the path reads them:
 *   type[s]   0 star, 1 galaxy (SrcParams.a);
 *             2 = a galaxy on the older per-profile route, gen_galaxy_prof_psf_image
 *             (celeste_galaxy_conditionals.py:134-182): shape[s] = theta, W00, W01, W11 with
 *             W = R R^T as the caller formed it (:151, R from the constant img.Ups_n, :33);
 *             theta = 1 / 0 is the 'exp' / 'dev' profile alone; its box is the int() box of
 *             :166-167 from the bounding radius (ERROR 1e-5) of the components that carry weight
 *   radec[s]  (ra, dec) degrees (SrcParams.u)
 *   counts[s*B + b] expected photons of source s in band b -- the multiplier that
 *             gen_src_image applies (celeste.py:35-62; the caller picks the flux convention)
 *   shape[s]  theta, sigma (arcsec), phi (DEGREES, celeste_galaxy_conditionals.py:97), rho */
int cel_sources_create(cel_ctx *ctx, int64_t capacity, int B, cel_sources **out);
int cel_sources_destroy(cel_sources *src);
int cel_sources_set(cel_sources *src, int64_t S, const int32_t *type, const double *radec,
                    const double *counts, const double *shape, int mem);
/* Replace n rows of the catalogue (host arrays, packed row-major as above; idx[i] = the row that packed row i goes to).
 * What a caller that changed ONE source between two evaluations uploads -- the slice steps and RJ moves of
 * CelestePy/util/infer/mcmc_transitions.py:37-152 call celeste_likelihood(list of SrcParams) after every such change. */
int cel_sources_set_rows(cel_sources *src, int64_t n, const int32_t *idx, const int32_t *type, const double *radec,
                         const double *counts, const double *shape);

/* ---- the hot path --------------------------------------------------------------------- */
/* gen_model_image (celeste.py:203-219) for every band + celeste_likelihood /
 * celeste_likelihood_multi_image (celeste.py:237-252), fused.  Galaxies are accumulated by
 * their own patches (models.py:88-108 semantics).
 *   lambda[b] = eps_b + sum_s counts[s][b] * unit_stamp(s, b)     (kept on device)
 *   ll_band[b] = sum_{y,x} nelec*log(lambda) - lambda              (host, B doubles, may be NULL)
 *   ll_total   = sum_b ll_band[b]                                  (host, may be NULL)
 * Synchronises the stream when ll_band or ll_total is non-NULL. */
int cel_render_field(cel_images *img, cel_sources *src, int flags, double *ll_band, double *ll_total);
/* work counters of the last cel_render_field: n_srcpix = sum of box areas (source-pixel
 * evaluations), n_gauss = sum of K*area, n_tile_entries = length of the tile lists */
int cel_field_stats(cel_images *img, double *n_srcpix, double *n_gauss, double *n_tile_entries);
/* diagnostic (CEL_OPT_TILE_TIMING): per render TILE i in [band][tile row][tile column] order -- whatever position the tile
 * order gave its block in the launch -- out[3i] = start, out[3i+1] = end (100 MHz wall clock ticks), out[3i+2] = packed work
 * counters (32x64 / 16x128 layouts: list length | pairs of groups << 12 | kept component-rows << 32; 64x32 layout:
 * (tile index << 32) | list length).  out == NULL: only *n_tiles is returned.  Never enabled in a timed run. */
/* the totals image the last cel_photon_split on the recurrence kernels drew from (every pixel's rate under the split's
 * strict boxes), B*H*W doubles to host memory: what the tests compare between the two ways of forming it (CEL_OPT_SPLIT_REUSE) */
int cel_debug_split_rates(cel_images *img, double *out);
int cel_debug_tile_timing(cel_images *img, uint64_t *out, int64_t *n_tiles);
/* diagnostic: how the last cel_render_field of this image set rendered -- *dirty_tiles = -1: every tile; >= 0: incrementally
 * (CEL_OPT_INCREMENTAL), that many tiles: those touched by the boxes of the rows cel_sources_set_rows changed since the
 * image set's previous render of the same catalogue.  What the tests use to know which path ran; replaces nothing in the
 * reference (its gen_model_image re-renders every source on every call, CelestePy/celeste.py:203-219). */
int cel_debug_last_render(cel_images *img, int64_t *dirty_tiles);

/* ---- stamps --------------------------------------------------------------------------- */
/* gen_point_source_psf_image (celeste.py:114-176) / gen_galaxy_psf_image
 * (celeste_galaxy_conditionals.py:185-214) for every source of `src` in one band.
 * Step 1: boxes[s*4..] = y0, y1, x0, x1 and status[s]: 1 = has a stamp, 0 = the box is empty,
 *         -1 = the reference's overlap test fails and it returns (None, None, None)
 *         (celeste.py:130-135) whatever limits the caller imposes.  Host arrays.
 * Step 2: the caller sizes a packed buffer (offsets[s+1]-offsets[s] = box area, or the area of
 *         the caller-imposed box in boxes_in) and the stamps are written into it, row-major.
 *   scaled: 0 = unit flux, 1 = multiplied by counts[s][band] (gen_src_image_with_fluxes,
 *           celeste.py:84-96)
 *   boxes_in: NULL = each source's own box; else S*4 caller limits (xlim/ylim arguments of
 *           celeste.py:145-152); a source with an empty box is skipped. */
int cel_stamp_boxes(cel_images *img, cel_sources *src, int band, int32_t *boxes, int32_t *status);
int cel_render_stamps(cel_images *img, cel_sources *src, int band, int scaled, const int32_t *boxes_in,
                      const int64_t *offsets, double *out, int mem);

/* ---- per-source conditional log-likelihoods --------------------------------------------- */
/* Source.log_likelihood (CelestePy/sources.py:134-183) and Source.log_likelihood_isolated
 * (:188-237) for a batch of P parameter proposals of ONE source -- what slice sampling / HMC
 * call 10-50 times per source per sweep (sources.py:308-319).
 *   src      P proposals (type, radec, counts[B], shape), usually one coordinate varied
 *   boxes    B*4 ints: the fixed limits y0,y1,x0,x1 of the source's sample patch per band
 *            (samp_img.y0/y1/x0/x1); an empty box = no sample image in that band
 *   offsets  B+1: packed position of each band's patch data; offsets[b+1]-offsets[b] = box area
 *   data     patch values, row-major per band: photons attributed to the source (mode 0) or
 *            the observed nelec patch (mode 1); `mem` says host or device
 *   mode 0   ll = sum_b [ sum_{m>0} log(m) z - counts_b sum(psf weights_b) ],  m = counts_b * stamp
 *            (a star failing the overlap test contributes -counts_b sum(weights_b), :160-163)
 *   mode 1   ll = sum_b [ sum log(m + eps_b) z - sum (m + eps_b) ]
 *   mode 2   ll = sum_b [ sum_{m>0} log(m) z - sum m ]   -- galaxy_source_like
 *            (celeste_galaxy_conditionals.py:15-42) on given limits
 *   mode 4   ll = sum_b [ sum_{z unmasked, m+bg>0} log(m + bg) z - (m + bg) ]   -- poisson_loglike of the observed box
 *            against background + model, the image_like closure of the star <-> galaxy move
 *            (CelestePy/sources.py:6-12,277-291).  Each band's patch data is TWO planes, z then bg
 *            (offsets[b+1]-offsets[b] = 2 x box area); a NaN z marks a masked pixel (invvar == 0); negative counts are data
 *   ll_out   P doubles (host) */
int cel_patch_loglik(cel_images *img, cel_sources *src, const int32_t *boxes, const int64_t *offsets,
                     const double *data, int mem, int mode, double *ll_out);
/* The same for proposals of MANY sources in one launch (a whole sweep of per-source updates):
 * NB patch sets (boxes NB*B*4, offsets NB*B+1, index set*B + band) and owner[p] = the patch set
 * proposal p is scored on (the source it is a proposal for).
 * RESIDENT form: boxes = offsets = data = NULL and NB = the S of the last resident
 * cel_photon_split: mode 0 scores against that split's device-resident sample patches, mode 1
 * against the observed image on the same boxes -- nothing crosses PCIe but the proposals. */
int cel_patch_loglik_multi(cel_images *img, cel_sources *src, const int32_t *owner, int64_t NB,
                           const int32_t *boxes, const int64_t *offsets, const double *data, int mem, int mode,
                           double *ll_out);

/* n standard Gamma(a[i]) variates, element i from its own counter-based streams keyed by (seed, i): the flux conditionals'
 * draws of Source.resample_fluxes (CelestePy/sources.py:341-345) for a whole catalogue (celeste_mcmc.gamma_by_stream is the
 * host form: the same streams and decisions, values equal to rounding).  Host arrays. */
int cel_gamma_streams(cel_ctx *ctx, int64_t n, const double *a, uint64_t seed, double *out);
/* Sum of every source's UNIT stamp over its own box, in every band: mass[s*B + b] -- what
 * Source.resample_fluxes multiplies by kappa/calib for the rate of its Gamma conditional
 * (CelestePy/sources.py:336-339) and celeste_em's sum_fs (celeste_em.py:89).  0 without a stamp.  Host output.
 * Right after a resident cel_photon_split of the same catalogue that took its totals from the model image on the device
 * (CEL_OPT_SPLIT_REUSE = 2) the values are read off the sums that split made (see the option); cel_stamp_mass_ready says
 * whether this call would: a caller that could also ask for a part of the catalogue only (a rank of a dealt chain) then asks
 * for the whole, which costs nothing and gives every rank the numbers the single-rank chain has. */
int cel_stamp_mass(cel_images *img, cel_sources *src, double *mass);
int cel_stamp_mass_ready(cel_images *img, cel_sources *src, int *ready);
/* The same in two halves: _begin queues the kernel and returns, _end waits and copies the S*B values out -- so that the host
 * can draw its Gamma variates while the device sums the stamps (the flux step of a Gibbs sweep) without a second thread.
 * Between the two, calls that touch neither the catalogue, the source records nor the photon split may run on this context
 * (cel_gamma_streams, cel_samples_fetch, cel_images_set_epsilon: what ModelGibbs does there); _end fails with
 * CEL_ERR_INVALID when the source records or the photon split were rebuilt in between. */
int cel_stamp_mass_begin(cel_images *img, cel_sources *src);
int cel_stamp_mass_end(cel_images *img, double *mass);
/* Source.resample_fluxes (CelestePy/sources.py:321-349) for every source of a catalogue whose resident photon split is on the
 * device, without a host round trip (round 6): element (s, L) of flux_new[S][5] (L = band letter u g r i z) =
 *     Gamma(a0 + photons of s in the images of letter L) / (b0 + sum over those images of mass(s, image) * kappa / calib),
 * the Gamma variate from element s * 5 + L's own streams (cel_gamma_streams' sampler), the masses cel_stamp_mass's (short cut
 * and leftovers alike), every operation in the order celeste_mcmc.ModelGibbs.resample_fluxes takes on the host: the same bits.
 * band_letter[b] in 0..4 names image b's letter; calib / kappa per image.  active[s] = 1 when the source has a sample patch in
 * some image: only those sources' expected counts (flux / calib * kappa) are rewritten in the catalogue's device array -- the
 * caller keeps the old flux of the others, as the reference leaves a source without a patch alone (sources.py:243). */
int cel_flux_conditionals(cel_images *img, cel_sources *src, uint64_t seed, double a0, double b0, const int32_t *band_letter,
                          const double *calib, const double *kappa, double *flux_new, int32_t *active);

/* Source.resample_location (CelestePy/sources.py:308-319) for EVERY source of `src` at once: slicesample
 * (CelestePy/util/infer/slicesample.py:89-227) with the options of that call -- component-wise, no stepping
 * out, interval width sigma (degrees) -- run as a lock-step state machine on the device against the
 * resident photon split of exactly these sources (cel_photon_split with offsets = NULL): every round each
 * unfinished chain's next point is scored by the conditional-likelihood kernel (mode 0 of
 * cel_patch_loglik_multi, resident form) and the chains advance; rounds are queued four at a time and
 * four counters (chains running, error bits, evaluations, rounds that had work) cross PCIe per batch.
 * A source without any sample patch is left where it is, and so is one with chain_ids[s] < 0 (when ONE
 * chain is dealt over several GPUs each rank updates only its own sources and the ranks exchange the
 * new locations afterwards).  Random numbers: one SplitMix64 stream per
 * chain keyed by (seed, chain_ids[s] or s), in the reference's draw order -- the same streams and
 * arithmetic as the host engine of the Python mirror (util/infer/slicesample.py), chain for chain.
 *   radec_out  S*2 (host, may be NULL): the new locations; they also REPLACE src's locations on the device
 *   llh_out    S (host, may be NULL): log-likelihood at the new location (NaN for a source left alone)
 *   stats      4 (host, may be NULL): rounds, likelihood evaluations, the ALGORITHMIC HBM bytes those evaluations
 *              read (per evaluation and band: 4 B per pixel of the photon rectangle it walks + one 128-B
 *              record), conditional-likelihood launches */
int cel_slice_locations(cel_images *img, cel_sources *src, const int32_t *chain_ids, double sigma, uint64_t seed,
                        int max_rounds, double *radec_out, double *llh_out, int64_t *stats);

/* slicesample (CelestePy/util/infer/slicesample.py:89-227) with the options cel_slice_locations does not run -- random
 * directions and stepping out by doubling with the `acceptable` test -- over every source's location (param 0: D = 2) or
 * every GALAXY's shape (param 1: D = 4: theta, sigma, phi, rho; the call of CelestePy/celeste_mcmc.py:229-239,
 * slice_sample_skew), as a lock-step state machine on the device against the resident photon split of exactly these
 * sources.  Every round each unfinished chain's next one or two points are scored (mode 0 of cel_patch_loglik_multi,
 * resident form) and the chains advance.  Same per-chain SplitMix64 streams, draw order and arithmetic as the host engine
 * of the Python mirror (util/infer/slicesample.py), chain for chain.
 *   dirs       S*numdir*D (host): the chains' random unit directions, drawn by the caller (the mirror draws them from each
 *              chain's normal stream); NULL = component-wise, the axes in each chain's own random order (numdir ignored)
 *   step_out   0 = none, 1 = doubling (at most max_steps_out doublings per direction)
 *   phi_max    param 1: log-prior galaxy_shape_prior_constrained (celeste_galaxy_conditionals.py:268-275) with this bound on
 *              phi is added to the conditional likelihood; a point outside its support scores -inf and is never rendered.
 *              <= 0: no prior.  (Any other prior: the host engine.)
 *   chain_ids  as cel_slice_locations; a star is left alone under param 1
 *   x_out      S*D (host, may be NULL): the new states; they also REPLACE src's on the device.  llh_out, stats: as
 *              cel_slice_locations (stats[2] = 0) */
int cel_slice_sample(cel_images *img, cel_sources *src, int param, const int32_t *chain_ids, const double *dirs, int numdir,
                     int step_out, int max_steps_out, double sigma, double phi_max, uint64_t seed, int max_rounds,
                     double *x_out, double *llh_out, int64_t *stats);

/* ---- photon split (Gibbs step) ------------------------------------------------------------ */
/* boxes[(b*S+s)*4..] = y0,y1,x0,x1 and status[b*S+s] (as cel_stamp_boxes) for every band at once */
int cel_source_boxes(cel_images *img, cel_sources *src, int32_t *boxes, int32_t *status);
/* sample_source_counts / sample_multinomial (CelestePy/celeste_sample_sources.pyx:61-156) for every
 * band image: each pixel's nelec photons are split among the sources whose patch strictly contains
 * the pixel (:50-51) and the sky, by conditional binomials in source order, sky last.
 *   offsets  S*B+1: packed position of the sample patch of (source s, band b) at index s*B+b (the
 *            patch-set layout of cel_patch_loglik_multi); offsets[i+1]-offsets[i] = that box's area
 *            (cel_source_boxes), 0 without a patch.
 *            NULL = RESIDENT form: the library lays the patches out itself and keeps them in
 *            device memory for cel_patch_loglik_multi / cel_samples_fetch; samp is ignored
 *   samp     packed sample patches (doubles holding integers, as NativePatch.data), zeroed here
 *   noise    B doubles (host): photons attributed to the sky, the reference's noise_sum
 * Random numbers: Philox4x32-10 keyed by (seed; band, pixel, source) -- a result depends only on
 * the seed and the inputs.  The reference's randomkit stream cannot be reproduced on a GPU:
 * parity with it is statistical. */
int cel_photon_split(cel_images *img, cel_sources *src, uint64_t seed, const int64_t *offsets, double *samp,
                     int mem, double *noise);

/* the resident split's bookkeeping: number of sources and of packed patch values */
int cel_samples_info(cel_images *img, int64_t *S, int64_t *total);
/* copy out what the resident split holds; every output may be NULL:
 *   boxes S*B*4 (y0,y1,x0,x1 at s*B+b), offsets S*B+1, data (offsets[S*B] doubles),
 *   sums S*B = photons attributed to source s in band b (the flux Gibbs step's statistic,
 *   sources.py:327-345), reduced on the device */
int cel_samples_fetch(cel_images *img, int32_t *boxes, int64_t *offsets, double *data, double *sums);
/* where the resident split put each source's photons: rects S*B*4 (y0,y1,x0,x1 at s*B+b), the smallest rectangle of the
 * patch of (s, b) that holds every pixel with a photon, all zeros when the patch holds none.  A proposal for source s has
 * a non-zero conditional probability only if its own box covers that rectangle: the model draws a source's photons on its
 * box and nowhere else (celeste.py:217-219 adds the patch at its limits; sources.py:134-183 scores the fixed data patch
 * whatever the proposal's limits -- ModelGibbs(conditional="exact") applies the rule, DESIGN Q20). */
int cel_samples_photon_rects(cel_images *img, int32_t *rects);
/* diagnostic: N independent Binomial(n, p) variates from the split's sampler (stream i = draw i) */
int cel_debug_binomial(cel_ctx *ctx, int64_t n, double p, uint64_t seed, int64_t N, int64_t *out);

/* ---- E-step sufficient statistics -------------------------------------------------------- */
/* What celeste_em (CelestePy/celeste_em.py:38-91) reduces gen_src_prob_layers
 * (celeste.py:222-234) to, without building the (S+1) x H x W responsibility tensor:
 *   xtilde[s*B+b] = sum_pixels nelec * F_s / lambda     (celeste_em.py:85)
 *   mass[s*B+b]   = sum_pixels unit stamp of s in band b (celeste_em.py:89, before the min(1, .))
 *   noise[b]      = sum_pixels nelec * eps / lambda      (celeste_em.py:62, before the / size)
 * Renders lambda for `src` first.  Host outputs; any of them may be NULL. */
int cel_estep_stats(cel_images *img, cel_sources *src, double *xtilde, double *mass, double *noise);

/* ---- generic evaluator ------------------------------------------------------------------ */
/* gmm_like_2d (util/like/gmm_like_fast.pyx:130-176; wrapper util/like/__init__.py:7-11):
 * probs[n] = sum_k ws[k] N(x[n]; mus[k], sigs[k]).  x: N*2, mus: K*2, sigs: K*4 covariances.
 * x and probs follow `mem`; ws/mus/sigs are host arrays. */
int cel_gmm_like_2d(cel_ctx *ctx, const double *x, int64_t N, const double *ws, const double *mus,
                    const double *sigs, int K, double *probs, int mem);

/* mog_loglike (util/dists/mog.py:5-21), the log-domain evaluator MixtureOfGaussians.logpdf /
 * evaluate_grid (:58-60, :102-112) and gen_point_source_psf_image (celeste.py:158) call:
 *   out[n] = logsumexp_k( -(x_n - mu_k)^T icov_k (x_n - mu_k) / 2 + logw[k] )
 * logw[k] = -log(2 pi) - log(dets[k]) / 2 + log(pis[k]), formed by the caller as the reference forms
 * it (a weight <= 0 gives NaN / -inf there; NaN propagates to every output).
 * x and out follow `mem`; means (K*2), icovs (K*4), logw (K) are host arrays. */
int cel_mog_loglike(cel_ctx *ctx, const double *x, int64_t N, const double *means, const double *icovs,
                    const double *logw, int K, double *out, int mem);

/* gen_galaxy_prof_psf_mixture_params (CelestePy/celeste_fast.pyx:100-140), for N sources sharing the
 * PSF and profile arrays: the convolved component tables, PSF-major (index k * J + j):
 *   weights[n][k*J+j] = image_ws[k] * amp[j]
 *   means  [n][k*J+j] = v_s[n] + image_means[k]
 *   covars [n][k*J+j] = image_covars[k] + sigs[j] * W[n]
 * W: N*4 (= R R^T), v_s: N*2, image_*: the PSF (K_psf components), amp/sigs: the profile (J).
 * gen_galaxy_psf_mixture_params (:29-94) is the same call with amp = [thetas[0] * exp_amp,
 * thetas[1] * dev_amp] and sigs = [exp_sigs, dev_sigs].  Host arrays in and out; built on the device. */
int cel_galaxy_mixture_params(cel_ctx *ctx, int64_t N, const double *W, const double *v_s, const double *image_ws,
                              const double *image_means, const double *image_covars, int K_psf, const double *amp,
                              const double *sigs, int J, double *weights, double *means, double *covars);

/* calc_bounding_radius (util/bound/bounding_box.py:9-31); host arithmetic, no device needed */
int cel_bounding_radius(const double *w, const double *mu, const double *cov, int K, double error,
                        const double *center, double *out);

/* ---- measurement -------------------------------------------------------------------------- */
/* With CEL_OPT_PROFILE = 1 (2: see the option) every launch of kernel `k` is bracketed by HIP events on the
 * context's stream; cel_profile_get synchronises and returns the mean duration. */
int cel_profile_reset(cel_ctx *ctx);
int cel_profile_get(cel_ctx *ctx, int kernel, double *mean_ms, int64_t *launches);

#ifdef __cplusplus
}
#endif
#endif /* CELESTE_HIP_H */
