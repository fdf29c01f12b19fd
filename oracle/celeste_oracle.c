/*
 * celeste_oracle.c -- CPU restatement of CelestePy's render + Poisson log-lik path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle: only tests/,
 * __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may load it.  The
 * product (desi-mcmc_amd/) never links, imports or calls anything in oracle/.
 *
 * Parity status: PINNED.  Every function below is checked in tests/test_oracle.py
 * against fixtures produced by running the reference's own Python in the build
 * container (tests/golden/make_golden.py), including the reference's own
 * known-answer test (CelestePy/test/test_gmm.py:63-105, seed 41, K=42).
 *
 * Each function cites the reference file:line (relative to /root/reference) it
 * restates.  All arithmetic is IEEE double, in the reference's operation order
 * where that order is observable (box edges, log-domain mixture sum).
 *
 * Build: see oracle/Makefile (gcc -O2 -fopenmp -shared).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* One band image's parameters: the FitsImage fields the path reads
 * (CelestePy/fits_image.py:85-155).  Plain doubles, 37 of them, no padding. */
typedef struct {
    double eps;          /* epsilon = SKY*GAIN            fits_image.py:113 */
    double kappa;        /* GAIN                          fits_image.py:112 */
    double calib;        /* CALIB (nmgy per count)        fits_image.py:116 */
    double w[3];         /* PSF weights                   fits_image.py:129 */
    double mu[3][2];     /* PSF means (x, y)              fits_image.py:130 */
    double cov[3][2][2]; /* PSF covariances               fits_image.py:135-137 */
    double rho[2];       /* CRPIX - 1                     fits_image.py:99  */
    double phi[2];       /* CRVAL                         fits_image.py:100 */
    double ups[2][2];    /* CD                            fits_image.py:101 */
    double ups_inv[2][2];/* inv(CD)                       fits_image.py:103 */
    double R;            /* star bounding radius, eps=1e-3 fits_image.py:151-155 */
} orc_band;

int orc_band_doubles(void) { return (int)(sizeof(orc_band) / sizeof(double)); }

int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void orc_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* exp / dev galaxy profile mixtures: Hogg & Lang amplitudes and variances
 * (CelestePy/mixture_profiles.py:9-19; amplitudes are normalised at :13,:19). */
static const double EXP_AMP_RAW[6] = {2.34853813e-03, 3.07995260e-02, 2.23364214e-01,
                                      1.17949102e+00, 4.33873750e+00, 5.99820770e+00};
static const double EXP_VAR[6] = {1.20078965e-03, 8.84526493e-03, 3.91463084e-02,
                                  1.39976817e-01, 4.60962500e-01, 1.50159566e+00};
static const double DEV_AMP_RAW[8] = {4.26347652e-02, 2.40127183e-01, 6.85907632e-01, 1.51937350e+00,
                                      2.83627243e+00, 4.46467501e+00, 5.72440830e+00, 5.60989349e+00};
static const double DEV_VAR[8] = {2.23759216e-04, 1.00220099e-03, 4.18731126e-03, 1.69432589e-02,
                                  6.84850479e-02, 2.87207080e-01, 1.33320254e+00, 8.40215071e+00};

#define K_PSF 3
#define K_EXP 6
#define K_DEV 8
#define K_PROF (K_EXP + K_DEV)
#define K_GAL (K_PROF * K_PSF)

/* numpy's pairwise-free np.sum over 6 / 8 elements is a plain left-to-right sum */
void orc_profile_tables(double *exp_amp, double *exp_var, double *dev_amp, double *dev_var) {
    double s = 0.0;
    for (int i = 0; i < K_EXP; i++) s += EXP_AMP_RAW[i];
    for (int i = 0; i < K_EXP; i++) { exp_amp[i] = EXP_AMP_RAW[i] / s; exp_var[i] = EXP_VAR[i]; }
    s = 0.0;
    for (int i = 0; i < K_DEV; i++) s += DEV_AMP_RAW[i];
    for (int i = 0; i < K_DEV; i++) { dev_amp[i] = DEV_AMP_RAW[i] / s; dev_var[i] = DEV_VAR[i]; }
}

/* ------------------------------------------------------------------ WCS -- */

/* fits_image.py:166-174 equa2pixel (linear WCS; returns [x, y]) */
void orc_equa2pixel(const orc_band *b, const double u[2], double v[2]) {
    double phi1rad = b->phi[1] / 180.0 * M_PI;
    double s0 = (u[0] - b->phi[0]) * cos(phi1rad);
    double s1 = (u[1] - b->phi[1]);
    v[0] = (b->ups_inv[0][0] * s0 + b->ups_inv[0][1] * s1) + b->rho[0];
    v[1] = (b->ups_inv[1][0] * s0 + b->ups_inv[1][1] * s1) + b->rho[1];
}

/* fits_image.py:176-181 pixel2equa */
void orc_pixel2equa(const orc_band *b, const double p[2], double u[2]) {
    double phi1rad = b->phi[1] / 180.0 * M_PI;
    double d0 = p[0] - b->rho[0], d1 = p[1] - b->rho[1];
    double i0 = b->ups[0][0] * d0 + b->ups[0][1] * d1;
    double i1 = b->ups[1][0] * d0 + b->ups[1][1] * d1;
    u[0] = i0 / cos(phi1rad) + b->phi[0];
    u[1] = i1 + b->phi[1];
}

/* fits_image.py:196-216 cd_at_pixel: 10-px finite difference of pixel2equa */
void orc_cd_at_pixel(const orc_band *b, double x, double y, double cd[4]) {
    const double step = 10.0;
    double p[2], e0[2], ex[2], ey[2];
    p[0] = x; p[1] = y; orc_pixel2equa(b, p, e0);
    p[0] = x + step; p[1] = y; orc_pixel2equa(b, p, ex);
    p[0] = x; p[1] = y + step; orc_pixel2equa(b, p, ey);
    double cosd = cos(e0[1] * (M_PI / 180.0));
    cd[0] = (ex[0] - e0[0]) / step * cosd;
    cd[1] = (ey[0] - e0[0]) / step * cosd;
    cd[2] = (ex[1] - e0[1]) / step;
    cd[3] = (ey[1] - e0[1]) / step;
}

/* ------------------------------------------------------ bounding radius -- */

/* util/bound/bounding_box.py:9-31 calc_bounding_radius.
 * scipy.stats.chi2.ppf(1 - error, 2) has the closed form -2 ln(error) for 2 dof;
 * `rsq` is passed in so that tests can hand over scipy's own value. */
double orc_bounding_radius_rsq(const double *w, const double *mu, const double *cov, int K,
                               double rsq, const double center[2]) {
    double minbound = -INFINITY;
    (void)w;
    for (int i = 0; i < K; i++) {
        const double *c = cov + 4 * i;
        double sigma1 = sqrt(c[0]);
        double sigma2 = sqrt(c[3]);
        double rho = c[1] / (sigma1 * sigma2);
        double A11 = sigma1;
        double A21 = rho * sigma2;
        double A22 = sigma2 * sqrt(1.0 - rho * rho);
        double An = 1.0 / rsq * (1.0 / (A11 * A11) + (A21 * A21) / (A22 * A22));
        double Bn = 1.0 / rsq * (-2.0 * A21 / (A11 * (A22 * A22)));
        double Cn = 1.0 / rsq * 1.0 / (A22 * A22);
        double majaxis = pow(0.5 * (An + Cn - sqrt(Bn * Bn + (An - Cn) * (An - Cn))), -0.5);
        double d0 = mu[2 * i] - center[0], d1 = mu[2 * i + 1] - center[1];
        double dist = sqrt(d0 * d0 + d1 * d1);
        double cand = majaxis + dist;
        if (cand > minbound) minbound = cand;
    }
    return minbound;
}

double orc_bounding_radius(const double *w, const double *mu, const double *cov, int K,
                           double error, const double center[2]) {
    /* chi2.ppf(1-error, 2) = -2 log(1 - (1 - error)); scipy evaluates it from q = 1 - error */
    double q = 1.0 - error;
    double rsq = -2.0 * log1p(-q);
    return orc_bounding_radius_rsq(w, mu, cov, K, rsq, center);
}

/* ------------------------------------------------------------ evaluators -- */

/* util/like/gmm_like_fast.pyx:130-176 gmm_like_2d: component-outer, pixel-inner
 * (prange) direct sum; in-loop 2x2 inverse; exp(-log2pi - .5 log det - .5 q) * w. */
void orc_gmm_like_2d(double *probs, const double *x, int64_t N, const double *ws, const double *mus,
                     const double *sigs, int K) {
    const double log2pi = log(2.0 * M_PI);
    int64_t n;
#pragma omp parallel for schedule(static) if (N >= 16384)
    for (n = 0; n < N; n++) probs[n] = 0.0;
    for (int k = 0; k < K; k++) {
        const double *s = sigs + 4 * k;
        double detk = s[0] * s[3] - s[1] * s[2];
        double invk_00 = s[3] / detk;
        double invk_11 = s[0] / detk;
        double invk_01 = -1 * s[1] / detk;
        double m0 = mus[2 * k], m1 = mus[2 * k + 1], wk = ws[k];
#pragma omp parallel for schedule(static) if (N >= 16384)   /* small grids: a parallel region per component costs more than it saves */
        for (n = 0; n < N; n++) {
            double x0 = x[2 * n] - m0;
            double x1 = x[2 * n + 1] - m1;
            double quad = x0 * x0 * invk_00 + x1 * x1 * invk_11 + 2. * x0 * x1 * invk_01;
            probs[n] += exp(-log2pi - .5 * log(detk) - .5 * quad) * wk;
        }
    }
}

/* util/dists/mog.py:5-21 mog_loglike: log sum_k pi_k N(x; mu_k, C_k) via logsumexp.
 * icovs are full 2x2 (the einsum 'ijk,lji->lki' contracts solved = icov . centered). */
static inline double mog_loglike_pt(double px, double py, const double *means, const double *icovs,
                                    const double *dets, const double *pis, int K, double *scratch) {
    const double log2pi = log(2 * M_PI);
    double mx = -INFINITY;
    for (int k = 0; k < K; k++) {
        double c0 = px - means[2 * k], c1 = py - means[2 * k + 1];
        const double *ic = icovs + 4 * k;
        double s0 = ic[0] * c0 + ic[1] * c1;
        double s1 = ic[2] * c0 + ic[3] * c1;
        double lp = -0.5 * (s0 * c0 + s1 * c1) - log2pi - 0.5 * log(dets[k]) + log(pis[k]);
        scratch[k] = lp;
        if (lp > mx) mx = lp;
    }
    if (!isfinite(mx)) mx = 0.0; /* scipy.special.logsumexp guards a non-finite max */
    double s = 0.0;
    for (int k = 0; k < K; k++) s += exp(scratch[k] - mx);
    return log(s) + mx;
}

void orc_mog_loglike(double *out, const double *x, int64_t N, const double *means, const double *icovs,
                     const double *dets, const double *pis, int K) {
    int64_t n;
#pragma omp parallel if (N >= 4096)
    {
        double *scratch = (double *)malloc(sizeof(double) * (size_t)(K > 0 ? K : 1));
#pragma omp for schedule(static)
        for (n = 0; n < N; n++)
            out[n] = mog_loglike_pt(x[2 * n], x[2 * n + 1], means, icovs, dets, pis, K, scratch);
        free(scratch);
    }
}

/* exp(mog_loglike) on the integer grid [x0,x1) x [y0,y1), row-major [y][x]
 * (celeste.py:141-144 / mog.py:102-112: meshgrid 'xy', ravel C). Serial: callers
 * parallelise over sources. */
static void eval_grid(double *patch, int x0, int x1, int y0, int y1, const double *means,
                      const double *icovs, const double *dets, const double *pis, int K) {
    double scratch[K_GAL];
    int nx = x1 - x0;
    for (int y = y0; y < y1; y++)
        for (int x = x0; x < x1; x++)
            patch[(int64_t)(y - y0) * nx + (x - x0)] =
                exp(mog_loglike_pt((double)x, (double)y, means, icovs, dets, pis, K, scratch));
}

static inline void inv2(const double c[4], double ic[4], double *det) {
    double d = c[0] * c[3] - c[1] * c[2];
    ic[0] = c[3] / d; ic[1] = -c[1] / d; ic[2] = -c[2] / d; ic[3] = c[0] / d;
    *det = d;
}

/* ------------------------------------------------------------ star stamp -- */

/* celeste.py:114-140: pixel location, overlap test (Q1, reproduced with its axis mix-up)
 * and the int()-truncated box.  Returns 0 when the reference returns (None, None, None),
 * 1 otherwise; box = {y0, y1, x0, x1} (may be empty/negative-width: Q1). */
int orc_star_box(const orc_band *b, int H, int W, const double u[2], double v[2], int box[4]) {
    orc_equa2pixel(b, u, v);
    int miss = (v[0] < -50 || v[0] > 2 * H || v[1] < -50 || v[0] > 2 * W); /* celeste.py:130-132 */
    if (miss) return 0;
    double bound = b->R;
    int lx = (int)(v[0] - bound), hx = (int)(v[0] + bound + 1);   /* int(): truncation toward 0 */
    int ly = (int)(v[1] - bound), hy = (int)(v[1] + bound + 1);
    box[2] = lx > 0 ? lx : 0; box[3] = hx < W ? hx : W;
    box[0] = ly > 0 ? ly : 0; box[1] = hy < H ? hy : H;
    return 1;
}

/* celeste.py:153-162: unit-flux PSF stamp on a given box (means = psf.mu + v,
 * icovs = inv(covars), dets = exp(logdets), pis = weights). */
void orc_star_patch(const orc_band *b, const double v[2], const int box[4], double *patch) {
    double means[2 * K_PSF], icovs[4 * K_PSF], dets[K_PSF];
    for (int k = 0; k < K_PSF; k++) {
        means[2 * k] = b->mu[k][0] + v[0];
        means[2 * k + 1] = b->mu[k][1] + v[1];
        double d;
        inv2(&b->cov[k][0][0], icovs + 4 * k, &d);
        dets[k] = exp(log(d)); /* logdets via slogdet, then np.exp(logdets): celeste.py:160 */
    }
    eval_grid(patch, box[2], box[3], box[0], box[1], means, icovs, dets, b->w, K_PSF);
}

/* ---------------------------------------------------------- galaxy stamp -- */

/* celeste_galaxy_conditionals.py:90-125: gen_galaxy_ra_dec_basis + gen_galaxy_transformation.
 * phi_s is in DEGREES (Q7).  Returns Tinv (pixels per r_e), row-major. */
void orc_galaxy_tinv(double sig_s, double rho_s, double phi_s, const double cd[4], double Tinv[4]) {
    double phi = (90. - phi_s) * M_PI / 180.;
    double re_deg = fmax(1. / 30, sig_s) / 3600.;
    double cp = cos(phi), sp = sin(phi);
    double G[4] = {re_deg * cp, re_deg * (sp * rho_s), re_deg * (-sp), re_deg * (cp * rho_s)};
    double Gi[4], T[4], d;
    inv2(G, Gi, &d);
    T[0] = Gi[0] * cd[0] + Gi[1] * cd[2];
    T[1] = Gi[0] * cd[1] + Gi[1] * cd[3];
    T[2] = Gi[2] * cd[0] + Gi[3] * cd[2];
    T[3] = Gi[2] * cd[1] + Gi[3] * cd[3];
    inv2(T, Tinv, &d);
}

/* celeste_galaxy_conditionals.py:193-203 + util/dists/mog.py:75-100:
 * convex_combine([exp, dev], [theta, 1-theta]) -> apply_affine(Tinv, [px,py]) -> convolve(psf).
 * Output order is galaxy-major: idx = j*3 + k (Q10).  th = {theta, sigma, phi, rho}. */
void orc_galaxy_table(const orc_band *b, const double th[4], const double u[2], double *pis,
                      double *means, double *covs, double pxy[2], double Tinv[4]) {
    double ea[K_EXP], ev[K_EXP], da[K_DEV], dv[K_DEV];
    orc_profile_tables(ea, ev, da, dv);
    orc_equa2pixel(b, u, pxy);
    double cd[4];
    orc_cd_at_pixel(b, pxy[0], pxy[1], cd);
    orc_galaxy_tinv(th[1], th[3], th[2], cd, Tinv);
    for (int j = 0; j < K_PROF; j++) {
        double pj = (j < K_EXP) ? th[0] * ea[j] : (1. - th[0]) * da[j - K_EXP];
        double var = (j < K_EXP) ? ev[j] : dv[j - K_EXP];
        /* apply_affine: A . (var I) . A^T, evaluated as dot(dot(A, c), A.T) */
        double Ac[4] = {Tinv[0] * var, Tinv[1] * var, Tinv[2] * var, Tinv[3] * var};
        double C[4] = {Ac[0] * Tinv[0] + Ac[1] * Tinv[1], Ac[0] * Tinv[2] + Ac[1] * Tinv[3],
                       Ac[2] * Tinv[0] + Ac[3] * Tinv[1], Ac[2] * Tinv[2] + Ac[3] * Tinv[3]};
        for (int k = 0; k < K_PSF; k++) {
            int i = j * K_PSF + k;
            pis[i] = pj * b->w[k];
            means[2 * i] = pxy[0] + b->mu[k][0];
            means[2 * i + 1] = pxy[1] + b->mu[k][1];
            covs[4 * i + 0] = C[0] + b->cov[k][0][0];
            covs[4 * i + 1] = C[1] + b->cov[k][0][1];
            covs[4 * i + 2] = C[2] + b->cov[k][1][0];
            covs[4 * i + 3] = C[3] + b->cov[k][1][1];
        }
    }
}

/* celeste_galaxy_conditionals.py:205-211: bound (error 1e-5, centre (px,py)) and the
 * floor/ceil box (Q6); box = {y0, y1, x0, x1}. Returns the bound. */
double orc_galaxy_box(int H, int W, const double *pis, const double *means, const double *covs,
                      const double pxy[2], int box[4]) {
    double bound = orc_bounding_radius(pis, means, covs, K_GAL, 1e-5, pxy);
    double xl = fmax(0.0, floor(pxy[0] - bound)), xh = fmin((double)W, ceil(pxy[0] + bound));
    double yl = fmax(0.0, floor(pxy[1] - bound)), yh = fmin((double)H, ceil(pxy[1] + bound));
    box[0] = (int)yl; box[1] = (int)yh; box[2] = (int)xl; box[3] = (int)xh;
    return bound;
}

/* mog.py:102-112 evaluate_grid on the box, with dets/icovs from update_params (:55-56). */
void orc_galaxy_patch(const double *pis, const double *means, const double *covs, const int box[4],
                      double *patch) {
    double icovs[4 * K_GAL], dets[K_GAL];
    for (int i = 0; i < K_GAL; i++) inv2(covs + 4 * i, icovs + 4 * i, dets + i);
    eval_grid(patch, box[2], box[3], box[0], box[1], means, icovs, dets, pis, K_GAL);
}

/* -------------------------------------------------- one source, one band -- */

/* Unit-flux patch of one source in one band: A8 (type 0) or A17 (type 1).
 * Returns the number of patch pixels written (0 => no contribution); box = {y0,y1,x0,x1}.
 * `patch` may be NULL to query the box only. */
int64_t orc_source_patch(const orc_band *b, int H, int W, int type, const double u[2],
                         const double shape[4], int box[4], double *patch) {
    if (type == 0) {
        double v[2];
        if (!orc_star_box(b, H, W, u, v, box)) { box[0] = box[1] = box[2] = box[3] = 0; return 0; }
        if (box[1] <= box[0] || box[3] <= box[2]) return 0;
        if (patch) orc_star_patch(b, v, box, patch);
    } else {
        double pis[K_GAL], means[2 * K_GAL], covs[4 * K_GAL], pxy[2], Tinv[4];
        orc_galaxy_table(b, shape, u, pis, means, covs, pxy, Tinv);
        orc_galaxy_box(H, W, pis, means, covs, pxy, box);
        if (box[1] <= box[0] || box[3] <= box[2]) return 0;
        if (patch) orc_galaxy_patch(pis, means, covs, box, patch);
    }
    return (int64_t)(box[1] - box[0]) * (box[3] - box[2]);
}

/* ----------------------------------------------------------- full field -- */

/* celeste.py:203-219 gen_model_image + :237-252 celeste_likelihood[_multi_image], with the
 * patch-accumulating extension for galaxies (models.py:88-108 semantics):
 *   lambda[b] = eps_b + sum_s counts[s][b] * unit_patch(s, b) placed at its own box
 *   ll_band[b] = sum_{y,x} nelec * log(lambda) - lambda
 * counts[s*B + b] is the expected-photon multiplier (the three flux conventions of
 * celeste.py:35-62 are applied by the caller).  Sources are accumulated in index order.
 * lambda (B*H*W) and ll_band (B) are outputs; nelec may be NULL (then ll_band is not written).
 * stats (may be NULL): {n_srcpix, n_gauss} summed over bands.
 * Parallelism: chunks of sources rendered in parallel into private patches, added serially. */
void orc_render_field(const orc_band *bands, int B, int H, int W, int64_t S, const int32_t *type,
                      const double *radec, const double *counts, const double *shape,
                      const double *nelec, double *lambda, double *ll_band, double *stats) {
    enum { CHUNK = 64 };
    double n_srcpix = 0.0, n_gauss = 0.0;
    int64_t maxpix = (int64_t)H * W;
    for (int b = 0; b < B; b++) {
        double *lam = lambda + (int64_t)b * H * W;
        for (int64_t i = 0; i < (int64_t)H * W; i++) lam[i] = 0.0;
        for (int64_t s0 = 0; s0 < S; s0 += CHUNK) {
            int64_t ns = (S - s0 < CHUNK) ? (S - s0) : CHUNK;
            double *patches[CHUNK];
            int boxes[CHUNK][4];
            int64_t npix[CHUNK];
            int64_t c;
#pragma omp parallel for schedule(dynamic, 1)
            for (c = 0; c < ns; c++) {
                int64_t s = s0 + c;
                npix[c] = orc_source_patch(&bands[b], H, W, type[s], radec + 2 * s, shape + 4 * s,
                                           boxes[c], NULL);
                patches[c] = NULL;
                if (npix[c] > 0 && npix[c] <= maxpix) {
                    patches[c] = (double *)malloc(sizeof(double) * (size_t)npix[c]);
                    orc_source_patch(&bands[b], H, W, type[s], radec + 2 * s, shape + 4 * s, boxes[c],
                                     patches[c]);
                }
            }
            for (c = 0; c < ns; c++) {
                if (!patches[c]) continue;
                int64_t s = s0 + c;
                double cnt = counts[s * B + b];
                int y0 = boxes[c][0], y1 = boxes[c][1], x0 = boxes[c][2], x1 = boxes[c][3];
                int nx = x1 - x0;
                for (int y = y0; y < y1; y++)
                    for (int x = x0; x < x1; x++)
                        lam[(int64_t)y * W + x] += patches[c][(int64_t)(y - y0) * nx + (x - x0)] * cnt;
                n_srcpix += (double)npix[c];
                n_gauss += (double)npix[c] * (type[s] == 0 ? K_PSF : K_GAL);
                free(patches[c]);
            }
        }
        double eps = bands[b].eps;
        for (int64_t i = 0; i < (int64_t)H * W; i++) lam[i] = eps + lam[i];
        if (nelec && ll_band) {
            const double *ne = nelec + (int64_t)b * H * W;
            /* np.sum is pairwise; at 1e-6 relative the order is immaterial, but long double
             * keeps the oracle's own error far below the tolerance */
            long double acc = 0.0L;
            for (int64_t i = 0; i < (int64_t)H * W; i++) acc += (long double)(ne[i] * log(lam[i]) - lam[i]);
            ll_band[b] = (double)acc;
        }
    }
    if (stats) { stats[0] = n_srcpix; stats[1] = n_gauss; }
}

/* Reference-faithful variant of gen_model_image for STARS (celeste.py:203-219): every source
 * allocates a zeroed full frame, embeds its patch, scales the whole frame and adds it.
 * Timed on a stated subsample to document the reference's O(S*H*W) behaviour. */
void orc_gen_model_image_fullframe(const orc_band *band, int H, int W, int64_t S, const double *radec,
                                   const double *counts_b, double *lambda) {
    int64_t n = (int64_t)H * W;
    double *f_s = (double *)calloc((size_t)n, sizeof(double));
    for (int64_t s = 0; s < S; s++) {
        double *grid = (double *)calloc((size_t)n, sizeof(double));  /* celeste.py:170-171 */
        int box[4];
        double v[2];
        if (orc_star_box(band, H, W, radec + 2 * s, v, box) && box[1] > box[0] && box[3] > box[2]) {
            int nx = box[3] - box[2];
            double *patch = (double *)malloc(sizeof(double) * (size_t)nx * (size_t)(box[1] - box[0]));
            orc_star_patch(band, v, box, patch);
            for (int y = box[0]; y < box[1]; y++)
                memcpy(grid + (int64_t)y * W + box[2], patch + (int64_t)(y - box[0]) * nx,
                       sizeof(double) * (size_t)nx);
            free(patch);
        }
        double c = counts_b[s];
        for (int64_t i = 0; i < n; i++) f_s[i] += grid[i] * c;            /* celeste.py:45,217 */
        free(grid);
    }
    for (int64_t i = 0; i < n; i++) lambda[i] = band->eps + f_s[i];       /* celeste.py:219 */
    free(f_s);
}

/* One image's term of Source.log_likelihood (sources.py:134-183, mode 0) or
 * Source.log_likelihood_isolated (:188-237, mode 1): the source's unit stamp on the FIXED patch
 * limits box = {y0,y1,x0,x1} (compute_scatter_on_pixels with xlim/ylim, :351-388), scaled by
 * counts = flux_in_image (:120-129), against the patch data.  mode 4: the type move's image_like. */
double orc_patch_loglik(const orc_band *b, int H, int W, int type, const double u[2],
                        const double shape[4], double counts, const int box[4], const double *data,
                        int mode) {
    double wsum = (b->w[0] + b->w[1]) + b->w[2];            /* np.sum(fits_img.weights) */
    int64_t n = (int64_t)(box[1] - box[0]) * (box[3] - box[2]);
    if (n <= 0) return 0.0;
    double *patch = (double *)malloc(sizeof(double) * (size_t)n);
    if (type == 0) {
        double v[2];
        int own[4];
        if (!orc_star_box(b, H, W, u, v, own)) {            /* psf_ns is None (:160-163) */
            free(patch);
            return -counts * wsum;
        }
        orc_star_patch(b, v, box, patch);
    } else {
        double pis[K_GAL], means[2 * K_GAL], covs[4 * K_GAL], pxy[2], Tinv[4];
        orc_galaxy_table(b, shape, u, pis, means, covs, pxy, Tinv);
        orc_galaxy_patch(pis, means, covs, box, patch);
    }
    long double a = 0.0L, msum = 0.0L;
    for (int64_t i = 0; i < n; i++) {
        double m = counts * patch[i];
        if (mode == 0) {
            if (m > 0.) a += (long double)(log(m) * data[i]);        /* mask = model_patch > 0 (:172-174) */
        } else if (mode == 4) {
            /* image_like of calculate_acceptance_logprob (sources.py:277-291): poisson_loglike (:6-12) of the
             * observed box against background_img + model_img; data = [observed (n), background (n)], an observed
             * value that is NaN stands for mask == 0 (a negative count, as in sky-subtracted data, is kept: :9) */
            m += data[n + i];
            if (m > 0. && data[i] == data[i]) {
                a += (long double)(log(m) * data[i]);
                msum += (long double)m;
            }
        } else {
            m += b->eps;                                              /* :219 */
            a += (long double)(log(m) * data[i]);
            msum += (long double)m;
        }
    }
    free(patch);
    return mode == 0 ? (double)a - counts * wsum : (double)(a - msum);
}

/* ------------------------------------- the older per-profile galaxy route (A16, A18) -- */

/* celeste_fast.pyx:100-140 gen_galaxy_prof_psf_mixture_params: one profile (amp[J], sigs[J])
 * convolved with the PSF (K_psf components); PSF-major output, cnt = k * J + j (:122-139). */
void orc_galaxy_prof_psf_mixture_params(const double W[4], const double v_s[2], const double *image_ws,
                                        const double *image_means, const double *image_covars, int K_psf,
                                        const double *amp, const double *sigs, int J, double *weights,
                                        double *means, double *covars) {
    int cnt = 0;
    for (int k = 0; k < K_psf; k++)
        for (int j = 0; j < J; j++) {
            weights[cnt] = image_ws[k] * amp[j];                         /* :124 */
            means[2 * cnt + 0] = v_s[0] + image_means[2 * k + 0];        /* :127-128 */
            means[2 * cnt + 1] = v_s[1] + image_means[2 * k + 1];
            for (int ii = 0; ii < 2; ii++)
                for (int jj = 0; jj < 2; jj++)                           /* :131-134 */
                    covars[4 * cnt + 2 * ii + jj] = image_covars[4 * k + 2 * ii + jj] + sigs[j] * W[2 * ii + jj];
            cnt++;
        }
}

/* celeste_fast.pyx:29-94 gen_galaxy_psf_mixture_params: for k (PSF) / for i in (exp, dev) / for j;
 * weights = image_ws[k] * thetas[i] * amp_ij (:77), evaluated left to right. */
void orc_galaxy_psf_mixture_params(const double thetas[2], const double W[4], const double v_s[2],
                                   const double *image_ws, const double *image_means, const double *image_covars,
                                   int K_psf, const double *exp_amp, const double *exp_sigs, int K_exp_,
                                   const double *dev_amp, const double *dev_sigs, int K_dev_, double *weights,
                                   double *means, double *covars) {
    int cnt = 0;
    for (int k = 0; k < K_psf; k++)
        for (int i = 0; i < 2; i++) {
            int Ki = (i == 0) ? K_exp_ : K_dev_;
            for (int j = 0; j < Ki; j++) {
                double amp_ij = (i == 0) ? exp_amp[j] : dev_amp[j];
                double var_ij = (i == 0) ? exp_sigs[j] : dev_sigs[j];
                weights[cnt] = image_ws[k] * thetas[i] * amp_ij;
                means[2 * cnt + 0] = v_s[0] + image_means[2 * k + 0];
                means[2 * cnt + 1] = v_s[1] + image_means[2 * k + 1];
                for (int ii = 0; ii < 2; ii++)
                    for (int jj = 0; jj < 2; jj++)
                        covars[4 * cnt + 2 * ii + jj] = image_covars[4 * k + 2 * ii + jj] + var_ij * W[2 * ii + jj];
                cnt++;
            }
        }
}

/* celeste_galaxy_conditionals.py:134-182 gen_galaxy_prof_psf_image: prof 0 = 'exp', 1 = 'dev'; R is the
 * shape matrix (row-major 2x2), W = R R^T (:151); amplitudes / variances are the profile's normalised
 * tables (what `.amp` / `.var[:,0,0]` of :155-156 mean).  bound with ERROR = 1e-5 about v_s (:160-161),
 * int() box (:166-167) unless lims = {y0,y1,x0,x1} is given (:162-164); values by gmm_like_2d (:173-176).
 * box = {y0,y1,x0,x1} out; patch may be NULL to query the box.  Returns the number of patch pixels. */
static int64_t prof_psf_image_W(const orc_band *b, int H, int W_, int prof, const double Wm[4], const double u[2],
                               const int *lims, int box[4], double *patch);

int64_t orc_galaxy_prof_psf_image(const orc_band *b, int H, int W_, int prof, const double R[4], const double u[2],
                                  const int *lims, int box[4], double *patch) {
    /* np.dot(R, R.T) */
    double Wm[4] = {R[0] * R[0] + R[1] * R[1], R[0] * R[2] + R[1] * R[3], R[2] * R[0] + R[3] * R[1], R[2] * R[2] + R[3] * R[3]};
    return prof_psf_image_W(b, H, W_, prof, Wm, u, lims, box, patch);
}

/* the same from W = R R^T on (what :151 forms and everything after it reads) */
static int64_t prof_psf_image_W(const orc_band *b, int H, int W_, int prof, const double Wm[4], const double u[2],
                               const int *lims, int box[4], double *patch) {
    double ea[K_EXP], ev[K_EXP], da[K_DEV], dv[K_DEV];
    orc_profile_tables(ea, ev, da, dv);
    const double *amp = prof == 0 ? ea : da, *sig = prof == 0 ? ev : dv;
    const int J = prof == 0 ? K_EXP : K_DEV;
    double v_s[2];
    orc_equa2pixel(b, u, v_s);
    double weights[K_PSF * K_DEV], means[2 * K_PSF * K_DEV], covars[4 * K_PSF * K_DEV];
    orc_galaxy_prof_psf_mixture_params(Wm, v_s, b->w, &b->mu[0][0], &b->cov[0][0][0], K_PSF, amp, sig, J, weights, means, covars);
    if (lims) {
        box[0] = lims[0]; box[1] = lims[1]; box[2] = lims[2]; box[3] = lims[3];
    } else {
        double bound = orc_bounding_radius(weights, means, covars, K_PSF * J, 0.00001, v_s);
        int lx = (int)(v_s[0] - bound), hx = (int)(v_s[0] + bound + 1);
        int ly = (int)(v_s[1] - bound), hy = (int)(v_s[1] + bound + 1);
        box[2] = lx > 0 ? lx : 0; box[3] = hx < W_ ? hx : W_;
        box[0] = ly > 0 ? ly : 0; box[1] = hy < H ? hy : H;
    }
    if (box[1] <= box[0] || box[3] <= box[2]) return 0;
    int64_t n = (int64_t)(box[1] - box[0]) * (box[3] - box[2]);
    if (!patch) return n;
    int nx = box[3] - box[2];
    double *pts = (double *)malloc(sizeof(double) * 2 * (size_t)n);
    for (int64_t i = 0; i < n; i++) {                                    /* meshgrid 'xy', C-order ravel (:170-171) */
        pts[2 * i + 0] = (double)(box[2] + (int)(i % nx));
        pts[2 * i + 1] = (double)(box[0] + (int)(i / nx));
    }
    orc_gmm_like_2d(patch, pts, n, weights, means, covars, K_PSF * J);
    free(pts);
    return n;
}

/* celeste_galaxy_conditionals.py:15-42 galaxy_source_like, one image's term on the photon patch's
 * limits box = {y0,y1,x0,x1}: R_s from the CONSTANT b->ups (:33), f = theta f_exp + (1 - theta) f_dev
 * (:34-36, both profiles on the same limits), lam = image_flux * f (:39-40),
 * sum Z log(lam) - lam over the pixels with lam > 0 (:41; a pixel the model does not reach is skipped
 * instead of contributing 0 * log 0).  th = {theta, sigma, phi, rho}. */
double orc_galaxy_source_like(const orc_band *b, int H, int W_, const double th[4], const double u[2],
                              double image_flux, const int box[4], const double *Z) {
    int64_t n = (int64_t)(box[1] - box[0]) * (box[3] - box[2]);
    if (n <= 0) return 0.0;
    double R[4];
    orc_galaxy_tinv(th[1], th[3], th[2], &b->ups[0][0], R);
    double *fe = (double *)malloc(sizeof(double) * (size_t)n), *fd = (double *)malloc(sizeof(double) * (size_t)n);
    int bx[4];
    orc_galaxy_prof_psf_image(b, H, W_, 0, R, u, box, bx, fe);
    orc_galaxy_prof_psf_image(b, H, W_, 1, R, u, box, bx, fd);
    long double a = 0.0L, m = 0.0L;
    for (int64_t i = 0; i < n; i++) {
        double lam = image_flux * (th[0] * fe[i] + (1. - th[0]) * fd[i]);
        if (lam > 0.) { a += (long double)(Z[i] * log(lam)); m += (long double)lam; }
    }
    free(fe); free(fd);
    return (double)(a - m);
}

/* The terms of orc_patch_loglik's mode 0 (Source.log_likelihood, sources.py:134-183) APART, for tests that must not
 * lose the photon term's digits where it cancels against the mass term:
 *   out[0] = sum_{m>0} log(m) z      out[1] = sum_{m>0} |log(m) z|   (the scale its rounding is relative to)
 *   out[2] = counts * sum(psf weights)                                (ll = out[0] - out[2])
 *   out[3] = what the SUBNORMAL range is worth in the photon term.  Where the unit stamp lies below 2^-1022 (a proposal
 *            hundreds of pixels from its photons: exponents between -708 and -750) every evaluator -- the reference's two
 *            included, mog_loglike's exp(logsumexp) and gmm_like_2d's sum of exp -- holds it to a few quanta of 2^-1074
 *            only; log() turns n quanta of error into n 2^-1074 / stamp, and below a handful of quanta whether a pixel
 *            counts at all (m > 0) depends on the order of the arithmetic.  With L = the stamp's logarithm formed in the
 *            log domain: a pixel with L < log(2^-1074) - log 64 adds nothing (zero everywhere), one with
 *            L < log(2^-1074) + log 64 its whole term |L + log counts| z, any other subnormal one 64 quanta.
 * type 0 star, 1 galaxy (shape = theta, sigma, phi, rho), 2 the older per-profile route's source as the product's ABI takes
 * it: shape = theta, W00, W01, W11 with W = R R^T of celeste_galaxy_conditionals.py:151 (constant Ups_n), the unit stamp
 * theta f_exp + (1 - theta) f_dev (:34-36) on the limits. */
void orc_patch_loglik_terms(const orc_band *b, int H, int W, int type, const double u[2], const double shape[4],
                            double counts, const int box[4], const double *data, double out[4]) {
    double wsum = (b->w[0] + b->w[1]) + b->w[2];
    out[0] = out[1] = out[3] = 0.0;
    out[2] = counts * wsum;
    int64_t n = (int64_t)(box[1] - box[0]) * (box[3] - box[2]);
    if (n <= 0) { out[2] = 0.0; return; }
    double *patch = (double *)malloc(sizeof(double) * (size_t)n);
    /* the same mixture as (pis, means, icovs, dets), for the log-domain value */
    double pis[K_GAL], means[2 * K_GAL], covs[4 * K_GAL], icovs[4 * K_GAL], dets[K_GAL], scratch[K_GAL];
    int K = K_GAL;
    if (type == 0) {
        double v[2];
        int own[4];
        if (!orc_star_box(b, H, W, u, v, own)) { free(patch); return; }     /* psf_ns is None (:160-163) */
        orc_star_patch(b, v, box, patch);
        K = K_PSF;
        for (int k = 0; k < K_PSF; k++) {
            pis[k] = b->w[k];
            means[2 * k] = b->mu[k][0] + v[0];
            means[2 * k + 1] = b->mu[k][1] + v[1];
            memcpy(covs + 4 * k, &b->cov[k][0][0], sizeof(double) * 4);
        }
    } else if (type == 1) {
        double pxy[2], Tinv[4];
        orc_galaxy_table(b, shape, u, pis, means, covs, pxy, Tinv);
        orc_galaxy_patch(pis, means, covs, box, patch);
    } else {
        double Wm[4] = {shape[1], shape[2], shape[2], shape[3]};
        double *fd = (double *)malloc(sizeof(double) * (size_t)n);
        int bx[4];
        prof_psf_image_W(b, H, W, 0, Wm, u, box, bx, patch);
        prof_psf_image_W(b, H, W, 1, Wm, u, box, bx, fd);
        for (int64_t i = 0; i < n; i++) patch[i] = shape[0] * patch[i] + (1. - shape[0]) * fd[i];
        free(fd);
        double ea[K_EXP], ev[K_EXP], da[K_DEV], dv[K_DEV], v_s[2];
        orc_profile_tables(ea, ev, da, dv);
        orc_equa2pixel(b, u, v_s);
        orc_galaxy_prof_psf_mixture_params(Wm, v_s, b->w, &b->mu[0][0], &b->cov[0][0][0], K_PSF, ea, ev, K_EXP, pis, means, covs);
        orc_galaxy_prof_psf_mixture_params(Wm, v_s, b->w, &b->mu[0][0], &b->cov[0][0][0], K_PSF, da, dv, K_DEV,
                                           pis + K_PSF * K_EXP, means + 2 * K_PSF * K_EXP, covs + 4 * K_PSF * K_EXP);
        for (int k = 0; k < K_GAL; k++) pis[k] *= (k < K_PSF * K_EXP) ? shape[0] : (1. - shape[0]);
    }
    for (int k = 0; k < K; k++) inv2(covs + 4 * k, icovs + 4 * k, dets + k);
    const double LOG_Q = -744.44007192138126;          /* log(2^-1074) */
    const double LOG_64 = 4.1588830833596715;
    const int nx = box[3] - box[2];
    long double a = 0.0L, aa = 0.0L, qq = 0.0L;
    for (int64_t i = 0; i < n; i++) {
        double m = counts * patch[i];
        if (m > 0.) {
            double t = log(m) * data[i];
            a += (long double)t;
            aa += (long double)fabs(t);
        }
        if (patch[i] < 2.2250738585072014e-308 && data[i] != 0.) {
            double L = mog_loglike_pt((double)(box[2] + (int)(i % nx)), (double)(box[0] + (int)(i / nx)), means, icovs, dets, pis, K, scratch);
            if (L < LOG_Q - LOG_64) continue;
            if (L < LOG_Q + LOG_64) qq += (long double)(fabs(L + log(counts)) * fabs(data[i]));
            else qq += (long double)(fabs(data[i]) * 64.0 * exp(LOG_Q - L));
        }
    }
    free(patch);
    out[0] = (double)a;
    out[1] = (double)aa;
    out[3] = (double)qq;
}

/* The reductions celeste_em.py:38-91 takes of gen_src_prob_layers (celeste.py:222-234):
 *   xtilde[s*B+b] = sum nelec * F_s / lambda, mass[s*B+b] = sum unit stamp, noise[b] = sum nelec * eps / lambda
 * lambda (B*H*W) must be the model image of exactly these sources (orc_render_field). */
void orc_estep_stats(const orc_band *bands, int B, int H, int W, int64_t S, const int32_t *type,
                     const double *radec, const double *counts, const double *shape, const double *nelec,
                     const double *lambda, double *xtilde, double *mass, double *noise) {
    for (int b = 0; b < B; b++) {
        const double *lam = lambda + (int64_t)b * H * W, *ne = nelec + (int64_t)b * H * W;
        long double z = 0.0L;
        for (int64_t i = 0; i < (int64_t)H * W; i++) z += (long double)(ne[i] * (bands[b].eps / lam[i]));
        noise[b] = (double)z;
        for (int64_t s = 0; s < S; s++) {
            int box[4];
            int64_t n = orc_source_patch(&bands[b], H, W, type[s], radec + 2 * s, shape + 4 * s, box, NULL);
            xtilde[s * B + b] = 0.0;
            mass[s * B + b] = 0.0;
            if (n <= 0) continue;
            double *patch = (double *)malloc(sizeof(double) * (size_t)n);
            orc_source_patch(&bands[b], H, W, type[s], radec + 2 * s, shape + 4 * s, box, patch);
            long double xt = 0.0L, ms = 0.0L;
            int nx = box[3] - box[2];
            for (int y = box[0]; y < box[1]; y++)
                for (int x = box[2]; x < box[3]; x++) {
                    double u = patch[(int64_t)(y - box[0]) * nx + (x - box[2])];
                    int64_t i = (int64_t)y * W + x;
                    xt += (long double)((u * counts[s * B + b]) / lam[i] * ne[i]);
                    ms += (long double)u;
                }
            xtilde[s * B + b] = (double)xt;
            mass[s * B + b] = (double)ms;
            free(patch);
        }
    }
}

/* sources.py:6-12 poisson_loglike on a patch with mask (mask may be NULL). */
double orc_poisson_loglike(const double *data, const double *model, const uint8_t *mask, int64_t n) {
    long double a = 0.0L, bsum = 0.0L;
    for (int64_t i = 0; i < n; i++) {
        if (model[i] > 0. && (!mask || mask[i])) {
            a += (long double)(log(model[i]) * data[i]);
            bsum += (long double)model[i];
        }
    }
    return (double)(a - bsum);
}
