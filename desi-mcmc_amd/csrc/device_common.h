// device_common.h -- shared device-side data of the celeste HIP kernels (included by celeste_hip.hip)
#pragma once
#define K_PSF 3
#define K_EXP 6
#define K_PROF 14
#define K_GAL 42
#define TILE_W 64
#define MAX_BANDS 16

// ------------------------------------------------------------------------------------------
// device-side data
// ------------------------------------------------------------------------------------------
struct BandDev {        // per band, SoA-friendly PSF so that lane k can index component k
    double eps;
    double w[K_PSF], mux[K_PSF], muy[K_PSF], cxx[K_PSF], cxy[K_PSF], cyy[K_PSF];
    double rho[2], phi[2], ups[4], ups_inv[4];
    double R;
};

struct alignas(16) SrcRec {   // one per (band, source); 128 bytes
    double px, py;            // pixel position (x = column, y = row)
    double scale;             // expected photons of this source in this band
    double w00, w01, w11;     // galaxy: Tinv Tinv^T (cov_j = var_j * W + psf_cov_k)
    double theta;             // exp-profile fraction
    int x0, x1, y0, y1;       // clipped box [x0,x1) x [y0,y1); empty when x1<=x0 or y1<=y0
    int type;                 // 0 star, 1 galaxy, -1 no contribution
    int pad[3];
    double rsv[5];
};
static_assert(sizeof(SrcRec) == 128, "SrcRec must be 128 bytes");

// exp/dev profile mixtures (Hogg & Lang; CelestePy/mixture_profiles.py:9-19), amplitudes
// normalised on the host exactly as the reference does (:13,:19) and uploaded once.
__constant__ double c_prof_amp[K_PROF];
__constant__ double c_prof_var[K_PROF];

// log() table of the render epilogue (log_tab in k_render.h): for c_j = 1 + (j + 1/2)/64,
// c_log_ic[j] = fl(1/c_j) and c_log_lc[j] = -log(c_log_ic[j]) evaluated in long double on the
// host, so that log(m) = lc[j] + log1p(m * ic[j] - 1) holds to the last bit for m in [1, 2).
__constant__ double c_log_ic[64];
__constant__ double c_log_lc[64];

static const double H_EXP_AMP[6] = {2.34853813e-03, 3.07995260e-02, 2.23364214e-01,
                                    1.17949102e+00, 4.33873750e+00, 5.99820770e+00};
static const double H_EXP_VAR[6] = {1.20078965e-03, 8.84526493e-03, 3.91463084e-02,
                                    1.39976817e-01, 4.60962500e-01, 1.50159566e+00};
static const double H_DEV_AMP[8] = {4.26347652e-02, 2.40127183e-01, 6.85907632e-01, 1.51937350e+00,
                                    2.83627243e+00, 4.46467501e+00, 5.72440830e+00, 5.60989349e+00};
static const double H_DEV_VAR[8] = {2.23759216e-04, 1.00220099e-03, 4.18731126e-03, 1.69432589e-02,
                                    6.84850479e-02, 2.87207080e-01, 1.33320254e+00, 8.40215071e+00};

// one photon-holding pixel of a device-resident sample patch (k_nz_compact, k_split.h): x | y << 16 (absolute
// pixel coordinates, both below 65 536) and the photons attributed to the source there
struct NzEntry { int xy, z; };

#define PI_D 3.14159265358979323846

// unit of the integer stamp-mass sums the photon split and k_strict_totals accumulate (cel_stamp_mass's short cut): 2^-60
#define MASS_FX 1152921504606846976.0
