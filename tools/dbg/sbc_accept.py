"""the shape step's calibration with the reference's `acceptable` and with Neal's sticky flag (host engine)"""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import desi_mcmc_amd as cel
import test_calibration as tc
ctx = cel.default_context(0)
for acc in ("reference", "neal"):
    t0 = time.time()
    ru, rf, rs = tc.pooled_ranks(cel, ctx, "host", 8, shapes=True, shape_args={"accept": acc})
    print(acc, "%.0f s" % (time.time() - t0), {n: (round(tc.chi2_pvalue(rs[:, i], tc.K_DRAWS)[1], 5), tc.chi2_pvalue(rs[:, i], tc.K_DRAWS)[2].astype(int).tolist())
                for i, n in enumerate(("theta", "sigma", "phi", "rho"))}, flush=True)
