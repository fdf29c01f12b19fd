"""exp / dev galaxy profile mixtures-of-Gaussians: the constants of
CelestePy/mixture_profiles.py:9-19 (Hogg & Lang fits), amplitudes normalised to sum 1 as at
:13 and :19.  Only the tables are on the hot path; the reference's tractor-style mixture class
and grid evaluator (:27-182) are dead code there and are not reproduced.
"""
import numpy as np

exp_amp = np.array([2.34853813e-03, 3.07995260e-02, 2.23364214e-01, 1.17949102e+00, 4.33873750e+00,
                    5.99820770e+00])
exp_var = np.array([1.20078965e-03, 8.84526493e-03, 3.91463084e-02, 1.39976817e-01, 4.60962500e-01,
                    1.50159566e+00])
exp_amp /= np.sum(exp_amp)

dev_amp = np.array([4.26347652e-02, 2.40127183e-01, 6.85907632e-01, 1.51937350e+00, 2.83627243e+00,
                    4.46467501e+00, 5.72440830e+00, 5.60989349e+00])
dev_var = np.array([2.23759216e-04, 1.00220099e-03, 4.18731126e-03, 1.69432589e-02, 6.84850479e-02,
                    2.87207080e-01, 1.33320254e+00, 8.40215071e+00])
dev_amp /= np.sum(dev_amp)
