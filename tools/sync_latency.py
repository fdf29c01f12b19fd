#!/usr/bin/env python3
"""Step time of the small star field with the HIP runtime's host-wait policy: default, spin, yield, blocking.  (diagnostic)"""
import ctypes as C, os, sys, time, subprocess
if len(sys.argv) > 1:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import desi_mcmc_amd as cel
    from desi_mcmc_amd import _lib, synth
    hip = C.CDLL(_lib.LIB_PATH)
    flag = int(sys.argv[1])
    if flag >= 0:
        print("hipSetDeviceFlags ->", hip.hipSetDeviceFlags(C.c_uint(flag)))
    ctx = cel.Context(0)
    for name in ("stars1k_512", "mixed10k_2048"):
        f = synth.SyntheticField.from_config(ctx, name)
        for _ in range(50):
            f.images.render(f.sources, loglik=True)
        t0 = time.perf_counter()
        n = 300
        for _ in range(n):
            f.images.render(f.sources, loglik=True)
        print(name, "flag", flag, "step %.4f ms" % ((time.perf_counter() - t0) / n * 1e3))
else:
    for flag in (-1, 1, 2, 4):      # default, hipDeviceScheduleSpin, Yield, BlockingSync
        subprocess.run([sys.executable, __file__, str(flag)])
