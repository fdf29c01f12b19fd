"""Per-block phase timeline of k_small_stars (diagnostic; needs a library built with -DSMALL_STAMPS, e.g.
    hipcc <Makefile flags> -DSMALL_STAMPS -o tools/bin/libcel_stamps.so desi-mcmc_amd/csrc/celeste_hip.hip
and CEL_HIP_LIBRARY pointing at it): CEL_SMALL_STAMPS=file makes the library dump, per block, six
100 MHz wall-clock stamps (start, tables ready, scan done, walk done, epilogue done, end), the star count and the XCC id.
    python tools/small_stamps.py [workload]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
path = os.path.join(ROOT, "gpurun_out", "small_stamps.bin")
os.makedirs(os.path.dirname(path), exist_ok=True)
os.environ["CEL_SMALL_STAMPS"] = path
import numpy as np
import desi_mcmc_amd as cel
from desi_mcmc_amd import synth
name = sys.argv[1] if len(sys.argv) > 1 else "stars1k_512"
ctx = cel.default_context(0)
f = synth.SyntheticField.from_config(ctx, name)
for _ in range(5):
    f.images.render(f.sources, loglik=True)
st = np.fromfile(path, dtype=np.uint64).reshape(-1, 8).astype(np.int64)
stage_us = (st[:, 6] >> 32) / 100.0
st[:, 6] &= 0xffffffff
t0 = st[:, 0].min()
us = (st[:, :6] - t0) / 100.0
print("blocks %d; kernel span %.2f us (first start -> last end)" % (len(st), us[:, 5].max()))
print("block start: median %.2f, p95 %.2f, max %.2f us" % (np.median(us[:, 0]), np.percentile(us[:, 0], 95), us[:, 0].max()))
d = np.diff(us, axis=1)
for k, nm in enumerate(["loads+tables", "scan", "stage+walk", "epilogue", "records"]):
    print("%-14s mean %.2f  median %.2f  p95 %.2f  max %.2f us" % (nm, d[:, k].mean(), np.median(d[:, k]), np.percentile(d[:, k], 95), d[:, k].max()))
print("  of stage+walk, the staging (barriers, boxes, sort): mean %.2f median %.2f p95 %.2f us" % (stage_us.mean(), np.median(stage_us), np.percentile(stage_us, 95)))
dur = us[:, 5] - us[:, 0]
print("block duration: mean %.2f median %.2f p95 %.2f max %.2f us; block end: median %.2f p95 %.2f max %.2f" %
      (dur.mean(), np.median(dur), np.percentile(dur, 95), dur.max(), np.median(us[:, 5]), np.percentile(us[:, 5], 95), us[:, 5].max()))
print("stars per block: mean %.1f max %d; corr(duration, stars) %.2f" % (st[:, 6].mean(), st[:, 6].max(), np.corrcoef(dur, st[:, 6])[0, 1]))
