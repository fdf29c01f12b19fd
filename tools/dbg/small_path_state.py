"""the one-launch small star path (k_small_stars) and the book-keeping of the image set it writes into: a general render of catalogue
A, a small-path render of a star-only catalogue B, then calls on A that may re-use what the image set holds"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np
import desi_mcmc_amd as cel
from desi_mcmc_amd import synth
ctx = cel.default_context(0)
f = synth.SyntheticField(ctx, 600, 5, 512, 512, frac_gal=0.5, seed=2)
g = synth.SyntheticField(ctx, 900, 5, 512, 512, frac_gal=0.0, seed=3)
A = f.sources
B_ = cel.SourceSet(ctx, g.S, 5).set(g.src["type"], g.src["radec"], g.src["counts"], g.src["shape"])
img = f.images
ref = cel.ImageSet(ctx, f.bands, f.H, f.W, nelec=f.nelec)
bad = 0
# 1. split of A after (render A, render B): the totals must be A's
img.render(A, loglik=True)
img.render(B_, loglik=True)
n1 = img.photon_split_resident(A, 5); s1 = img.sample_sums()
n2 = ref.photon_split_resident(A, 5); s2 = ref.sample_sums()
d = np.abs(s1 - s2)
print("split of A after B's small-path render: photons differ by up to %g (sum %g)" % (d.max(), d.sum()))
bad += d.max() > 3
# 2. boxes of A
img.render(A, loglik=True); img.render(B_, loglik=True)
b1 = img.source_boxes(A)[0]; b2 = ref.source_boxes(A)[0]
print("boxes of A equal:", np.array_equal(b1, b2)); bad += not np.array_equal(b1, b2)
# 3. an incremental render of A after B's small render
cur = {k: np.array(f.src[k], copy=True) for k in ("type", "radec", "counts", "shape")}
img.render(A, loglik=True); img.render(B_, loglik=True)
rows = np.array([3], dtype=np.int32); cur["counts"][3] *= 1.1
A.set_rows(rows, cur["type"][rows], cur["radec"][rows], cur["counts"][rows], cur["shape"][rows])
l1 = img.render(A, loglik=True); d_ = img.last_render_dirty_tiles()
l2 = ref.render(cel.SourceSet(ctx, f.S, 5).set(cur["type"], cur["radec"], cur["counts"], cur["shape"]), loglik=True)
eq = np.array_equal(img.model_images(), ref.model_images()) and np.array_equal(l1[1], l2[1])
print("render of A (one row changed) after B's small render: dirty %d, equal to a fresh render: %s" % (d_, eq)); bad += not eq
print("ok" if not bad else "MISMATCH: %d" % bad)
sys.exit(1 if bad else 0)
