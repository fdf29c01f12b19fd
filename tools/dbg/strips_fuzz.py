"""random frames cut into random row strips (2-8 ranks, equal or random tile-aligned edges, random tile parts): every strip's model
image is the whole frame's rows and the strips' log-likelihoods add up to the whole frame's, band by band -- to what the drop rule
allows (a window's tiles start on other rows, so other components fall below eps e^-T on a tile: 2e-9 at the default T = 24, 1e-12
at the strict T = 32)"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np
import desi_mcmc_amd as cel
from desi_mcmc_amd import synth, dist
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ctx = cel.default_context(0)
L = cel._lib
bad = 0
for seed in range(first, first + N):
    rs = np.random.RandomState(3000 + seed)
    H, W = int(rs.randint(130, 900)), int(rs.randint(60, 500))
    S = int(rs.randint(1, 600))
    B = int(rs.randint(1, 6))
    f = synth.SyntheticField(ctx, S, B, H, W, frac_gal=rs.rand(), seed=seed)
    ctx.set_option(L.CEL_OPT_TILE_PARTS, int(rs.choice([0, 1, 2, 4])))
    strict = bool(rs.randint(2))
    ctx.set_tail_log("strict" if strict else "default")
    tol = 1e-12 if strict else 2e-9
    try:
        ll, llb = f.images.render(f.sources, loglik=True)
        lam = f.images.model_images()
        world = int(rs.randint(2, 9))
        nrow = (H + 63) // 64
        if rs.rand() < 0.5 or nrow <= world:
            edges = dist.strip_edges(H, world)
        else:
            cuts = np.sort(rs.choice(np.arange(1, nrow), world - 1, replace=False)) * 64
            edges = [0] + cuts.tolist() + [H]
        tot = np.zeros(B)
        for r in range(world):
            y0, y1 = int(edges[r]), int(edges[r + 1])
            if y1 <= y0:
                continue
            win = cel.ImageSet(ctx, f.bands, y1 - y0, W, nelec=f.nelec[:, y0:y1])
            win.set_window(y0, H)
            _, lw = win.render(f.sources, loglik=True)
            tot += lw
            d = np.abs(win.model_images() / lam[:, y0:y1] - 1).max()
            if d > tol:
                bad += 1; print("seed %d rank %d/%d rows [%d, %d): model image off by %g" % (seed, r, world, y0, y1, d))
            win.close()
        scale = (np.abs(f.nelec * np.log(lam)) + lam).sum(axis=(1, 2))
        if np.any(np.abs(tot - llb) > tol * scale):
            bad += 1; print("seed %d (%dx%d, S=%d, %d strips %s): log-likelihoods %r against %r" % (seed, H, W, S, world, list(edges), tot, llb))
    finally:
        ctx.set_option(L.CEL_OPT_TILE_PARTS, 0)
        ctx.set_tail_log("default")
    f.images.close()
    if seed % 50 == 49:
        print("seed %d: %d disagreements so far" % (seed, bad), flush=True)
print("ok: %d frames" % N if not bad else "MISMATCH in %d" % bad)
sys.exit(1 if bad else 0)
