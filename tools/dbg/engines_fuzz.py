"""random small scenes at the edges (one source, no galaxy, no star, sources on and off the border, empty patches, sky over three
decades): two Gibbs sweeps with the galaxies' shape step on the device engine and on the host engine -- locations, fluxes and shapes
must be equal bit for bit, sweep after sweep.   python tools/dbg/engines_fuzz.py [N] [first seed]"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np
import desi_mcmc_amd as cel
from desi_mcmc_amd import synth, celeste_mcmc
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ctx = cel.default_context(0)
bad = 0
for seed in range(first, first + N):
    rs = np.random.RandomState(7000 + seed)
    H, W = int(rs.randint(64, 300)), int(rs.randint(64, 300))
    S = int(rs.choice([1, 2, 3, 8, 30]))
    bands = synth.make_bands(H, W, 5)
    bands[:, 0] *= 10.0 ** rs.uniform(-1, 2)
    pix = np.column_stack([rs.uniform(-8, W + 8, S), rs.uniform(-8, H + 8, S)])
    typ = (rs.rand(S) < rs.choice([0.0, 0.5, 1.0])).astype(np.int32)
    shape = np.column_stack([rs.uniform(0.05, 0.95, S), np.exp(rs.uniform(np.log(0.3), np.log(2.5), S)), rs.uniform(1, 179, S), rs.uniform(0.1, 0.95, S)])
    shape[typ == 0] = 0.0
    flux = np.exp(rs.uniform(np.log(0.3), np.log(300.0), (S, 5)))
    radec = synth.pixel2equa(bands[0], pix)
    counts = flux / bands[None, :, 2] * bands[None, :, 1]
    iset = cel.ImageSet(ctx, bands, H, W)
    iset.render(cel.SourceSet(ctx, S, 5).set(typ, radec, counts, shape), loglik=False)
    iset.set_nelec(rs.poisson(iset.model_images()).astype(np.float64))
    # the samplers' options the device engines take: the shape step along the axes or along 1-4 random directions, stepping
    # out by doubling or not at all, any interval; the location step's interval
    shape_args = dict(compwise=bool(rs.randint(2)), numdir=int(rs.randint(1, 5)), sigma=float(rs.choice([1.0, 0.3, 0.05])))
    if rs.rand() < 0.4:
        shape_args["step_out"] = False
    slice_args = None if rs.rand() < 0.5 else dict(sigma=float(rs.choice([1e-3, 3e-4, 5e-3])))
    out = {}
    err = {}
    for engine in ("device", "host"):
        for b in range(5):
            iset.set_epsilon(b, bands[b, 0])
        gf = celeste_mcmc.GibbsField(iset, list(range(5)), bands[:, 2], bands[:, 1], H * W)
        g = celeste_mcmc.ModelGibbs([gf], typ, radec, flux, shape, seed=seed, engine=engine, shape_args=shape_args, slice_args=slice_args)
        tr = []
        try:
            for k in range(2):
                g.sweep(shapes=True)
                tr.append((g.u.copy(), g.fluxes.copy(), g.shape.copy(), g.log_likelihood()))
        except Exception as e:
            err[engine] = "%s: %s" % (type(e).__name__, str(e)[:80])
        out[engine] = tr
    if err.get("device") != err.get("host") or len(out["device"]) != len(out["host"]):
        bad += 1; print("seed %d (S=%d %dx%d): the engines end differently: %s" % (seed, S, H, W, err))
        continue
    for k, (a, b_) in enumerate(zip(out["device"], out["host"])):
        for name, x, y in zip(("u", "fluxes", "shape", "ll"), a, b_):
            if not np.array_equal(np.asarray(x), np.asarray(y), equal_nan=True):
                bad += 1
                print("seed %d (S=%d %dx%d) sweep %d: %s differs by %g" % (seed, S, H, W, k, name, np.nanmax(np.abs(np.asarray(x) - np.asarray(y)))))
                break
    if seed % 25 == 24:
        print("seed %d: %d disagreements so far%s" % (seed, bad, ("; last error on both engines: %s" % err["device"]) if err else ""), flush=True)
print("ok: %d scenes" % N if not bad else "MISMATCH in %d" % bad)
sys.exit(1 if bad else 0)
