"""random sequences of what a caller does to a LIST of SrcParams between likelihood calls -- assignments, in-place edits followed by
an assignment or touch(), objects replaced, lists re-ordered, shortened, extended, two lists and two image groups in turn -- with
celeste_likelihood_multi_image / gen_model_image after each: the value must be, bit for bit, what the same list gives with the list
cache off (every source re-read, whole upload, every tile rendered).   python tools/dbg/list_cache_stress.py [STEPS] [seed] [big] [stamps]
(the default mode, "exact", also gets in-place edits that nobody announces)"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np
import desi_mcmc_amd as cel
from desi_mcmc_amd import celeste, celeste_src, synth
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ctx = cel.default_context(0)
celeste.set_device(0)
rs = np.random.RandomState(seed)
mode = "stamps" if "stamps" in sys.argv else "exact"      # the list mode under test (celeste.list_cache): the default re-reads every source
celeste.list_cache(mode)
big = "big" in sys.argv                   # a frame of 5 120 tiles: the one-wave-per-tile kernel, whose dirty-tile path the row uploads reach
fa = synth.SyntheticField(ctx, 2500, 5, 1024, 2048, frac_gal=0.5, seed=5) if big else synth.SyntheticField(ctx, 600, 5, 500, 420, frac_gal=0.5, seed=5)
imgs_a = synth.fits_images(fa)
imgs_b = imgs_a[1:4]                     # another image group
ref_a = synth.fits_images(fa)            # the cache-off values come from image sets of their own (other FitsImage objects)
ref_b = ref_a[1:4]
incremental = [0, 0]
src = fa.src
cat = cel.SrcCatalog((src["type"] == 1).astype(np.int64), src["radec"], fa.flux5(), src["shape"])


def fresh(p):
    return cel.SrcParams(u=np.array(p.u, copy=True), a=p.a, fluxes=np.array(p.fluxes, copy=True), theta=p.theta, sigma=p.sigma, phi=p.phi, rho=p.rho)


lists = [[fresh(p) for p in cat], [fresh(p) for p in list(cat)[:len(cat) // 2]]]
log = []


def mutate(L):
    op = rs.choice(["assign_u", "assign_u", "assign_u", "inplace_then_assign", "inplace_touch", "flux", "shape", "type", "replace", "swap", "pop", "append", "many", "noop_assign"] +
                   (["inplace_only", "inplace_only", "inplace_flux_only"] if mode == "exact" else []))
    i = int(rs.randint(len(L)))
    p = L[i]
    if op == "assign_u":
        p.u = p.u + rs.normal(0, 2e-5, 2)
    elif op == "inplace_then_assign":
        p.u[0] += 1e-5; p.u = p.u
    elif op == "inplace_only":                       # nobody tells the library: the default mode must see it all the same
        p.u[int(rs.randint(2))] += 1e-5 * rs.normal()
    elif op == "inplace_flux_only":
        p.fluxes[int(rs.randint(5))] *= 1.01
    elif op == "inplace_touch":
        p.u[1] -= 1e-5; celeste_src.touch(p)
    elif op == "flux":
        fl = np.array(p.fluxes, copy=True); fl *= np.exp(rs.normal(0, 0.2, fl.shape)); p.fluxes = fl
    elif op == "shape" and p.a == 1:
        p.sigma = float(p.sigma * np.exp(rs.normal(0, 0.2))); p.rho = float(np.clip(p.rho + rs.normal(0, 0.05), 0.05, 1.0))
    elif op == "type":
        if p.a == 1:
            p.a = 0
        else:
            p.a = 1; p.theta, p.sigma, p.phi, p.rho = 0.5, 1.0, 30.0, 0.7
    elif op == "replace":
        L[i] = fresh(L[int(rs.randint(len(L)))])
    elif op == "swap":
        j = int(rs.randint(len(L))); L[i], L[j] = L[j], L[i]
    elif op == "pop" and len(L) > 100 and rs.rand() < 0.3:
        L.pop(i)
    elif op == "append" and len(L) < 3000 and rs.rand() < 0.3:
        L.append(fresh(L[i]))
    elif op == "many":
        for j in rs.choice(len(L), int(rs.choice([3, 40, 90, 200])), replace=False):
            L[j].u = L[j].u + rs.normal(0, 1e-5, 2)
    elif op == "noop_assign":
        p.u = p.u
    return op


def evaluate(L, imgs, what):
    ref = ref_a if imgs is imgs_a else ref_b
    if what == "ll":
        got = celeste.celeste_likelihood_multi_image(L, imgs)
        d = celeste._image_set(tuple(imgs)).last_render_dirty_tiles()
        incremental[0 if d >= 0 else 1] += 1
    else:
        got = celeste.gen_model_image(L, imgs[0])
    # (a TUPLE of fresh copies is gathered source by source on every call: celeste._source_arrays caches lists only)
    if what == "ll":
        want = celeste.celeste_likelihood_multi_image(tuple(fresh(p) for p in L), ref)
    else:
        want = celeste.gen_model_image(tuple(fresh(p) for p in L), ref[0])
    ok = (got == want) if what == "ll" else np.array_equal(got, want)
    if not ok:
        print("MISMATCH at step %d (%s): %r against %r; last ops %s" % (len(log), what, got if what == "ll" else "image", want if what == "ll" else "image", log[-15:]))
        sys.exit(1)


n_eval = 0
for step in range(STEPS):
    L = lists[int(rs.rand() < 0.25)]
    r = rs.rand()
    if r < 0.55:
        log.append(mutate(L) + ("@%d" % (L is lists[1])))
    else:
        imgs = imgs_a if rs.rand() < 0.7 else imgs_b
        what = "ll" if rs.rand() < 0.85 else "image"
        evaluate(L, imgs, what)
        n_eval += 1
        log.append("%s(list %d, %d images)" % (what, L is lists[1], len(imgs)))
    if step % 500 == 499:
        print("step %d: %d evaluations" % (step + 1, n_eval), flush=True)
print("ok: list_cache %r, %d steps, %d evaluations equal to the cache-off values; likelihood renders: %d of dirty tiles only, %d of every tile" % (mode, STEPS, n_eval, incremental[0], incremental[1]))
