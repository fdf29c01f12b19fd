"""One rank of a Gibbs chain dealt over several processes (helper of tests/test_gibbs.py; not collected).
Run with RANK / WORLD_SIZE / MASTER_* set:  python tests/_dealt_chain_rank.py OUT.npz S SIZE SWEEPS ENGINE SHAPES [strips]
Builds the synthetic field (every rank the same), runs ModelGibbs with a dist.SourceDeal over the gloo
process group (the ranks may share one GPU) and writes the chain's state after every sweep."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run_chain(S, size, sweeps, engine, deal=None, seed=4, shapes=False, strips=None):
    """strips = (world, rank): the strip-partitioned chain's rank (the deal is built here)"""
    import desi_mcmc_amd as cel
    from desi_mcmc_amd import celeste_mcmc, synth
    ctx = cel.default_context(0)
    f = synth.SyntheticField(ctx, S, 5, size, size, frac_gal=0.5, seed=9)
    if strips is not None:
        boxes, status = f.images.source_boxes(f.sources)              # (B, S, 4) y0, y1, x0, x1 on the whole frame
        deal, gf = celeste_mcmc.strip_gibbs_field(ctx, f.bands, f.nelec, f.src["pix"][:, 1], boxes, status, *strips)
    else:
        gf = celeste_mcmc.GibbsField(f.images, list(range(5)), f.bands[:, 2], f.bands[:, 1], size * size)
    g = celeste_mcmc.ModelGibbs([gf], f.src["type"], f.src["radec"], f.flux5(), f.src["shape"], seed=seed,
                                engine=engine, deal=deal)
    us, fls, eps, lls, shs, sums = [], [], [], [], [], []
    for _ in range(sweeps):
        g.sweep(shapes=shapes)
        sums.append(np.where(g.deal.mask[:, None], gf.sums, 0.0) if g.deal is not None else gf.sums.copy())
        shs.append(g.shape.copy())
        us.append(g.u.copy())
        fls.append(g.fluxes.copy())
        eps.append(gf.epsilon.copy())
        lls.append(g.log_likelihood())
    return dict(u=np.array(us), fluxes=np.array(fls), eps=np.array(eps), ll=np.array(lls), shape=np.array(shs), sums=np.array(sums),
                noise=np.array(g.noise_sums[0]), nelec_sum=f.nelec.reshape(5, -1).sum(axis=1),
                evals=np.array(g.timing["evals"]), shape_evals=np.array(g.timing["shape_evals"]), active=g.active.copy())


if __name__ == "__main__":
    out, S, size, sweeps, engine = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    shapes = sys.argv[6] == "1"
    strips = len(sys.argv) > 7 and sys.argv[7] == "strips"
    from desi_mcmc_amd import dist
    rank, world, _ = dist.init_from_env(backend="gloo")
    if strips:
        res = run_chain(S, size, sweeps, engine, shapes=shapes, strips=(world, rank))
    else:
        res = run_chain(S, size, sweeps, engine, deal=dist.SourceDeal(S, world, rank), shapes=shapes)
    np.savez(out, **res)
    dist.barrier()
    import torch.distributed as td
    td.destroy_process_group()
