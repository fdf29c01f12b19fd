"""N > 1 path on CPU: world_size-2 gloo.  The per-rank compute is stood in by the CPU oracle
(tests may use it); what is under test is the partition arithmetic and the one collective of
desi-mcmc_amd/dist.py -- the same code bench.py runs over RCCL on GPUs."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT, load_golden


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, mode, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from desi_mcmc_amd import dist
    from oracle import oracle as orc
    r, w, _ = dist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    g = load_golden("mini_field.npz")
    B = orc.pack_bands(g)
    H, W = int(g["H"]), int(g["W"])
    counts = g["flux"] / g["calib"][None, :] * g["kappa"][None, :]
    lam, _, _ = orc.render_field(B, H, W, g["is_gal"], g["radec"], counts, g["shape"], None)
    if mode == "strips":
        y0, y1 = dist.strip_rows(H, world, rank)
        part = np.array([orc.poisson_loglike(g["nelec"][b, y0:y1], lam[b, y0:y1]) for b in range(5)])
    elif mode == "strips_measured":
        # strip edges from a MEASURED per-band cost: every rank measures something else (here: a cost that depends on the
        # rank), the cut is rank 0's (dist.agree_on_edges) -- the strips still tile the frame exactly once
        nb = -(-H // 32)
        cost = np.ones(nb) + (rank + 1) * np.arange(nb)[::-1]
        mine_edges = dist.strip_edges(H, world, cost)
        edges = dist.agree_on_edges(mine_edges)
        assert edges == dist.strip_edges(H, world, np.ones(nb) + np.arange(nb)[::-1]) and len(edges) == world + 1
        assert mine_edges == dist.strip_edges(H, world, cost)               # (a rank's own measurement: possibly another cut)
        y0, y1 = edges[rank], edges[rank + 1]
        part = np.array([orc.poisson_loglike(g["nelec"][b, y0:y1], lam[b, y0:y1]) for b in range(5)])
    else:  # fields: 8 fields dealt to the ranks as bench.py --workload fields8_2048 deals them
        # (field k = the same sky observed with +k electrons per pixel)
        mine = dist.field_shard(8, world, rank)
        assert len(mine) == 8 // world and sorted(sum((dist.field_shard(8, world, r) for r in range(world)), [])) == list(range(8))
        part = np.zeros(5)
        for f in mine:
            part += np.array([orc.poisson_loglike(g["nelec"][b] + f, lam[b]) for b in range(5)])
    tot = dist.allreduce_loglik(part)
    tot_det = dist.allreduce_loglik(part, deterministic=True)
    # the pipelined form bench.py uses: results come back in submission order, one step late
    red = dist.LoglikReducer(5, depth=2)
    got = []
    for k in range(4):
        red.submit(part * (k + 1))
        if len(red.pending) > 1:
            got.append(red.result())
    got += red.drain()
    assert len(got) == 4 and not red.pending
    for k in range(4):
        np.testing.assert_allclose(got[k], tot * (k + 1), rtol=1e-14)
    with pytest.raises(RuntimeError):
        red.result()
    dist.barrier()
    np.save(os.path.join(out_dir, "%s_%d.npy" % (mode, rank)), np.stack([tot, tot_det, part]))
    import torch.distributed as td
    td.destroy_process_group()


def _real_set_worker(rank, world, port, out_dir):
    """the real-data field set (tests/golden/real_fields.npz) dealt to the ranks: each rank scores ITS fields (the oracle
    stands in for the GPU here) and one all-reduce of the 5 per-band sums gives every rank the survey's log-likelihood"""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import real_fields
    from desi_mcmc_amd import dist
    from oracle import oracle as orc
    dist.init_from_env(backend="gloo")
    _, fields = real_fields()
    mine = dist.field_shard(len(fields), world, rank)
    part = np.zeros(5)
    for k in mine:
        f = fields[k]
        S = len(f["radec"])
        _, ll, _ = orc.render_field(orc.pack_bands(f["rec"]), f["H"], f["W"], np.zeros(S, np.int32), f["radec"].reshape(S, 2),
                                    (f["flux"] * f["rec"]["kappa"][None, :]).reshape(S, 5), np.zeros((S, 4)), f["nelec"])
        part += ll
    tot = dist.allreduce_loglik(part)
    tot_det = dist.allreduce_loglik(part, deterministic=True)
    np.save(os.path.join(out_dir, "realset_%d.npy" % rank), np.stack([tot, tot_det, part, np.full(5, float(len(mine)))]))
    dist.barrier()
    import torch.distributed as td
    td.destroy_process_group()


def test_real_field_set_world2_gloo(tmp_path):
    """BASELINE configs[3] on the real data the reference ships: 100 fields dealt round-robin to two ranks, one all-reduce;
    the sum is the reference's own (celeste_likelihood per image of every field, summed)"""
    import torch.multiprocessing as mp
    from conftest import real_fields
    world = 2
    mp.spawn(_real_set_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    res = [np.load(os.path.join(str(tmp_path), "realset_%d.npy" % r)) for r in range(world)]
    _, fields = real_fields()
    want = np.sum([f["ll_band"] for f in fields], axis=0)
    for r in range(world):
        np.testing.assert_allclose(res[r][0], want, rtol=1e-12)
        np.testing.assert_allclose(res[r][1], want, rtol=1e-12)
        assert np.array_equal(res[r][1], res[0][1])
        assert res[r][3][0] == 50.0
    np.testing.assert_allclose(res[0][2] + res[1][2], want, rtol=1e-12)
    assert not np.allclose(res[0][2], res[1][2])


def _deal_target(idx, P):
    """a per-chain log-density (chain c is a Gaussian about (c, -c) of width 1 + c / 10): the stand-in for the
    conditional likelihood of source c given the photon split"""
    c = np.asarray(idx, dtype=np.float64)
    return -0.5 * np.sum((P - np.stack([c, -c], axis=1)) ** 2, axis=1) / (1.0 + c / 10.0) ** 2


def _deal_worker(rank, world, port, out_dir):
    """ONE chain's per-source updates dealt to the ranks (dist.SourceDeal): every rank updates its own
    sources with the lock-step slice sampler and the rows are exchanged with one all-gather"""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    from desi_mcmc_amd import dist
    from desi_mcmc_amd.util.infer.slicesample import slicesample_lockstep
    dist.init_from_env(backend="gloo")
    S = 37                                                        # not a multiple of the world: the last rows are padded
    deal = dist.SourceDeal(S, world, rank)
    assert deal.mine.tolist() == list(range(rank, S, world)) and deal.mask.sum() == deal.mine.size
    ids = deal.chain_ids()
    assert np.array_equal(ids[deal.mine], deal.mine) and np.all(ids[~deal.mask] == -1)
    X = np.column_stack([np.arange(S, dtype=np.float64), -np.arange(S, dtype=np.float64)]) + 0.25
    extra = np.arange(S * 5, dtype=np.float64).reshape(S, 5)
    for sweep in range(3):
        new, _ = slicesample_lockstep(X[deal.mine], lambda i, P: _deal_target(deal.mine[i], P), sigma=1.5,
                                      seed=100 + sweep, chain_ids=deal.mine)
        X[deal.mine] = new
        extra[deal.mine] += 1000.0 * (sweep + 1)                  # the "fluxes": also only this rank's rows
        both = deal.merge(np.concatenate([X, extra], axis=1))
        X, extra = both[:, :2].copy(), both[:, 2:].copy()
    with pytest.raises(ValueError):
        deal.merge(np.zeros((S + 1, 2)))
    np.savez(os.path.join(out_dir, "deal_%d.npz" % rank), X=X, extra=extra)
    dist.barrier()
    import torch.distributed as td
    td.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_one_chain_dealt_to_the_ranks_equals_the_single_rank_chain(tmp_path, world):
    """the partition of SURVEY 8e for config 5: sources dealt round-robin, one all-gather per sweep; every
    rank ends with the same state, and it is the state a single rank computes -- bit for bit (a chain's
    random stream does not depend on which chains run beside it)"""
    import torch.multiprocessing as mp
    from desi_mcmc_amd import dist
    from desi_mcmc_amd.util.infer.slicesample import slicesample_lockstep
    mp.spawn(_deal_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    S = 37
    X = np.column_stack([np.arange(S, dtype=np.float64), -np.arange(S, dtype=np.float64)]) + 0.25
    extra = np.arange(S * 5, dtype=np.float64).reshape(S, 5)
    for sweep in range(3):
        X, _ = slicesample_lockstep(X, _deal_target, sigma=1.5, seed=100 + sweep)
        extra += 1000.0 * (sweep + 1)
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), "deal_%d.npz" % r))
        assert np.array_equal(got["X"], X) and np.array_equal(got["extra"], extra)
    one = dist.SourceDeal(S)                                      # a world of one: merge is a copy, every source is mine
    assert one.mine.size == S and np.array_equal(one.merge(X), X)
    with pytest.raises(ValueError):
        dist.SourceDeal(S, 2, 2)


def _strip_worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    from desi_mcmc_amd import dist
    dist.init_from_env(backend="gloo")
    H, S = 300, 41
    rows = np.random.RandomState(5).uniform(-3.0, H + 3.0, S)           # a few sources just off the frame
    deal = dist.StripDeal(rows, H, world, rank, halo=50)
    y0, y1 = deal.strip
    assert (y0, y1) == dist.strip_rows(H, world, rank)
    inside = (np.clip(np.floor(rows), 0, H - 1) >= y0) & (np.clip(np.floor(rows), 0, H - 1) < y1)
    assert np.array_equal(deal.mask, inside) and deal.mine.size == inside.sum()
    assert deal.window == (max(0, y0 - 64), min(H, y1 + 64))              # the halo in whole 32-row tiles
    assert deal.noise_rows() == (y0 - deal.window[0], y1 - deal.window[0])
    state = np.zeros((S, 3))
    for sweep in range(2):
        state[deal.mine] += (rank + 1) * 10.0 ** sweep + deal.mine[:, None] * 0.001        # only this rank's rows
        state = deal.merge(state)
    tot = deal.rank_sum(np.array([float(deal.mine.size), 1.0]))
    assert tot.tolist() == [float(S), float(world)]
    Hw = deal.window[1] - deal.window[0]
    boxes = np.zeros((2, S, 4), dtype=np.int32)
    boxes[:, :, 0], boxes[:, :, 1] = 5, Hw - 5
    ones = np.ones((2, S), dtype=np.int32)
    assert deal.check_boxes(boxes, ones) is None                          # inside every window: fine (a collective)
    got = deal.check_boxes(boxes, ones, extra=np.array([1.0, float(rank)]))
    assert got.tolist() == [float(world), float(sum(range(world)))]         # the caller's vector rides along, summed
    # ONE rank's window cuts one of its boxes: every rank raises (a rank raising alone would leave the others blocked in
    # the sweep's next collective)
    assert deal.boxes_cut(boxes, ones) == 0
    if rank == 1:
        assert deal.window[0] > 0 and deal.mine.size
        boxes[0, deal.mine[0], 0] = 0
        assert deal.boxes_cut(boxes, ones) == 1
    with pytest.raises(RuntimeError, match="halo"):
        deal.check_boxes(boxes, ones)
    with pytest.raises(ValueError, match="own no rows"):
        dist.StripDeal(rows, 64, world, rank)                             # 2 tile rows for 3 ranks
    np.savez(os.path.join(out_dir, "strip_%d.npz" % rank), state=state, owner=deal.owner)
    dist.barrier()
    import torch.distributed as td
    td.destroy_process_group()


def test_strip_deal_world3_gloo(tmp_path):
    """dist.StripDeal: ownership by row strip, windows with halos, the merge of unequal shares and the rank-ordered sums"""
    import torch.multiprocessing as mp
    world = 3
    mp.spawn(_strip_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    res = [np.load(os.path.join(str(tmp_path), "strip_%d.npz" % r)) for r in range(world)]
    for r in range(1, world):
        assert np.array_equal(res[r]["state"], res[0]["state"]) and np.array_equal(res[r]["owner"], res[0]["owner"])
    owner, state = res[0]["owner"], res[0]["state"]
    assert set(owner.tolist()) == {0, 1, 2}
    want = (owner[:, None] + 1) * 11.0 + 2 * np.arange(len(owner))[:, None] * 0.001
    np.testing.assert_allclose(state, np.repeat(want, 3, axis=1).reshape(len(owner), 3), rtol=1e-15)


@pytest.mark.parametrize("mode", ["strips", "strips_measured", "fields"])
def test_world2_gloo_loglik_allreduce(tmp_path, mode):
    import torch.multiprocessing as mp
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, mode, str(tmp_path)), nprocs=world, join=True)
    res = [np.load(os.path.join(str(tmp_path), "%s_%d.npy" % (mode, r))) for r in range(world)]
    g = load_golden("mini_field.npz")
    if mode.startswith("strips"):
        expect = g["ll_band"]
    else:
        from oracle import oracle as orc
        expect = sum(np.array([orc.poisson_loglike(g["nelec"][b] + k, g["lam"][b]) for b in range(5)]) for k in range(8))
    for r in range(world):
        np.testing.assert_allclose(res[r][0], expect, rtol=1e-12)      # all-reduce
        np.testing.assert_allclose(res[r][1], expect, rtol=1e-12)      # all-gather + ordered sum
        assert np.array_equal(res[r][1], res[0][1])                     # identical on every rank
    assert not np.allclose(res[0][2], res[1][2])                        # the ranks did different work


# ---- bench.py --gpus N without a launcher ---------------------------------------------------------
def _run_bench(extra_env, *argv):
    import subprocess
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), env=env,
                          capture_output=True, text=True, timeout=600)


def test_bench_self_launch_spawns_n_ranks_gloo_dry_run():
    """`python bench.py --gpus 2` starts its own two ranks (torch.distributed.run child), they
    rendezvous and run the bench's collectives; rank 0 reports n_gpus = 2.  Dry run: no GPU here."""
    import json
    r = _run_bench({"CEL_BENCH_BACKEND": "gloo", "CEL_BENCH_DRYRUN": "1"}, "--gpus", "2")
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["ranks"] == 2
    assert out["allreduce_check"] == out["expected"] == 3.0
    # the roll call: the collective itself counted two ranks, each reported where it sits
    assert out["ranks_seen_by_collective"] == 2 and len(out["device_uuid"]) == 2 and len(set(out["device_uuid"])) == 2
    # the N = 1 point of the weak-scaling curve is the headline: rank 0 of the 2-rank job runs what `--gpus 1` runs
    assert out["rank0_job"] == out["n1_job"] == {"function": "run_render", "workload": "mixed10k_2048", "field_seed": 42, "scaling": "weak"}


def test_roll_call_refuses_a_short_group_and_shared_devices(tmp_path):
    """dist.roll_call in a 2-rank gloo group: the right count passes; a job started for 3 ranks whose collective reaches 2, and two
    ranks on one device, raise on every rank (bench.py turns that into a non-zero exit before any number is printed)"""
    import torch.multiprocessing as mp
    mp.spawn(_roll_call_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert sorted(os.listdir(str(tmp_path))) == ["ok_0", "ok_1"]


def _roll_call_worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    from desi_mcmc_amd import dist
    dist.init_from_env(backend="gloo")
    out = dist.roll_call(2, 0)
    assert out["ranks_seen_by_collective"] == 2 and len(set(out["device_uuid"])) == 2 and out["backend"] == "gloo"
    with pytest.raises(RuntimeError, match="reached 2 rank"):
        dist.roll_call(3, 0)
    real = dist.device_identity
    dist.device_identity = lambda local: "GPU-same"
    try:
        with pytest.raises(RuntimeError, match="share a device"):
            dist.roll_call(2, 0)
        assert dist.roll_call(2, 0, allow_shared_devices=True)["device_uuid"] == ["GPU-same", "GPU-same"]
    finally:
        dist.device_identity = real
    open(os.path.join(out_dir, "ok_%d" % rank), "w").close()
    dist.barrier()
    import torch.distributed as td
    td.destroy_process_group()


def test_bench_gpus_n_fails_loudly_with_fewer_devices():
    """Fewer GPUs than --gpus: non-zero exit and a message, never a silent 1-GPU line."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs visible")
    r = _run_bench({}, "--gpus", "2")
    assert r.returncode != 0
    assert "--gpus 2 requested but only" in r.stderr
    assert not any(l.startswith("{") for l in r.stdout.splitlines())


def test_bench_rejects_world_size_mismatch():
    r = _run_bench({"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0", "CEL_BENCH_DRYRUN": "1"}, "--gpus", "2")
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)
