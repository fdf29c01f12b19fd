"""Multi-GPU layer: one process per GPU, partition by pixels, ONE scalar-sized collective.

log lambda is non-linear in the per-pixel sum over sources, so every pixel's lambda has to be
complete on one GPU before the log.  Two partitions satisfy that without any
data-path exchange:

  fields  independent fields (or exposures) are dealt to ranks; each rank renders and scores its
          own fields                                                   -> weak scaling
  strips  one field is cut into row strips aligned to the 32-row tile; a source is replicated to
          every strip its box touches (the k_bin pass does that implicitly: every rank holds the
          catalogue, bins only its rows)                               -> strong scaling

Either way the only communication is the sum of B per-band log-likelihood doubles: one
all-reduce over RCCL/xGMI on GPUs (backend "nccl" is RCCL on ROCm), gloo on CPU test rigs.
"""
import os

import numpy as np

TILE_ROWS = 32


def strip_rows(H, world, rank, align=TILE_ROWS):
    """Rows [y0, y1) of rank's strip: contiguous, tile-aligned, covering [0, H) exactly once."""
    tiles = (H + align - 1) // align
    lo = (tiles * rank) // world
    hi = (tiles * (rank + 1)) // world
    return min(lo * align, H), min(hi * align, H)


def field_shard(n_fields, world, rank):
    """Indices of the fields rank owns (round-robin: equal counts when world divides n_fields)."""
    return list(range(rank, n_fields, world))


class SourceDeal(object):
    """The sources of ONE Gibbs chain dealt to the ranks: given the photon split, the
    per-source updates (Source.resample_fluxes / resample_location, CelestePy/sources.py:308-349) are
    independent of each other, so rank r updates the sources s = r (mod world) -- a round-robin deal keeps
    stars, galaxies and bright sources evenly spread -- and the new rows are exchanged with ONE all-gather
    per merge (10 000 x 7 doubles per sweep at config 5: 560 KB over xGMI).  The merged arrays are identical
    on every rank, to the bit, and identical to what a single rank computes (the per-chain random streams
    do not depend on which other chains run beside them)."""

    def __init__(self, S, world=1, rank=0, device=None):
        self.S, self.world, self.rank = int(S), int(world), int(rank)
        if not 0 <= self.rank < self.world:
            raise ValueError("rank %d outside a world of %d" % (rank, world))
        self.device = device
        self.mine = np.arange(self.rank, self.S, self.world)              # indices of this rank's sources
        self.mask = np.zeros(self.S, dtype=bool)
        self.mask[self.mine] = True
        self.per_rank = (self.S + self.world - 1) // self.world           # rows every rank contributes (padded)

    def chain_ids(self):
        """cel_slice_locations' chain_ids: a source's own index where it is this rank's, -1 elsewhere"""
        return np.where(self.mask, np.arange(self.S), -1).astype(np.int32)

    def merge(self, arr):
        """arr (S, k) with this rank's rows up to date -> (S, k) with every row taken from its owner"""
        arr = np.ascontiguousarray(arr, dtype=np.float64)
        if self.world == 1:
            return arr.copy()
        import torch
        import torch.distributed as dist
        if arr.ndim != 2 or arr.shape[0] != self.S:
            raise ValueError("merge takes an (S, k) array")
        k = arr.shape[1]
        send = np.zeros((self.per_rank, k))
        send[:self.mine.size] = arr[self.mine]
        t = torch.from_numpy(send)
        on_gpu = dist.get_backend() == "nccl"
        if on_gpu:
            t = t.cuda(self.device if self.device is not None else torch.cuda.current_device())
        recv = torch.empty((self.world * self.per_rank, k), dtype=t.dtype, device=t.device)      # rank-major concatenation
        dist.all_gather_into_tensor(recv, t)
        got = recv.cpu().numpy().reshape(self.world, self.per_rank, k)
        out = np.empty_like(arr)
        for r in range(self.world):
            rows = np.arange(r, self.S, self.world)
            out[rows] = got[r, :rows.size]
        return out


def init_from_env(backend=None):
    """torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun's contract).
    -> (rank, world, local_rank).  No-op (0, 1, 0) when WORLD_SIZE is unset or 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world <= 1:
        return 0, 1, local
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def allreduce_loglik(ll_band, device=None, deterministic=False, force=False):
    """Sum per-band log-likelihoods over ranks.  ll_band: (B,) float64 numpy -> (B,) numpy.

    deterministic=False: one all-reduce (sum) of B doubles -- the collective north_star names.
    deterministic=True : all-gather + fixed rank-order sum on the host, bitwise reproducible for
                         any topology."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force):
        return np.asarray(ll_band, dtype=np.float64).copy()
    t = torch.from_numpy(np.ascontiguousarray(ll_band, dtype=np.float64).copy())
    if dist.get_backend() == "nccl":
        t = t.cuda(device if device is not None else torch.cuda.current_device())
    if deterministic:
        parts = [torch.empty_like(t) for _ in range(dist.get_world_size())]
        dist.all_gather(parts, t)
        out = np.zeros(t.numel())
        for p in parts:            # rank order
            out += p.cpu().numpy()
        return out
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()


class LoglikReducer(object):
    """The same all-reduce, pipelined: submit() starts the sum of one evaluation's B per-band
    doubles and returns at once; result() hands back the oldest outstanding sum.  With fields
    dealt to ranks the chains of different fields never wait for the global log-likelihood (it
    is a diagnostic of the whole survey), so the collective of evaluation k can ride under the
    render of evaluation k+1 instead of adding its latency (launch + xGMI hop + D2H, ~0.1 ms
    against a 1.9 ms step) to every step.  Buffers are allocated once: a ring of `depth` device
    tensors and pinned host mirrors."""

    def __init__(self, B, device=None, depth=2, force=False):
        import torch
        import torch.distributed as dist
        self.B, self.depth = int(B), int(depth)
        # force: run the collective even in a one-rank group (a rehearsal of the RCCL path on one GPU)
        self.active = dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force)
        self.pending = []          # (slot, work handle)
        self.slot = 0
        self.gpu = self.active and dist.get_backend() == "nccl"
        if not self.active:
            self.host = [np.zeros(self.B) for _ in range(self.depth)]
            return
        self.dev = torch.device("cuda", device if device is not None else torch.cuda.current_device()) if self.gpu else None
        self.host_t = [torch.zeros(self.B, dtype=torch.float64, pin_memory=self.gpu) for _ in range(self.depth)]
        self.dev_t = [torch.zeros(self.B, dtype=torch.float64, device=self.dev) for _ in range(self.depth)] if self.gpu else None

    def submit(self, ll_band):
        """Start summing ll_band (B doubles, numpy) over ranks.  At most `depth` may be outstanding."""
        if len(self.pending) >= self.depth:
            raise RuntimeError("LoglikReducer: %d sums outstanding, call result() first" % len(self.pending))
        s = self.slot
        self.slot = (s + 1) % self.depth
        if not self.active:
            self.host[s][:] = ll_band
            self.pending.append((s, None))
            return
        import torch
        import torch.distributed as dist
        self.host_t[s].copy_(torch.from_numpy(np.ascontiguousarray(ll_band, dtype=np.float64)))
        if self.gpu:
            self.dev_t[s].copy_(self.host_t[s], non_blocking=True)
            work = dist.all_reduce(self.dev_t[s], op=dist.ReduceOp.SUM, async_op=True)
        else:
            work = dist.all_reduce(self.host_t[s], op=dist.ReduceOp.SUM, async_op=True)
        self.pending.append((s, work))

    def result(self):
        """The oldest outstanding sum -> (B,) numpy (a copy)."""
        if not self.pending:
            raise RuntimeError("LoglikReducer: nothing outstanding")
        s, work = self.pending.pop(0)
        if not self.active:
            return self.host[s].copy()
        work.wait()
        if self.gpu:
            self.host_t[s].copy_(self.dev_t[s])        # blocking D2H: orders after the collective on the current stream
        return self.host_t[s].numpy().copy()

    def drain(self):
        out = []
        while self.pending:
            out.append(self.result())
        return out


def barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
