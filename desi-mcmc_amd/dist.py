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


def strip_edges(H, world, row_cost=None, align=TILE_ROWS):
    """The world + 1 row edges of the strips: equal tile rows (strip_rows) without `row_cost`; with it -- the cost of every
    `align`-row band of the frame, e.g. strip_cost_from_tiles() of a whole-frame render's measured tile durations -- the
    cut that evens out the strips' summed cost (each strip at least one band).  Edges are multiples of `align`, the last
    one H.  Every rank must cut from the SAME numbers (measure on one rank and broadcast, or use an estimate that does not
    depend on the rank)."""
    if row_cost is None:
        return [strip_rows(H, world, r, align)[0] for r in range(world)] + [H]
    cost = np.asarray(row_cost, dtype=np.float64)
    nb = (H + align - 1) // align
    if cost.shape != (nb,):
        raise ValueError("row_cost must hold one value per %d-row band of the frame (%d)" % (align, nb))
    if world > nb:
        raise ValueError("%d strips for %d bands of %d rows" % (world, nb, align))
    cum = np.concatenate([[0.0], np.cumsum(np.maximum(cost, 0.0) + 1e-12 * max(float(cost.max()), 1.0))])
    edges = [0]
    for r in range(1, world):
        k = int(np.searchsorted(cum, cum[-1] * r / world))
        # the nearer of the two band edges around the ideal cut, leaving every strip a band
        if k > 0 and abs(cum[k - 1] - cum[-1] * r / world) <= abs(cum[min(k, nb)] - cum[-1] * r / world):
            k -= 1
        k = min(max(k, edges[-1] + 1), nb - (world - r))
        edges.append(k)
    edges.append(nb)
    return [min(e * align, H) for e in edges]


def strip_cost_from_tiles(tile_cost, B, nty, ntx, tile_rows, align=TILE_ROWS, H=None):
    """per-`align`-row cost of the frame from per-tile costs laid out [band][tile row][tile column] (the render's
    measured tile durations -- ImageSet.tile_timing, whose row i is TILE i whatever the launch order -- or the binning pass's
    estimates): a tile row's cost spread over the bands of `align` rows it covers.  With H, exactly the ceil(H / align) bands
    strip_edges asks for (a frame whose height is no multiple of tile_rows ends in a partial tile row)."""
    c = np.asarray(tile_cost, dtype=np.float64).reshape(B, nty, ntx).sum(axis=(0, 2))
    per = max(tile_rows // align, 1)
    out = np.repeat(c / per, per)
    return out if H is None else out[: (int(H) + align - 1) // align]


def agree_on_edges(edges, src=0):
    """every rank gets rank `src`'s strip edges (a measured cost differs from rank to rank; the cut must not): one broadcast of
    world + 1 integers.  No process group: the edges as they are."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [int(e) for e in edges]
    t = torch.tensor([int(e) for e in edges], dtype=torch.int64)
    if dist.get_backend() == "nccl":
        t = t.cuda(torch.cuda.current_device())
    dist.broadcast(t, src=src)
    return [int(e) for e in t.cpu().tolist()]


def field_shard(n_fields, world, rank):
    """Indices of the fields rank owns (round-robin: equal counts when world divides n_fields)."""
    return list(range(rank, n_fields, world))


class SourceDeal(object):
    """The sources of ONE Gibbs chain dealt to the ranks: given the photon split, the
    per-source updates (Source.resample_fluxes / resample_location, CelestePy/sources.py:308-349) are
    independent of each other, so rank r updates the sources s = r (mod world) -- a round-robin deal keeps
    stars, galaxies and bright sources evenly spread -- and the new rows are exchanged with ONE all-gather
    per merge (10 000 x 11 doubles per sweep at config 5: 880 KB over xGMI).  The merged arrays are identical
    on every rank, to the bit, and identical to what a single rank computes (the per-chain random streams
    do not depend on which other chains run beside them).  Every rank runs the whole photon split here
    (`kind` = "replicated"); StripDeal partitions that too."""

    kind = "replicated"

    def __init__(self, S, world=1, rank=0, device=None, owner=None, solo=False):
        self.S, self.world, self.rank = int(S), int(world), int(rank)
        # solo: this process plays rank `rank` of `world` WITHOUT a process group (bench.py --as-rank k --of N: one rank's step
        # timed on one GPU).  The exchanges return this rank's own contribution in every rank's place -- the numbers of the
        # other ranks' sources then go stale, which the timing of this rank's work does not depend on.
        self.solo = bool(solo)
        if not 0 <= self.rank < self.world:
            raise ValueError("rank %d outside a world of %d" % (rank, world))
        self.device = device
        # owner[s] = the rank that updates source s
        self.owner = (np.arange(self.S) % self.world) if owner is None else np.asarray(owner, dtype=np.int64)
        if self.owner.shape != (self.S,) or (self.S and (self.owner.min() < 0 or self.owner.max() >= self.world)):
            raise ValueError("owner must name a rank for every source")
        self.rows_of = [np.nonzero(self.owner == r)[0] for r in range(self.world)]
        self.mine = self.rows_of[self.rank]                               # indices of this rank's sources
        self.mask = np.zeros(self.S, dtype=bool)
        self.mask[self.mine] = True
        self.per_rank = max([r.size for r in self.rows_of] + [1])         # rows every rank contributes (padded)
        self._bufs = {}                                                   # (n, k) -> persistent pinned / device tensors of _gather

    def chain_ids(self):
        """cel_slice_locations' chain_ids: a source's own index where it is this rank's, -1 elsewhere"""
        return np.where(self.mask, np.arange(self.S), -1).astype(np.int32)

    def _gather(self, send):
        """(n, k) of this rank -> (world, n, k) of every rank (one all_gather_into_tensor).  Under RCCL the send / receive
        tensors and their pinned host mirrors are allocated once per shape and the copies are asynchronous on the current
        stream: the rows themselves are formed on the host (the flux conditionals' Gamma draws are host-side), so one
        pinned H2D of this rank's rows (110 KB at config 5 on 8 ranks) is inherent; nothing is staged through pageable memory."""
        import torch
        import torch.distributed as dist
        send = np.ascontiguousarray(send, dtype=np.float64)
        if self.solo:
            return np.broadcast_to(send, (self.world,) + send.shape).copy()
        if dist.get_backend() != "nccl":
            t = torch.from_numpy(send)
            recv = torch.empty((self.world * t.shape[0], t.shape[1]), dtype=t.dtype)      # rank-major concatenation
            dist.all_gather_into_tensor(recv, t)
            return recv.numpy().reshape(self.world, t.shape[0], t.shape[1])
        key = send.shape
        buf = self._bufs.get(key)
        if buf is None:
            dev = torch.device("cuda", self.device if self.device is not None else torch.cuda.current_device())
            buf = (torch.empty(key, dtype=torch.float64, pin_memory=True), torch.empty(key, dtype=torch.float64, device=dev),
                   torch.empty((self.world * key[0], key[1]), dtype=torch.float64, device=dev),
                   torch.empty((self.world * key[0], key[1]), dtype=torch.float64, pin_memory=True))
            self._bufs[key] = buf
        h_send, d_send, d_recv, h_recv = buf
        h_send.numpy()[...] = send
        d_send.copy_(h_send, non_blocking=True)
        dist.all_gather_into_tensor(d_recv, d_send)
        h_recv.copy_(d_recv, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        return h_recv.numpy().reshape(self.world, key[0], key[1]).copy()

    def merge(self, arr):
        """arr (S, k) with this rank's rows up to date -> (S, k) with every row taken from its owner"""
        arr = np.ascontiguousarray(arr, dtype=np.float64)
        if arr.ndim != 2 or arr.shape[0] != self.S:
            raise ValueError("merge takes an (S, k) array")
        if self.world == 1 or self.solo:
            return arr.copy()                 # (solo: the other ranks' rows stay as they are)
        send = np.zeros((self.per_rank, arr.shape[1]))
        send[:self.mine.size] = arr[self.mine]
        got = self._gather(send)
        out = np.empty_like(arr)
        for r in range(self.world):
            out[self.rows_of[r]] = got[r, :self.rows_of[r].size]
        return out

    def rank_sum(self, x):
        """sum over ranks of a small vector, in rank order on every rank (identical bits everywhere)"""
        x = np.atleast_1d(np.asarray(x, dtype=np.float64))
        if self.world == 1:
            return x.copy()
        got = self._gather(x.reshape(1, -1))
        out = np.zeros(x.shape[0])
        for r in range(self.world):
            out += got[r, 0]
        return out


class StripDeal(SourceDeal):
    """ONE chain partitioned over the ranks as SURVEY 8e prescribes for config 5: "per-source conditional updates ... are
    embarrassingly parallel over sources; the photon split is per-pixel and needs complete per-pixel source lists, so it
    uses the same spatial partition."  The frame is cut into tile-aligned row strips (strip_rows); rank r owns the sources
    whose (initial) pixel row lies in strip r, and holds the images on the WINDOW = its strip plus a halo as tall as its own
    sources' boxes reach (cel_images_set_window), so that (a) every pixel of an own source's box sees its complete list of
    sources -- every rank holds the whole catalogue, the binning cuts it to the window -- and the own sources' sample
    patches are complete, and (b) the split costs each rank window / frame of the whole.  A pixel's draws are keyed by its
    FULL-FRAME index and the source index, so whoever splits a pixel draws the same photons.  The sky photons of the strip
    rows only are counted (cel_images_set_noise_rows) and the strips' sums added over the ranks for the sky level's Gamma
    conditional (CelestePy/models.py:155-160); the trace's log-likelihood is the strips' sum likewise.
    The window's row origin enters the pixel arithmetic (positions are window-relative), so this chain agrees with the
    single-rank one to rounding (1e-12 on a sweep's fluxes), not bit for bit as the replicated deal does."""

    kind = "strips"

    def __init__(self, rows, H, world=1, rank=0, halo=0, device=None, edges=None, solo=False, align=TILE_ROWS):
        rows = np.asarray(rows, dtype=np.float64)
        if int(world) > -(-int(H) // TILE_ROWS):
            raise ValueError("StripDeal: %d ranks for a frame of %d rows = %d tile rows: a rank would own no rows"
                             % (world, H, -(-int(H) // TILE_ROWS)))
        # edges: the world + 1 strip edges (strip_edges: equal tile rows, or evened out by a measured cost); the same on every rank
        # align: strip edges (when not given) and the halo are multiples of it; 64 -- the default layout's tile height -- lets a
        # rank add its strip's log-likelihood from a render of its whole window (celeste_mcmc.strip_gibbs_field)
        if edges is None:
            edges = strip_edges(H, world, align=align if int(world) <= -(-int(H) // align) else TILE_ROWS)
        edges = [int(e) for e in edges]
        if len(edges) != world + 1 or edges[0] != 0 or edges[-1] != int(H) or any(b <= a for a, b in zip(edges, edges[1:])):
            raise ValueError("StripDeal: edges must be %d increasing rows from 0 to %d" % (world + 1, H))
        if any(e % TILE_ROWS for e in edges[:-1]):
            # the halo below is rounded to whole tiles RELATIVE to the edges: an edge off the tile grid would put the window,
            # and with it every pixel's tile, off the grid the split's and the render's tiles are cut on
            raise ValueError("StripDeal: strip edges must be multiples of %d rows (dist.strip_edges gives such edges): %s"
                             % (TILE_ROWS, edges))
        ends = np.array(edges[1:])
        owner = np.minimum(np.searchsorted(ends, np.clip(np.floor(rows), 0, H - 1), side="right"), world - 1)
        SourceDeal.__init__(self, rows.shape[0], world, rank, device=device, owner=owner, solo=solo)
        self.H = int(H)
        self.edges = edges
        self.strip = (edges[rank], edges[rank + 1])
        halo = int(-(-int(halo) // align) * align)                       # whole tiles
        self.window = (max(0, self.strip[0] - halo), min(self.H, self.strip[1] + halo))

    def noise_rows(self):
        """the strip's rows inside the window (window-relative): what the split's noise sums count"""
        return self.strip[0] - self.window[0], self.strip[1] - self.window[0]

    def boxes_cut(self, boxes, status):
        """boxes (B, S, 4) = y0, y1, x0, x1 relative to the window, status (B, S) -> how many of this rank's source boxes the
        window cuts (a box may end at the frame's edge)"""
        y0, y1 = boxes[:, self.mine, 0], boxes[:, self.mine, 1]
        has = status[:, self.mine] > 0
        top_cut = has & (y0 <= 0) & (self.window[0] > 0)
        bot_cut = has & (y1 >= self.window[1] - self.window[0]) & (self.window[1] < self.H)
        return int(top_cut.sum() + bot_cut.sum())

    def check_boxes(self, boxes, status, extra=None):
        """COLLECTIVE: every rank calls it in every sweep.  The ranks' cut counts are summed (riding with `extra`, a small
        vector the caller wants summed anyway) and EVERY rank raises when any rank's window cuts one of its boxes -- a rank
        raising alone would leave the others blocked in the sweep's next collective.
        -> the rank sum of `extra` (or None)"""
        cut = self.boxes_cut(boxes, status)
        vec = np.append(np.atleast_1d(np.asarray(extra, dtype=np.float64)) if extra is not None else np.zeros(0), float(cut))
        tot = self.rank_sum(vec)
        if tot[-1] > 0:
            raise RuntimeError("StripDeal: %d source boxes reach beyond their rank's window (%d on rank %d, window rows %s): "
                               "enlarge the halo" % (int(tot[-1]), cut, self.rank, self.window))
        return tot[:-1] if extra is not None else None


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


class _DeviceDoubles(object):
    """n doubles at a raw device address, for torch.as_tensor (zero copy through __cuda_array_interface__)"""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f8", "data": (int(ptr), False), "version": 2}


def _fetch_device_doubles(ptr, n):
    import torch
    return torch.as_tensor(_DeviceDoubles(ptr, n), device="cuda").cpu().numpy()


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

    def submit_device(self, image_sets):
        """The same for sums that are ALREADY on the device: the per-band log-likelihoods the last render of each of
        `image_sets` left in the library's memory (ImageSet.loglik_device_ptr; their renders have returned, i.e. their
        stream was synchronised).  Under RCCL the sums are added into this reducer's device slot by device-to-device
        operations and all-reduced from there: nothing crosses PCIe before the collective.  Without a GPU process group
        (gloo rigs, one rank) the values are fetched and take submit()'s path."""
        image_sets = list(image_sets)
        if not (self.active and self.gpu):
            tot = np.zeros(self.B)
            for iset in image_sets:
                tot += _fetch_device_doubles(iset.loglik_device_ptr(), self.B)
            return self.submit(tot)
        if len(self.pending) >= self.depth:
            raise RuntimeError("LoglikReducer: %d sums outstanding, call result() first" % len(self.pending))
        import torch
        import torch.distributed as dist
        s = self.slot
        self.slot = (s + 1) % self.depth
        slot = self.dev_t[s]
        for k, iset in enumerate(image_sets):
            src = torch.as_tensor(_DeviceDoubles(iset.loglik_device_ptr(), self.B), device=self.dev)
            if k == 0:
                slot.copy_(src)
            else:
                slot.add_(src)
        # the library's buffers are rewritten by the next render on ITS stream: the device-to-device copies above (a few
        # microseconds on torch's stream) must have run before this call returns; the collective itself stays asynchronous
        torch.cuda.current_stream(self.dev).synchronize()
        work = dist.all_reduce(slot, op=dist.ReduceOp.SUM, async_op=True)
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


def device_identity(local):
    """what tells two GPUs apart on one node: the HIP device's UUID and PCI address as far as torch exposes them; "cpu:<pid>"
    without a GPU; "unknown:<pid>" when the runtime exposes neither (never equal between two ranks: the roll call cannot prove
    a shared device then, and must not invent one)"""
    import torch
    if not torch.cuda.is_available():
        return "cpu:%d" % os.getpid()
    p = torch.cuda.get_device_properties(local)
    parts = []
    uuid = getattr(p, "uuid", None)
    if uuid is not None and str(uuid).replace("0", "").replace("-", "").strip():        # (an all-zero UUID says nothing)
        parts.append(str(uuid))
    pci = [getattr(p, k, None) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id")]
    if any(v is not None for v in pci):
        parts.append("pci:%s:%s:%s" % tuple("?" if v is None else v for v in pci))
    return "/".join(parts) if parts else "unknown:%d" % os.getpid()


def roll_call(expected, local=0, allow_shared_devices=False):
    """First thing an N-rank job does with its process group: an all-reduce of ones (how many ranks the COLLECTIVE reaches -- not
    what the environment claims) and an all-gather of every rank's device identity.  Raises RuntimeError -- on every rank alike,
    so the launcher exits non-zero -- when the collective sees another number of ranks than `expected`, or when two ranks sit
    on the same device (unless allow_shared_devices: the gloo rehearsals on a box with fewer GPUs than ranks).
    -> dict(ranks_seen_by_collective, device_uuid=[per rank], backend).  Without a process group: one rank, its own device."""
    import torch
    import torch.distributed as td
    me = device_identity(local)
    if not (td.is_available() and td.is_initialized()):
        seen, ids, backend = 1, [me], "none"
    else:
        backend = td.get_backend()
        one = torch.ones(1, dtype=torch.float64)
        if backend == "nccl":
            one = one.cuda(local)
        td.all_reduce(one, op=td.ReduceOp.SUM)
        seen = int(round(float(one.item())))
        ids = [None] * td.get_world_size()
        td.all_gather_object(ids, me)
    out = {"ranks_seen_by_collective": seen, "device_uuid": ids, "backend": backend}
    if seen != expected or len(ids) != expected:
        raise RuntimeError("roll call: the collective reached %d rank(s) (%d identities), the job was started for %d" % (seen, len(ids), expected))
    if len(set(ids)) != len(ids) and not allow_shared_devices:
        raise RuntimeError("roll call: two ranks share a device: %s" % ids)
    return out


def barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
