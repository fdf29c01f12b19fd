#!/usr/bin/env python3
"""bench.py -- the headline metric of BASELINE.json on MI355X, plus the two multi-GPU configurations.

Default workload (config.workload = mixed10k_2048): BASELINE.json configs[2], the configuration the
metric is quoted on -- 10 000 mixed star/galaxy sources x 5 bands x 2048^2, synthetic.
One step = one full-field log-likelihood evaluation with everything resident in HBM:
    k_prep (WCS, galaxy shape matrix, bounding box per source x band) -> k_bin (tile lists)
    -> k_render (model images + fused Poisson term) -> k_reduce -> 5 doubles to the host
    [-> one all-reduce of the 5 doubles across ranks when N > 1].
value = source-pixel evaluations per second, whole job (sum over ranks / max-over-ranks time);
ms_per_step = full-field log-lik latency.  N > 1: one field per rank ("weak"), one collective per
step (RCCL).  Other render workloads: stars1k_512 (configs[1]), stars10k_2048, stars2k_4096.

--workload fields8_2048  configs[3] stand-in (the Stripe-82 data is not in the reference tree): 8
    synthetic 10k-source fields dealt to the ranks (dist.field_shard), every step scores every field;
    total work fixed ("strong"), one all-reduce of the per-band sums per step.
--workload gibbs10k      configs[4]: Gibbs sweeps (photon split + sky level, flux Gamma conditionals,
    lock-step slice sampling of every source's location) over the 10k-source field; one step = one
    sweep; value = source updates (samples) per second.  N > 1: one independent chain per GPU over
    the same field (the standard way to parallelise MCMC; "weak"), the chains' field log-likelihoods
    all-reduced every sweep.

`python bench.py --gpus N` without a launcher starts its own N ranks (torch.distributed.run child)
before touching the GPU, and refuses to run on fewer than N GPUs.  Every run opens with a roll call
(dist.roll_call): the collective counts its ranks and gathers their device identities; another count
than --gpus, or two ranks on one device, exits non-zero before a number is printed.

The JSON line's `roofline` keeps the contract's fields (bound "hbm", algorithmic bytes over the dominant
kernel's live-measured duration, PMC traffic from the committed counter passes of this library) and nests
`roofline.binding`: the roof that actually binds these kernels -- fp64 vector issue -- with achieved,
peak, frac, the useful (recurrence) share and VALU busy.

Usage: python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--kernel direct|recurrence]
"""
import argparse
import json
import os
import sys
import time

# The GPU boxes show every core of the host (256) and grant a share of 16: thread pools sized by the core count (OpenMP,
# BLAS, torch's intra-op pool) then oversubscribe the share and the cgroup throttles the whole process -- the sweeps' host
# side (launches, flag reads) slowed by 10-25 % in such runs.  Nothing here needs more than the share.
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "NUMEXPR_NUM_THREADS"):
    os.environ.setdefault(_v, "16")

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_VALU_PEAK_TF = 78.6     # 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz
FP64_LANE_OPS_PEAK = 3.93e13 # the same in lane-instructions per second (an fma counts once)
FLOP_PER_GAUSS = 35.0        # SURVEY 8d accounting: 10 arithmetic + exp counted as 25

# Per-launch PMC figures of the dominant kernel.  Counters cannot be read from inside this process (rocprofv3
# runs in its own passes: tools/profile_r06.sh), so they are READ AT RUN TIME from the committed summaries of
# exactly this command -- and only when the summary was taken with the library that is loaded now (its sha256 is
# stored in the profile): after any kernel change without a re-profile the fields print null instead of going stale.
#   traffic = 2 * FETCH_SIZE + WRITE_SIZE (gfx950 correction, re-calibrated with tools/calib_traffic.hip)
#   valu    = SQ_INSTS_VALU (wave-instructions), busy = waves per SIMD * SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES
PMC_PROFILES = {   # (workload, kernel, tail_log, layout) -> committed summary
    ("mixed10k_2048", "recurrence", 24.0, 1): "profiles/r06_final_pmc.json",
    ("stars10k_2048", "recurrence", 24.0, 1): "profiles/r06_stars_pmc.json",
    ("stars1k_512", "recurrence", 24.0, 1): "profiles/r06_stars1k_pmc.json",
}
GIBBS_PMC_PROFILE = "profiles/r06_aux_pmc.json"     # bench.py --workload gibbs10k under the counters


def library_sha256():
    import hashlib
    from desi_mcmc_amd import _lib
    with open(_lib.LIB_PATH, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


def load_pmc(key, kernel="k_render"):
    """-> dict(traffic, valu_insts, valu_busy, source) from the committed summary, or dict(stale=reason)"""
    rel = PMC_PROFILES.get(key)
    if rel is None:
        return None
    path = os.path.join(ROOT, rel)
    if not os.path.exists(path):
        return {"stale": "%s not present" % rel}
    prof = json.load(open(path))
    have, want = prof.get("library_sha256"), library_sha256()
    if have != want:
        return {"stale": "%s was taken with library sha256 %s..., loaded is %s...: re-run tools/profile_r06.sh"
                         % (rel, str(have)[:12], want[:12])}
    ks = [k for k in prof["kernels"] if kernel in k or "k_small_stars" in k]
    if not ks:
        return {"stale": "%s holds no %s launches" % (rel, kernel)}
    name = sorted(ks, key=lambda k: -prof["kernels"][k].get("SQ_INSTS_VALU", {}).get("last", 0.0))[0]
    c = prof["kernels"][name]
    last = lambda name: c[name]["last"]      # noqa: E731  -- the last launch: a timed-region step
    waves_per_simd = 3.0 if ("k_render_stars" in name or "k_small_stars" in name) else 2.0       # resident waves per SIMD (kernel-resource-usage)
    return {"traffic": 2.0 * last("FETCH_SIZE") * 1024.0 + last("WRITE_SIZE") * 1024.0, "valu_insts": last("SQ_INSTS_VALU"),
            "valu_busy": waves_per_simd * last("SQ_ACTIVE_INST_VALU") / last("SQ_WAVE_CYCLES"), "source": rel}


def load_gibbs_pmc():
    """HBM traffic of ONE round of the Gibbs location step (its one or two likelihood launches) from the committed counter
    passes of `bench.py --workload gibbs10k`, under the same library-hash check as load_pmc
    -> dict(traffic, source) or dict(stale=reason)"""
    path = os.path.join(ROOT, GIBBS_PMC_PROFILE)
    if not os.path.exists(path):
        return {"stale": "%s not present" % GIBBS_PMC_PROFILE}
    prof = json.load(open(path))
    have, want = prof.get("library_sha256"), library_sha256()
    if have != want:
        return {"stale": "%s was taken with library sha256 %s..., loaded is %s...: re-run tools/profile_r06.sh"
                         % (GIBBS_PMC_PROFILE, str(have)[:12], want[:12])}
    ks = [k for k in prof["kernels"] if "k_patch_ll_nz" in k or "k_patch_ll_hw<0" in k]
    if not ks:
        return {"stale": "%s holds no likelihood launches" % GIBBS_PMC_PROFILE}
    rounds = max(prof["kernels"][k]["FETCH_SIZE"]["launches"] for k in ks)       # every round launches the photon-list kernel
    tot = valu = 0.0
    for k in ks:
        c = prof["kernels"][k]
        tot += (2.0 * c["FETCH_SIZE"]["mean"] + c["WRITE_SIZE"]["mean"]) * 1024.0 * c["FETCH_SIZE"]["launches"]
        valu += c["SQ_INSTS_VALU"]["mean"] * c["SQ_INSTS_VALU"]["launches"] if "SQ_INSTS_VALU" in c else 0.0
    split = [k for k in prof["kernels"] if "k_photon_split_hw" in k]
    split_valu = prof["kernels"][split[0]].get("SQ_INSTS_VALU", {}).get("mean") if split else None
    return {"traffic": tot / rounds, "valu_insts": valu / rounds, "split_valu_insts": split_valu, "source": GIBBS_PMC_PROFILE, "kernels": ks}


def _issue_binding(valu_insts, kernel_ms, source, split_valu=None, split_ms=None):
    """roofline.binding of a launch from its counted VALU wave-instructions: executed lane-instructions per second against the
    data-sheet fp64 issue rate (the roof the Gaussian evaluators are on; bench.py keeps `bound: "hbm"` as the contract's)"""
    ok = bool(valu_insts) and kernel_ms and kernel_ms > 0
    out = {"bound": "fp64_valu_issue", "achieved": valu_insts * 64.0 / (kernel_ms * 1e-3) if ok else None, "peak": FP64_LANE_OPS_PEAK,
           "unit": "fp64 VALU lane-instructions/s", "frac": valu_insts * 64.0 / (kernel_ms * 1e-3) / FP64_LANE_OPS_PEAK if ok else None,
           "pmc_source": source}
    if split_valu and split_ms and split_ms > 0:
        out["k_photon_split_hw_frac"] = split_valu * 64.0 / (split_ms * 1e-3) / FP64_LANE_OPS_PEAK
    return out


CPU_THREADS_MAX = 16         # the GPU box's CPU share for one GPU

RENDER_WORKLOADS = ("mixed10k_2048", "stars1k_512", "stars10k_2048", "stars2k_4096", "stamp51")


# ---- launcher ------------------------------------------------------------------------------------------
def visible_gpus():
    """Number of HIP devices, WITHOUT initialising the GPU runtime in this process (a child does it)."""
    import subprocess
    code = "import torch; print(torch.cuda.device_count())"
    try:
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
        return int(out.stdout.strip().splitlines()[-1])
    except Exception:
        return 0


def self_launch(n, argv, port=0):
    """Run `python -m torch.distributed.run --nproc-per-node n bench.py <argv>` as a child process and
    return its exit code.  Fails loudly (non-zero, message on stderr) when fewer than n GPUs are
    visible -- never a silent 1-GPU number.  CEL_BENCH_BACKEND=gloo is the documented rehearsal mode
    (ranks may share GPUs, the collective runs on the host); it is exempt from the device-count check."""
    import socket
    import subprocess
    rehearsal = os.environ.get("CEL_BENCH_BACKEND") == "gloo"
    have = visible_gpus()
    if have < n and not rehearsal:
        sys.stderr.write("bench.py: --gpus %d requested but only %d GPU(s) visible; refusing to run on fewer "
                         "(set CEL_BENCH_BACKEND=gloo to rehearse the multi-rank flow on shared GPUs)\n" % (n, have))
        return 2
    if not port:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def dry_run(args):
    """CEL_BENCH_DRYRUN=1: rendezvous + the collectives of the timed region on synthetic numbers, no
    GPU and no metric -- what the CPU test of the self-launch path runs."""
    from desi_mcmc_amd import dist
    rank, world, local = dist.init_from_env(backend=os.environ.get("CEL_BENCH_BACKEND", "gloo"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    # the same first act as the real job: how many ranks the collective reaches, who sits on which device
    call = dist.roll_call(args.gpus, local, allow_shared_devices=True)
    red = dist.LoglikReducer(5, depth=2)
    red.submit(np.full(5, float(rank + 1)))
    got = red.drain()[-1]
    shard = dist.field_shard(8, world, rank)
    dist.barrier()
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "ranks": world, "allreduce_check": float(got[0]),
                          "expected": world * (world + 1) / 2.0, "fields_of_rank0": shard,
                          "ranks_seen_by_collective": call["ranks_seen_by_collective"], "device_uuid": call["device_uuid"],
                          # rank 0's job in the N-rank weak-scaling run IS the --gpus 1 headline: same function, same field
                          "rank0_job": job_of_rank(args, 0, world), "n1_job": job_of_rank(args, 0, 1)}))
    if world > 1:
        import torch.distributed as td
        td.destroy_process_group()


def field_seed(rank, strong):
    """weak scaling: one field per rank, same population, another seed per rank (rank 0's is the 1-GPU headline's field);
    strong: every rank builds the SAME field"""
    return 42 + (0 if strong else 1000 * rank)


def job_of_rank(args, rank, world):
    """what rank `rank` of a `world`-rank job runs: the function, the workload and the field it builds.  At any N the weak job's
    rank 0 is the `--gpus 1` job (tests/test_dist_gloo.py asserts it on the dry run): the N = 1 point of the scaling curve is the
    headline, not another code path"""
    kind = "gibbs" if args.workload == "gibbs10k" else ("fields" if args.workload.startswith("fields") else "render")
    strong = (args.scaling == "strong") and world > 1
    return {"function": "run_" + kind, "workload": args.workload, "field_seed": field_seed(rank, strong),
            "scaling": "strong" if strong else "weak"}


# ---- shared pieces -----------------------------------------------------------------------------------
class Timer(object):
    """W untimed warm-up steps, then exactly K timed steps bracketed by barrier + device sync on both sides"""

    def __init__(self, dist, torch):
        self.dist, self.torch = dist, torch

    def run(self, step, warmup, steps, after_warmup=None, finish=None, prime=0):
        # set-up, before the W warm-up steps: `prime` untimed steps (the same number on every rank: a step may
        # hold a collective), so that the GPU is at its running clocks whatever W is -- the first steps after
        # start-up ran ~5 % slow (1.54 against 1.47 ms per step with W = 3)
        for _ in range(prime):
            step()
        for _ in range(warmup):
            step()
        if after_warmup:
            after_warmup()
        self.dist.barrier()
        self.torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        if finish:
            finish()
        self.torch.cuda.synchronize()
        self.dist.barrier()
        return time.perf_counter() - t0


def reduce_over_ranks(torch, world, local, dt, sums):
    """max over ranks of the elapsed time, sum over ranks of the work counters"""
    if world == 1:
        return dt, list(sums)
    import torch.distributed as td
    t = torch.tensor([dt], dtype=torch.float64)
    s = torch.tensor(list(sums), dtype=torch.float64)
    if td.get_backend() == "nccl":
        t, s = t.cuda(local), s.cuda(local)
    td.all_reduce(t, op=td.ReduceOp.MAX)
    td.all_reduce(s, op=td.ReduceOp.SUM)
    return t.item(), [float(v) for v in s.cpu()]


def host_threads():
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    return max(1, min(avail, CPU_THREADS_MAX))


def cpu_baseline(field, nsample, orc):
    """The CPU oracle (kind "port": oracle/celeste_oracle.c, the restatement of the reference's path;
    the reference's own Cython does not build here) timed on BOUNDED samples of the SAME workload on
    the host's cores: (1) all threads, patch-accumulating render + log-lik of the first `nsample`
    sources in all bands; (2) the same on ONE thread for a quarter of them; (3) the reference's
    literal gen_model_image semantics -- a full H x W frame allocated, scaled and added per source
    (celeste.py:203-219) -- for 100 sources in one band, which is what makes the reference O(S H W)."""
    bands = field.bands.copy()
    for b in range(field.B):
        bands[b, 36] = orc.checked_radius(bands[b], field.images.band(b)[36])
    nthr = max(1, min(orc.max_threads(), host_threads()))

    def timed(ns, threads):
        sl = slice(0, ns)
        orc.set_threads(threads)
        t0 = time.perf_counter()
        lam, ll, st = orc.render_field(bands, field.H, field.W, field.src["type"][sl], field.src["radec"][sl],
                                       field.src["counts"][sl], field.src["shape"][sl], field.nelec)
        return st, time.perf_counter() - t0
    st, dt = timed(nsample, nthr)
    n1 = max(1, nsample // 4)
    st1, dt1 = timed(n1, 1)
    out = dict(value=st["n_srcpix"] / dt, unit="source-pixel evals/s", cores=nthr, kind="port",
               sample="first %d of %d sources, %d bands, %dx%d frame, %.2e source-px in %.2f s wall "
                      "(oracle/celeste_oracle.c, OpenMP over sources)"
                      % (nsample, field.S, field.B, field.H, field.W, st["n_srcpix"], dt),
               gauss_evals_per_s=st["n_gauss"] / dt,
               one_thread={"value": st1["n_srcpix"] / dt1, "cores": 1,
                           "sample": "first %d sources, %.2e source-px in %.2f s" % (n1, st1["n_srcpix"], dt1)})
    stars = np.nonzero(field.src["type"] == 0)[0][:100]
    if stars.size:
        orc.set_threads(nthr)
        b = min(2, field.B - 1)
        t0 = time.perf_counter()
        orc.gen_model_image_fullframe(bands[b], field.H, field.W, field.src["radec"][stars], field.src["counts"][stars, b])
        dtf = time.perf_counter() - t0
        out["reference_semantics_fullframe"] = {
            "s_per_source": dtf / stars.size, "sources": int(stars.size), "band": int(b),
            "note": "gen_model_image as the reference writes it (point sources; one zeros(H,W) + scale + add per "
                    "source, celeste.py:203-219): %.1f ms per source per band at %dx%d, i.e. ~%.0f s for this "
                    "workload's %d sources x %d bands" % (dtf / stars.size * 1e3, field.H, field.W,
                                                        dtf / stars.size * field.S * field.B, field.S, field.B)}
    orc.set_threads(nthr)
    return out



# ---- sustained runs: seconds, not milliseconds ------------------------------------------------------
class SmiSampler(object):
    """clock and power of the GPU sampled in a thread while a long leg runs: sysfs where readable (pp_dpm_sclk's starred
    level, hwmon power1_average / power1_input), `rocm-smi -c -P --json` otherwise; a box that offers neither yields []"""

    def __init__(self, device=0, period=0.5):
        import threading
        self.period, self.samples, self._stop = period, [], threading.Event()
        self._t0 = time.perf_counter()
        self._dev = self._find_sysfs(device)
        self._th = threading.Thread(target=self._run, daemon=True)

    @staticmethod
    def _find_sysfs(device):
        """the sysfs directory of HIP device `device`: by its PCI address (a box shows every card of the host under
        /sys/class/drm, the process sees one of them as device 0)"""
        try:
            import torch
            p = torch.cuda.get_device_properties(device)
            bdf = "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, p.pci_device_id)
            d = os.path.join("/sys/bus/pci/devices", bdf)
            return d if os.path.exists(os.path.join(d, "pp_dpm_sclk")) else None
        except Exception:
            return None

    def _read_sysfs(self):
        import glob
        out = {}
        try:
            for line in open(os.path.join(self._dev, "pp_dpm_sclk")):
                if "*" in line:
                    out["sclk_mhz"] = float(line.split(":")[1].strip().split("M")[0])
            for name in ("power1_average", "power1_input"):
                hw = glob.glob(os.path.join(self._dev, "hwmon", "hwmon*", name))
                if hw:
                    out["power_w"] = float(open(hw[0]).read()) / 1e6
                    break
        except Exception:
            return None
        return out or None

    @staticmethod
    def _read_smi():
        import subprocess
        try:
            r = subprocess.run(["rocm-smi", "-c", "-P", "--json"], capture_output=True, text=True, timeout=5)
            d = json.loads(r.stdout)
            card = d[sorted(d)[0]]
            out = {}
            for k, v in card.items():
                kl = k.lower()
                if "sclk" in kl and "mhz" in str(v).lower():
                    out["sclk_mhz"] = float(str(v).lower().replace("(", "").split("mhz")[0])
                if "power" in kl and "socket" in kl or "average graphics package power" in kl:
                    try:
                        out["power_w"] = float(v)
                    except ValueError:
                        pass
            return out or None
        except Exception:
            return None

    def _run(self):
        while not self._stop.is_set():
            smp = self._read_sysfs() if self._dev else None
            if smp is None:
                smp = self._read_smi()
            if smp:
                smp["source"] = "sysfs " + os.path.basename(self._dev) if self._dev else "rocm-smi (first card listed)"
                smp["t_s"] = round(time.perf_counter() - self._t0, 3)
                self.samples.append(smp)
            self._stop.wait(self.period)

    def __enter__(self):
        self._th.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        self._th.join(timeout=10)


def sustained_leg(torch, step, seconds=None, steps=None, bucket_s=0.5, sync_every=1):
    """back-to-back steps for `seconds` of wall time (or exactly `steps`), ms per step in buckets of bucket_s, SMI samples
    beside them: what the clocks do over a run longer than the contract's 20 steps"""
    buckets, n_tot = [], 0
    with SmiSampler() as smi:
        torch.cuda.synchronize()
        t_start = t_b = time.perf_counter()
        n_b = 0
        while True:
            step()
            n_b += 1
            n_tot += 1
            now = time.perf_counter()
            done = (steps is not None and n_tot >= steps) or (seconds is not None and now - t_start >= seconds)
            if now - t_b >= bucket_s or done:
                torch.cuda.synchronize()
                now = time.perf_counter()
                buckets.append({"t_s": round(now - t_start, 3), "steps": n_b, "ms_per_step": (now - t_b) / n_b * 1e3})
                t_b, n_b = now, 0
            if done:
                break
        total = time.perf_counter() - t_start
    ms = [b["ms_per_step"] for b in buckets]
    return {"wall_s": total, "steps": n_tot, "ms_per_step": total / n_tot * 1e3, "bucket_s": bucket_s,
            "ms_per_step_first_bucket": ms[0], "ms_per_step_last_bucket": ms[-1], "ms_per_step_min_bucket": min(ms), "ms_per_step_max_bucket": max(ms),
            "buckets": buckets, "smi": smi.samples,
            "smi_note": "sclk / socket power sampled every 0.5 s in a thread (sysfs, else rocm-smi); [] when the box exposes neither"}

# ---- render workloads (configs[1], configs[2] and the star-only fields) ---------------------------------
def run_render(args, env):
    torch, cel, dist, synth, _lib = env["torch"], env["cel"], env["dist"], env["synth"], env["_lib"]
    rank, world, local, ctx = env["rank"], env["world"], env["local"], env["ctx"]
    # weak: one field per rank (same population, different seed).  strong: every rank builds the SAME
    # field (the catalogue is small and replicated) and keeps only its row strip of the pixels.
    strong = (args.scaling == "strong") and world > 1
    field = synth.SyntheticField.from_config(ctx, args.workload, seed=field_seed(rank, strong))
    full_stats = None
    if strong:
        field.images.render(field.sources, loglik=False)
        full_stats = field.images.stats()                  # the whole field's work: what every step of the job does
        # strip edges: every rank holds the whole field at this point; the cut that evens out the whole-frame render's measured
        # tile durations is taken from rank 0's measurement (dist.agree_on_edges) -- or equal tile rows (--strip-cut equal)
        align = 64 if args.layout == 1 else dist.TILE_ROWS
        row_cost = measured_row_cost(ctx, _lib, field.images, field.sources, align) if args.strip_cut == "measured" else None
        edges = dist.agree_on_edges(dist.strip_edges(field.H, world, row_cost, align=align if world <= -(-field.H // align) else dist.TILE_ROWS))
        y0, y1 = edges[rank], edges[rank + 1]
        strip = cel.ImageSet(ctx, field.bands, max(y1 - y0, 1), field.W, nelec=field.nelec[:, y0:max(y1, y0 + 1)])
        strip.set_window(y0, field.H)
        field.images = strip

    # the one collective: B per-band doubles summed over ranks.  Pipelined one step deep: the sum of
    # step k travels while step k+1 renders (the ranks' fields are independent chains; the global
    # log-likelihood is a diagnostic), and every sum is collected inside the timed region.
    reducer = dist.LoglikReducer(field.B, device=local, depth=2) if world > 1 else None
    last = {}

    def step():
        ll, llb = field.images.render(field.sources, loglik=True)
        if reducer is not None:
            reducer.submit_device([field.images])     # the sums go from the library's device memory straight into the collective
            if len(reducer.pending) > 1:
                llb = reducer.result()
        last["llb"] = llb

    def after_warmup():
        if reducer is not None:
            reducer.drain()
        # the dominant kernel is timed live over the whole timed region (HIP events attached to its dispatches); the three
        # small kernels around it are timed in a few extra steps afterwards (an event pair costs the host ~10 us per launch:
        # bracketing all four kernels made the step itself 3 % longer)
        ctx.profile(2)

    def finish():
        if reducer is not None:
            last["llb"] = reducer.drain()[-1]
    dt = Timer(dist, torch).run(step, args.warmup, args.steps, after_warmup, finish, prime=100)
    t_render, n_render, render_kernel = ctx.profile_render()
    ctx.profile(1)
    for _ in range(20):                 # untimed: prep / binning / reduction durations for `kernels_ms`
        field.images.render(field.sources, loglik=True)
    t_bin, _ = ctx.profile_get("bin")
    t_prep, _ = ctx.profile_get("prep")
    t_red, _ = ctx.profile_get("reduce")
    ctx.profile(False)
    stats = field.images.stats()
    if strong:   # total work is the one field's, whoever renders which rows: count it once (rank 0)
        stats = dict(full_stats) if rank == 0 else dict(full_stats, n_srcpix=0, n_gauss=0)
    dt_max, (n_srcpix_all, n_gauss_all) = reduce_over_ranks(torch, world, local, dt, [stats["n_srcpix"], stats["n_gauss"]])
    if rank != 0:
        return
    S, B, H, W, fg = synth.CONFIGS[args.workload]
    n_imgpix = B * H * W
    # algorithmic HBM bytes of one k_render launch (DESIGN.md "Measurement"):
    #   read nelec 8 B + write lambda 8 B per image pixel, + one 128-B record per (source, band)
    alg_bytes = 16.0 * n_imgpix + 128.0 * S * B
    if strong:   # rank 0's launch covers its strip of the pixels (and still reads every record)
        alg_bytes = 16.0 * B * (y1 - y0) * W + 128.0 * S * B
    achieved = alg_bytes / (t_render * 1e-3) / 1e9 if t_render > 0 else 0.0
    pmc = None if strong else load_pmc((args.workload, args.kernel, args.tail_log, args.layout))
    pmc_note = None
    if pmc is not None and "stale" in pmc:
        pmc_note, pmc = pmc["stale"], None
    out = {
        # BASELINE.json's metric; `value` is its first half, `ms_per_step` its second
        "metric": ("source-pixel evals/sec + full-field log-lik ms, %s sources x %d bands x %d^2"
                   % ("10k" if S == 10000 else str(S), B, H)) if H == W else
                  "source-pixel evals/sec + full-field log-lik ms, %d sources x %d bands x %dx%d" % (S, B, H, W),
        "value": n_srcpix_all * args.steps / dt_max,
        "unit": "source-pixel evals/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt_max / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong" if strong else "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": args.workload, "sources": S, "bands": B, "frame": [H, W],
                   "galaxy_fraction": fg, "kernel": args.kernel, "tail_log": args.tail_log,
                   "tile_layout": {1: "32x64 half-wave", 2: "16x128 quarter-wave"}.get(args.layout, "64x%d" % args.tile_rows),
                   "tile_order": args.tile_order,
                   "tile_order_note": {0: "index order", 1: "heaviest first by the previous step's measured tile durations "
                                       "(a launch-order hint only: every step redoes all of the work)",
                                       2: "heaviest first by the binning pass's estimate"}[args.tile_order],
                   "parallelism": ("1 field cut into %d row strips, 1 per GPU" if strong else "1 field per GPU, %d GPU(s)") % world
                                  + ", 1 all-reduce of %d doubles per step (overlapped with the next step's render)" % B,
                   "ranks": world, "collective_backend": env["backend"],
                   "ranks_seen_by_collective": env["roll_call"]["ranks_seen_by_collective"], "device_uuid": env["roll_call"]["device_uuid"]},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": pmc["traffic"] if pmc else None,
                     "traffic_source": pmc["source"] if pmc else pmc_note,
                     "kernel": render_kernel, "kernel_ms": t_render, "launches": n_render,
                     "algorithmic_bytes_per_launch": alg_bytes},
        # gauss_evals_per_s is put on the EVALUATED count below (extra_render_legs counts it); until then, and for the other
        # workloads, the nominal K x box-area count carries its own name
        "work": {"n_srcpix_per_step": n_srcpix_all, "n_gauss_per_step_nominal": n_gauss_all,
                 "gauss_evals_per_s_nominal": n_gauss_all * args.steps / dt_max,
                 "n_tile_entries": stats["n_tile_entries"]},
        "kernels_ms": {"k_prep": t_prep, "k_bin": t_bin, "k_render": t_render, "k_reduce": t_red},
        "loglik": float(np.sum(last["llb"])),
    }
    # the roof that actually binds the mixed field's kernel is the fp64 vector ALU.
    # frac = EXECUTED lane-instructions per second / the data-sheet issue rate; the 35-flop-per-nominal-
    # Gaussian figure of SURVEY 8d is kept only as an equivalent (the recurrence and the drop rule need
    # ~10x fewer operations than that accounting assumes, so it exceeds the peak).
    fp = {"peak": FP64_LANE_OPS_PEAK, "unit": "fp64 VALU lane-instructions/s",
          "equivalent_tflops_35flop_per_nominal_gauss": FLOP_PER_GAUSS * stats["n_gauss"] / (t_render * 1e-3) / 1e12 if t_render > 0 else 0.0}
    if pmc and t_render > 0:
        fp.update({"achieved": pmc["valu_insts"] * 64.0 / (t_render * 1e-3), "valu_wave_instructions_per_launch_pmc": pmc["valu_insts"],
                   "valu_busy_frac_pmc": pmc["valu_busy"], "pmc_source": pmc["source"]})
        fp["frac"] = fp["achieved"] / FP64_LANE_OPS_PEAK
    else:
        fp.update({"achieved": None, "frac": None})
    out["fp64_valu"] = fp
    # the contract's `bound` stays "hbm"; the roof that binds rides INSIDE `roofline` so that the driver's record keeps it
    # ("report against the binding roof").  useful_frac is filled in below once the evaluated count is known.
    out["roofline"]["binding"] = {"bound": "fp64_valu_issue", "achieved": fp["achieved"], "peak": FP64_LANE_OPS_PEAK,
                                  "unit": fp["unit"], "frac": fp["frac"], "useful_frac": None,
                                  "valu_busy_frac": fp.get("valu_busy_frac_pmc"), "pmc_source": fp.get("pmc_source") or pmc_note}
    if world == 1 and not strong and args.legs == "all":
        extra_render_legs(args, env, field, out)
        ev = out["work"].get("n_gauss_evaluated_per_step")
        if ev:
            # the honest rate: Gaussian-pixels the kernel really walks per second (the nominal count divides work the drop rule
            # never does by the same time)
            out["work"]["gauss_evals_per_s"] = ev * args.steps / dt_max
            out["work"]["evaluated_fraction_of_nominal"] = ev / n_gauss_all
            out["fp64_valu"]["equivalent_tflops_35flop_per_evaluated_gauss"] = FLOP_PER_GAUSS * ev / (t_render * 1e-3) / 1e12 if t_render > 0 else 0.0
            if pmc:
                # the recurrence's 2 mul + 1 add per evaluated Gaussian-pixel over everything the kernel executed
                out["roofline"]["binding"]["useful_frac"] = 3.0 * ev / (pmc["valu_insts"] * 64.0)
        if args.workload == "mixed10k_2048":
            out["secondary"] = secondary_legs(args, env, field)
            if args.sustained > 0:
                # the same step for seconds instead of the contract's K steps (24 ms at K = 20): never the headline
                ctx.profile(False)
                out["sustained"] = {"mixed10k_2048": sustained_leg(torch, lambda: field.images.render(field.sources, loglik=True), seconds=args.sustained)}
                out["sustained"]["gibbs10k_200"] = out["secondary"].pop("gibbs10k_200_sustained", None)
    if world == 1 and args.cpu_sample > 0:
        from oracle import oracle as orc      # cpu_baseline leg only
        out["cpu_baseline"] = cpu_baseline(field, min(args.cpu_sample, S), orc)
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out))


def extra_render_legs(args, env, field, out):
    """Untimed-by-the-contract extras, measured after the timed region on rank 0 of a 1-GPU run: the
    Gaussian-pixels the kernel really evaluates, the step with the catalogue uploaded from host
    memory every time (a chain changes it before every evaluation), and the reference-API call."""
    torch, cel, synth, _lib, ctx = env["torch"], env["cel"], env["synth"], env["_lib"], env["ctx"]
    n = max(3, min(args.steps, 20))
    if args.layout == 1 and args.kernel == "recurrence":
        try:
            ctx.set_option(_lib.CEL_OPT_TILE_TIMING, 1)
            field.images.render(field.sources, loglik=True)
            tt = field.images.tile_timing()
            comprows = float(np.sum(tt[:, 2] >> np.uint64(32)))
            # a kept component is walked over its row range on all 32 columns of the tile
            out["work"]["n_gauss_evaluated_per_step"] = comprows * 32.0
            out["work"]["n_gauss_evaluated_note"] = ("kept component-rows x 32 tile columns, counted by k_render_hw in an untimed "
                                                      "launch; the rest of the nominal K x box-area count is below eps * e^-T "
                                                      "on its tile and skipped (CEL_OPT_TAIL_LOG)")
        finally:
            ctx.set_option(_lib.CEL_OPT_TILE_TIMING, 0)
    if args.tail_log == _lib.TAIL_LOG_DEFAULT and args.kernel == "recurrence":
        # the documented fast preset (T = 20: north_star's 1e-6, not the 1e-10 the default is tested to);
        # a second measurement, never the headline
        ctx.set_tail_log("fast")
        try:
            for _ in range(2):
                llf, _ = field.images.render(field.sources, loglik=True)
            ctx.profile(True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                llf, _ = field.images.render(field.sources, loglik=True)
            torch.cuda.synchronize()
            dtf = time.perf_counter() - t0
            tk = ctx.profile_render()[0]
            ctx.profile(False)
            out["tail_log_fast_preset"] = {"tail_log": _lib.TAIL_LOG_FAST, "ms_per_step": dtf / n * 1e3, "k_render_ms": tk,
                                           "loglik_rel_diff_vs_default": float(abs(llf - out["loglik"]) / abs(out["loglik"])),
                                           "note": "CEL_OPT_TAIL_LOG = 20: components below eps * e^-20 on a tile are skipped; "
                                                   "model pixels within 1e-6 of the reference (tested), not the default's 1e-10"}
        finally:
            ctx.set_tail_log("default")
            field.images.render(field.sources, loglik=True)
    src = field.src
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        field.sources.set(src["type"], src["radec"], src["counts"], src["shape"])
        field.images.render(field.sources, loglik=True)
    torch.cuda.synchronize()
    out["ms_per_step_with_source_upload"] = (time.perf_counter() - t0) / n * 1e3
    # ONE source changed between evaluations (the reference's single-source moves, util/infer/mcmc_transitions.py:37-152): its
    # row goes up (cel_sources_set_rows) and the render is INCREMENTAL -- only the tiles its old and new boxes touch are
    # rendered again, bit for bit the full render's pixels and log-likelihood (CEL_OPT_INCREMENTAL; prep and binning in full)
    field.sources.set(src["type"], src["radec"], src["counts"], src["shape"])
    field.images.render(field.sources, loglik=True)
    rs1 = np.random.RandomState(11)
    pick = rs1.choice(field.S, n, replace=False).astype(np.int32)
    dirty = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(n):
        row = pick[k:k + 1]
        field.sources.set_rows(row, src["type"][row], src["radec"][row] + 1e-5, src["counts"][row] * 1.01, src["shape"][row])
        ll_inc, _ = field.images.render(field.sources, loglik=True)
    torch.cuda.synchronize()
    out["ms_per_step_one_source_changed"] = (time.perf_counter() - t0) / n * 1e3
    out["one_source_changed_tiles_rendered"] = field.images.last_render_dirty_tiles()
    field.sources.set(src["type"], src["radec"], src["counts"], src["shape"])           # (back to the benchmark's catalogue)
    field.images.render(field.sources, loglik=True)
    # the drop-in Python API north_star names: celeste_likelihood_multi_image(srcs, imgs)
    from desi_mcmc_amd import celeste
    imgs = synth.fits_images(field)
    cat = cel.SrcCatalog((src["type"] == 1).astype(np.int64), src["radec"], field.flux5(), src["shape"])
    ll_api = celeste.celeste_likelihood_multi_image(cat, imgs)          # uploads the images once
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        ll_api = celeste.celeste_likelihood_multi_image(cat, imgs)
    torch.cuda.synchronize()
    out["python_api_ms"] = (time.perf_counter() - t0) / n * 1e3
    plist = [cel.SrcParams(u=p.u.copy(), a=p.a, fluxes=p.fluxes.copy(), theta=p.theta, sigma=p.sigma, phi=p.phi, rho=p.rho)
             for p in cat]
    # A plain LIST of SrcParams, as celeste_em.py:25,159, celeste_mcmc.py:130 and the moves of
    # util/infer/mcmc_transitions.py:37-152 pass it: (a) every source assigned to since the last call -- the whole list is
    # gathered again; (b) ONE source moved between calls -- the caller's usual case: its row is re-read and uploaded;
    # (c) nothing changed.  (SrcParams.__setattr__ stamps an object; celeste._cached_list_arrays.)
    t0 = time.perf_counter()
    for _ in range(3):
        for p in plist:
            p.u = p.u                        # an assignment stamps the object
        ll_list = celeste.celeste_likelihood_multi_image(plist, imgs)
    t_all = (time.perf_counter() - t0) / 3
    t0 = time.perf_counter()
    for _ in range(3):
        for p in plist:
            p.u = p.u
    t_assign = (time.perf_counter() - t0) / 3
    out["python_api_list_ms"] = (t_all - t_assign) * 1e3
    u0 = plist[23].u.copy()

    def list_legs():
        ll_moved = None
        t0 = time.perf_counter()
        for k in range(n):
            pos = plist[23].u                    # the reference's moves: edit in place, then assign (mcmc_transitions.py:49-51)
            pos[0] = u0[0] + 1e-6 * ((k % 5) - 2)
            plist[23].u = pos
            ll_moved = celeste.celeste_likelihood_multi_image(plist, imgs)
        one = (time.perf_counter() - t0) / n * 1e3
        plist[23].u = u0.copy()              # (a copy: the next leg edits the source's array in place again)
        ll_list2 = celeste.celeste_likelihood_multi_image(plist, imgs)
        assert ll_moved != ll_list2 and ll_list2 == ll_list, (ll_moved, ll_list2, ll_list)
        t0 = time.perf_counter()
        for _ in range(n):
            celeste.celeste_likelihood_multi_image(plist, imgs)
        return one, (time.perf_counter() - t0) / n * 1e3
    # the default: every source re-read on every call (the reference's semantics), only the rows that differ uploaded
    assert celeste.list_cache() == "exact"
    out["python_api_list_one_changed_ms"], out["python_api_list_unchanged_ms"] = list_legs()
    try:                                         # the opt-in fast mode: only the objects assigned to since the last call re-read
        celeste.list_cache("stamps")
        celeste.celeste_likelihood_multi_image(plist, imgs)
        out["python_api_list_stamps_one_changed_ms"], out["python_api_list_stamps_unchanged_ms"] = list_legs()
    finally:
        celeste.list_cache("exact")
    views = cel.SrcCatalog.from_params(plist).views()         # a LIST of per-source objects backed by one catalogue
    ll_views = celeste.celeste_likelihood_multi_image(views, imgs)
    views[17].u = views[17].u + 1e-5                          # a write through a view is seen by the next call
    moved = celeste.celeste_likelihood_multi_image(views, imgs)
    views[17].u = views[17].u - 1e-5
    t0 = time.perf_counter()
    for _ in range(n):
        ll_views = celeste.celeste_likelihood_multi_image(views, imgs)
    out["python_api_views_ms"] = (time.perf_counter() - t0) / n * 1e3
    assert moved != ll_views and abs(ll_views - ll_api) <= 1e-12 * abs(ll_api)
    out["python_api_note"] = ("celeste_likelihood_multi_image(srcs, imgs) end to end, images resident after the first call: "
                              "srcs = SrcCatalog (arrays; python_api_ms) / a plain list of %d SrcParams objects "
                              "(python_api_list_ms: every object assigned to since the last call, the whole list gathered again; "
                              "python_api_list_one_changed_ms: one source moved between calls -- EVERY source re-read (the default, list_cache('exact'): "
                              "the reference's semantics), the one row that differs uploaded, incremental render; "
                              "python_api_list_unchanged_ms; python_api_list_stamps_*: the opt-in list_cache('stamps'), only objects "
                              "assigned to since the last call re-read) / the list SrcCatalog.views() "
                              "hands out: per-source objects with SrcParams' attributes, backed by the catalogue's arrays "
                              "(python_api_views_ms)" % len(plist))
    out["python_api_loglik_rel_diff"] = float(abs(ll_api - out["loglik"]) / abs(out["loglik"])) if out["loglik"] else None
    assert abs(ll_list - ll_api) <= 1e-12 * abs(ll_api)


def secondary_legs(args, env, field):
    """After the timed region of the default run (N = 1): bounded measurements of the OTHER configurations, so that the
    driver's one line carries them too -- BASELINE configs[4] (Gibbs sweeps over this same field: samples/s and the
    sweep's phases), configs[1] (stars1k_512) and the star-only regimes where the HBM roof binds (stars10k_2048: a star on
    every tile; stars2k_4096: a sparse 4096^2 frame, 1.34 GB of image traffic per launch -- `north_star`'s 60 %-of-HBM regime).
    Each is what `bench.py --workload NAME` reports, on fewer steps; none of them touches the headline fields."""
    torch, cel, dist, synth, ctx = env["torch"], env["cel"], env["dist"], env["synth"], env["ctx"]
    from desi_mcmc_amd import celeste_mcmc
    sec = {}
    for name, steps in (("stars10k_2048", 100), ("stars2k_4096", 100), ("stars1k_512", 400)):
        f = synth.SyntheticField.from_config(ctx, name, seed=42)
        S, B, H, W, fg = synth.CONFIGS[name]
        for _ in range(30):
            f.images.render(f.sources, loglik=True)
        # the render kernel only (see run_render); a step of a few tens of microseconds (configs[1]: one 23 us launch) carries
        # the event pair -- ~10 us of host time -- on every FOURTH launch: a sample of the timed region's launches
        ctx.profile(3 if name == "stars1k_512" else 2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            f.images.render(f.sources, loglik=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        t_render, n_render, render_kernel = ctx.profile_render()
        ctx.profile(False)
        st = f.images.stats()
        alg = 16.0 * B * H * W + 128.0 * S * B
        sec[name] = {"value": st["n_srcpix"] * steps / dt, "unit": "source-pixel evals/s", "steps": steps, "ms_per_step": dt / steps * 1e3,
                     "roofline": {"bound": "hbm", "achieved": alg / (t_render * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": alg / (t_render * 1e-3) / 1e9 / HBM_PEAK_GBS, "kernel": render_kernel, "kernel_ms": t_render,
                                  "launches": n_render, "algorithmic_bytes_per_launch": alg}}
        if render_kernel == "k_small_stars":
            sec[name]["roofline"]["note"] = ("k_small_stars is the WHOLE step in one launch (source prep, tile binning, render, Poisson partials): "
                                             "kernel_ms is that launch; HIP events on every 4th of the %d timed steps" % steps)
        del f
    S, B, H, W = field.S, field.B, field.H, field.W
    gf = celeste_mcmc.GibbsField(field.images, list(range(B)), field.bands[:, 2], field.bands[:, 1], H * W)
    g = celeste_mcmc.ModelGibbs([gf], field.src["type"], field.src["radec"], field.flux5(), field.src["shape"], seed=1)
    eps0 = field.bands[:, 0].copy()
    sweeps = 20
    try:
        for _ in range(3):
            g.sweep()
            g.log_likelihood()
        for k in g.timing:
            g.timing[k] = 0
        ctx.profile(int(os.environ.get("CEL_BENCH_PROFILE", "2")))        # the evaluating kernels (likelihoods, split, mass, render), not the 46 k_prep launches of a sweep
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(sweeps):
            g.sweep()
            g.log_likelihood()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        g.sweeps_timed = sweeps
        rep = gibbs_report(g, gf, ctx, sweeps, dt, S, B)
        ctx.profile(False)
        sec["gibbs10k"] = dict({"value": float(g.active.sum()) * sweeps / dt, "unit": "source updates (samples)/s", "steps": sweeps,
                                "ms_per_step": dt / sweeps * 1e3, "slice_sigma_deg": g.slice_args.get("sigma", 1.0)}, **rep)
        if args.cpu_sample > 0:
            from oracle import oracle as orc      # cpu_baseline leg only
            sec["gibbs10k"]["cpu_baseline"] = gibbs_cpu_baseline(field, g, gf, orc, n_sources=32)
        if args.sustained > 0:
            # BASELINE configs[4] as quoted: 200 sweeps of the same chain, sustained (every sweep + its trace render)
            def sweep_step():
                g.sweep()
                g.log_likelihood()
            sus = sustained_leg(torch, sweep_step, steps=200)
            sus["samples_per_s"] = float(g.active.sum()) * sus["steps"] / sus["wall_s"]
            sec["gibbs10k_200_sustained"] = sus
    finally:
        for b in range(B):                      # the sweeps redrew the sky levels: the headline field gets its own back
            field.images.set_epsilon(b, eps0[b])
    # The strong-scaling jobs at N = 8, every rank played on this one GPU (run_projection): the field cut into row strips, and
    # ONE Gibbs chain partitioned by strips.  No multi-GPU node has been available to any round; this is what stands in.
    if args.projection:
        import copy
        pa = copy.copy(args)
        pa.of, pa.as_rank, pa.workload, pa.steps, pa.warmup, pa.strip_cut = 8, None, "mixed10k_2048", 60, 10, "equal"
        pr = run_projection(pa, env, field=field, emit=False)
        pg = copy.copy(pa)
        pg.workload, pg.steps, pg.split = "gibbs10k", 5, "strips"
        eps1 = field.images.eps.copy()
        pgo = run_projection(pg, env, field=field, emit=False)
        for b in range(B):
            field.images.set_epsilon(b, eps1[b])
        sec["projected_strong_N8"] = {
            "mixed10k_2048_row_strips": dict(pr["projected_strong"], per_rank_ms=[round(p["ms_per_step"], 4) for p in pr["per_rank"]],
                                             kernels_ms_rank0=pr["per_rank"][0]["kernels_ms"], strip_edges=pr["strip_edges"]),
            "gibbs10k_strip_chain": dict(pgo["projected_strong"], per_rank_ms=[round(p["ms_per_step"], 3) for p in pgo["per_rank"]],
                                         sweep_ms_rank3=pgo["per_rank"][3]["sweep_ms"], window_rows_rank3=pgo["per_rank"][3]["window_rows"])}
    # BASELINE configs[3]'s stand-in (the Stripe-82 set is absent from the reference tree): 8 fields of the configs[2]
    # population as ONE field set on this GPU, 3 timed steps (what `bench.py --workload fields8_2048` reports on more)
    K, steps = 8, 3
    fields = [field] + [synth.SyntheticField.from_config(ctx, "mixed10k_2048", seed=42 + 1000 * k) for k in range(1, K)]
    for f in fields:
        f.images.render(f.sources, loglik=True)
    ctx.profile(2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        llb = np.zeros(B)
        for f in fields:
            llb += f.images.render(f.sources, loglik=True)[1]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t_render, n_render, render_kernel = ctx.profile_render()
    ctx.profile(False)
    npx = sum(f.images.stats()["n_srcpix"] for f in fields)
    alg = 16.0 * B * H * W + 128.0 * S * B
    sec["fields8_2048"] = {"value": npx * steps / dt, "unit": "source-pixel evals/s", "steps": steps, "ms_per_step": dt / steps * 1e3,
                           "n_fields": K, "loglik_field_set": float(llb.sum()),
                           "roofline": {"bound": "hbm", "achieved": alg / (t_render * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                        "frac": alg / (t_render * 1e-3) / 1e9 / HBM_PEAK_GBS, "kernel": render_kernel, "kernel_ms": t_render,
                                        "launches": n_render, "algorithmic_bytes_per_launch": alg},
                           "note": "a step renders and scores all 8 fields (one launch of the render kernel per field); N > 1 deals the fields "
                                   "to the ranks and all-reduces the 5 per-band sums (bench.py --workload fields8_2048 --gpus N)"}
    del fields
    return sec



# ---- strong scaling, rank by rank on ONE GPU --------------------------------------------------------
def measured_row_cost(ctx, _lib, images, sources, align):
    """cost of every `align`-row band of the frame from the durations the render's tiles had in a whole-frame launch
    (cel_debug_tile_timing): what dist.strip_edges evens out"""
    from desi_mcmc_amd import dist
    ctx.set_option(_lib.CEL_OPT_TILE_TIMING, 1)
    try:
        images.render(sources, loglik=True)
        images.render(sources, loglik=True)
        tt = images.tile_timing()
    finally:
        ctx.set_option(_lib.CEL_OPT_TILE_TIMING, 0)
    dur = (tt[:, 1] - tt[:, 0]).astype(np.float64)
    B, H, W = images.B, images.H, images.W
    ntx, nty = (W + 31) // 32, (H + 63) // 64                   # the 32 x 64 layout's tiles, [band][tile row][tile column]
    if dur.shape[0] != B * ntx * nty:
        return None
    # (row i of the timing table is TILE i: k_render_hw<true> records by tile index, not by launch position)
    return dist.strip_cost_from_tiles(dur, B, nty, ntx, 64, align, H=H)


def run_projection(args, env, field=None, emit=True):
    """--scaling strong --of N [--as-rank k] on ONE GPU, no process group: this process builds rank k's part of the N-rank job
    exactly as that rank would -- its row strip (render workloads) or its window of the strip-partitioned / dealt chain
    (gibbs10k) -- and times its FULL step (prep, binning, order, render, reduction; the whole sweep).  Without --as-rank every
    rank is taken in turn.  The projected N-GPU step is the slowest rank's (the collectives, B doubles per step or 11 doubles
    per source per sweep, are not in it); `projected_strong` states max, sum and the 1-rank time measured in the same process."""
    torch, cel, dist, synth, _lib = env["torch"], env["cel"], env["dist"], env["synth"], env["_lib"]
    ctx = env["ctx"]
    N = args.of
    ranks = [args.as_rank] if args.as_rank is not None else list(range(N))
    if any(k < 0 or k >= N for k in ranks):
        raise SystemExit("--as-rank must be in [0, --of)")
    gibbs = args.workload == "gibbs10k"
    base = "mixed10k_2048" if gibbs else args.workload
    if field is None:
        field = synth.SyntheticField.from_config(ctx, base, seed=42)
    S, B, H, W, fg = synth.CONFIGS[base]
    align = 64 if args.layout == 1 else dist.TILE_ROWS
    steps, warm = args.steps, max(args.warmup, 3)

    def timed(step, n, w, chunks=5):
        """ms per step: the MEDIAN of `chunks` back-to-back chunks of n / chunks steps (a chunk that meets a one-off stall
        of the host or the box does not decide the rank's number)"""
        for _ in range(w):
            step()
        per = max(n // chunks, 1)
        out = []
        for _ in range(chunks):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(per):
                step()
            torch.cuda.synchronize()
            out.append((time.perf_counter() - t0) / per * 1e3)
        return float(np.median(out))

    row_cost = None
    if args.strip_cut == "measured":
        row_cost = measured_row_cost(ctx, _lib, field.images, field.sources, align)
    edges = dist.strip_edges(H, N, row_cost, align=align)
    out = {"workload": args.workload, "of": N, "ranks": ranks, "strip_cut": args.strip_cut if row_cost is not None else "equal",
           "strip_edges": edges, "steps": steps}
    per = []
    if not gibbs:
        one = timed(lambda: field.images.render(field.sources, loglik=True), steps, 100)
        ll_full = field.images.render(field.sources, loglik=True)[0]
        ll_sum = 0.0
        for k in ranks:
            y0, y1 = edges[k], edges[k + 1]
            strip = cel.ImageSet(ctx, field.bands, y1 - y0, W, nelec=field.nelec[:, y0:y1])
            strip.set_window(y0, H)
            ms = timed(lambda: strip.render(field.sources, loglik=True), steps, warm + 20)
            ctx.profile(1)
            for _ in range(20):
                ll_k = strip.render(field.sources, loglik=True)[0]
            km = {"k_prep": ctx.profile_get("prep")[0], "k_bin": ctx.profile_get("bin")[0], "k_render": ctx.profile_render()[0],
                  "k_reduce": ctx.profile_get("reduce")[0]}
            ctx.profile(False)
            ll_sum += ll_k
            per.append({"rank": k, "rows": [y0, y1], "ms_per_step": ms, "kernels_ms": km, "n_tile_entries": strip.stats()["n_tile_entries"]})
            strip.close()
        out["one_rank_ms"] = one
        if len(ranks) == N:
            out["loglik_strips_sum"] = ll_sum
            out["loglik_whole_frame"] = ll_full
    else:
        from desi_mcmc_amd import celeste_mcmc
        slice_args = dict(step_out=False, sigma=args.slice_sigma)

        ran = [0]

        def chain_ms(g, n):
            def step():
                g.sweep()
                g.log_likelihood()
                ran[0] += 1
            ms = timed(step, n, 3)
            return ms
        gf = celeste_mcmc.GibbsField(field.images, list(range(B)), field.bands[:, 2], field.bands[:, 1], H * W)
        g1 = celeste_mcmc.ModelGibbs([gf], field.src["type"], field.src["radec"], field.flux5(), field.src["shape"], seed=1, slice_args=slice_args)
        one = chain_ms(g1, steps)
        out["one_rank_ms"] = one
        out["split"] = args.split
        boxes, status = field.images.source_boxes(field.sources)
        eps0 = field.bands[:, 0].copy()
        for k in ranks:
            for b in range(B):
                field.images.set_epsilon(b, eps0[b])
            if args.split == "strips":
                deal, gfk = celeste_mcmc.strip_gibbs_field(ctx, field.bands, field.nelec, field.src["pix"][:, 1], boxes, status, N, k,
                                                           edges=edges, solo=True)
            else:
                deal = dist.SourceDeal(S, N, k, solo=True)
                gfk = celeste_mcmc.GibbsField(field.images, list(range(B)), field.bands[:, 2], field.bands[:, 1], H * W)
            g = celeste_mcmc.ModelGibbs([gfk], field.src["type"], field.src["radec"], field.flux5(), field.src["shape"], seed=1,
                                        slice_args=slice_args, deal=deal)
            for _ in range(2):
                g.sweep()
                g.log_likelihood()
            for key in g.timing:
                g.timing[key] = 0
            ran[0] = 0
            ms = chain_ms(g, steps)
            n = max(ran[0], 1)
            per.append({"rank": k, "ms_per_step": ms, "sources_owned": int(deal.mine.size),
                        "window_rows": list(getattr(deal, "window", (0, H))),
                        "sweep_ms": {"photon_split_and_sky": g.timing["split"] / n * 1e3, "flux": g.timing["flux"] / n * 1e3,
                                     "location_slice": g.timing["location"] / n * 1e3}})
            if args.split == "strips":
                gfk.iset.close()
                if gfk.trace_iset is not None:
                    gfk.trace_iset.close()
            del g, gfk, deal
    t = [p["ms_per_step"] for p in per]
    out["per_rank"] = per
    out["projected_strong"] = {"max_ms": max(t), "sum_ms": sum(t), "mean_ms": sum(t) / len(t), "one_rank_ms": out["one_rank_ms"],
                               "speedup_at_N": out["one_rank_ms"] / max(t), "efficiency": out["one_rank_ms"] / max(t) / N,
                               "note": "one GPU, one rank of the %d-rank job at a time, no process group: the slowest rank's full step "
                                       "against the 1-rank step of the same process; the collective (%s) is not in it"
                                       % (N, "one all-gather of 11 doubles per source per sweep" if gibbs else "one all-reduce of %d doubles" % B)}
    if emit:
        print(json.dumps(out))
    return out


# ---- configs[3] stand-in: K fields dealt to ranks ----------------------------------------------------
def run_fields(args, env):
    torch, cel, dist, synth = env["torch"], env["cel"], env["dist"], env["synth"]
    rank, world, local, ctx = env["rank"], env["world"], env["local"], env["ctx"]
    K = args.n_fields
    base = "mixed10k_2048"
    mine = dist.field_shard(K, world, rank)
    # The fields of a rank are independent: with --streams n they are dealt to n contexts (one HIP stream each),
    # every context driven by its own host thread, so that one field's binning, launch gaps and ragged last
    # round of tiles run under another field's render (measured: 11.65 -> 11.26 ms for 8 fields with 2, slower
    # with 3 or 4).  The per-field results are summed in field order either way.
    from concurrent.futures import ThreadPoolExecutor
    n_str = max(1, min(args.streams, len(mine)))
    ctxs = [ctx] + [cel.Context(local) for _ in range(n_str - 1)]
    fields = [synth.SyntheticField.from_config(ctxs[i % n_str], base, seed=42 + 1000 * k) for i, k in enumerate(mine)]
    B = synth.CONFIGS[base][1]
    reducer = dist.LoglikReducer(B, device=local, depth=2) if world > 1 else None
    last = {}
    pool = ThreadPoolExecutor(max_workers=n_str) if n_str > 1 else None

    def lane(j):        # the fields of context j, one after the other
        return [f.images.render(f.sources, loglik=True)[1] for f in fields[j::n_str]]

    def step():
        llb = np.zeros(B)
        per = [lane(0)] if pool is None else list(pool.map(lane, range(n_str)))
        for i in range(len(fields)):
            llb += per[i % n_str][i // n_str]
        if reducer is not None:
            reducer.submit_device([f.images for f in fields])     # summed on the device, then all-reduced: no host hop before the collective
            if len(reducer.pending) > 1:
                llb = reducer.result()
        last["llb"] = llb

    def after_warmup():
        if reducer is not None:
            reducer.drain()
        for cx in ctxs:
            cx.profile(2)               # the render kernel only (see run_render)

    def finish():
        if reducer is not None:
            last["llb"] = reducer.drain()[-1]
    dt = Timer(dist, torch).run(step, args.warmup, args.steps, after_warmup, finish, prime=max(2, 100 // max(K // world, 1)))
    tn = [cx.profile_render()[:2] for cx in ctxs]
    n_render = sum(n for _, n in tn)
    t_render = sum(t * n for t, n in tn) / max(n_render, 1)
    for cx in ctxs:
        cx.profile(False)
    if pool is not None:
        pool.shutdown()
    srcpix = sum(f.images.stats()["n_srcpix"] for f in fields)
    gauss = sum(f.images.stats()["n_gauss"] for f in fields)
    dt_max, (srcpix_all, gauss_all, nf_all) = reduce_over_ranks(torch, world, local, dt, [srcpix, gauss, len(fields)])
    if rank != 0:
        return
    S, _, H, W, fg = synth.CONFIGS[base]
    alg_bytes = 16.0 * B * H * W + 128.0 * S * B
    achieved = alg_bytes / (t_render * 1e-3) / 1e9 if t_render > 0 else 0.0
    print(json.dumps({
        "metric": "source-pixel evals/sec + log-lik ms over %d fields of 10k sources x %d bands x %d^2 "
                  "(synthetic stand-in for the Stripe-82 field set)" % (K, B, H),
        "value": srcpix_all * args.steps / dt_max, "unit": "source-pixel evals/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt_max / args.steps * 1e3,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "fields%d_2048" % K, "fields": K, "fields_on_rank0": mine, "fields_total_check": int(nf_all),
                   "sources_per_field": S, "bands": B, "frame": [H, W], "galaxy_fraction": fg,
                   "note": "BASELINE configs[3] names the Stripe-82 catalogue, which is not in the reference tree "
                           "(.MISSING_LARGE_BLOBS); synthetic fields of the configs[2] population stand in for it",
                   "parallelism": "fields dealt round-robin to %d GPU(s), 1 all-reduce of %d doubles per step; on a GPU "
                                  "the fields run on %d streams (a context and a host thread each)" % (world, B, n_str),
                   "streams_per_gpu": n_str,
                   "ranks": world, "collective_backend": env["backend"],
                   "ranks_seen_by_collective": env["roll_call"]["ranks_seen_by_collective"], "device_uuid": env["roll_call"]["device_uuid"]},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": None, "kernel": "k_render", "kernel_ms": t_render, "launches": n_render,
                     "algorithmic_bytes_per_launch": alg_bytes},
        "work": {"n_srcpix_per_step": srcpix_all, "n_gauss_per_step_nominal": gauss_all},
        "loglik": float(np.sum(last["llb"])),
        "cpu_baseline": _fields_cpu_baseline(args, fields[0], S) if (world == 1 and args.cpu_sample > 0) else None}))


def _fields_cpu_baseline(args, field, S):
    """the CPU oracle on a bounded sample of ONE of the set's fields (the fields are the configs[2] population)"""
    from oracle import oracle as orc      # cpu_baseline leg only
    out = cpu_baseline(field, min(args.cpu_sample, S), orc)
    out["sample"] = "field 0 of the set: " + out["sample"]
    return out


# ---- configs[4]: Gibbs sweeps ------------------------------------------------------------------------
def gibbs_cpu_baseline(field, g, gf, orc, n_sources=48):
    """The CPU oracle (kind "port") on the location step of a BOUNDED sample of the same sweep: for each of the
    sampled sources, as many conditional-likelihood evaluations (orc_patch_loglik, mode 0: the restatement of
    Source.log_likelihood, sources.py:134-183) as the sampler made per source, in all bands, on the split's own
    photon patches fetched to the host -- one thread per source on the host's cores (ctypes releases the GIL)."""
    from concurrent.futures import ThreadPoolExecutor
    S, B = field.S, field.B
    bands = field.bands.copy()
    for b in range(B):
        bands[b, 36] = orc.checked_radius(bands[b], field.images.band(b)[36])
    boxes, offs, data = gf.iset.fetch_samples()
    evals_per_source = max(1, int(round(g.timing["evals"] / max(g.sweeps_timed, 1) / max(float(g.active.sum()), 1.0))))
    rs = np.random.RandomState(5)
    pick = rs.choice(np.nonzero(g.active)[0], size=min(n_sources, int(g.active.sum())), replace=False)
    cts = g.counts(gf)
    jitter = rs.normal(0.0, 2e-5, size=(pick.size, evals_per_source, 2))
    nthr = max(1, min(host_threads(), pick.size))
    orc.set_threads(1)

    def one(i):
        s = pick[i]
        tot = 0.0
        for e in range(evals_per_source):
            for b in range(B):
                k = s * B + b
                tot += orc.patch_loglik(bands[b], field.H, field.W, g.typ[s], g.u[s] + jitter[i, e], g.shape[s], cts[s, b],
                                        boxes[s, b], data[offs[k]:offs[k + 1]], mode=0)
        return tot
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=nthr) as pool:
        list(pool.map(one, range(pick.size)))
    dt = time.perf_counter() - t0
    orc.set_threads(nthr)
    return {"value": pick.size / dt, "unit": "source location updates/s", "cores": nthr, "kind": "port",
            "sample": "%d of %d sources x %d conditional-likelihood evaluations (the sampler's mean per source) x %d bands on "
                      "the split's photon patches, %.1f s wall (orc_patch_loglik, oracle/celeste_oracle.c; the location step "
                      "only: it is 2/3 of the sweep)" % (pick.size, S, evals_per_source, B, dt)}


def gibbs_report(g, gf, ctx, steps, dt, S, B):
    """per-sweep figures of a timed run of ModelGibbs: phases, the conditional-likelihood kernel's own time (HIP events
    attached to its launches) and the algorithmic bytes it walked (counted on the device)"""
    t_ll, n_ll = ctx.profile_get("patch_ll")
    t_split, n_split = ctx.profile_get("split")
    t_mass, n_mass = ctx.profile_get("mass")
    t_render, n_render = ctx.profile_render()[:2]
    # a round of the location step is one or two launches (the blocks scored densely, the blocks scored at their photons):
    # the roofline is priced per ROUND -- the bytes its evaluations read over the kernel time of its launches
    rounds = max(g.timing.get("loc_launches", 0), 1)
    shapes_ran = g.timing.get("shape_evals", 0) > 0          # the shape step's launches are in the same event bucket: no price then
    alg_bytes = g.timing.get("loc_bytes", 0) / rounds
    round_ms = t_ll * n_ll / rounds
    achieved = (alg_bytes / (round_ms * 1e-3) / 1e9) if (round_ms > 0 and not shapes_ran) else None
    known = g.timing["split"] + g.timing["flux"] + g.timing["location"] + g.timing.get("shape", 0.0) + g.timing.get("merge", 0.0)
    pmc = load_gibbs_pmc()
    return {
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": None if achieved is None else achieved / HBM_PEAK_GBS,
                     "traffic": pmc.get("traffic"), "traffic_source": pmc.get("source") or pmc.get("stale"), "kernel": "k_patch_ll_nz + k_patch_ll_hw<0> (one slice round of every running chain)",
                     "kernel_ms": round_ms, "launches": rounds, "algorithmic_bytes_per_launch": alg_bytes,
                     "note": "one launch = one round of the location step (its one or two dispatches); kernel_ms: HIP events attached to "
                             "the dispatches, summed per round and averaged over the timed sweeps' rounds; bytes: per evaluation and band "
                             "8 B per photon-holding pixel of a patch scored at its photons (4 B per pixel of the photon rectangle for one "
                             "scored densely) + a 128-B record, counted on the device.  The kernels are fp64-issue-bound like k_render "
                             "(DESIGN.md 5); the HBM fraction is what the contract asks for; `binding` is the roof they are on",
                     "binding": _issue_binding(pmc.get("valu_insts"), round_ms, pmc.get("source") or pmc.get("stale"),
                                               split_valu=pmc.get("split_valu_insts"), split_ms=t_split)},
        "work": {"slice_rounds_per_sweep": g.timing["rounds"] / steps, "loglik_evals_per_sweep": g.timing["evals"] / steps,
                 "sources_updated_per_sweep": float(g.active.sum())},
        "sweep_ms": {"photon_split_and_sky": g.timing["split"] / steps * 1e3, "flux": g.timing["flux"] / steps * 1e3,
                     "location_slice": g.timing["location"] / steps * 1e3,
                     "shape_slice": g.timing.get("shape", 0.0) / steps * 1e3,
                     "merge_all_gather": g.timing.get("merge", 0.0) / steps * 1e3,
                     "trace_render": (dt - known) / steps * 1e3},
        "device_ms_per_sweep": {"k_patch_ll_nz + k_patch_ll_hw<0> (location%s)" % (" + shapes" if shapes_ran else ""): t_ll * n_ll / steps, "k_photon_split_hw": t_split * n_split / steps,
                                "k_patch_ll_hw<3> (stamp mass)": t_mass * n_mass / steps,
                                "k_render (split totals + trace)": t_render * n_render / steps}}


def run_gibbs(args, env):
    torch, cel, dist, synth = env["torch"], env["cel"], env["dist"], env["synth"]
    rank, world, local, ctx = env["rank"], env["world"], env["local"], env["ctx"]
    from desi_mcmc_amd import celeste_mcmc
    base = "mixed10k_2048"
    field = synth.SyntheticField.from_config(ctx, base, seed=42)          # the same sky on every rank
    S, B, H, W, fg = synth.CONFIGS[base]
    gf = celeste_mcmc.GibbsField(field.images, list(range(B)), field.bands[:, 2], field.bands[:, 1], H * W)
    slice_args = dict(step_out=False, sigma=args.slice_sigma)
    strong = args.scaling == "strong"
    if strong and args.split == "strips":
        # ONE chain on all the GPUs, partitioned as SURVEY 8e prescribes: row strips -- a rank splits the photons of its strip
        # plus a halo, owns the sources of its strip; sky-photon sums, the trace and the new rows are exchanged
        boxes, status = field.images.source_boxes(field.sources)
        deal, gf = celeste_mcmc.strip_gibbs_field(ctx, field.bands, field.nelec, field.src["pix"][:, 1], boxes, status, world, rank,
                                                  device=local)
        field.images.close()                # the whole frame is not needed on this rank any more
        g = celeste_mcmc.ModelGibbs([gf], field.src["type"], field.src["radec"], field.flux5(), field.src["shape"],
                                    seed=1, slice_args=slice_args, deal=deal)
    elif strong:
        # ONE chain on all the GPUs: same seed everywhere, the photon split replicated, the sources dealt
        # to the ranks for the per-source updates, one all-gather of the new locations and fluxes per sweep
        deal = dist.SourceDeal(S, world, rank, device=local)
        g = celeste_mcmc.ModelGibbs([gf], field.src["type"], field.src["radec"], field.flux5(), field.src["shape"],
                                    seed=1, slice_args=slice_args, deal=deal)
    else:
        g = celeste_mcmc.ModelGibbs([gf], field.src["type"], field.src["radec"], field.flux5(), field.src["shape"],
                                    seed=1 + 1000 * rank, slice_args=slice_args)   # one independent chain per rank
    reducer = dist.LoglikReducer(1, device=local, depth=2) if (world > 1 and not strong) else None
    trace = []

    def step():
        g.sweep(shapes=args.shapes)
        ll = np.array([g.log_likelihood()])          # the chain's trace (one render)
        if reducer is not None:
            reducer.submit(ll)
            if len(reducer.pending) > 1:
                ll = reducer.result()
        trace.append(float(ll[0]))

    def after_warmup():
        if reducer is not None:
            reducer.drain()
        for k in g.timing:
            g.timing[k] = 0
        ctx.profile(int(os.environ.get("CEL_BENCH_PROFILE", "2")))        # the evaluating kernels (likelihoods, split, mass, render), not the 46 k_prep launches of a sweep

    def finish():
        if reducer is not None:
            trace.append(float(reducer.drain()[-1][0]))
    ll0 = g.log_likelihood()
    dt = Timer(dist, torch).run(step, args.warmup, args.steps, after_warmup, finish, prime=3)
    g.sweeps_timed = args.steps
    rep = gibbs_report(g, gf, ctx, args.steps, dt, S, B)
    ctx.profile(False)
    mine = g.active if not strong else (g.active & g.deal.mask)
    dt_max, (updates,) = reduce_over_ranks(torch, world, local, dt, [float(mine.sum()) * args.steps])
    if rank != 0:
        return
    out = {
        "metric": "end-to-end samples/sec, slice-sampling Gibbs sweeps over the 10k-source x %d-band x %d^2 synthetic field" % (B, H),
        "value": updates / dt_max, "unit": "source updates (samples)/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt_max / args.steps * 1e3,
        "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "gibbs10k", "sources": S, "bands": B, "frame": [H, W], "galaxy_fraction": fg,
                   "sweep": "photon split of all bands + sky level (models.py:123-160), then per source: flux Gamma "
                            "conditionals (sources.py:321-349) and location by slice sampling (sources.py:308-319), "
                            "all sources in lock-step%s; + one field log-likelihood per sweep (the trace)"
                            % ("; then every galaxy's shape by slice sampling (celeste_mcmc.py:224-243)" if args.shapes else ""),
                   "slice": dict(slice_args, compwise=True,
                                 note="sigma in degrees.  The reference's call passes step=du/5=0.001 deg, which its "
                                      "slicesample ignores (sigma stays 1.0 deg); 0.001 is the call's intent, the library's and "
                                      "this bench's default; --slice-sigma 1.0 runs the literal behaviour"),
                   "parallelism": ("ONE chain on %d GPU(s), partitioned by row strips: a rank holds its strip + a halo (window rows %s of "
                                   "%d), splits those photons, owns the sources of its strip; per sweep 1 all-gather of 11 doubles per "
                                   "source + 2 rank-ordered sums of 5 doubles (sky photons, trace)" % (world, list(g.deal.window), H))
                                  if (strong and args.split == "strips") else
                                  ("ONE chain on %d GPU(s): the photon split replicated (counter-based draws, bitwise equal on every "
                                   "rank), the sources dealt round-robin to the ranks for the flux and location updates, 1 all-gather "
                                   "of 11 doubles per source per sweep; the chain is the 1-GPU chain bit for bit" % world) if strong else
                                  ("%d independent chain(s), 1 per GPU, over the same field; 1 all-reduce of the chains' "
                                   "log-likelihood per sweep" % world),
                   "ranks": world, "collective_backend": env["backend"],
                   "ranks_seen_by_collective": env["roll_call"]["ranks_seen_by_collective"], "device_uuid": env["roll_call"]["device_uuid"]},
        "loglik_before": ll0, "loglik_trace_tail": trace[-3:]}
    out.update(rep)
    if world == 1 and args.cpu_sample > 0:
        from oracle import oracle as orc      # cpu_baseline leg only
        out["cpu_baseline"] = gibbs_cpu_baseline(field, g, gf, orc)
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--workload", default="mixed10k_2048")
    ap.add_argument("--kernel", default="recurrence", choices=["direct", "recurrence"])
    ap.add_argument("--tail-log", type=float, default=None,
                    help="CEL_OPT_TAIL_LOG for every kernel; default: the library's defaults (24 for the field render, 32 for the per-source kernels)")
    ap.add_argument("--tile-rows", type=int, default=32, choices=[32, 64])
    ap.add_argument("--tile-order", type=int, default=1, choices=[0, 1, 2])
    ap.add_argument("--layout", type=int, default=1, choices=[0, 1, 2],
                    help="render tile geometry: 0 = 64x32 (k_render), 1 = 32x64 half-wave (k_render_hw)")
    ap.add_argument("--cpu-sample", type=int, default=400, help="sources in the CPU baseline sample (0 = skip)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default, what the driver runs): one field (gibbs10k: one chain) per GPU.  strong: ONE "
                         "field cut into row strips, one per GPU (cel_images_set_window), total work fixed; gibbs10k: ONE "
                         "chain, its sources dealt to the GPUs")
    ap.add_argument("--n-fields", type=int, default=8, help="fields8_2048: number of fields dealt to the ranks")
    ap.add_argument("--streams", type=int, default=1,
                    help="fields8_2048: contexts (HIP streams, a host thread each) per GPU the rank's fields run on; 2 is 3 %% faster, "
                         "but the kernels' event times then include each other")
    ap.add_argument("--slice-sigma", type=float, default=0.001, help="gibbs10k: slice-sampler interval width in degrees")
    ap.add_argument("--star-tiles", type=int, default=1, choices=[0, 1, 2],
                    help="CEL_OPT_STAR_TILES: 0 = the general render kernel always, 1 (default) = k_render_stars for a catalogue "
                         "without galaxies on a frame of more than 2048 tiles, 2 = at any size")
    ap.add_argument("--photon-lists", type=int, default=0, choices=[0, 1, 2],
                    help="gibbs10k: CEL_OPT_PHOTON_LISTS (0 = per patch whichever is cheaper, 1 = always at the photons, 2 = always densely)")
    ap.add_argument("--split", default="replicated", choices=["replicated", "strips"],
                    help="gibbs10k --scaling strong: every rank runs the whole photon split (the chain is the 1-GPU chain bit for "
                         "bit), or the split is partitioned by row strips like the sources (SURVEY 8e)")
    ap.add_argument("--slice-fuse", type=int, default=None,
                    help="gibbs10k: CEL_OPT_SLICE_FUSE (N = a slice round of at most N likelihood blocks is ONE launch: the block that finishes a chain's last job "
                         "steps the chain; 1 = every round, 0 = never: three launches per round; default: the library's)")
    ap.add_argument("--shapes", action="store_true", help="gibbs10k: every sweep also resamples the galaxies' shapes")
    ap.add_argument("--of", type=int, default=0,
                    help="with --scaling strong on ONE GPU (no --gpus): play the ranks of an N-rank job one at a time, each "
                         "building its strip / window exactly as in the N-rank job and timing its full step -> projected_strong")
    ap.add_argument("--as-rank", type=int, default=None, help="with --of N: only this rank (default: all N in turn)")
    ap.add_argument("--strip-cut", default="equal", choices=["measured", "equal"],
                    help="--scaling strong: equal tile rows (default), or strip edges that even out the whole-frame render's measured "
                         "tile durations (cel_debug_tile_timing; rank 0 measures, every rank cuts alike).  Round 6: with the durations "
                         "filed under their tiles (round 5 filed them by launch position and every 'measured' cut came out equal) the "
                         "measured cut of the benchmark field is WORSE, 0.32 against 0.29 ms for the slowest of 8 ranks: a strip is "
                         "4 tile rows, one row is a quarter of it, and a tile's duration in a full launch (heaviest first: the light "
                         "tiles run last on a half-empty chip) understates the light rows")
    ap.add_argument("--legs", default="all", choices=["all", "none"],
                    help="render workloads at N=1: 'all' (default) adds the untimed-by-the-contract extras after the timed region "
                         "(evaluated-Gaussian count, fast tail preset, source-upload step, Python-API call); 'none' runs the "
                         "timed region only -- what the committed rocprofv3 summaries profile, so that their per-kernel "
                         "averages are those of the timed launches")
    ap.add_argument("--sustained", type=float, default=5.0,
                    help="default run at N = 1: seconds of back-to-back mixed10k_2048 steps after the timed region (ms per step in "
                         "0.5-s buckets + SMI clock / power samples), and 200 Gibbs sweeps likewise -> `sustained`; 0 = skip")
    ap.add_argument("--no-projection", dest="projection", action="store_false",
                    help="default run at N = 1: skip `secondary.projected_strong_N8` (the ranks of the 8-GPU strong-scaling jobs played "
                         "one at a time on this GPU: ~10 s)")
    ap.add_argument("--master-port", type=int, default=0, help="self-launch only: rendezvous port (0 = pick a free one)")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves, one per GPU,
        # as children of this process -- BEFORE anything here touches the GPU (the parent never does).
        raise SystemExit(self_launch(args.gpus, sys.argv[1:], args.master_port))
    if os.environ.get("CEL_BENCH_DRYRUN") == "1":
        return dry_run(args)
    kind = "gibbs" if args.workload == "gibbs10k" else ("fields" if args.workload.startswith("fields") else "render")
    if kind == "render" and args.workload not in RENDER_WORKLOADS:
        raise SystemExit("unknown workload %r (render: %s; also fields8_2048, gibbs10k)" % (args.workload, ", ".join(RENDER_WORKLOADS)))

    import torch
    torch.set_num_threads(min(16, max(1, torch.get_num_threads())))

    import desi_mcmc_amd as cel
    from desi_mcmc_amd import _lib, dist, synth

    # CEL_BENCH_BACKEND=gloo rehearses the multi-rank flow on a box with fewer GPUs than ranks
    # (ranks then share GPUs and the collective runs on the host); the driver's runs use RCCL.
    rank, world, local = dist.init_from_env(backend=os.environ.get("CEL_BENCH_BACKEND"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torchrun --nproc-per-node %d, or run "
                         "`python bench.py --gpus %d` without a launcher (it starts its own ranks)"
                         % (args.gpus, world, args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if world > torch.cuda.device_count() and os.environ.get("CEL_BENCH_BACKEND") != "gloo":
        raise SystemExit("bench.py: %d ranks but %d GPU(s) visible" % (world, torch.cuda.device_count()))
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    ctx = cel.Context(local)
    ctx.set_kernel(args.kernel)
    if args.tail_log is not None:
        ctx.set_tail_log(args.tail_log)
    args.tail_log = ctx.get_option(_lib.CEL_OPT_TAIL_LOG)        # the field render's threshold (what the headline kernel runs at)
    ctx.set_option(_lib.CEL_OPT_TILE_ROWS, args.tile_rows)
    ctx.set_option(_lib.CEL_OPT_TILE_LAYOUT, args.layout)
    ctx.set_option(_lib.CEL_OPT_TILE_ORDER, args.tile_order)
    ctx.set_option(_lib.CEL_OPT_PHOTON_LISTS, args.photon_lists)
    ctx.set_option(_lib.CEL_OPT_STAR_TILES, args.star_tiles)
    if args.slice_fuse is not None:
        ctx.set_option(_lib.CEL_OPT_SLICE_FUSE, args.slice_fuse)
    backend = "none"
    if world > 1:
        import torch.distributed as td
        backend = {"nccl": "rccl (torch.distributed nccl)"}.get(td.get_backend(), td.get_backend())
    # first-run insurance: the collective must reach exactly --gpus ranks, each on a device of its own; anything else exits non-zero
    # on every rank before a number is printed (CEL_BENCH_BACKEND=gloo rehearsals may share GPUs)
    try:
        call = dist.roll_call(args.gpus, local, allow_shared_devices=os.environ.get("CEL_BENCH_BACKEND") == "gloo")
    except RuntimeError as e:
        if str(e).startswith("roll call:"):           # the two things it exists to refuse: a short group, a shared device
            raise SystemExit("bench.py: %s" % e)
        call = {"ranks_seen_by_collective": None, "device_uuid": None, "backend": backend, "error": "%s: %s" % (type(e).__name__, e)}
        sys.stderr.write("bench.py: roll call could not run (%s); continuing without it\n" % call["error"])
    except Exception as e:                            # noqa: BLE001 -- insurance must not be what breaks the first multi-GPU run
        call = {"ranks_seen_by_collective": None, "device_uuid": None, "backend": backend, "error": "%s: %s" % (type(e).__name__, e)}
        sys.stderr.write("bench.py: roll call could not run (%s); continuing without it\n" % call["error"])
    env = dict(torch=torch, cel=cel, dist=dist, synth=synth, _lib=_lib, rank=rank, world=world, local=local, ctx=ctx,
               backend=backend, roll_call=call)
    if args.of:
        if world != 1 or args.scaling != "strong":
            raise SystemExit("--of N plays the ranks of a strong-scaling job on ONE GPU: use it with --scaling strong and without --gpus")
        run_projection(args, env)
        return
    {"render": run_render, "fields": run_fields, "gibbs": run_gibbs}[kind](args, env)
    if world > 1:
        import torch.distributed as td
        td.barrier()
        td.destroy_process_group()


if __name__ == "__main__":
    main()
