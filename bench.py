#!/usr/bin/env python3
"""bench.py -- the headline metric of BASELINE.json on MI355X.

Workload (config.workload): BASELINE.json configs[2], the configuration the metric is quoted
on -- 10 000 mixed star/galaxy sources x 5 bands x 2048^2, synthetic (data "synthetic").
One step = one full-field log-likelihood evaluation with everything resident in HBM:
    k_prep (WCS, galaxy shape matrix, bounding box per source x band) -> k_bin (tile lists)
    -> k_render (model images + fused Poisson term) -> k_reduce -> 5 doubles to the host
    [-> one all-reduce of the 5 doubles across ranks when N > 1].
value = source-pixel evaluations per second, whole job (sum over ranks / max-over-ranks time);
ms_per_step = full-field log-lik latency.  N > 1: one field per rank ("weak"), launched by
torchrun, one collective per step (RCCL).

Usage: python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--kernel direct|recurrence]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_VALU_PEAK_TF = 78.6     # 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz
FLOP_PER_GAUSS = 35.0        # SURVEY 8d accounting: 10 arithmetic + exp counted as 25

# HBM bytes per k_render launch from the PMC counters (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
# separate passes, gfx950 correction 2*FETCH + WRITE re-calibrated for this access pattern with
# tools/calib_traffic.hip).  Counters cannot be read from inside this process: the number is the
# committed measurement of exactly this command and is reported only for the configuration it
# was taken on; any other configuration gets null.
PMC_TRAFFIC = {   # (workload, kernel, tail_log, layout) -> (HBM bytes per k_render launch, source)
    ("mixed10k_2048", "recurrence", 32.0, 1): (386329760.0, "profiles/r01_final_pmc.json"),
}
# same provenance: 2 * SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES (two waves per SIMD) and SQ_INSTS_VALU of that launch
PMC_VALU = {
    ("mixed10k_2048", "recurrence", 32.0, 1): (0.82, 6.64e8, "profiles/r01_final_pmc.json"),
}
CPU_THREADS_MAX = 16         # the GPU box's CPU share for one GPU


def cpu_baseline(field, nsample, orc):
    """The CPU oracle timed on a bounded sample of the SAME workload (first `nsample` sources,
    all bands, full frame) on the host's cores.  Reported beside the GPU number; not the target."""
    sl = slice(0, nsample)
    bands = field.bands.copy()
    for b in range(field.B):
        bands[b, 36] = field.images.band(b)[36]
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    orc.set_threads(max(1, min(orc.max_threads(), avail, CPU_THREADS_MAX)))
    t0 = time.perf_counter()
    lam, ll, st = orc.render_field(bands, field.H, field.W, field.src["type"][sl], field.src["radec"][sl],
                                   field.src["counts"][sl], field.src["shape"][sl], field.nelec)
    dt = time.perf_counter() - t0
    return dict(value=st["n_srcpix"] / dt, unit="source-pixel evals/s", cores=orc.max_threads(), kind="port",
                sample="first %d of %d sources, %d bands, %dx%d frame, %.2e source-px in %.2f s wall "
                       "(oracle/celeste_oracle.c, OpenMP over sources)"
                       % (nsample, field.S, field.B, field.H, field.W, st["n_srcpix"], dt),
                gauss_evals_per_s=st["n_gauss"] / dt)


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
    """CEL_BENCH_DRYRUN=1: rendezvous + the two collectives of the timed region on synthetic
    numbers, no GPU and no metric -- what the CPU test of the self-launch path runs."""
    from desi_mcmc_amd import dist
    rank, world, local = dist.init_from_env(backend=os.environ.get("CEL_BENCH_BACKEND", "gloo"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    red = dist.LoglikReducer(5, depth=2)
    red.submit(np.full(5, float(rank + 1)))
    got = red.drain()[-1]
    dist.barrier()
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "ranks": world, "allreduce_check": float(got[0]),
                          "expected": world * (world + 1) / 2.0}))
    if world > 1:
        import torch.distributed as td
        td.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="mixed10k_2048")
    ap.add_argument("--kernel", default="recurrence", choices=["direct", "recurrence"])
    ap.add_argument("--tail-log", type=float, default=32.0)
    ap.add_argument("--tile-rows", type=int, default=32, choices=[32, 64])
    ap.add_argument("--tile-order", type=int, default=1, choices=[0, 1, 2])
    ap.add_argument("--layout", type=int, default=1, choices=[0, 1, 2],
                    help="render tile geometry: 0 = 64x32 (k_render), 1 = 32x64 half-wave (k_render_hw)")
    ap.add_argument("--cpu-sample", type=int, default=400, help="sources in the CPU baseline sample (0 = skip)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default, what the driver runs): one field per GPU.  strong: ONE field cut into row "
                         "strips, one per GPU (cel_images_set_window), total work fixed")
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

    import torch

    import desi_mcmc_amd as cel
    from desi_mcmc_amd import dist, synth

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
    ctx.set_tail_log(args.tail_log)
    from desi_mcmc_amd import _lib
    ctx.set_option(_lib.CEL_OPT_TILE_ROWS, args.tile_rows)
    ctx.set_option(_lib.CEL_OPT_TILE_LAYOUT, args.layout)
    ctx.set_option(_lib.CEL_OPT_TILE_ORDER, args.tile_order)

    # weak: one field per rank (same population, different seed).  strong: every rank builds the SAME
    # field (the catalogue is small and replicated) and keeps only its row strip of the pixels.
    strong = (args.scaling == "strong") and world > 1
    field = synth.SyntheticField.from_config(ctx, args.workload, seed=42 + (0 if strong else 1000 * rank))
    stats = None
    if strong:
        y0, y1 = dist.strip_rows(field.H, world, rank)
        full_stats = None
        field.images.render(field.sources, loglik=False)
        full_stats = field.images.stats()                  # the whole field's work: what every step of the job does
        strip = cel.ImageSet(ctx, field.bands, max(y1 - y0, 1), field.W, nelec=field.nelec[:, y0:max(y1, y0 + 1)])
        strip.set_window(y0, field.H)
        field.images = strip

    # the one collective: B per-band doubles summed over ranks.  Pipelined one step deep: the sum of
    # step k travels while step k+1 renders (the ranks' fields are independent chains; the global
    # log-likelihood is a diagnostic), and every sum is collected inside the timed region.
    reducer = dist.LoglikReducer(field.B, device=local, depth=2) if world > 1 else None

    def step():
        ll, llb = field.images.render(field.sources, loglik=True)
        if reducer is not None:
            reducer.submit(llb)
            if len(reducer.pending) > 1:
                llb = reducer.result()
        return llb

    for _ in range(args.warmup):
        step()
    if reducer is not None:
        reducer.drain()
    stats = field.images.stats()
    if strong:   # total work is the one field's, whoever renders which rows: count it once (rank 0)
        stats = dict(full_stats) if rank == 0 else dict(full_stats, n_srcpix=0, n_gauss=0)
    ctx.profile(True)

    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        llb = step()
    if reducer is not None:
        llb = reducer.drain()[-1]
    torch.cuda.synchronize()
    dist.barrier()
    dt = time.perf_counter() - t0

    t_render, n_render = ctx.profile_get("render")
    t_bin, _ = ctx.profile_get("bin")
    t_prep, _ = ctx.profile_get("prep")
    t_red, _ = ctx.profile_get("reduce")
    ctx.profile(False)

    # max over ranks of the elapsed time, sum over ranks of the work
    agg = torch.tensor([dt, stats["n_srcpix"], stats["n_gauss"]], dtype=torch.float64)
    if world > 1:
        import torch.distributed as td
        if td.get_backend() == "nccl":
            agg = agg.cuda(local)
        tmax = agg[:1].clone()
        td.all_reduce(tmax, op=td.ReduceOp.MAX)
        tsum = agg[1:].clone()
        td.all_reduce(tsum, op=td.ReduceOp.SUM)
        dt_max, n_srcpix_all, n_gauss_all = tmax.item(), tsum[0].item(), tsum[1].item()
    else:
        dt_max, n_srcpix_all, n_gauss_all = dt, stats["n_srcpix"], stats["n_gauss"]

    if rank == 0:
        S, B, H, W, fg = synth.CONFIGS[args.workload]
        n_imgpix = B * H * W
        # algorithmic HBM bytes of one k_render launch (DESIGN.md "Measurement"):
        #   read nelec 8 B + write lambda 8 B per image pixel, + one 128-B record per (source, band)
        alg_bytes = 16.0 * n_imgpix + 128.0 * S * B
        if strong:   # rank 0's launch covers its strip of the pixels (and still reads every record)
            alg_bytes = 16.0 * B * (y1 - y0) * W + 128.0 * S * B
        achieved = alg_bytes / (t_render * 1e-3) / 1e9 if t_render > 0 else 0.0
        pmc = None if strong else PMC_TRAFFIC.get((args.workload, args.kernel, args.tail_log, args.layout))
        out = {
            # BASELINE.json's metric; `value` is its first half, `ms_per_step` its second
            "metric": "source-pixel evals/sec + full-field log-lik ms, %s sources x %d bands x %d^2"
                      % ("10k" if S == 10000 else str(S), B, H) if H == W else
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
                                      + ", 1 all-reduce of %d doubles per step (overlapped with the next step's render)" % B},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": pmc[0] if pmc else None,
                         "traffic_source": pmc[1] if pmc else None,
                         "kernel": "k_render", "kernel_ms": t_render, "launches": n_render,
                         "algorithmic_bytes_per_launch": alg_bytes},
            # the roof that actually binds this kernel: fp64 vector ALU
            "fp64_valu": {"achieved": FLOP_PER_GAUSS * stats["n_gauss"] / (t_render * 1e-3) / 1e12 if t_render > 0 else 0.0,
                          "peak": FP64_VALU_PEAK_TF, "unit": "TFLOP/s (35 flop per Gaussian-eval accounting)"},
            "work": {"n_srcpix_per_step": n_srcpix_all, "n_gauss_per_step": n_gauss_all,
                     "gauss_evals_per_s": n_gauss_all * args.steps / dt_max,
                     "n_tile_entries": stats["n_tile_entries"]},
            "kernels_ms": {"k_prep": t_prep, "k_bin": t_bin, "k_render": t_render, "k_reduce": t_red},
            "loglik": float(np.sum(llb)),
        }
        out["fp64_valu"]["frac"] = out["fp64_valu"]["achieved"] / FP64_VALU_PEAK_TF
        vp = PMC_VALU.get((args.workload, args.kernel, args.tail_log, args.layout))
        if vp:
            # how busy the vector ALUs really are, and the executed instruction stream, from the
            # committed PMC profile of this configuration (not re-measured here)
            out["fp64_valu"].update({"valu_busy_frac_pmc": vp[0], "valu_wave_instructions_per_launch_pmc": vp[1],
                                     "pmc_source": vp[2]})
        if world == 1 and args.cpu_sample > 0:
            from oracle import oracle as orc      # cpu_baseline leg only
            out["cpu_baseline"] = cpu_baseline(field, min(args.cpu_sample, S), orc)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as td
        td.barrier()
        td.destroy_process_group()


if __name__ == "__main__":
    main()
