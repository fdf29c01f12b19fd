#!/usr/bin/env python3
"""Diagnostic: where does k_render's wall time go?  Renders the benchmark field with
CEL_OPT_TILE_TIMING and prints the tile-duration distribution and the occupancy timeline
(how many tile-waves are resident over the launch).  Not a timed run: stamps cost cycles.

    python tools/tile_timeline.py [--workload mixed10k_2048] [--tile-order 1]
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="mixed10k_2048")
ap.add_argument("--tile-order", type=int, default=1)
ap.add_argument("--tail-log", type=float, default=32.0)
args = ap.parse_args()

import desi_mcmc_amd as cel  # noqa: E402
from desi_mcmc_amd import _lib, synth  # noqa: E402

ctx = cel.Context(0)
ctx.set_tail_log(args.tail_log)
ctx.set_option(_lib.CEL_OPT_TILE_ORDER, args.tile_order)
f = synth.SyntheticField.from_config(ctx, args.workload)
for _ in range(3):
    f.images.render(f.sources, loglik=True)
ctx.set_option(6, 1.0)     # CEL_OPT_TILE_TIMING
f.images.render(f.sources, loglik=True)
n = C.c_int64(0)
_lib.check(_lib.lib().cel_debug_tile_timing(f.images._h, None, C.byref(n)))
buf = np.zeros(3 * n.value, dtype=np.uint64)
_lib.check(_lib.lib().cel_debug_tile_timing(f.images._h, buf.ctypes.data, C.byref(n)))
t = buf.reshape(-1, 3)
start, end = t[:, 0].astype(np.int64), t[:, 1].astype(np.int64)
cnt = (t[:, 2] & np.uint64(0xfff)).astype(np.int64)
pairs = ((t[:, 2] >> np.uint64(12)) & np.uint64(0xfffff)).astype(np.int64)
comprows = (t[:, 2] >> np.uint64(32)).astype(np.int64)
print("work: sources %d  pairs of groups %d (%.2f per source)  kept component-rows %.3e (%.1f per pair, "
      "x32 columns = %.3e kept Gaussian-pixel evaluations)"
      % (cnt.sum(), pairs.sum(), pairs.sum() / max(cnt.sum(), 1), comprows.sum(), comprows.sum() / max(pairs.sum(), 1),
         comprows.sum() * 32.0))
t0 = start.min()
start, end = (start - t0) / 100.0, (end - t0) / 100.0            # microseconds (100 MHz clock)
dur = end - start
total = end.max()
print("tiles %d   launch span %.1f us   sum of tile durations %.1f us   mean %.2f us" % (len(dur), total, dur.sum(), dur.mean()))
print("duration percentiles us: p50 %.1f p90 %.1f p99 %.1f max %.1f" % tuple(np.percentile(dur, [50, 90, 99, 100])))
print("list length: mean %.1f max %d ; duration/entry mean %.2f us" % (cnt.mean(), cnt.max(), dur.sum() / max(cnt.sum(), 1)))
# resident tile-waves over time
edges = np.linspace(0, total, 41)
for i in range(40):
    a, b = edges[i], edges[i + 1]
    occ = np.sum(np.clip(np.minimum(end, b) - np.maximum(start, a), 0, None)) / (b - a)
    print("%7.0f-%7.0f us  resident waves %7.1f  %s" % (a, b, occ, "#" * int(occ / 64)))
# launch-order check: when does each decile of blocks start?
print("start time of block deciles (us):", np.round(np.percentile(start, np.arange(0, 101, 10)), 1))
