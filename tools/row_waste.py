#!/usr/bin/env python3
"""Diagnostic: how much of k_render_hw's column walk is needed?  A pair of groups (12 components)
walks the union of its components' row ranges on all 32 columns of the tile; this prints the walked
component-rows next to the sum of the components' own row ranges and their column-clipped areas
(CEL_OPT_TILE_TIMING counters; CEL_OPT_DEBUG bit 128 selects the second set).

    python tools/row_waste.py [--workload mixed10k_2048]
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import desi_mcmc_amd as cel  # noqa: E402
from desi_mcmc_amd import _lib, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="mixed10k_2048")
args = ap.parse_args()
ctx = cel.Context(0)
f = synth.SyntheticField.from_config(ctx, args.workload)
for _ in range(3):
    f.images.render(f.sources, loglik=True)


def counters(dbg):
    ctx.set_option(_lib.CEL_OPT_DEBUG, float(dbg))
    ctx.set_option(_lib.CEL_OPT_TILE_TIMING, 1.0)
    f.images.render(f.sources, loglik=True)
    n = C.c_int64(0)
    _lib.check(_lib.lib().cel_debug_tile_timing(f.images._h, None, C.byref(n)))
    buf = np.zeros(3 * n.value, dtype=np.uint64)
    _lib.check(_lib.lib().cel_debug_tile_timing(f.images._h, buf.ctypes.data, C.byref(n)))
    return buf.reshape(-1, 3)[:, 2]


t = counters(0)
walked = (t >> np.uint64(32)).astype(np.int64).sum()
t = counters(128)
own_rows = (t & np.uint64(0xffffffff)).astype(np.int64).sum()
own_area = (t >> np.uint64(32)).astype(np.int64).sum()
print("walked component-rows (x 32 columns) %.3e ; sum of the components' own row ranges %.3e (%.2f) ; "
      "their column-clipped rectangles / 32 %.3e (%.2f)" % (walked, own_rows, own_rows / walked, own_area, own_area / walked))
