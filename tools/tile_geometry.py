#!/usr/bin/env python3
"""What would another render-tile geometry cost?  From the benchmark field's boxes: (source, tile)
pairs (set-up work), lane-rows executed (walk work, masked lanes included) and lane-seeds
(seed work per component), for several TW x TH.  Diagnostic."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import desi_mcmc_amd as cel
from desi_mcmc_amd import synth
ctx = cel.Context(0)
f = synth.SyntheticField.from_config(ctx, sys.argv[1] if len(sys.argv) > 1 else "mixed10k_2048")
boxes, status = f.images.source_boxes(f.sources)
ok = status > 0
y0, y1, x0, x1 = [boxes[..., i][ok].astype(np.int64) for i in range(4)]
area = ((y1 - y0) * (x1 - x0)).sum()
print("boxes %d, area %.3e, mean %.0f x %.0f" % (ok.sum(), area, (x1 - x0).mean(), (y1 - y0).mean()))
for TW, TH in ((32, 64), (16, 128), (64, 32), (32, 32), (16, 64), (8, 256)):
    ntx = (x1 - 1) // TW - x0 // TW + 1
    nty = (y1 - 1) // TH - y0 // TH + 1
    entries = (ntx * nty).sum()
    lane_rows = (ntx * TW * (y1 - y0)).sum()          # every lane of a touched tile column walks the box's rows
    lane_seeds = (ntx * TW * nty).sum()               # one seed per lane per vertical tile per component
    print("%3d x %3d: entries %8d  lane-rows %.3e (x%.2f of area)  lane-seeds %.3e" % (TW, TH, entries, lane_rows, lane_rows / area, lane_seeds))
