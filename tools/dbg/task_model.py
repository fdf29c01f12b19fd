#!/usr/bin/env python3
"""CPU model (diagnostic, no GPU): what would k_render_hw's general path walk if a galaxy's groups of components were dealt
to the lanes as (group, column) TASKS -- only the columns a group can matter on -- instead of pairs of groups on all 32
columns of the tile?  Applies the kernel's drop rule (T + log(A / eps), rows on the tile's columns, slots by row class) to a
sample of the config-3 galaxies and prints, for both schemes, seeds and walked rows in units of one wave-step of <= 6
components per lane.

    python tools/dbg/task_model.py [--every 20] [--T 24]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from desi_mcmc_amd import synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--every", type=int, default=20)
ap.add_argument("--T", type=float, default=24.0)
ap.add_argument("--tight", action="store_true", help="row ranges by ceil / floor + 1 instead of floor / ceil + 1")
ap.add_argument("--classes", type=int, default=8, help="row-count classes of the slot order (0: exact order by rows)")
ap.add_argument("--order", default="class", help="slot order inside a class: class (component order) | rlo | mid")
ap.add_argument("--chords", action="store_true", help="tasks walk the union chord of their group's ellipses at their column")
args = ap.parse_args()
S, B, H, W, fg = synth.CONFIGS["mixed10k_2048"]
bands = synth.make_bands(H, W, B)
src = synth.make_sources(S, H, W, bands, fg, 42)
for b in range(B):
    bands[b, 36] = orc.band_radius(bands[b])
TW, TH, G = 32, 64, 6
gal = np.nonzero(src["type"] == 1)[0][::args.every]

tot = dict(entries=0, pairs=0, pair_rows=0.0, pair_seed=0.0, trips=0, trip_rows=0.0, trip_seed=0.0, comprows=0.0, tasks=0,
           pair_cost=0.0, trip_cost=0.0)


def qmin_rect(a, b, c, x1, x2, y1, y2):
    inside = (x1 <= 0) & (x2 >= 0) & (y1 <= 0) & (y2 >= 0)
    best = np.full(a.shape, np.inf)
    for xe in (x1, x2):
        y = np.clip(-b * xe / c, y1, y2)
        best = np.minimum(best, a * xe * xe + 2 * b * xe * y + c * y * y)
    for ye in (y1, y2):
        x = np.clip(-b * ye / a, x1, x2)
        best = np.minimum(best, a * x * x + 2 * b * x * ye + c * ye * ye)
    return np.where(inside, 0.0, best)


for s in gal:
    for b in range(B):
        band = bands[b]
        eps = band[0]
        patch, (y0, y1), (x0, x1) = orc.source_patch(band, H, W, 1, src["radec"][s], src["shape"][s])
        if y1 <= y0 or x1 <= x0:
            continue
        pis, means, covs, pxy, tinv = orc.galaxy_table(band, src["shape"][s], src["radec"][s])
        det = covs[:, 0, 0] * covs[:, 1, 1] - covs[:, 0, 1] ** 2
        qa, qb, qc = covs[:, 1, 1] / det, -covs[:, 0, 1] / det, covs[:, 0, 0] / det
        A = src["counts"][s, b] * pis / (2 * np.pi * np.sqrt(det))
        Tk = args.T + np.log(np.abs(A) / eps)
        rhs = 2 * np.maximum(Tk, 0.0)
        hx = np.sqrt(rhs * covs[:, 0, 0])                      # x half-extent of the threshold ellipse
        for ty in range(y0 // TH, (y1 - 1) // TH + 1):
            for tx in range(x0 // TW, (x1 - 1) // TW + 1):
                X0, Y0 = tx * TW, ty * TH
                xa, xb = max(x0, X0), min(x1, X0 + TW) - 1
                ya, yb = max(y0, Y0), min(y1, Y0 + TH) - 1
                qmin = qmin_rect(qa, qb, qc, xa - means[:, 0], xb - means[:, 0], ya - means[:, 1], yb - means[:, 1])
                keep = 0.5 * qmin <= Tk
                # rows of each component on this tile's columns (numerically over the integer columns)
                xs = np.arange(xa, xb + 1)[:, None] - means[None, :, 0]
                disc = rhs[None, :] - (qa - qb * qb / qc)[None, :] * xs ** 2
                ok = disc >= 0
                hh = np.sqrt(np.maximum(disc, 0.0) / qc[None, :])
                cy = means[None, :, 1] - (qb / qc)[None, :] * xs
                if args.tight:
                    lo = np.where(ok, np.ceil(cy - hh - 0.02), 1e9)
                    hi = np.where(ok, np.floor(cy + hh + 0.02) + 1, -1e9)
                else:
                    lo = np.where(ok, np.floor(cy - hh), 1e9)
                    hi = np.where(ok, np.ceil(cy + hh) + 1, -1e9)
                clo = np.maximum(lo, ya)
                chi = np.minimum(hi, yb + 1)                    # per (column, component) chord, clipped to the box rows on this tile
                rlo = clo.min(axis=0)
                rhi = chi.max(axis=0)
                keep &= rhi > rlo
                if not keep.any():
                    tot["entries"] += 1
                    tot["empty"] = tot.get("empty", 0) + 1
                    continue
                tot.setdefault("kk", []).append(int(keep.sum()))
                tot["entries"] += 1
                idx = np.nonzero(keep)[0]
                nrows = (rhi - rlo)[idx]
                if args.classes == 0:
                    cls = -nrows
                else:
                    w_ = 64 // args.classes
                    cls = (args.classes - 1) - np.minimum((nrows - 1) // w_, args.classes - 1)
                if args.order == "rlo":
                    order = idx[np.lexsort((rlo[idx], cls))]
                elif args.order == "mid":
                    order = idx[np.lexsort(((rlo + rhi)[idx], cls))]
                else:
                    order = idx[np.argsort(cls, kind="stable")]
                Kk = len(order)
                tot["comprows"] += float(nrows.sum())
                # now: pairs of 12
                for p0 in range(0, Kk, 2 * G):
                    m = order[p0:p0 + 2 * G]
                    R = len(m)
                    gA = (R + 1) // 2
                    rows = rhi[m].max() - rlo[m].min()
                    tot["pairs"] += 1
                    if xb - xa + 1 <= 16:
                        tot["narrow_rows"] = tot.get("narrow_rows", 0.0) + rows
                        tot["narrow_pairs"] = tot.get("narrow_pairs", 0) + 1
                    tot["pair_rows"] += rows
                    tot["pair_cost"] += gA * 37 + rows * (3 * gA - 1 + 2)
                    # phased walk: the pair's larger half of the slots on the whole union, the smaller half on ITS union only
                    nb = (R + 1) // 2
                    big, small = m[:nb], m[nb:]
                    rs_ = (rhi[small].max() - rlo[small].min()) if len(small) else 0
                    tot["walk_now"] = tot.get("walk_now", 0.0) + rows * R
                    tot["walk_phased"] = tot.get("walk_phased", 0.0) + rows * len(big) + rs_ * len(small)
                    tot["walk_own"] = tot.get("walk_own", 0.0) + float((rhi[m] - rlo[m]).sum())
                    # the best of the splits 2 / 3 / 4 big slots per half (pairs of 12 only; the others as above)
                    best = rows * len(big) + rs_ * len(small)
                    if R == 12:
                        for nbh in (1, 2, 4, 5):
                            sm = m[2 * nbh:]
                            c_ = rows * 2 * nbh + (rhi[sm].max() - rlo[sm].min()) * len(sm)
                            best = min(best, c_)
                    tot["walk_best"] = tot.get("walk_best", 0.0) + best
                    # three nested sets: thirds
                    n3 = (R + 2) // 3
                    a, bq, c = m[:n3], m[n3:2 * n3], m[2 * n3:]
                    rb_ = (rhi[np.concatenate([bq, c])].max() - rlo[np.concatenate([bq, c])].min()) if len(bq) + len(c) else 0
                    rc_ = (rhi[c].max() - rlo[c].min()) if len(c) else 0
                    tot["walk_thirds"] = tot.get("walk_thirds", 0.0) + rows * len(a) + rb_ * len(bq) + rc_ * len(c)
                # tasks: balanced groups of <= 6
                ng = (Kk + G - 1) // G
                Gs = (Kk + ng - 1) // ng
                tasks = []       # (rows, )
                for g in range(ng):
                    m = order[g * Gs:(g + 1) * Gs]
                    colok = ok[:, m] & (chi[:, m] > clo[:, m])
                    cols = colok.any(axis=1)
                    if args.chords:
                        l = np.where(colok, clo[:, m], 1e9).min(axis=1)
                        h = np.where(colok, chi[:, m], -1e9).max(axis=1)
                        r = (h - l)[cols]
                    else:
                        r = np.full(int(cols.sum()), rhi[m].max() - rlo[m].min())
                    tasks.append(r)
                tasks = np.concatenate(tasks) if tasks else np.zeros(0)
                tasks = -np.sort(-tasks, kind="stable")
                tot["tasks"] += len(tasks)
                for t0 in range(0, len(tasks), 64):
                    rows = tasks[t0]
                    tot["trips"] += 1
                    tot["trip_rows"] += rows
                    tot["trip_cost"] += Gs * 37 + 12 + rows * (3 * Gs - 1 + 2)

n = len(gal) * B
print("sampled galaxy (source, band) pairs:", n, " tile entries:", tot["entries"], " per pair %.2f" % (tot["entries"] / n))
print("now  : pairs %d (%.2f per entry)  pair-rows %.3e  own component-rows %.3e  cost %.3e" %
      (tot["pairs"], tot["pairs"] / tot["entries"], tot["pair_rows"], tot["comprows"], tot["pair_cost"]))
print("tasks: trips %d (%.2f per entry)  trip-rows %.3e  tasks %d (%.1f per trip)  cost %.3e" %
      (tot["trips"], tot["trips"] / tot["entries"], tot["trip_rows"], tot["tasks"], tot["tasks"] / max(tot["trips"], 1), tot["trip_cost"]))
print("walk component-rows: now %.3e  own %.3e (%.3f)  two nested sets %.3e (%.3f)  three %.3e (%.3f)" %
      (tot["walk_now"], tot["walk_own"], tot["walk_own"] / tot["walk_now"], tot["walk_phased"], tot["walk_phased"] / tot["walk_now"],
       tot["walk_thirds"], tot["walk_thirds"] / tot["walk_now"]))
print("best split per pair of 12 (1, 2, 3, 4 or 5 big slots per half): %.3e (%.3f)" % (tot["walk_best"], tot["walk_best"] / tot["walk_now"]))
print("pairs on entries whose box covers <= 16 of the tile's columns: %d of %d (%.1f %%), their rows %.1f %%" %
      (tot.get("narrow_pairs", 0), tot["pairs"], 100.0 * tot.get("narrow_pairs", 0) / tot["pairs"], 100.0 * tot.get("narrow_rows", 0) / tot["pair_rows"]))
kk = np.array(tot["kk"])
print("entries without a kept component: %d of %d (%.1f %%); kept components per entry with any: mean %.1f, <= 6: %.1f %%, <= 12: %.1f %%" %
      (tot.get("empty", 0), tot["entries"], 100.0 * tot.get("empty", 0) / tot["entries"], kk.mean(), 100.0 * (kk <= 6).mean(), 100.0 * (kk <= 12).mean()))
print("ratio tasks / now: steps %.3f  rows %.3f  cost %.3f" % (tot["trips"] / tot["pairs"], tot["trip_rows"] / tot["pair_rows"],
                                                              tot["trip_cost"] / tot["pair_cost"]))
