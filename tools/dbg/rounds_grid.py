"""per slice round of the last sweep in a rocprofv3 kernel trace: the likelihood launch's grid (blocks), its duration, and what
the duration would be if the round were throughput-bound at the full rounds' rate (block-seconds per block of round 0)"""
import csv, glob, os, sys
d = sys.argv[1]
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = []
with open(f) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0), int(r.get("Workgroup_Size", r.get("Workgroup_Size_X", 64)) or 64)))
rows.sort()
i_split = max(i for i, r in enumerate(rows) if "k_photon_split_hw" in r[2])
nz = [r for r in rows[i_split:] if "k_patch_ll_nz" in r[2]]
print("round  blocks   nz_us   us per 1000 blocks")
for i, (s, e, k, g, w) in enumerate(nz):
    blocks = g // max(w, 1) if g >= w else g
    print("%4d  %7d  %7.1f  %8.2f" % (i, blocks, (e - s) / 1e3, (e - s) / 1e3 / max(blocks, 1) * 1000))
