"""The location step's slice rounds on the device, one line per round: k_patch_ll_nz's duration and the wall time from
that launch's start to the next round's (from the same kernel trace tools/sweep_gaps.py reads).
    python tools/sweep_rounds.py gpurun_out/gaps"""
import csv, glob, os, sys
d = sys.argv[1]
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = []
with open(f) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60]))
rows.sort()
# the last sweep: from the last k_photon_split_hw on
i_split = max(i for i, r in enumerate(rows) if "k_photon_split_hw" in r[2])
last = rows[i_split:]
nz = [r for r in last if "k_patch_ll_nz" in r[2]]
print("last sweep: %d slice rounds; k_patch_ll_nz total %.2f ms; first round starts %.2f ms after the split's start, last ends %.2f ms after"
      % (len(nz), sum(e - s for s, e, _ in nz) / 1e6, (nz[0][0] - last[0][0]) / 1e6, (nz[-1][1] - last[0][0]) / 1e6))
t0 = nz[0][0]
busy_all = 0
print("round  start_ms  nz_us  round_wall_us  busy_us(all kernels)")
for i, (s, e, _) in enumerate(nz):
    nxt = nz[i + 1][0] if i + 1 < len(nz) else max(r[1] for r in last if r[0] < e + 200000 and r[0] >= s)
    busy = sum(min(r[1], nxt) - r[0] for r in last if s <= r[0] < nxt)
    print("%4d  %8.3f  %6.1f  %8.1f  %8.1f" % (i, (s - t0) / 1e6, (e - s) / 1e3, (nxt - s) / 1e3, busy / 1e3))

# every sweep of the trace: the location step's span on the device (first likelihood launch -> the end of the last kernel before
# the trace render) and its likelihood-kernel time
splits = [i for i, r in enumerate(rows) if "k_photon_split_hw" in r[2]] + [len(rows)]
print("per sweep: rounds, location span ms (first k_patch_ll_nz start -> last k_patch_ll_nz / k_slice_step end), nz kernel ms")
for a, b in zip(splits[:-1], splits[1:]):
    sw = rows[a:b]
    nzs = [r for r in sw if "k_patch_ll_nz" in r[2]]
    if not nzs:
        continue
    tail = [r for r in sw if "k_patch_ll_nz" in r[2] or "k_slice_step" in r[2]]
    print("   %3d rounds  span %8.3f ms   nz %8.3f ms   launches of k_slice_step %d" % (len(nzs), (max(r[1] for r in tail) - nzs[0][0]) / 1e6,
          sum(e - s for s, e, _ in nzs) / 1e6, sum(1 for r in sw if "k_slice_step" in r[2])))
