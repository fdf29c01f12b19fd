"""Where a Gibbs sweep's wall time goes on the device: kernels and the idle gaps between them, from a rocprofv3 kernel trace.
    (GPU box)  cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/gaps -- python3 $ROOT/bench.py --workload gibbs10k --steps 6 --warmup 2 --cpu-sample 0
    python tools/sweep_gaps.py gpurun_out/gaps"""
import csv, glob, os, sys, collections
d = sys.argv[1]
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = []
with open(f) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60]))
rows.sort()
# the timed sweeps: the last 60 % of the trace
t_lo = rows[0][0] + 0.45 * (rows[-1][1] - rows[0][0])
rows = [r for r in rows if r[0] >= t_lo]
span = rows[-1][1] - rows[0][0]
busy = collections.Counter(); n = collections.Counter(); gaps = collections.Counter(); ngaps = collections.Counter()
end = rows[0][0]
prev = None
for s, e, k in rows:
    busy[k] += e - s; n[k] += 1
    if prev is not None and s > end:
        gaps[(prev, k)] += s - end; ngaps[(prev, k)] += 1
    end = max(end, e); prev = k
nsplit = max(1, n[[k for k in n if "k_photon_split_hw" in k][0]])
print("trace window %.1f ms, %d sweeps (split launches): %.2f ms per sweep; busy %.2f ms, idle %.2f ms per sweep"
      % (span / 1e6, nsplit, span / 1e6 / nsplit, sum(busy.values()) / 1e6 / nsplit, (span - sum(busy.values())) / 1e6 / nsplit))
print("-- kernels (ms per sweep, launches per sweep)")
for k, v in busy.most_common(18):
    print("  %-62s %7.3f  %6.1f" % (k, v / 1e6 / nsplit, n[k] / nsplit))
print("-- idle gaps by (previous kernel -> next kernel) (ms per sweep, count per sweep, mean us)")
for k, v in gaps.most_common(18):
    print("  %-40s -> %-40s %7.3f %6.1f %7.1f" % (k[0][:40], k[1][:40], v / 1e6 / nsplit, ngaps[k] / nsplit, v / 1e3 / ngaps[k]))
