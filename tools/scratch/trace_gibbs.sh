#!/bin/bash
set -e
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $root/gpurun_out/trace_gibbs
(cd $root && rocprofv3 --kernel-trace --output-format csv -d $root/gpurun_out/trace_gibbs -- python3 bench.py --workload gibbs10k --steps 1 --warmup 1 --legs none > $root/gpurun_out/trace_gibbs.json 2> $root/gpurun_out/trace_gibbs.log)
f=$(find $root/gpurun_out/trace_gibbs -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print(rows[0].keys())
sel = [r for r in rows if r["Kernel_Name"].startswith("void k_patch_ll_hw") or "k_patch_ll_hw" in r["Kernel_Name"]]
print(len(sel))
for r in sel[-60:]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print(r["Kernel_Name"][:30], r.get("Grid_Size", r.get("Grid_Size_X")), r.get("Workgroup_Size", r.get("Workgroup_Size_X")), "%.1f us" % d)
PY
