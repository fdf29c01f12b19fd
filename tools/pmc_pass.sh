#!/bin/bash
# usage: tools/pmc_pass.sh <tag> "<counters of pass 1>" "<counters of pass 2>" ...
# One rocprofv3 --pmc pass per argument over a short bench.py run (or PMC_PROG="script.py args"); per-kernel sums are printed and
# written to gpurun_out/pmc_<tag>.json.  (Counters only with --kernel-trace: see the gpurun rules.)
set -e
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "$@"; do
  i=$((i+1))
  rm -rf $root/gpurun_out/pmc_${tag}_$i
  (cd $root && rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $root/gpurun_out/pmc_${tag}_$i -- python3 ${PMC_PROG:-bench.py --steps 3 --warmup 1 --cpu-sample 0} > $root/gpurun_out/pmc_${tag}_$i.log 2>&1)
done
python3 - "$root" "$tag" <<'PY'
import csv, glob, json, sys, collections
root, tag = sys.argv[1], sys.argv[2]
# per kernel, per counter: {dispatch id: value summed over the counter's hardware instances}
per = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
for f in sorted(glob.glob("%s/gpurun_out/pmc_%s_*/**/*counter_collection.csv" % (root, tag), recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        per[k][r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
res = {}
for k in per:
    res[k] = {c: [per[k][c][d] for d in sorted(per[k][c])] for c in sorted(per[k])}
json.dump(res, open("%s/gpurun_out/pmc_%s.json" % (root, tag), "w"), indent=1)
for k in sorted(res):
    if not any(t in k for t in ("k_render", "k_patch", "k_photon", "k_estep", "k_small", "k_strict")):
        continue
    print(k)
    for c in sorted(res[k]):
        v = res[k][c]
        print("   %-28s launches %2d  mean %16.0f  last %16.0f" % (c, len(v), sum(v) / len(v), v[-1]))
PY
