#!/bin/bash
# usage: tools/pmc_pass.sh <tag> "<counters of pass 1>" "<counters of pass 2>" ...
# One rocprofv3 --pmc pass per argument over a short bench.py run; per-kernel sums are printed and
# written to gpurun_out/pmc_<tag>.json.  (Counters only with --kernel-trace: see the gpurun rules.)
set -e
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "$@"; do
  i=$((i+1))
  rm -rf $root/gpurun_out/pmc_${tag}_$i
  (cd $root && rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $root/gpurun_out/pmc_${tag}_$i -- python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 > $root/gpurun_out/pmc_${tag}_$i.log 2>&1)
done
python3 - "$root" "$tag" <<'PY'
import csv, glob, json, sys, collections
root, tag = sys.argv[1], sys.argv[2]
out = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("%s/gpurun_out/pmc_%s_*/**/*counter_collection.csv" % (root, tag), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        out[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]] += 1
res = {}
for k in out:
    if "k_render" not in k and "k_patch" not in k and "k_photon" not in k:
        continue
    # per launch: a dispatch contributes one row per counter (dimension instances are summed by the tool or listed; sum either way)
    res[k] = {c: out[k][c] for c in sorted(out[k])}
    res[k]["_rows"] = {c: cnt[k][c] for c in sorted(cnt[k])}
json.dump(res, open("%s/gpurun_out/pmc_%s.json" % (root, tag), "w"), indent=1)
for k in res:
    print(k)
    for c in sorted(out[k]):
        print("   %-28s %16.0f  rows %d" % (c, out[k][c], cnt[k][c]))
PY
