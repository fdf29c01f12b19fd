#!/bin/bash
# rocprofv3 kernel stats of the default bench line for each library given (CEL_HIP_LIBRARY; "default" = the shipped one):
# the small kernels' mean durations.   gpurun -- bash tools/prof_libs.sh default tools/bin/x.so ...
cd /tmp && export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT
for lib in "$@"; do
  tag=$(basename $lib .so)
  if [ "$lib" = default ]; then unset CEL_HIP_LIBRARY; else export CEL_HIP_LIBRARY=$root/$lib; fi
  rm -rf $root/gpurun_out/pl_$tag
  (cd $root && rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/pl_$tag -- python3 bench.py --steps 100 --warmup 10 --cpu-sample 0 --legs none ${AB_ARGS} > /dev/null 2> $root/gpurun_out/pl_$tag.log)
  f=$(find $root/gpurun_out/pl_$tag -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "$tag" <<'P'
import csv, sys
print("==", sys.argv[2])
for r in csv.DictReader(open(sys.argv[1])):
    if int(r['Calls']) > 50:
        print("  %-30s calls %5s avg_us %9.2f min %8.2f" % (r['Name'][:30], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3))
P
  rm -rf $root/gpurun_out/pl_$tag
done
