#!/bin/bash
# kernel trace of a few Gibbs sweeps (bench.py --workload gibbs10k) -> gpurun_out/<tag>/ ; then tools/sweep_rounds.py gpurun_out/<tag>
#   gpurun -- bash tools/trace_gibbs.sh r5_gaps [extra bench args]
tag=${1:-gaps}; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $root/gpurun_out/$tag
(cd $root && rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/$tag -- python3 bench.py --workload gibbs10k --steps 4 --warmup 1 --cpu-sample 0 "$@" > $root/gpurun_out/$tag.json 2> $root/gpurun_out/$tag.log)
(cd $root && python3 tools/sweep_rounds.py gpurun_out/$tag > gpurun_out/${tag}_rounds.txt 2>&1)
f=$(find $root/gpurun_out/$tag -name "*kernel_stats.csv" | head -1); cp "$f" $root/gpurun_out/${tag}_kernel_stats.csv
# keep the trace of the last sweep only (the merge-back limit is 64 MiB)
find $root/gpurun_out/$tag -name "*kernel_trace.csv" -size +30M -delete
head -70 $root/gpurun_out/${tag}_rounds.txt
