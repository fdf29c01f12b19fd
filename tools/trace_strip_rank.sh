#!/bin/bash
# kernel stats of ONE rank of the strip-partitioned chain, played on one GPU:  gpurun -- bash tools/trace_strip_rank.sh <tag> <rank> <of>
tag=${1:-r5_strip}; rank=${2:-3}; of=${3:-8}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $root/gpurun_out/$tag
(cd $root && rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/$tag -- python3 bench.py --workload gibbs10k --scaling strong --split strips --of $of --as-rank $rank --steps 6 > $root/gpurun_out/$tag.json 2> $root/gpurun_out/$tag.log)
f=$(find $root/gpurun_out/$tag -name "*kernel_stats.csv" | head -1); cp "$f" $root/gpurun_out/${tag}_kernel_stats.csv
rm -rf $root/gpurun_out/$tag
head -30 $root/gpurun_out/${tag}_kernel_stats.csv | cut -c1-160
