#!/bin/bash
# Round-5 profiles: rocprofv3 kernel stats + PMC passes of the bench workloads, summarised into profiles/r05_*.
#   gpurun -- tools/profile_r05.sh [final|stars|stars1k|gibbs|all]     (on the GPU box; everything lands under gpurun_out/r05_*)
#   tools/profile_r05.sh collect                                (here, afterwards: the summaries to judge -> profiles/)
# Counters in their own passes with --kernel-trace only (gpurun rule); the program right after `--`.
set -e
what=${1:-all}
root=${GRAFT_REPO_ROOT:-$(pwd)}
if [ $what = collect ]; then
  for f in $root/gpurun_out/r05_*_kernel_stats.csv $root/gpurun_out/r05_*_bench.json $root/gpurun_out/r05_*_pmc.json $root/gpurun_out/r05_proj_*.json; do
    [ -f "$f" ] && cp "$f" $root/profiles/
  done
  ls $root/profiles/r05_*
  exit 0
fi
cd /tmp && export TMPDIR=/tmp
stats() {   # tag, bench args
  tag=$1; shift
  rm -rf $root/gpurun_out/r05_${tag}_stats
  (cd $root && rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/r05_${tag}_stats -- python3 bench.py "$@" > $root/gpurun_out/r05_${tag}_bench.json 2> $root/gpurun_out/r05_${tag}_stats.log)
  f=$(find $root/gpurun_out/r05_${tag}_stats -name "*kernel_stats.csv" | head -1)
  cp "$f" $root/gpurun_out/r05_${tag}_kernel_stats.csv
  echo "== $tag"; head -8 $root/gpurun_out/r05_${tag}_kernel_stats.csv
}
pmc() {     # tag, PMC_PROG
  tag=$1; prog=$2
  (cd $root && PMC_PROG="$prog" tools/pmc_pass.sh r05_$tag "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE" > gpurun_out/r05_${tag}_pmc.txt 2>&1)
  (cd $root && python3 tools/pmc_summarise.py r05_$tag "PMC_PROG=\"$prog\" tools/pmc_pass.sh r05_$tag ..." > gpurun_out/r05_${tag}_pmc.json)
  # the bench run that follows reads its PMC figures from profiles/ (and only when the library hash matches): hand it this pass
  cp $root/gpurun_out/r05_${tag}_pmc.json $root/profiles/r05_${tag}_pmc.json
}
if [ $what = final ] || [ $what = all ]; then
  pmc final "bench.py --steps 3 --warmup 1 --cpu-sample 0 --legs none"
  stats final --steps 200 --warmup 30 --cpu-sample 0 --legs none
fi
if [ $what = stars ] || [ $what = all ]; then
  pmc stars "bench.py --workload stars10k_2048 --steps 3 --warmup 1 --cpu-sample 0 --legs none"
  stats stars --workload stars10k_2048 --steps 200 --warmup 30 --cpu-sample 0 --legs none
fi
if [ $what = stars1k ] || [ $what = all ]; then
  pmc stars1k "bench.py --workload stars1k_512 --steps 3 --warmup 1 --cpu-sample 0 --legs none"
  stats stars1k --workload stars1k_512 --steps 400 --warmup 30 --cpu-sample 0 --legs none
fi
if [ $what = proj ] || [ $what = all ]; then
  for n in 2 4 8; do
    (cd $root && python3 bench.py --scaling strong --of $n --steps 200 --warmup 20 > gpurun_out/r05_proj_render_N$n.json 2> gpurun_out/r05_proj_render_N$n.log)
  done
  (cd $root && python3 bench.py --workload gibbs10k --scaling strong --of 8 --split strips --steps 8 > gpurun_out/r05_proj_gibbs_strips_N8.json 2> gpurun_out/r05_proj_gibbs_strips_N8.log)
  (cd $root && python3 bench.py --workload gibbs10k --scaling strong --of 8 --split replicated --steps 8 > gpurun_out/r05_proj_gibbs_replicated_N8.json 2> gpurun_out/r05_proj_gibbs_replicated_N8.log)
  echo "== projections"; ls $root/gpurun_out/r05_proj_*.json
fi
if [ $what = gibbs ] || [ $what = all ]; then
  stats gibbs --workload gibbs10k --steps 10 --warmup 2 --cpu-sample 0
  pmc aux "bench.py --workload gibbs10k --steps 2 --warmup 1 --cpu-sample 0"
fi
