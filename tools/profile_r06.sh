#!/bin/bash
# Round-6 profiles: PMC passes FIRST (the bench lines that follow read their traffic / instruction counts from these summaries,
# and only while the library's hash matches), then rocprofv3 kernel stats, the bench lines, the rank-by-rank projections and
# the diagnostic breakdowns, all summarised into profiles/r06_*.
#   gpurun -- tools/profile_r06.sh [final|stars|stars1k|gibbs|proj|diag|all]     (on the GPU box; everything lands under gpurun_out/r06_*)
#   tools/profile_r06.sh collect                                                 (here, afterwards: the summaries to judge -> profiles/)
# Counters in their own passes with --kernel-trace only (gpurun rule); the program right after `--`.
set -e
what=${1:-all}
root=${GRAFT_REPO_ROOT:-$(pwd)}
if [ $what = collect ]; then
  for f in $root/gpurun_out/r06_*_kernel_stats.csv $root/gpurun_out/r06_*_bench.json $root/gpurun_out/r06_*_pmc.json $root/gpurun_out/r06_proj_*.json \
           $root/gpurun_out/r06_*.txt $root/gpurun_out/r06_default_bench_line.json; do
    [ -f "$f" ] && cp "$f" $root/profiles/
  done
  ls $root/profiles/r06_*
  exit 0
fi
cd /tmp && export TMPDIR=/tmp
stats() {   # tag, bench args
  tag=$1; shift
  rm -rf $root/gpurun_out/r06_${tag}_stats
  (cd $root && rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/r06_${tag}_stats -- python3 bench.py "$@" > $root/gpurun_out/r06_${tag}_bench.json 2> $root/gpurun_out/r06_${tag}_stats.log)
  f=$(find $root/gpurun_out/r06_${tag}_stats -name "*kernel_stats.csv" | head -1)
  cp "$f" $root/gpurun_out/r06_${tag}_kernel_stats.csv
  rm -rf $root/gpurun_out/r06_${tag}_stats
  echo "== $tag"; head -8 $root/gpurun_out/r06_${tag}_kernel_stats.csv
}
pmc() {     # tag, PMC_PROG
  tag=$1; prog=$2
  (cd $root && PMC_PROG="$prog" tools/pmc_pass.sh r06_$tag "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
       "SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE" "SQ_THREAD_CYCLES_VALU" > gpurun_out/r06_${tag}_pmc_passes.log 2>&1)
  (cd $root && python3 tools/pmc_summarise.py r06_$tag "PMC_PROG=\"$prog\" tools/pmc_pass.sh r06_$tag ..." > gpurun_out/r06_${tag}_pmc.json)
  rm -rf $root/gpurun_out/pmc_r06_${tag}_*
  # the bench run that follows reads its PMC figures from profiles/ (and only when the library hash matches): hand it this pass
  cp $root/gpurun_out/r06_${tag}_pmc.json $root/profiles/r06_${tag}_pmc.json
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
if [ $what = gibbs ] || [ $what = all ]; then
  pmc aux "bench.py --workload gibbs10k --steps 2 --warmup 1 --cpu-sample 0"
  stats gibbs --workload gibbs10k --steps 10 --warmup 2 --cpu-sample 0
  (cd $root && bash tools/trace_gibbs.sh r06_gaps > /dev/null 2>&1; cp gpurun_out/r06_gaps_rounds.txt gpurun_out/r06_gibbs_rounds.txt; rm -rf gpurun_out/r06_gaps)
fi
if [ $what = proj ] || [ $what = all ]; then
  for n in 2 4 8; do
    (cd $root && python3 bench.py --scaling strong --of $n --steps 200 --warmup 20 > gpurun_out/r06_proj_render_N$n.json 2> gpurun_out/r06_proj_render_N$n.log)
  done
  (cd $root && python3 bench.py --workload gibbs10k --scaling strong --of 8 --split strips --steps 8 > gpurun_out/r06_proj_gibbs_strips_N8.json 2> gpurun_out/r06_proj_gibbs_strips_N8.log)
  (cd $root && python3 bench.py --workload gibbs10k --scaling strong --of 8 --split replicated --steps 8 > gpurun_out/r06_proj_gibbs_replicated_N8.json 2> gpurun_out/r06_proj_gibbs_replicated_N8.log)
  echo "== projections"; ls $root/gpurun_out/r06_proj_*.json
fi
if [ $what = diag ] || [ $what = all ]; then
  # where the split's time goes and what its queue looks like; the render's work counters; the default line as the driver runs it
  (cd $root && python3 tools/ablate_split.py > gpurun_out/r06_split_ablation.txt 2>&1)
  (cd $root && python3 tools/tile_timeline.py --workload mixed10k_2048 --tail-log 24 > gpurun_out/r06_render_work.txt 2>&1)
  (cd $root && python3 bench.py --steps 20 --warmup 5 > gpurun_out/r06_default_bench_line.json 2> gpurun_out/r06_default_bench_line.log)
  echo "== diag"; head -20 $root/gpurun_out/r06_split_ablation.txt; head -4 $root/gpurun_out/r06_render_work.txt
fi
