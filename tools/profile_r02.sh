#!/bin/bash
# Round-2 profiles: rocprofv3 kernel stats + PMC passes of the three bench workloads.
#   tools/profile_r02.sh            (on the GPU box; results under gpurun_out/r02_*)
# Counters in their own passes with --kernel-trace only (gpurun rule); the program right after `--`.
set -e
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
stats() {   # tag, bench args
  tag=$1; shift
  rm -rf $root/gpurun_out/r02_${tag}_stats
  (cd $root && rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/r02_${tag}_stats -- python3 bench.py "$@" > $root/gpurun_out/r02_${tag}_bench.json 2> $root/gpurun_out/r02_${tag}_stats.log)
  f=$(find $root/gpurun_out/r02_${tag}_stats -name "*kernel_stats.csv" | head -1)
  cp "$f" $root/gpurun_out/r02_${tag}_kernel_stats.csv
  echo "== $tag"; head -12 $root/gpurun_out/r02_${tag}_kernel_stats.csv
}
stats final --steps 200 --warmup 30 --cpu-sample 0 --legs none
stats stars --workload stars10k_2048 --steps 200 --warmup 30 --cpu-sample 0 --legs none
stats stars1k --workload stars1k_512 --steps 200 --warmup 30 --cpu-sample 0 --legs none
stats gibbs --workload gibbs10k --steps 5 --warmup 2
cd $root
PMC_PROG="bench.py --steps 3 --warmup 1 --cpu-sample 0 --legs none" tools/pmc_pass.sh r02_final "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE" > gpurun_out/r02_final_pmc.txt 2>&1
PMC_PROG="bench.py --workload stars10k_2048 --steps 3 --warmup 1 --cpu-sample 0 --legs none" tools/pmc_pass.sh r02_stars "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" > gpurun_out/r02_stars_pmc.txt 2>&1
PMC_PROG="bench.py --workload gibbs10k --steps 2 --warmup 1" tools/pmc_pass.sh r02_aux "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" > gpurun_out/r02_aux_pmc.txt 2>&1
tail -40 gpurun_out/r02_final_pmc.txt
