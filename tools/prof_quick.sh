cd /tmp && export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT
rm -rf $root/gpurun_out/r5_p1
(cd $root && rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/r5_p1 -- python3 bench.py --steps 50 --warmup 5 --cpu-sample 0 --legs none > $root/gpurun_out/r5_p1.json 2> $root/gpurun_out/r5_p1.log)
f=$(find $root/gpurun_out/r5_p1 -name "*kernel_stats.csv" | head -1)
cp "$f" $root/gpurun_out/r5_p1_kernel_stats.csv
head -12 $root/gpurun_out/r5_p1_kernel_stats.csv
rm -rf $root/gpurun_out/r5_p1
