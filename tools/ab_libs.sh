#!/bin/bash
# A/B of library builds on ONE box: the default bench line's step and kernel times for each library given (CEL_HIP_LIBRARY),
# twice each, interleaved.   gpurun -- bash tools/ab_libs.sh tools/bin/a.so tools/bin/b.so ...   ("default" = the shipped one)
for rep in 1 2; do
  for lib in "$@"; do
    if [ "$lib" = default ]; then unset CEL_HIP_LIBRARY; else export CEL_HIP_LIBRARY=$PWD/$lib; fi
    python bench.py --steps ${AB_STEPS:-300} --warmup 30 --cpu-sample 0 --legs none ${AB_ARGS} > /tmp/ab.json 2>/dev/null
    python -c "
import json;d=json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1]);print('%-28s ms_per_step %.4f kernel_ms %.4f' % ('$lib', d['ms_per_step'], d['roofline']['kernel_ms']))"
  done
done
