#!/bin/bash
# A/B of the star-tile kernel (CEL_OPT_STAR_TILES through bench.py --star-tiles) on the star fields
set -e
mkdir -p gpurun_out
for w in ${WORKLOADS:-stars10k_2048 stars2k_4096 stars1k_512}; do
  for v in ${VARIANTS:-0 1}; do
    python bench.py --workload $w --steps 200 --warmup 20 --legs none --cpu-sample 0 --star-tiles $v > gpurun_out/star_${w}_$v.json 2> gpurun_out/star_${w}_$v.err
    python - $w $v <<'PY'
import json, sys
w, v = sys.argv[1:3]
d = json.loads(open(f"gpurun_out/star_{w}_{v}.json").read().strip().splitlines()[-1])
print(w, "star_tiles", v, "step %.4f ms" % d["ms_per_step"], d["roofline"]["kernel"], "%.4f ms" % d["roofline"]["kernel_ms"], "%.0f GB/s" % d["roofline"]["achieved"], "frac %.3f" % d["roofline"]["frac"])
PY
  done
done
