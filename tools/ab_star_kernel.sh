#!/bin/bash
# A/B of the star-tile kernel (CEL_STAR_KERNEL = number of column parts, 0 = k_render_hw) on the star fields
set -e
mkdir -p gpurun_out
for v in ${VARIANTS:-0 1 2 4}; do
  CEL_STAR_KERNEL=$v python bench.py --workload stars10k_2048 --steps 200 --warmup 20 --legs none > gpurun_out/star_kernel_$v.json 2> gpurun_out/star_kernel_$v.err
  CEL_STAR_KERNEL=$v python bench.py --workload stars1k_512 --steps 200 --warmup 20 --legs none > gpurun_out/star_kernel_1k_$v.json 2>> gpurun_out/star_kernel_$v.err
done
python - <<'PY'
import json, os
for v in os.environ.get("VARIANTS", "0 1 2 4").split():
    for n in (f"star_kernel_{v}", f"star_kernel_1k_{v}"):
        d=json.loads(open(f"gpurun_out/{n}.json").read().strip().splitlines()[-1])
        print(n, round(d["ms_per_step"],4), d["roofline"].get("kernel_ms"), round(d["roofline"]["achieved"],1))
PY
