#!/bin/bash
for bias in 1 2 4 8 1000; do
  echo "bias $bias"
  CEL_NZ_BIAS=$bias python bench.py --workload gibbs10k --steps 20 --warmup 3 --cpu-sample 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['sweep_ms']['location_slice'], d['device_ms_per_sweep']['k_patch_ll_hw<0> (location)'])"
done
