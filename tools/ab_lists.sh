#!/bin/bash
# A/B of CEL_OPT_PHOTON_LISTS on the Gibbs sweep (on the GPU box)
for m in 2 1 0; do
  echo "photon lists mode $m"
  python bench.py --workload gibbs10k --steps 20 --warmup 3 --cpu-sample 0 --photon-lists $m 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['sweep_ms'], d['device_ms_per_sweep'], d['loglik_trace_tail'])"
done
