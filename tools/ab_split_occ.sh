#!/bin/bash
for lib in "" tools/bin/libceleste_pad5.so tools/bin/libceleste_pad4.so; do
  echo "lib=$lib"
  if [ -n "$lib" ]; then export CEL_HIP_LIBRARY=$PWD/$lib; else unset CEL_HIP_LIBRARY; fi
  python bench.py --workload gibbs10k --steps 20 --warmup 3 --cpu-sample 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['device_ms_per_sweep'])"
done
