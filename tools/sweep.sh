#!/bin/bash
# usage: sweep.sh "<flags1>" "<flags2>" ...   (each run prints kernel ms)
for f in "$@"; do
  timeout -k 10 200 python bench.py --steps 10 --warmup 2 --cpu-sample 0 $f 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$f', '| step %.3f ms | render %.3f bin %.3f prep %.3f | val %.3e | ll %.10e' % (d['ms_per_step'], d['kernels_ms']['k_render'], d['kernels_ms']['k_bin'], d['kernels_ms']['k_prep'], d['value'], d['loglik']))
"
done
