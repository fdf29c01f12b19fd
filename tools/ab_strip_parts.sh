for cfg in "default 0" "default 4" "tools/bin/libcel_pe8.so 0" "tools/bin/libcel_pe18.so 0" "tools/bin/libcel_pe24.so 0" "tools/bin/libcel_pe18.so 4" "tools/bin/libcel_pe24.so 4"; do
  set -- $cfg
  if [ "$1" = default ]; then unset CEL_HIP_LIBRARY; else export CEL_HIP_LIBRARY=$PWD/$1; fi
  CEL_TILE_PARTS=$2 python bench.py --scaling strong --of 8 --steps 200 --warmup 20 > /tmp/p.json 2>/dev/null
  python -c "
import json;d=json.loads(open('/tmp/p.json').read().strip().splitlines()[-1]);p=d.get('projected_strong') or d;print('%-28s parts=%s  max_ms %.4f mean %.4f one_rank %.4f speedup %.3f' % ('$1','$2',p['max_ms'],p['mean_ms'],p['one_rank_ms'],p['speedup_at_N']))"
done
