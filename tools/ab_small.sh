#!/bin/bash
# timing-only ablations of k_small_stars (results are WRONG with a bit set): build one library per variant into tools/bin, time each
#   tools/ab_small.sh build   (here)      tools/ab_small.sh run   (on the GPU box)
cd "$(dirname "$0")/.."
VARIANTS="0 1 2 4 7"
if [ "$1" = build ]; then
  mkdir -p tools/bin
  for v in $VARIANTS; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 -fno-gpu-rdc -DSMALL_ABL=$v $EXTRA -o tools/bin/libcel_small_$v.so desi-mcmc_amd/csrc/celeste_hip.hip 2>/dev/null &
  done
  wait
  ls -la tools/bin/libcel_small_*.so
else
  for v in $VARIANTS; do
    echo "SMALL_ABL=$v"
    CEL_HIP_LIBRARY=$PWD/tools/bin/libcel_small_$v.so python tools/step_breakdown.py stars1k_512 500 | grep kernels
  done
fi
