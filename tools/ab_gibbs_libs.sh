#!/bin/bash
# A/B of library builds on ONE box by Gibbs sweeps of the benchmark field (tools/ab_nz.py), twice each, interleaved.
#   gpurun -- bash tools/ab_gibbs_libs.sh tools/bin/a.so tools/bin/b.so ...   ("default" = the shipped one)
for rep in 1 2; do
  for lib in "$@"; do
    if [ "$lib" = default ]; then python tools/ab_nz.py; else python tools/ab_nz.py $lib; fi
  done
done
