#!/bin/bash
# gpurun -- bash tools/ab_scratch.sh   : the shipped kernels against their zero-scratch variants (ms per launch, three repeats)
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
python3 tools/ab_scratch.py 2>&1 | grep -v amdgpu.ids
CEL_HIP_LIBRARY=$root/tools/bin/libceleste_hip_noscratch.so python3 tools/ab_scratch.py 2>&1 | grep -v amdgpu.ids
python3 tools/ab_scratch.py 2>&1 | grep -v amdgpu.ids
CEL_HIP_LIBRARY=$root/tools/bin/libceleste_hip_noscratch.so python3 tools/ab_scratch.py 2>&1 | grep -v amdgpu.ids
