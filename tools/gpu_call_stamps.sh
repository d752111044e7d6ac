#!/bin/bash
# phase stamps of the wave kernels (diagnostic build) -- usage: gpu_call_stamps.sh <tag>
TAG=${1:-r02c}
mkdir -p gpurun_out
for cfg in "32 f32 256" "32 f64 256" "25 f64 256" "25 f32 256" "32 f64 4096" "25 f64 4096"; do
  set -- $cfg
  timeout -k 10 120 python tools/stamp_profile.py --win-ms $1 --compute $2 --batch $3 > gpurun_out/${TAG}_stamps_n$1_$2_b$3.log 2>&1
  grep -v amdgpu.ids gpurun_out/${TAG}_stamps_n$1_$2_b$3.log
done
