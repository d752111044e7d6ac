#!/bin/bash
# launch timeline (stamps build) -- usage: gpu_call_tl.sh <tag>
TAG=${1:-r02u}
mkdir -p gpurun_out
rocm-smi --showclocks 2>&1 | grep -i "sclk\|mclk\|fclk" | head -4
for cfg in "25 f64 256" "25 f64 4096" "25 f32 4096"; do
  set -- $cfg
  timeout -k 10 120 python tools/stamp_profile.py --win-ms $1 --compute $2 --batch $3 > gpurun_out/${TAG}_stamps_n$1_$2_b$3.log 2>&1
  grep "kernel \|GHz\|wave lifetime" gpurun_out/${TAG}_stamps_n$1_$2_b$3.log
done
