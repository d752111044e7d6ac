#!/bin/bash
# phase stamps + launch timeline (stamps build) -- usage: gpu_call_tl.sh <tag> ["win compute batch" ...]
TAG=${1:-r02u}; shift
mkdir -p gpurun_out
[ $# -eq 0 ] && set -- "25 f64 256" "25 f64 4096" "25 f32 256"
for cfg in "$@"; do
  set -- $cfg
  timeout -k 10 200 python tools/stamp_profile.py --win-ms $1 --compute $2 --batch $3 > gpurun_out/${TAG}_stamps_n$1_$2_b$3.log 2>&1
  grep -v "amdgpu.ids\|0-1 \|1-2 \|2-3 \|RuntimeWarning\|np.percentile" gpurun_out/${TAG}_stamps_n$1_$2_b$3.log | head -40
done
