#!/bin/bash
# launch timeline (stamps build) + the bench line -- usage: gpu_call_tl.sh <tag>
TAG=${1:-r02u}
mkdir -p gpurun_out
for cfg in "25 f64 256" "25 f64 4096" "25 f32 256"; do
  set -- $cfg
  timeout -k 10 120 python tools/stamp_profile.py --win-ms $1 --compute $2 --batch $3 > gpurun_out/${TAG}_stamps_n$1_$2_b$3.log 2>&1
  grep -v "amdgpu.ids\|0-1 \|1-2 \|2-3 " gpurun_out/${TAG}_stamps_n$1_$2_b$3.log
done
timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver_style.json 2> gpurun_out/${TAG}_bench_driver_style.err && cat gpurun_out/${TAG}_bench_driver_style.json
