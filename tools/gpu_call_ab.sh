#!/bin/bash
# A/B of the N=512 / N=400 kernel variants + the gpu-tier tests that cover them.  usage: gpu_call_ab.sh <tag> <win_ms>
TAG=${1:-r02b}; WIN=${2:-32}
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/${TAG}_pytest.log 2>&1; tail -3 gpurun_out/${TAG}_pytest.log
for c in f32 f64; do
  timeout -k 10 120 python tools/ab_bench.py --win-ms $WIN --compute $c > gpurun_out/${TAG}_ab_$c.log 2>&1; grep -v amdgpu.ids gpurun_out/${TAG}_ab_$c.log
  timeout -k 10 120 python tools/ab_bench.py --win-ms $WIN --compute $c --batch 4096 --rounds 7 --launches 50 > gpurun_out/${TAG}_ab_${c}_big.log 2>&1; grep -v amdgpu.ids gpurun_out/${TAG}_ab_${c}_big.log
done
