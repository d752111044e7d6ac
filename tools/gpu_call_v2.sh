#!/bin/bash
TAG=${1:-r02d}
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests -q -m gpu -p no:cacheprovider -x > gpurun_out/${TAG}_pytest.log 2>&1; tail -3 gpurun_out/${TAG}_pytest.log
for cfg in "25 f64 256" "25 f32 256"; do
  set -- $cfg
  timeout -k 10 120 python tools/stamp_profile.py --win-ms $1 --compute $2 --batch $3 > gpurun_out/${TAG}_stamps_n$1_$2_b$3.log 2>&1
  grep -v amdgpu.ids gpurun_out/${TAG}_stamps_n$1_$2_b$3.log | head -12
done
for w in 25; do for c in f64 f32; do
  timeout -k 10 120 python tools/ab_bench.py --win-ms $w --compute $c > gpurun_out/${TAG}_ab_n${w}_$c.log 2>&1; grep -v amdgpu.ids gpurun_out/${TAG}_ab_n${w}_$c.log | head -9
  timeout -k 10 120 python tools/ab_bench.py --win-ms $w --compute $c --batch 4096 --rounds 7 --launches 50 > gpurun_out/${TAG}_ab_n${w}_${c}_big.log 2>&1; grep -v amdgpu.ids gpurun_out/${TAG}_ab_n${w}_${c}_big.log | head -9
done; done
