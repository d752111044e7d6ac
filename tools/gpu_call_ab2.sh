#!/bin/bash
TAG=${1:-r02o}
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests -q -m gpu -p no:cacheprovider -x > gpurun_out/${TAG}_pytest.log 2>&1; tail -2 gpurun_out/${TAG}_pytest.log
for w in 25 32; do for c in f64 f32; do
  timeout -k 10 120 python tools/ab_bench.py --win-ms $w --compute $c > gpurun_out/${TAG}_ab_n${w}_$c.log 2>&1; grep -v "amdgpu.ids\|skip r16" gpurun_out/${TAG}_ab_n${w}_$c.log | head -7
  timeout -k 10 120 python tools/ab_bench.py --win-ms $w --compute $c --batch 4096 --rounds 7 --launches 50 > gpurun_out/${TAG}_ab_n${w}_${c}_big.log 2>&1; grep -v "amdgpu.ids\|skip r16" gpurun_out/${TAG}_ab_n${w}_${c}_big.log | head -7
done; done
