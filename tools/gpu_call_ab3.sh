#!/bin/bash
# quick A/B of the N = 400 default kernel (no tests): gpu_call_ab3.sh <tag>
TAG=${1:-r03m}
mkdir -p gpurun_out
for c in f64 f32; do
  timeout -k 10 200 python tools/ab_bench.py --win-ms 25 --compute $c --batch 256 > gpurun_out/${TAG}_ab_n25_${c}.log 2>&1; grep "w20 (default)\|w20 one\|w20 pers\|w25 (25" gpurun_out/${TAG}_ab_n25_${c}.log
  timeout -k 10 200 python tools/ab_bench.py --win-ms 25 --compute $c --batch 4096 --rounds 7 --launches 50 > gpurun_out/${TAG}_ab_n25_${c}_big.log 2>&1; grep "w20 (default)\|w20 one\|w20 pers\|w25 (25" gpurun_out/${TAG}_ab_n25_${c}_big.log
done
