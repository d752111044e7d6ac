#!/bin/bash
# round-2 second GPU call: VALU issue rates, the bench in f32/f64 x N=400/512, A/B of the variants
set -u
mkdir -p gpurun_out
step() { local name=$1 secs=$2; shift 2; echo "[call2] $name"; timeout -k 10 "$secs" "$@" > "gpurun_out/r02_${name}.log" 2>&1; local rc=$?; echo "[call2] $name rc=$rc"; tail -2 "gpurun_out/r02_${name}.log"; if [ $rc -eq 124 ] || [ $rc -ge 128 ]; then return $rc; fi; return 0; }
step valu_rates 60 ./tools/ubench/valu_rates &&
step bench_f32_n512 200 python bench.py --steps 2000 --no-cpu-baseline &&
step bench_f32_n400 200 python bench.py --steps 2000 --win-ms 25 --no-cpu-baseline &&
step bench_f64_n512 200 python bench.py --steps 2000 --compute f64 --no-cpu-baseline &&
step bench_f64_n400 200 python bench.py --steps 2000 --win-ms 25 --compute f64 --no-cpu-baseline &&
step bench_f32_n512_cpu 300 python bench.py --steps 2000 &&
step ab_n512_f32 200 python tools/ab_bench.py &&
step ab_n512_f64 200 python tools/ab_bench.py --compute f64 &&
step ab_n400_f32 200 python tools/ab_bench.py --win-ms 25 &&
step ab_n400_f64 200 python tools/ab_bench.py --win-ms 25 --compute f64 &&
step ab_n512_f32_big 200 python tools/ab_bench.py --batch 4096 --rounds 7 --launches 50 &&
step ab_n400_f64_big 200 python tools/ab_bench.py --win-ms 25 --compute f64 --batch 4096 --rounds 7 --launches 50
echo "[call2] done"
