#!/bin/bash
# cfg4 (mel + gabor): tests, then the gabor kernel variants -- usage: gpu_call_gabor.sh <tag>
TAG=${1:-r03p}
mkdir -p gpurun_out
timeout -k 10 200 python -m pytest tests -q -m gpu -k "gabor or process_batch or sndenv" -p no:cacheprovider > gpurun_out/${TAG}_pytest_gabor.log 2>&1; rc=$?; tail -2 gpurun_out/${TAG}_pytest_gabor.log
[ $rc -eq 0 ] || exit $rc
for v in 0 2 4 1; do
  timeout -k 10 300 python bench.py --workload cfg4 --no-cpu-baseline --option gabor_lds=$v > gpurun_out/${TAG}_bench_cfg4_f64_g$v.json 2> gpurun_out/${TAG}_bench_cfg4_f64_g$v.err; echo "cfg4 f64 gabor variant $v rc=$?"
done
for v in 0 2 4; do
  timeout -k 10 300 python bench.py --workload cfg4 --no-cpu-baseline --option gabor_lds=$v --streams 1 > gpurun_out/${TAG}_bench_cfg4_f64_g${v}_1s.json 2> gpurun_out/${TAG}_bench_cfg4_f64_g${v}_1s.err; echo "cfg4 f64 gabor variant $v one stream rc=$?"
done
timeout -k 10 300 python bench.py --workload cfg4 --compute f32 --no-cpu-baseline --report-anyway --option gabor_lds=2 > gpurun_out/${TAG}_bench_cfg4_f32_g2.json 2> gpurun_out/${TAG}_bench_cfg4_f32_g2.err; echo "cfg4 f32 rc=$?"
python - "$TAG" <<'PY'
import json,sys
for f in ("cfg4_f64_g0","cfg4_f64_g2","cfg4_f64_g4","cfg4_f64_g1","cfg4_f64_g0_1s","cfg4_f64_g2_1s","cfg4_f64_g4_1s","cfg4_f32_g2"):
    try:
        d=json.loads(open("gpurun_out/%s_bench_%s.json" % (sys.argv[1], f)).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "no json", e); continue
    print(f, d["value"], d["steps"], d["us_per_step_device"]["mean"], d["config"]["kernel"], d["parity"]["max_scaled_err"], d["parity"]["n_past_1e-5"])
PY
