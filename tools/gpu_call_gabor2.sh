#!/bin/bash
TAG=${1:-r03q}
mkdir -p gpurun_out
timeout -k 10 200 python -m pytest tests -q -m gpu -k "gabor or process_batch or sndenv" -p no:cacheprovider > gpurun_out/${TAG}_pytest_gabor.log 2>&1; rc=$?; tail -2 gpurun_out/${TAG}_pytest_gabor.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --workload cfg4 --no-cpu-baseline > gpurun_out/${TAG}_bench_cfg4_f64.json 2> gpurun_out/${TAG}_bench_cfg4_f64.err; echo "cfg4 f64 rc=$?"
timeout -k 10 300 python bench.py --workload cfg4 --no-cpu-baseline --streams 1 > gpurun_out/${TAG}_bench_cfg4_f64_1s.json 2> gpurun_out/${TAG}_bench_cfg4_f64_1s.err; echo "cfg4 f64 1 stream rc=$?"
timeout -k 10 300 python bench.py --workload cfg4 --compute f32 --no-cpu-baseline --report-anyway > gpurun_out/${TAG}_bench_cfg4_f32.json 2> gpurun_out/${TAG}_bench_cfg4_f32.err; echo "cfg4 f32 rc=$?"
python - "$TAG" <<'PY'
import json,sys
for f in ("cfg4_f64","cfg4_f64_1s","cfg4_f32"):
    try:
        d=json.loads(open("gpurun_out/%s_bench_%s.json" % (sys.argv[1], f)).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "no json", e); continue
    print(f, d["value"], d["steps"], d["us_per_step_device"]["mean"], d["config"]["kernel"], d["parity"]["max_scaled_err"], d["parity"]["n_past_1e-5"])
PY
