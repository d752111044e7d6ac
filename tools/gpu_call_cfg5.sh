#!/bin/bash
# N = 2048 (BASELINE configs[4]): parity tests, then the bench rows -- usage: gpu_call_cfg5.sh <tag>
TAG=${1:-r03f}
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests -q -m gpu -k "n2048 or cfg5 or stereo or workgroup_order" -p no:cacheprovider > gpurun_out/${TAG}_pytest_n2048.log 2>&1; rc=$?; tail -3 gpurun_out/${TAG}_pytest_n2048.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 500 python bench.py --workload cfg5 --batch 1280 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/${TAG}_bench_cfg5_f64.json 2> gpurun_out/${TAG}_bench_cfg5_f64.err; echo "cfg5 f64 rc=$?"; tail -c 300 gpurun_out/${TAG}_bench_cfg5_f64.err
timeout -k 10 500 python bench.py --workload cfg5 --batch 1280 --steps 20 --warmup 3 --compute f32 --no-cpu-baseline --report-anyway > gpurun_out/${TAG}_bench_cfg5_f32.json 2> gpurun_out/${TAG}_bench_cfg5_f32.err; echo "cfg5 f32 rc=$?"
timeout -k 10 500 python bench.py --workload cfg5 --batch 1280 --steps 20 --warmup 3 --no-cpu-baseline --option kernel=2 > gpurun_out/${TAG}_bench_cfg5_f64_r1024.json 2> gpurun_out/${TAG}_bench_cfg5_f64_r1024.err; echo "cfg5 f64 tile kernel rc=$?"
python - "$TAG" <<'PY'
import json,sys
for f in ("cfg5_f64","cfg5_f32","cfg5_f64_r1024"):
    try:
        d=json.loads(open("gpurun_out/%s_bench_%s.json" % (sys.argv[1], f)).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "no json", e); continue
    print(f, d["value"], d["steps"], d["us_per_step_device"], d["roofline"]["frac"], d["config"]["kernel"], d["parity"]["max_scaled_err"], d["parity"]["n_past_1e-5"], d["config"].get("streams"))
PY
