#!/bin/bash
# N = 2048: parity tests, bench rows, stamps -- usage: gpu_call_cfg5b.sh <tag>
TAG=${1:-r03h}
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests -q -m gpu -k "n2048 or cfg5 or stereo or workgroup_order" -p no:cacheprovider > gpurun_out/${TAG}_pytest_n2048.log 2>&1; rc=$?; tail -3 gpurun_out/${TAG}_pytest_n2048.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 500 python bench.py --workload cfg5 --batch 1280 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/${TAG}_bench_cfg5_f64.json 2> gpurun_out/${TAG}_bench_cfg5_f64.err; echo "cfg5 f64 rc=$?"; tail -c 300 gpurun_out/${TAG}_bench_cfg5_f64.err
timeout -k 10 500 python bench.py --workload cfg5 --batch 1280 --steps 20 --warmup 3 --compute f32 --no-cpu-baseline --report-anyway > gpurun_out/${TAG}_bench_cfg5_f32.json 2> gpurun_out/${TAG}_bench_cfg5_f32.err; echo "cfg5 f32 rc=$?"
python - "$TAG" <<'PY'
import json,sys
for f in ("cfg5_f64","cfg5_f32"):
    try:
        d=json.loads(open("gpurun_out/%s_bench_%s.json" % (sys.argv[1], f)).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "no json", e); continue
    print(f, d["value"], d["steps"], d["us_per_step_device"], d["roofline"]["frac"], d["config"]["kernel"], d["parity"]["max_scaled_err"], d["parity"]["n_past_1e-5"], d["config"].get("streams"))
PY
bash tools/gpu_call_tl.sh $TAG "46.44 f64 64" "46.44 f32 64" | grep "kernel\|^[0-9]-[0-9]\|lifetime\|GHz\|waves per CU\| us  "
